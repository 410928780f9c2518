import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_fixture, plan_from_fixture
from pygsti_amd import _lib
fx = load_fixture(sys.argv[1] if len(sys.argv) > 1 else "smq1Q_XYI_L128_depol")
pl = plan_from_fixture(fx)
J = pl.fill_dprobs(param_idx=fx["dprobs_cols"], mode=_lib.DERIV_ANALYTIC)
rows = fx["matrix_rows"]
d = np.abs(J[rows] - fx["dprobs_matrix"])
print("max err", d.max())
bad_rows = np.flatnonzero(d.max(axis=1) > 1e-8)
print("bad rows", len(bad_rows), "of", len(rows), bad_rows[:20])
bad_cols = np.flatnonzero(d.max(axis=0) > 1e-8)
print("bad cols", bad_cols)
lens = np.diff(fx["circ_ptr"])
el_c = fx["el_circuit"][rows[bad_rows]]
print("circuit lengths of bad rows", np.unique(lens[el_c])[:30], "min", lens[el_c].min() if len(el_c) else None)
good_c = np.setdiff1d(np.arange(len(lens)), el_c)
print("max length among good circuits", lens[good_c].max())
r = bad_rows[0]
print("row", r, "got", J[rows[r]][:20], "\nexp", fx["dprobs_matrix"][r][:20])
