"""The reference's only published end-to-end workload (test/performance/mpi_2D_scaling/run_me_with_mpirun.py:12-59: smq2Q_XYICNOT
GST, L <= 64, 13,958 circuits, chi^2 stages then logL; 3,113 s on one rank, reference-8955f5d/timings.json:7), RECORDED in the
build container (tests/golden/make_golden_fit2q.py: 73 LM iterations, per iterate the parameter vector and checksums of the
normal equations numpy formed from the reference's own dlsvec / lsvec arrays) and REPLAYED on the device
(tools/fit_replay2q.py): vector -> bit-exact FD Jacobian -> objective rows -> J_s^T J_s, J_s^T lsvec.

CPU half: the record's meaning is pinned without a GPU -- the oracle (the reference's C++ reps when built) forms the Jacobian
of two iterates of the small (L = 1) record, the numpy objective maps the row factors, and the checksums come out."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, design_checker

sys.path.insert(0, os.path.join(ROOT, "tools"))
import fit_replay2q as FR          # noqa: E402


def test_fit_record_checksums_on_the_oracle(oracle_built):
    from oracle import objective_oracle as OO
    from pygsti_amd import modelpacks
    from pygsti_amd.layout import HipCOPALayout
    fx = FR.load("fit_smq2Q_XYICNOT_L1_full")
    pack = modelpacks.smq2Q_XYICNOT
    model = pack.target_model()
    assert np.allclose(model.to_vector(), fx["start_vec"], rtol=0, atol=1e-15)
    assert [str(l) for l in fx["op_labels"]] == [str(l) for l in pack.gate_labels]
    ops = [str(l) for l in fx["op_labels"]]
    ptr, g = fx["s0_circ_ptr"], fx["s0_circ_gates"]
    circuits = [tuple(ops[k] for k in g[ptr[i]:ptr[i + 1]]) for i in range(len(ptr) - 1)]
    assert circuits == [tuple(c) for c in pack.create_gst_circuits(1, lite=True)]
    lay = HipCOPALayout(circuits, model, num_atoms=1)
    eff = [str(l) for l in fx["effect_labels"]]
    order = [eff.index(str(l).split("_", 1)[-1]) for l in model.effect_labels]
    nE, nP = lay.global_num_elements, int(fx["nP"])
    counts, totals = np.empty(nE), np.empty(nE)
    for i in range(len(circuits)):
        sl = lay.indices_for_index(i)
        counts[sl] = fx["s0_counts"][i, order]; totals[sl] = fx["s0_totals"][i]
    u = fx["u"]
    n_obj = int(fx["n_obj"])
    kinds = [int(fx["ob%d_kind" % k]) for k in range(n_obj)]
    for k in (0, kinds.index(1)):                                   # the first chi^2 iterate and the first logL one
        model.from_vector(fx["ob%d_vec" % k])
        orc = design_checker(oracle_built, pack, model, circuits, lay)
        J, p = orc.dprobs(np.arange(nP), eps=float(fx["derivative_eps"]), return_probs=True)
        kind = kinds[k]
        mpc = float(fx["ob%d_mpcw" % k]) if kind == 0 else float(fx["ob%d_min_p" % k])
        rad = 1e-4 if kind == 0 else float(fx["ob%d_radius" % k])
        p = np.clip(p, float(fx["ob%d_clip_lo" % k]), float(fx["ob%d_clip_hi" % k]))
        t, ls, dt, rs = OO.objective_rows(OO.CHI2 if kind == 0 else OO.DLOGL, p, counts, totals, mpc, rad)
        Js = J * rs[:, None]
        jtj = Js.T @ Js
        scale = np.abs(fx["ob%d_diag" % k]).max()
        tol = 1e-11 if kind == 0 else 1e-9
        assert np.abs(np.diag(jtj) - fx["ob%d_diag" % k]).max() <= tol * scale, k
        assert np.abs(jtj @ u - fx["ob%d_jtju" % k]).max() <= tol * scale, k
        assert abs(np.trace(jtj) - float(fx["ob%d_trace" % k])) <= tol * float(fx["ob%d_trace" % k])
        assert abs(np.linalg.norm(jtj) - float(fx["ob%d_fro" % k])) <= tol * float(fx["ob%d_fro" % k])
        fs = float(fx["ob%d_fsum" % k])
        assert abs(t.sum() - fs) <= tol * fs
        assert np.abs(Js.T @ ls - fx["ob%d_jtf" % k]).max() <= tol * np.sqrt(scale * fs)


@pytest.mark.gpu
@pytest.mark.parametrize("name,n_expected", [("fit_smq2Q_XYICNOT_L1_full", 24), ("fit_smq2Q_XYICNOT_L64_full", 73)])
def test_gpu_replay_of_the_recorded_2q_fit(name, n_expected):
    r = FR.replay(name)
    assert r["iterations"] == n_expected
    w = r["worst_relative_deviation"]
    # chi^2 rows are bit-identical maps of bit-identical probabilities: only the summation order of the products differs
    # (1e-12); the logL rows carry log() rounding (1e-10)
    assert w["diag"] < 1e-9 and w["jtju"] < 1e-9 and w["trace"] < 1e-10 and w["fro"] < 1e-10 and w["fsum"] < 1e-10, w
    assert w["jtf"] < 1e-9, w
    assert r["device_ms_sum"] < 1e3 * r["reference_dlsvec_seconds_same_calls"]          # (the device is not slower than the CPU run)
