"""Analytic derivative mode (GST_DERIV_ANALYTIC) on the GPU against the analytic oracles: the golden vectors
of the reference's MatrixForwardSimulator (tolerance 1e-8, the north-star bar; observed ~1e-12) and the numpy
forward/backward Jacobian of oracle/oracle.py on the deep 2Q circuits where Matrix is infeasible."""
import numpy as np
import pytest

from conftest import load_fixture, assert_bitwise, plan_from_fixture
from pygsti_amd import _lib

pytestmark = pytest.mark.gpu
TOL = 1e-8


@pytest.mark.parametrize("name", ["smq1Q_XYI_L4_depol", "smq1Q_XYI_L4_kick", "smq1Q_XYI_L128_depol",
                                  "smq2Q_XYICNOT_L2_depol"])
def test_analytic_dprobs_vs_matrix_simulator(name):
    fx = load_fixture(name)
    pl = plan_from_fixture(fx)
    pr = np.empty(int(fx["nE"]))
    J = pl.fill_dprobs(param_idx=fx["dprobs_cols"], probs_out=pr, mode=_lib.DERIV_ANALYTIC)
    rows = fx["matrix_rows"]
    assert np.abs(J[rows] - fx["dprobs_matrix"]).max() < TOL
    assert_bitwise(pr, fx["probs"], "probabilities of the analytic call are the base pass's")
    # and it is NOT the finite-difference Jacobian (which carries O(eps * p'') truncation error)
    assert np.abs(J - fx["dprobs_map"]).max() > 1e-9


def test_analytic_deep_circuits_vs_numpy_oracle(oracle_built):
    fx = load_fixture("smq2Q_XYICNOT_L1024_deep")     # depth up to 1030, |J| up to 65
    pl = plan_from_fixture(fx)
    cols = np.concatenate([np.arange(0, 100), fx["dprobs_cols"], np.arange(1100, 1130), np.arange(1360, 1616)])
    J = pl.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
    Jo, _ = oracle_built.analytic_dprobs(fx, cols)
    scale = max(1.0, np.abs(Jo).max())
    assert np.abs(J - Jo).max() < TOL * scale
    none = fx["pkind"][cols] == -1
    assert none.any() and (J[:, none] == 0).all()


def test_analytic_column_window_and_permutation():
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    pl = plan_from_fixture(fx)
    nE = int(fx["nE"])
    full = pl.fill_dprobs(param_idx=np.arange(1616), mode=_lib.DERIV_ANALYTIC)
    rng = np.random.default_rng(3)
    cols = rng.permutation(1616)[:300]
    out = np.full((nE, 320), -3.0)
    pl.fill_dprobs(out=out, param_idx=cols, dest_idx=np.arange(300) + 7, mode=_lib.DERIV_ANALYTIC)
    assert np.array_equal(out[:, 7:307], full[:, cols])
    assert (out[:, :7] == -3.0).all() and (out[:, 307:] == -3.0).all()


def test_analytic_valu_kernel_fallback(monkeypatch):
    """GST_ANALYTIC_MFMA=0 (read when the plan is created) keeps the one-wavefront-per-circuit VALU kernel alive at D = 16:
    same vectors, same tolerance, and the two paths agree far below it."""
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    J_mfma = plan_from_fixture(fx).fill_dprobs(param_idx=fx["dprobs_cols"], mode=_lib.DERIV_ANALYTIC)
    monkeypatch.setenv("GST_ANALYTIC_MFMA", "0")
    J_valu = plan_from_fixture(fx).fill_dprobs(param_idx=fx["dprobs_cols"], mode=_lib.DERIV_ANALYTIC)
    rows = fx["matrix_rows"]
    assert np.abs(J_valu[rows] - fx["dprobs_matrix"]).max() < TOL
    assert np.abs(J_valu - J_mfma).max() < 1e-12
    assert not np.array_equal(J_valu, J_mfma)          # (different summation orders: they are different kernels)
