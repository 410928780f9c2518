"""Log-depth level passes on hardware (gst_kernels_levels.hip, GST_OPT_FAST_CHAINS / GST_OPT_FAST_PROBS): the modes WITHOUT
an ordering contract take the trie states from matrix squaring + doubling over the germ-power paths instead of the
sequential walk.  Bars (north star): probabilities <= 1e-10 against the reference's, exact derivatives <= 1e-8
against MatrixForwardSimulator's; against the sequential walk's own states the difference is re-association only.  The
finite-difference mode never uses the level pass and stays bit-identical."""
import numpy as np
import pytest

from conftest import load_fixture, assert_bitwise, plan_from_fixture
from pygsti_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["smq2Q_XYICNOT_L1024_deep", "smq2Q_XYICNOT_L2_depol"])
def test_fast_probs_within_1e10_of_the_reference(name):
    fx = load_fixture(name)
    pl = plan_from_fixture(fx)
    exact = pl.fill_probs()
    assert_bitwise(exact, fx["probs"], "default probabilities stay bit-identical")
    assert pl.stats()["last_levels"] == 0
    pl.set_option(_lib.OPT_FAST_PROBS, 1)
    pl.set_option(_lib.OPT_FAST_CHAINS, 2)            # (the L <= 2 design has nothing periodic: forced, for coverage)
    fast = pl.fill_probs()
    assert pl.stats()["last_levels"] == 1
    err = np.abs(fast - fx["probs"]).max()
    assert err < 1e-10, err
    assert err < 1e-12, err                           # (observed: a few 1e-15)
    # device-resident variant, and the FD mode is untouched by either option
    nE = int(fx["nE"])
    d = pl.device_malloc(nE * 8)
    pl.fill_probs_dev(d); pl.sync()
    assert np.array_equal(pl.memcpy_d2h(np.empty(nE), d), fast)
    pl.device_free(d)
    cols = fx["dprobs_cols"][:40]
    pr = np.empty(nE)
    J = pl.fill_dprobs(param_idx=cols, eps=float(fx["derivative_eps"]), probs_out=pr)
    assert_bitwise(J, fx["dprobs_map"][:, :40], "FD columns with the fast options set")
    assert_bitwise(pr, fx["probs"], "FD base probabilities with the fast options set")
    pl.set_option(_lib.OPT_FAST_PROBS, 0)
    assert_bitwise(pl.fill_probs(), fx["probs"], "option off again")


def test_analytic_jacobian_through_level_passes_deep_families(oracle_built):
    """Depth-1,030 germ-power families: the default analytic fill now takes BOTH chain passes from the level programs
    (worthwhile there); against the numpy analytic oracle <= 1e-8 absolute, against the sequential-walk form of the same
    fill <= 1e-10 (|J| reaches 65)."""
    fx = load_fixture("smq2Q_XYICNOT_L1024_deep")
    cols = np.concatenate([np.arange(0, 100), np.arange(336, 400), np.arange(1100, 1130), np.arange(1360, 1616)])
    pl = plan_from_fixture(fx)
    pr = np.empty(int(fx["nE"]))
    J = pl.fill_dprobs(param_idx=cols, probs_out=pr, mode=_lib.DERIV_ANALYTIC)
    assert pl.stats()["last_levels"] == 1
    Jo, po = oracle_built.analytic_dprobs(fx, cols)
    assert np.abs(J - Jo).max() < 1e-8, np.abs(J - Jo).max()
    assert np.abs(pr - fx["probs"]).max() < 1e-10
    seq = plan_from_fixture(fx)
    seq.set_option(_lib.OPT_FAST_CHAINS, 0)
    pr2 = np.empty(int(fx["nE"]))
    J2 = seq.fill_dprobs(param_idx=cols, probs_out=pr2, mode=_lib.DERIV_ANALYTIC)
    assert seq.stats()["last_levels"] == 0
    assert_bitwise(pr2, fx["probs"], "sequential analytic fill: the base pass's probabilities")
    assert np.abs(J - J2).max() < 1e-10, np.abs(J - J2).max()
    # switching the option on a plan whose reversed plan already exists (its level program is built with it, whatever the
    # option said at that moment: a late switch once read a state graph that had been dropped)
    seq.set_option(_lib.OPT_FAST_CHAINS, 1)
    J3 = seq.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
    assert seq.stats()["last_levels"] == 1
    assert np.array_equal(J3, J)


def test_forced_level_passes_on_a_design_fixture_jacobian_and_hessian():
    """GST_OPT_FAST_CHAINS = 2 on the L <= 2 design (no periodic path: every state comes from level-by-level tiles): the
    exact Jacobian against the Matrix simulator's vectors, and an exact Hessian block -- whose derivative walks read BOTH
    caches the level passes filled -- against the Matrix simulator's block."""
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    pl = plan_from_fixture(fx)
    pl.set_option(_lib.OPT_FAST_CHAINS, 2)
    J = pl.fill_dprobs(param_idx=fx["dprobs_cols"], mode=_lib.DERIV_ANALYTIC)
    assert pl.stats()["last_levels"] == 1
    rows = fx["matrix_rows"]
    assert np.abs(J[rows] - fx["dprobs_matrix"]).max() < 1e-8
    H = pl.fill_hprobs(idx1=fx["mh0_idx1"], idx2=fx["mh0_idx2"], mode=_lib.DERIV_ANALYTIC)
    assert np.abs(H[rows] - fx["mh0_hprobs"]).max() < 1e-8
    base = plan_from_fixture(fx)
    base.set_option(_lib.OPT_FAST_CHAINS, 0)
    J0 = base.fill_dprobs(param_idx=fx["dprobs_cols"], mode=_lib.DERIV_ANALYTIC)
    assert np.abs(J - J0).max() < 1e-12


def test_level_passes_on_the_bench_design():
    """The benchmarked workload (136,275 circuits): fast probabilities against the bit-exact ones <= 1e-10 on all 545,100
    rows; analytic Jacobian checksums (J^T f, reduced on the device) with and without the level passes agree to 1e-10 of
    their scale."""
    from pygsti_amd import modelpacks as MP
    from pygsti_amd.layout import HipCOPALayout
    pack = MP.smq2Q_XYICNOT
    model = pack.target_model().depolarize(0.01, 0.01)
    layout = HipCOPALayout(pack.create_gst_circuits(1024, lite=False), model, num_atoms=1, devices=[0], rank=0, size=1)
    plan = layout.atoms[0].plan()
    plan.set_model(*layout.model_arrays(model)); plan.set_param_map(*layout.param_map(model))
    nE, nP = layout.num_elements, model.num_params
    exact = plan.fill_probs()
    plan.set_option(_lib.OPT_FAST_PROBS, 1)
    fast = plan.fill_probs()
    assert plan.stats()["last_levels"] == 1
    assert np.abs(fast - exact).max() < 1e-10, np.abs(fast - exact).max()
    pidx = np.arange(nP, dtype=np.int64)
    f = np.random.default_rng(3).standard_normal(nE)
    bufs = [plan.device_malloc(n) for n in (nE * nP * 8, nE * 8, nE * 8, nP * 8)]
    d_J, d_p, d_f, d_y = bufs
    try:
        plan.memcpy_h2d(d_f, f)
        ys = {}
        for fc in (1, 0):
            plan.set_option(_lib.OPT_FAST_CHAINS, fc)
            plan.fill_dprobs_dev(d_J, nP, pidx, None, 1e-7, d_p, _lib.DERIV_ANALYTIC)
            plan.sync()
            assert plan.stats()["last_levels"] == fc
            plan.fill_jtf_dev(d_J, nE, nP, nP, d_f, d_y)
            ys[fc] = plan.memcpy_d2h(np.empty(nP), d_y)
        assert np.abs(ys[1] - ys[0]).max() <= 1e-10 * np.abs(ys[0]).max(), np.abs(ys[1] - ys[0]).max()
    finally:
        for b in bufs:
            plan.device_free(b)


def test_three_qubit_walk_on_the_matrix_cores():
    """D = 64 (gst_kernels_chain64.hip): the modes without an ordering contract walk a task as one [start vectors][64] row
    block on the matrix cores instead of a wavefront per (task, start vector).  GST_OPT_FAST_PROBS probabilities <= 1e-10 from
    the reference's (the default fill stays bit-identical); exact Jacobians within re-association of the SAME plan's exact
    Jacobian through the sequential row kernel (GST_OPT_FAST_CHAINS = 0), which the Matrix fixtures pin
    (test_gpu_analytic.py); tasks with save slots and depth-256 circuits included (3q_explicit_L64)."""
    fx = load_fixture("3q_explicit_L64")
    pl = plan_from_fixture(fx)
    nE = int(fx["nE"])
    exact = pl.fill_probs()
    assert_bitwise(exact, fx["probs"], "default 3Q probabilities stay bit-identical")
    assert pl.stats()["last_levels"] == 0
    pl.set_option(_lib.OPT_FAST_PROBS, 1)
    fast = pl.fill_probs()
    assert pl.stats()["last_levels"] == 1
    err = np.abs(fast - fx["probs"]).max()
    assert 0 < err < 1e-10 or err == 0.0, err
    assert err < 1e-13, err
    pl.set_option(_lib.OPT_FAST_PROBS, 0)
    # exact Jacobian columns spread over the state preparation, the effects and every gate
    rng = np.random.default_rng(5)
    cols = np.sort(rng.choice(int(fx["nP"]), 300, replace=False))
    pr1 = np.empty(nE); pr0 = np.empty(nE)
    J1 = pl.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC, probs_out=pr1)
    assert pl.stats()["last_levels"] == 1
    pl.set_option(_lib.OPT_FAST_CHAINS, 0)
    J0 = pl.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC, probs_out=pr0)
    assert pl.stats()["last_levels"] == 0
    assert_bitwise(pr0, fx["probs"], "sequential chain passes: probabilities of an exact fill are the reference's bits")
    assert np.abs(pr1 - pr0).max() < 1e-13
    scale = max(1.0, np.abs(J0).max())
    assert np.abs(J1 - J0).max() < 1e-11 * scale, np.abs(J1 - J0).max()
    assert np.abs(J0).max() > 1e-3
    # ... and against the finite-difference columns of the reference at the usual FD accuracy
    fd_cols = fx["dprobs_cols"][:64]
    pl.set_option(_lib.OPT_FAST_CHAINS, 1)
    Ja = pl.fill_dprobs(param_idx=fd_cols, mode=_lib.DERIV_ANALYTIC)
    assert np.abs(Ja - fx["dprobs_map"][:, :64]).max() < 5e-5
    # FD mode untouched
    J = pl.fill_dprobs(param_idx=fd_cols, eps=float(fx["derivative_eps"]))
    assert_bitwise(J, fx["dprobs_map"][:, :64], "3Q FD columns with the matrix-core walk available")
