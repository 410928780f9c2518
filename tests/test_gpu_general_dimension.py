"""D = 9 (the reference's qutrit model pack) on the device, zero-padded at 16 behind the C ABI: every fill of the Map path
bit for bit, exact derivatives <= 1e-8 against the Matrix simulator, the general routes (model sets, derivative matrices,
the fused LM step) through the same padding."""
import numpy as np
import pytest

from conftest import load_fixture, assert_bitwise, plan_from_fixture

pytestmark = pytest.mark.gpu
NAME = "qutrit_XYIMS_L8_depol"


def test_qutrit_fills_are_the_reference_map_simulators_bit_for_bit():
    from pygsti_amd import _lib
    fx = load_fixture(NAME)
    pl = plan_from_fixture(fx)
    nE, nP = int(fx["nE"]), int(fx["nP"])
    eps = float(fx["derivative_eps"])
    assert_bitwise(pl.fill_probs(), fx["probs"], "qutrit probs")
    cols = fx["dprobs_cols"]
    pr = np.empty(nE)
    J = pl.fill_dprobs(param_idx=cols, eps=eps, probs_out=pr)
    assert_bitwise(pr, fx["probs"], "qutrit probs beside the Jacobian")
    assert_bitwise(J, fx["dprobs_map"], "qutrit FD dprobs")
    Jall = pl.fill_dprobs(eps=eps)                                      # all 360 columns, every launch form's default
    assert_bitwise(Jall[:, cols], fx["dprobs_map"], "qutrit FD dprobs (all columns)")
    H = pl.fill_hprobs(idx1=fx["hprobs_rows"], idx2=fx["hprobs_cols"], eps=float(fx["hessian_eps"]))
    assert_bitwise(H, fx["hprobs_map"], "qutrit FD-of-FD hprobs")
    # exact derivatives vs the Matrix simulator (rows in the Matrix layout's order: matrix_rows)
    Ja = pl.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
    rows = fx["matrix_rows"]
    assert np.abs(Ja[rows] - fx["dprobs_matrix"]).max() < 1e-8
    assert np.abs(pl.fill_probs()[rows] - fx["probs_matrix"]).max() < 1e-10
    # device-resident fills into a wider array
    d = pl.device_malloc(nE * (nP + 3) * 8); dp = pl.device_malloc(nE * 8)
    pl.memcpy_h2d(d, np.full(nE * (nP + 3), np.nan))
    pl.fill_dprobs_dev(d, nP + 3, np.arange(nP), None, eps, dp, _lib.DERIV_FD); pl.sync()
    W = pl.memcpy_d2h(np.empty((nE, nP + 3)), d)
    assert np.array_equal(W[:, :nP], Jall) and np.isnan(W[:, nP:]).all()
    assert_bitwise(pl.memcpy_d2h(np.empty(nE), dp), fx["probs"], "device probs")
    for b in (d, dp):
        pl.device_free(b)


def test_qutrit_general_routes_through_the_padding(oracle_built):
    from pygsti_amd import _lib
    fx = load_fixture(NAME)
    pl = plan_from_fixture(fx)
    nE, nP, D = int(fx["nE"]), int(fx["nP"]), 9
    eps = float(fx["derivative_eps"])
    cols = fx["dprobs_cols"][::4]
    # host-stepped dense model sets (9 x 9 gates per set)
    G = np.repeat(fx["gates"][None], len(cols), 0); R = np.repeat(fx["rhos"][None], len(cols), 0); E = np.repeat(fx["effects"][None], len(cols), 0)
    for c, q in enumerate(cols):
        k, o, e = int(fx["pkind"][q]), int(fx["pobj"][q]), int(fx["pelem"][q])
        tgt = (G, R, E)[k]
        tgt[c, o].flat[e] = tgt[c, o].flat[e] + eps
    Jm = pl.fill_dprobs_models(G, R, E, eps=eps)
    ref = pl.fill_dprobs(param_idx=cols, eps=eps)
    assert_bitwise(Jm, ref, "qutrit model sets vs the element-map walk")
    # derivative matrices (the chain-rule route) with the identity parameterisation = the exact element route
    objs = []
    for k, n_obj, n_el in ((1, 1, D), (2, len(fx["effects"]), D), (0, len(fx["gates"]), D * D)):
        for o in range(n_obj):
            sel = np.nonzero((fx["pkind"] == k) & (fx["pobj"] == o))[0]
            dm = np.zeros((n_el, len(sel))); dm[fx["pelem"][sel], np.arange(len(sel))] = 1.0
            objs.append((k, o, sel, dm))
    Ja = pl.fill_dprobs(param_idx=np.arange(nP), mode=_lib.DERIV_ANALYTIC)
    pl2 = plan_from_fixture(fx)
    pl2.set_derivs(nP, objs)
    Jd = pl2.fill_dprobs(param_idx=np.arange(nP), mode=_lib.DERIV_ANALYTIC)
    assert np.abs(Jd - Ja).max() < 1e-12
    # the fused LM step
    rng = np.random.default_rng(1)
    N = np.full(nE, 1000.0); cnt = rng.multinomial(1000, np.full(3, 1 / 3), size=nE // 3).astype(float).ravel()
    total, jtj, jtf = pl.lsq_step(nP, cnt, N, "chi2", eps, _lib.DERIV_FD)
    from oracle import objective_oracle as OO
    Jf = pl.fill_dprobs(eps=eps)
    t, ls, _, rs = OO.objective_rows(OO.CHI2, fx["probs"], cnt, N, 1e-4, 1e-4)
    Js = Jf * rs[:, None]
    assert np.abs(jtj - Js.T @ Js).max() <= 1e-11 * np.abs(jtj).max() and abs(total - t.sum()) <= 1e-11 * t.sum()
