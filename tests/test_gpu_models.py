"""Finite differences over general parameterisations (gst_fill_dprobs_models): the caller steps its model on the host
(set_parameter_value + to_dense, exactly the loop of mapfill_dprobs_atom, mapforwardsim_calc_densitymx.pyx:362-381) and
the device evaluates every perturbed dense model.  Fixtures: CPTPLND models (static target composed with an exponentiated
Lindblad error generator) of smq1Q_XYI and smq2Q_XYICNOT with the reference Map simulator's own FD Jacobian and the
dense model after every step.

Two bars:
  * vs the CPU oracle fed the same dense model sets: BIT FOR BIT (same arithmetic order);
  * vs the reference's vectors: to rounding.  The reference propagates a composed member factor by factor
    (opcreps.cpp:242-524), the device through its dense product, so probabilities differ by ~1e-16 and the quotients by
    ~1e-16 / eps = 1e-9 .. 1e-8: the tolerance is written below with the observed figures."""
import numpy as np
import pytest

from conftest import load_fixture, assert_bitwise, plan_from_fixture

pytestmark = pytest.mark.gpu

# observed (oracle == device bit for bit): probs 2.8e-16 / 3.3e-16, dprobs 4.4e-9 / 6.1e-9 (1Q / 2Q) -- inside the
# north star's "probabilities <= 1e-10, dprobs <= 1e-8"
CASES = [("smq1Q_XYI_L4_CPTPLND", 1e-15, 1e-8), ("smq2Q_XYICNOT_L1_CPTPLND", 1e-15, 1e-8)]


def _oracle_fd(O, fx, eps):
    orc = O.from_fixture({k: np.array(v) for k, v in fx.items()})       # (set_model writes into the oracle's arrays)
    base = orc.probs()
    cols = []
    for g, r, e in zip(fx["mm_gates"], fx["mm_rhos"], fx["mm_effects"]):
        orc.set_model(g, r, e)
        cols.append((orc.probs() - base) / eps)
    return np.array(cols).T, base


@pytest.mark.parametrize("name,ptol,jtol", CASES)
def test_models_fd_vs_oracle_bitwise_and_reference_to_rounding(oracle_built, name, ptol, jtol):
    fx = load_fixture(name)
    eps = float(fx["derivative_eps"])
    pl = plan_from_fixture(fx)                      # (the fixture's parameter map is all "none": not used by this path)
    pr = np.empty(int(fx["nE"]))
    J = pl.fill_dprobs_models(fx["mm_gates"], fx["mm_rhos"], fx["mm_effects"], eps=eps, probs_out=pr)
    Jo, po = _oracle_fd(oracle_built, fx, eps)
    assert_bitwise(pr, po, "base probabilities vs oracle")
    assert_bitwise(J, Jo, "FD columns of the model sets vs oracle")
    dp = np.abs(pr - fx["probs"]).max(); dj = np.abs(J - fx["dprobs_map"]).max()
    print("%s: max|probs - reference| = %.2e, max|dprobs - reference| = %.2e (max|J| = %.2f)" % (name, dp, dj, np.abs(J).max()))
    assert dp <= ptol and dj <= jtol


def test_models_fd_destination_window_and_chunks(oracle_built, monkeypatch):
    fx = load_fixture("smq1Q_XYI_L4_CPTPLND")
    eps = float(fx["derivative_eps"])
    pl = plan_from_fixture(fx)
    Jo, _ = _oracle_fd(oracle_built, fx, eps)
    n = len(fx["mm_gates"])
    wide = np.full((int(fx["nE"]), n + 9), -3.0)
    pl.fill_dprobs_models(fx["mm_gates"][10:30], fx["mm_rhos"][10:30], fx["mm_effects"][10:30], out=wide,
                          dest_idx=np.arange(14, 34), eps=eps)
    assert_bitwise(wide[:, 14:34], Jo[:, 10:30], "window")
    assert (wide[:, :14] == -3.0).all() and (wide[:, 34:] == -3.0).all()
    scat = np.full((int(fx["nE"]), n), -3.0)
    dest = np.array([5, 0, 59, 17])
    pl.fill_dprobs_models(fx["mm_gates"][:4], fx["mm_rhos"][:4], fx["mm_effects"][:4], out=scat, dest_idx=dest, eps=eps)
    assert_bitwise(scat[:, dest], Jo[:, :4], "scattered destination columns")
    # no model sets: only the probabilities
    pr = np.empty(int(fx["nE"]))
    pl.fill_dprobs_models(np.zeros((0, 3, 4, 4)), np.zeros((0, 1, 4)), np.zeros((0, 2, 4)), out=np.empty((int(fx["nE"]), 0)),
                          eps=eps, probs_out=pr)
    assert_bitwise(pr, oracle_built.from_fixture(fx).probs(), "probs only")


def test_models_fd_equals_element_fd_on_a_full_model(oracle_built):
    """On a fully parameterised model the general path and the lane-per-model FD kernel must give the same bits: the
    perturbed model of parameter i is the base with one element + eps."""
    fx = load_fixture("smq2Q_XYICNOT_L2_depol")
    pl = plan_from_fixture(fx)
    cols = fx["dprobs_cols"][::3]
    eps = 1e-7
    G, R, E = [], [], []
    for c in cols:
        g, r, e = fx["gates"].copy(), fx["rhos"].copy(), fx["effects"].copy()
        k, o, el = int(fx["pkind"][c]), int(fx["pobj"][c]), int(fx["pelem"][c])
        tgt = (g if k == 0 else r if k == 1 else e)[o].reshape(-1)
        tgt[el] = tgt[el] + eps
        G.append(g); R.append(r); E.append(e)
    J = pl.fill_dprobs_models(np.array(G), np.array(R), np.array(E), eps=eps)
    assert_bitwise(J, fx["dprobs_map"][:, ::3], "general path == reference Map FD on a full model")


def test_models_fd_of_fd_hessian_block(oracle_built):
    """FD-of-FD Hessian block of the CPTPLND model from two-level model sets (fixture `mm2_*`: the dense model at
    theta + eps e_i and after every column step from there): the device composition equals the CPU oracle's bit for bit,
    and the reference's Map vectors to rounding / eps^2 (1e-16 / 1e-10: the bound is 2e-5 absolute; observed 6.7e-6 at max|H| = 2.0)."""
    fx = load_fixture("smq1Q_XYI_L4_CPTPLND")
    eps = float(fx["hessian_eps"])
    pl = plan_from_fixture(fx)
    H = pl.fill_hprobs_models(fx["mm2_gates"], fx["mm2_rhos"], fx["mm2_effects"], eps=eps)
    # the same composition on the CPU oracle
    O = oracle_built
    orc = O.from_fixture({k: np.array(v) for k, v in fx.items()})
    d = []
    for k in range(fx["mm2_gates"].shape[0]):
        orc.set_model(fx["mm2_gates"][k, 0], fx["mm2_rhos"][k, 0], fx["mm2_effects"][k, 0]); base = orc.probs()
        cols = []
        for c in range(1, fx["mm2_gates"].shape[1]):
            orc.set_model(fx["mm2_gates"][k, c], fx["mm2_rhos"][k, c], fx["mm2_effects"][k, c])
            cols.append((orc.probs() - base) / eps)
        d.append(np.array(cols).T)
    Ho = np.stack([(d[k] - d[0]) / eps for k in range(1, len(d))], axis=1)
    assert_bitwise(H, Ho, "FD-of-FD Hessian from model sets vs oracle")
    rows = [list(fx["hprobs_rows"]).index(r) for r in fx["mm2_rows"]]
    cols = [list(fx["hprobs_cols"]).index(c) for c in fx["mm2_cols"]]
    ref = fx["hprobs_map"][:, rows][:, :, cols]
    err = np.abs(H - ref).max()
    print("max|hprobs - reference| = %.2e (max|H| = %.2f)" % (err, np.abs(ref).max()))
    assert err <= 2e-5
