"""D = 64 (three qubits, BASELINE configs[4] shape: 10 dense 64x64 superoperators, 8 outcomes, 41,536
parameters): the row-per-lane kernel against the CPU oracle, bit for bit, on seeded random circuits."""
import numpy as np
import pytest

from conftest import assert_bitwise
from pygsti_amd import _lib

pytestmark = pytest.mark.gpu


def _make(n_circ=60, max_len=256, seed=0, max_slots=0):
    rng = np.random.default_rng(seed)
    D, nG, nEl = 64, 10, 8
    gates = np.eye(D)[None] + 0.04 * rng.standard_normal((nG, D, D))
    rhos = np.zeros((1, D)); rhos[0, 0] = 1.0 / np.sqrt(8); rhos[0] += 0.01 * rng.standard_normal(D)
    effects = 0.1 * rng.standard_normal((nEl, D)); effects[:, 0] += 1.0 / np.sqrt(8)
    lens = rng.integers(0, max_len + 1, n_circ)
    lens[:3] = [0, 1, max_len]
    circs = [rng.integers(0, nG, L) for L in lens]
    # some shared prefixes so that the trie / save slots are exercised
    for k in range(3, n_circ, 4):
        circs[k] = np.concatenate([circs[k - 1][:len(circs[k - 1]) // 2], circs[k][:8]])
    # a balanced binary tree of prefixes, 4 levels deep: walking it needs a save slot per level
    seg = [[np.concatenate([[2 * lvl + b], rng.integers(0, nG, 3)]) for b in (0, 1)] for lvl in range(4)]
    for leaf in range(16):
        circs[4 + leaf] = np.concatenate([seg[lvl][(leaf >> lvl) & 1] for lvl in range(4)])
    ptr = np.zeros(n_circ + 1, np.int64); ptr[1:] = np.cumsum([len(c) for c in circs])
    g = np.concatenate(circs).astype(np.int32)
    nE = n_circ * nEl
    eff_ptr = np.arange(n_circ + 1, dtype=np.int64) * nEl
    eff_label = np.tile(np.arange(nEl, dtype=np.int32), n_circ)
    eff_dest = np.arange(nE, dtype=np.int32)
    nP = D + nEl * D + nG * D * D
    kind = np.concatenate([np.full(D, 1), np.full(nEl * D, 2), np.full(nG * D * D, 0)]).astype(np.int32)
    obj = np.concatenate([np.zeros(D), np.repeat(np.arange(nEl), D), np.repeat(np.arange(nG), D * D)]).astype(np.int32)
    elem = np.concatenate([np.arange(D), np.tile(np.arange(D), nEl), np.tile(np.arange(D * D), nG)]).astype(np.int32)
    assert nP == 41536
    pl = _lib.Plan.from_circuits(D, nG, 1, nEl, nE, np.zeros(n_circ, np.int32), ptr, g, eff_ptr, eff_label, eff_dest,
                                 max_slots=max_slots)
    pl.set_model(gates, rhos, effects)
    pl.set_param_map(kind, obj, elem)
    tbl = dict(D=D, nE=nE, cache_size=0, t_dest=np.arange(n_circ), t_start=-np.ones(n_circ), t_cache=-np.ones(n_circ),
               t_rho=np.zeros(n_circ), row_ptr=ptr, gate_idx=g, eff_ptr=eff_ptr, eff_label=eff_label, eff_dest=eff_dest)
    mdl = dict(gates=gates, rhos=rhos, effects=effects, pkind=kind, pobj=obj, pelem=elem)
    return pl, tbl, mdl, nP


@pytest.mark.parametrize("max_slots", [0, 8])
def test_3q_probs_dprobs_hprobs_vs_oracle(oracle_built, max_slots):
    """max_slots=0 (default, <= 2 slots): the register-blocked walk_quad64_kernel; max_slots=8: plans that use more
    save slots run on walk_rows_shared_kernel (LDS slots).  Both bit-identical to the oracle."""
    pl, tbl, mdl, nP = _make(max_slots=max_slots)
    if max_slots == 8:
        assert pl.stats()["max_slots"] > 2, "the fixture must exercise the LDS-slot kernel"
    orc = oracle_built.Oracle(tbl, mdl)
    assert_bitwise(pl.fill_probs(), orc.probs(), "3Q probs")
    rng = np.random.default_rng(5)
    cols = np.sort(np.concatenate([[0, 63, 64, 575, 576, 577, 576 + 4095, 576 + 4096, nP - 1],
                                   rng.choice(nP, 40, replace=False)]))
    pr = np.empty(tbl["nE"])
    J = pl.fill_dprobs(param_idx=cols, eps=1e-7, probs_out=pr)
    Jo, po = orc.dprobs(cols, eps=1e-7, return_probs=True)
    assert_bitwise(pr, po, "3Q pr")
    assert_bitwise(J, Jo, "3Q dprobs")
    i1 = np.array([0, 70, 576, 600, 576 + 64, 9000]); i2 = np.array([0, 1, 70, 576, 577, 576 + 64 + 1, 30000])
    H = pl.fill_hprobs(idx1=i1, idx2=i2, eps=1e-5)
    assert_bitwise(H, orc.hprobs(i1, i2, eps=1e-5), "3Q hprobs")


def test_3q_plan_stats():
    pl, tbl, mdl, nP = _make(n_circ=40, max_len=64, seed=3)
    st = pl.stats()
    assert st["max_slots"] <= 8 and st["applies_per_pass"] <= st["sum_depth"]


def test_3q_analytic_dprobs_vs_numpy_oracle(oracle_built):
    """GST_DERIV_ANALYTIC at D = 64: backward states over the suffix trie + 64x64 MFMA blocks, against the numpy
    forward/backward Jacobian (the restatement pinned to MatrixForwardSimulator vectors at D = 4 and 16)."""
    pl, tbl, mdl, nP = _make(n_circ=40, max_len=96, seed=7)
    fx = dict(tbl); fx.update(mdl)
    rng = np.random.default_rng(11)
    cols = np.sort(np.concatenate([np.arange(0, 64), np.arange(64, 64 + 512), np.arange(576, 576 + 4096),
                                   576 + 4096 * 3 + rng.choice(4096, 300, replace=False), [nP - 1]]))
    pr = np.empty(tbl["nE"])
    J = pl.fill_dprobs(param_idx=cols, probs_out=pr, mode=_lib.DERIV_ANALYTIC)
    Jo, po = oracle_built.analytic_dprobs(fx, cols)
    scale = max(1.0, np.abs(Jo).max())
    assert np.abs(J - Jo).max() < 1e-8 * scale
    assert np.abs(pr - po).max() < 1e-10
    # the finite-difference Jacobian is close but not equal (truncation error of the FD step)
    Jfd = pl.fill_dprobs(param_idx=cols[:200], eps=1e-7)
    assert np.abs(Jfd - J[:, :200]).max() < 1e-4 * scale


def test_3q_analytic_hprobs_vs_numpy_oracle(oracle_built):
    """gst_fill_hprobs_analytic at D = 64 (dwalk64_kernel + the 64x64 MFMA contraction, two launches per row) against
    the exact numpy Hessian."""
    pl, tbl, mdl, nP = _make(n_circ=24, max_len=48, seed=9)
    fx = dict(tbl); fx.update(mdl)
    i1 = np.array([0, 63, 64 + 5, 64 + 64 * 3 + 7, 576, 576 + 65, 576 + 4096 * 4 + 130, nP - 1])
    i2 = np.concatenate([[1, 62, 64, 64 + 64 * 3 + 7, 575], 576 + np.arange(0, 40), 576 + 4096 * 4 + np.arange(120, 140), [nP - 1]])
    H = pl.fill_hprobs(idx1=i1, idx2=i2, mode=_lib.DERIV_ANALYTIC)
    Ho = oracle_built.analytic_hprobs(fx, i1, i2)
    scale = max(1.0, np.abs(Ho).max())
    assert np.abs(H - Ho).max() < 1e-8 * scale
    assert np.abs(Ho).max() > 1e-3


def test_3q_analytic_dprobs_with_register_resident_backward_walk(oracle_built, monkeypatch):
    """The opt-in backward walk of the exact Jacobian (chain64_resident_kernel: every gate's MFMA operands resident in
    registers, two tasks interleaved per workgroup; off by default because it is slower overall) gives the
    same Jacobian: against the numpy oracle (1e-8) and against the default walk (re-association only)."""
    pl0, tbl, mdl, nP = _make(n_circ=60, max_len=128, seed=13)
    fx = dict(tbl); fx.update(mdl)
    rng = np.random.default_rng(17)
    cols = np.sort(np.concatenate([np.arange(0, 64), np.arange(64, 64 + 512), 576 + 4096 * 2 + rng.choice(4096, 500, replace=False),
                                   576 + 4096 * 9 + rng.choice(4096, 300, replace=False), [nP - 1]]))
    J0 = pl0.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
    pl0.close()
    monkeypatch.setenv("GST_TEST_FORCE", "chain_resident=1")
    pl1, _, _, _ = _make(n_circ=60, max_len=128, seed=13)
    pr = np.empty(tbl["nE"])
    J1 = pl1.fill_dprobs(param_idx=cols, probs_out=pr, mode=_lib.DERIV_ANALYTIC)
    J1b = pl1.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)          # (the pop counter is reset per launch)
    Jo, po = oracle_built.analytic_dprobs(fx, cols)
    scale = max(1.0, np.abs(Jo).max())
    assert np.abs(J1 - Jo).max() < 1e-8 * scale
    assert np.abs(J1 - J0).max() < 1e-11 * scale
    assert np.array_equal(J1, J1b)
    assert np.abs(pr - po).max() < 1e-10 * max(1.0, np.abs(po).max())       # (seeded random gates: |p| grows with depth)
