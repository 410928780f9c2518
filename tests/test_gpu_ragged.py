"""General (non-GST) inputs through the C ABI against the CPU oracle, bit for bit: several state preparations,
ragged effect lists (a different subset of effects per circuit, in any order), duplicate circuits, empty circuits,
shared and unshared prefixes, odd parameter subsets in odd orders, slot budgets 1..4 -- the shapes the reference's
layout can hand over (elbl_indices_by_expcircuit is per circuit, maplayout.py:117-126)."""
import numpy as np
import pytest

from conftest import assert_bitwise
from pygsti_amd import _lib

pytestmark = pytest.mark.gpu


def _random_case(D, seed, n_circ=150, nG=5, nR=3, nEl=6, max_len=40):
    rng = np.random.default_rng(seed)
    gates = np.eye(D)[None] * 0.9 + 0.1 * rng.standard_normal((nG, D, D))
    rhos = rng.standard_normal((nR, D))
    effects = rng.standard_normal((nEl, D))
    circs, rho = [], []
    for k in range(n_circ):
        r = rng.random()
        if r < 0.15 and circs:                       # duplicate of an earlier circuit (same or other rho)
            j = rng.integers(len(circs)); circs.append(circs[j].copy()); rho.append(rho[j] if rng.random() < .5 else rng.integers(nR))
        elif r < 0.55 and circs:                     # extension of an earlier circuit
            j = rng.integers(len(circs)); circs.append(np.concatenate([circs[j], rng.integers(0, nG, rng.integers(0, 6))])); rho.append(rho[j])
        else:
            circs.append(rng.integers(0, nG, rng.integers(0, max_len))); rho.append(rng.integers(nR))
    ptr = np.zeros(n_circ + 1, np.int64); ptr[1:] = np.cumsum([len(c) for c in circs])
    g = np.concatenate(circs).astype(np.int32) if ptr[-1] else np.zeros(0, np.int32)
    eff_ptr = [0]; eff_label = []
    for k in range(n_circ):
        m = rng.integers(0, nEl + 1)                 # possibly NO element at all for a circuit
        eff_label.extend(rng.permutation(nEl)[:m]); eff_ptr.append(len(eff_label))
    nE = len(eff_label)
    eff_dest = rng.permutation(nE).astype(np.int32)  # arbitrary element order
    nP = nR * D + nEl * D + nG * D * D
    kind = np.concatenate([np.full(nR * D, 1), np.full(nEl * D, 2), np.full(nG * D * D, 0)]).astype(np.int32)
    obj = np.concatenate([np.repeat(np.arange(nR), D), np.repeat(np.arange(nEl), D), np.repeat(np.arange(nG), D * D)]).astype(np.int32)
    elem = np.concatenate([np.tile(np.arange(D), nR), np.tile(np.arange(D), nEl), np.tile(np.arange(D * D), nG)]).astype(np.int32)
    perm = rng.permutation(nP)                       # parameters in scrambled order, some mapped to nothing
    kind, obj, elem = kind[perm], obj[perm], elem[perm]
    kind[rng.choice(nP, 5, replace=False)] = -1
    args = dict(D=D, nG=nG, nR=nR, nEl=nEl, nE=nE, rho=np.array(rho, np.int32), ptr=ptr, g=g,
                eff_ptr=np.array(eff_ptr, np.int64), eff_label=np.array(eff_label, np.int32), eff_dest=eff_dest)
    tbl = dict(D=D, nE=nE, cache_size=0, t_dest=np.arange(n_circ), t_start=-np.ones(n_circ), t_cache=-np.ones(n_circ),
               t_rho=args["rho"], row_ptr=ptr, gate_idx=g, eff_ptr=args["eff_ptr"], eff_label=args["eff_label"], eff_dest=eff_dest)
    mdl = dict(gates=gates, rhos=rhos, effects=effects, pkind=kind, pobj=obj, pelem=elem)
    return args, tbl, mdl, nP


@pytest.mark.parametrize("D,seed,max_slots,target_tasks", [(4, 1, 0, 0), (4, 2, 1, 5), (16, 3, 0, 0), (16, 4, 2, 3), (16, 5, 4, 64)])
def test_ragged_inputs_vs_oracle(oracle_built, D, seed, max_slots, target_tasks):
    a, tbl, mdl, nP = _random_case(D, seed)
    pl = _lib.Plan.from_circuits(D, a["nG"], a["nR"], a["nEl"], a["nE"], a["rho"], a["ptr"], a["g"], a["eff_ptr"],
                                 a["eff_label"], a["eff_dest"], max_slots=max_slots, target_tasks=target_tasks)
    pl.set_model(mdl["gates"], mdl["rhos"], mdl["effects"])
    pl.set_param_map(mdl["pkind"], mdl["pobj"], mdl["pelem"])
    orc = oracle_built.Oracle(tbl, mdl)
    assert_bitwise(pl.fill_probs(), orc.probs(), "probs")
    rng = np.random.default_rng(seed + 100)
    cols = rng.permutation(nP)[: min(nP, 200)]
    J = pl.fill_dprobs(param_idx=cols, eps=1e-7)
    assert_bitwise(J, orc.dprobs(cols, eps=1e-7), "dprobs")
    i1 = rng.permutation(nP)[:7]; i2 = rng.permutation(nP)[:70]
    i2[:3] = i1[:3]                                   # same parameter in both blocks
    H = pl.fill_hprobs(idx1=i1, idx2=i2, eps=1e-5)
    assert_bitwise(H, orc.hprobs(i1, i2, eps=1e-5), "hprobs")
    # analytic mode against a central-difference-free check: agrees with FD to O(eps) and is column-exact zero
    # where the oracle's FD columns are exactly zero
    Ja = pl.fill_dprobs(param_idx=cols, mode=_lib.DERIV_ANALYTIC)
    Jfd = orc.dprobs(cols, eps=1e-7)
    assert np.abs(Ja - Jfd).max() < 1e-4 * max(1.0, np.abs(Jfd).max())
    assert (Ja[:, mdl["pkind"][cols] == -1] == 0).all()
    # ... and against the exact numpy forward/backward Jacobian (six effects: two backward passes at D = 16; several
    # preparations; circuits without outcomes; duplicates)
    fx = dict(tbl); fx.update(mdl)
    Jo, _ = oracle_built.analytic_dprobs(fx, cols)
    assert np.abs(Ja - Jo).max() < 1e-8 * max(1.0, np.abs(Jo).max())


def test_degenerate_requests():
    a, tbl, mdl, nP = _random_case(16, 9, n_circ=20)
    pl = _lib.Plan.from_circuits(16, a["nG"], a["nR"], a["nEl"], a["nE"], a["rho"], a["ptr"], a["g"], a["eff_ptr"],
                                 a["eff_label"], a["eff_dest"])
    pl.set_model(mdl["gates"], mdl["rhos"], mdl["effects"])
    pl.set_param_map(mdl["pkind"], mdl["pobj"], mdl["pelem"])
    pr = np.empty(a["nE"])
    J = pl.fill_dprobs(param_idx=np.zeros(0, np.int64), probs_out=pr)      # zero columns: (nE, 0) and the probabilities
    assert J.shape == (a["nE"], 0)
    assert_bitwise(pr, pl.fill_probs(), "probs with zero columns")
    H = pl.fill_hprobs(idx1=np.zeros(0, np.int64), idx2=np.arange(3))
    assert H.shape == (a["nE"], 0, 3)
    with pytest.raises(ValueError):
        pl.fill_dprobs(param_idx=np.array([nP + 3]))
    J1 = pl.fill_dprobs(param_idx=np.array([5, 5, 5]))                       # the same parameter three times
    assert_bitwise(J1[:, 0], J1[:, 2], "repeated column")
