"""Development: the 3Q exact Jacobian (tools/bench_configs.py three_q workload) as K atoms on ONE GPU whose fills are issued without
waiting for each other -- every plan has streams of its own, so the chain passes of one atom run beside the contraction of another."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsti_amd import _lib

rng = np.random.default_rng(0)
D, nG, nEl, n_circ, max_len = 64, 10, 8, 4000, 256
gates = np.eye(D)[None] + 0.04 * rng.standard_normal((nG, D, D))
rhos = np.zeros((1, D)); rhos[0, 0] = 1.0 / np.sqrt(8)
effects = 0.1 * rng.standard_normal((nEl, D)); effects[:, 0] += 1.0 / np.sqrt(8)
circs = [rng.integers(0, nG, L) for L in rng.integers(1, max_len + 1, n_circ)]
nP = D + nEl * D + nG * D * D
kind = np.concatenate([np.full(D, 1), np.full(nEl * D, 2), np.full(nG * D * D, 0)]).astype(np.int32)
obj = np.concatenate([np.zeros(D), np.repeat(np.arange(nEl), D), np.repeat(np.arange(nG), D * D)]).astype(np.int32)
elem = np.concatenate([np.arange(D), np.tile(np.arange(D), nEl), np.tile(np.arange(D * D), nG)]).astype(np.int32)
allc = np.arange(nP, dtype=np.int64)
order = sorted(range(n_circ), key=lambda i: tuple(circs[i]))          # prefix order: atoms = contiguous runs (what a layout deals)
out = {}
for K in (1, 2, 4, 8):
    plans, bufs = [], []
    for a in range(K):
        sel = order[a * n_circ // K:(a + 1) * n_circ // K]
        cs = [circs[i] for i in sel]
        n = len(cs); nE = n * nEl
        ptr = np.zeros(n + 1, np.int64); ptr[1:] = np.cumsum([len(c) for c in cs])
        pl = _lib.Plan.from_circuits(D, nG, 1, nEl, nE, np.zeros(n, np.int32), ptr, np.concatenate(cs).astype(np.int32), np.arange(n + 1, dtype=np.int64) * nEl,
                                     np.tile(np.arange(nEl, dtype=np.int32), n), np.arange(nE, dtype=np.int32))
        pl.set_model(gates, rhos, effects); pl.set_param_map(kind, obj, elem)
        plans.append(pl); bufs.append((pl.device_malloc(nE * nP * 8), pl.device_malloc(nE * 8)))
    def fill():
        for pl, (dJ, dp) in zip(plans, bufs):
            pl.fill_dprobs_dev(dJ, nP, allc, None, 1e-7, dp, _lib.DERIV_ANALYTIC)
    def sync():
        for pl in plans: pl.sync()
    fill(); sync(); fill(); sync()
    t0 = time.perf_counter()
    for _ in range(5): fill()
    sync()
    out[K] = 1e3 * (time.perf_counter() - t0) / 5
    for pl, (dJ, dp) in zip(plans, bufs):
        pl.device_free(dJ); pl.device_free(dp); pl.close()
print(json.dumps({"exact_jacobian_ms_by_atoms_on_one_gpu": out, "GBps": {k: 8.0 * n_circ * nEl * nP / (v * 1e-3) / 1e9 for k, v in out.items()}}))
