"""Quick timing of the 3-qubit exact Jacobian (development aid): step, contraction kernel, chain passes."""
import sys, os, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsti_amd import _lib

n_circ, max_len = 4000, 256
rng = np.random.default_rng(0)
D, nG, nEl = 64, 10, 8
gates = np.eye(D)[None] + 0.04 * rng.standard_normal((nG, D, D))
rhos = np.zeros((1, D)); rhos[0, 0] = 1.0 / np.sqrt(8)
effects = 0.1 * rng.standard_normal((nEl, D)); effects[:, 0] += 1.0 / np.sqrt(8)
circs = [rng.integers(0, nG, L) for L in rng.integers(1, max_len + 1, n_circ)]
ptr = np.zeros(n_circ + 1, np.int64); ptr[1:] = np.cumsum([len(c) for c in circs])
g = np.concatenate(circs).astype(np.int32)
nE = n_circ * nEl
nP = D + nEl * D + nG * D * D
kind = np.concatenate([np.full(D, 1), np.full(nEl * D, 2), np.full(nG * D * D, 0)]).astype(np.int32)
obj = np.concatenate([np.zeros(D), np.repeat(np.arange(nEl), D), np.repeat(np.arange(nG), D * D)]).astype(np.int32)
elem = np.concatenate([np.arange(D), np.tile(np.arange(D), nEl), np.tile(np.arange(D * D), nG)]).astype(np.int32)
plan = _lib.Plan.from_circuits(D, nG, 1, nEl, nE, np.zeros(n_circ, np.int32), ptr, g, np.arange(n_circ + 1, dtype=np.int64) * nEl,
                               np.tile(np.arange(nEl, dtype=np.int32), n_circ), np.arange(nE, dtype=np.int32), timing=1)
plan.set_model(gates, rhos, effects); plan.set_param_map(kind, obj, elem)
d_p = plan.device_malloc(nE * 8)
allc = np.arange(nP, dtype=np.int64)
d_J = plan.device_malloc(nE * nP * 8)
out = {}
for rep in range(2):
    plan.fill_dprobs_dev(d_J, nP, allc, None, 1e-7, d_p, _lib.DERIV_ANALYTIC); plan.sync()
ts, ks = [], []
for rep in range(4):
    t0 = time.perf_counter()
    plan.fill_dprobs_dev(d_J, nP, allc, None, 1e-7, d_p, _lib.DERIV_ANALYTIC); plan.sync()
    ts.append(time.perf_counter() - t0)
    st = plan.stats(); ks.append((st["last_kernel_ms"], st["last_total_ms"]))
out["step_ms"] = 1e3 * min(ts); out["kernel_total_ms"] = ks
out["GBps"] = 8.0 * nE * nP / min(ts) / 1e9
if len(sys.argv) > 1:          # a cheap checksum against the previous build
    J = plan.memcpy_d2h(np.empty((64, nP)), d_J)
    out["checksum"] = float(np.abs(J).sum())
print(json.dumps(out))
