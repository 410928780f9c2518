"""Timing of the probability fill and of the analytic Jacobian with and without the log-depth level passes (bench design)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsti_amd import modelpacks as MP, _lib
from pygsti_amd.layout import HipCOPALayout

pack = MP.smq2Q_XYICNOT
model = pack.target_model().depolarize(0.01, 0.01)
design = sys.argv[1] if len(sys.argv) > 1 else "full"
layout = HipCOPALayout(pack.create_gst_circuits(1024, lite=(design == "lite")), model, num_atoms=1, devices=[0], rank=0, size=1)
plan = layout.atoms[0].plan()
G, R, E = layout.model_arrays(model)
plan.set_model(G, R, E); plan.set_param_map(*layout.param_map(model))
nE, nP = layout.num_elements, model.num_params
d_p = plan.device_malloc(nE * 8); d_J = plan.device_malloc(nE * nP * 8)
pidx = np.arange(nP, dtype=np.int64)


def timed(fn, n=10):
    fn(); plan.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    plan.sync()
    return 1e3 * (time.perf_counter() - t0) / n


for fast in (0, 1):
    plan.set_option(_lib.OPT_FAST_PROBS, fast)
    ms = timed(lambda: (plan.set_model(G, R, E), plan.fill_probs_dev(d_p)))
    print("probs fast=%d: %.3f ms  (last_levels=%d kernel_ms=%.3f)" % (fast, ms, plan.stats()["last_levels"], plan.stats()["last_kernel_ms"]))
for fc in (0, 1):
    plan.set_option(_lib.OPT_FAST_CHAINS, fc)
    ms = timed(lambda: (plan.set_model(G, R, E), plan.fill_dprobs_dev(d_J, nP, pidx, None, 1e-7, d_p, _lib.DERIV_ANALYTIC)), 5)
    print("analytic fast_chains=%d: %.3f ms  (last_levels=%d kernel_ms=%.3f)" % (fc, ms, plan.stats()["last_levels"], plan.stats()["last_kernel_ms"]))
lp = plan.level_program(0)
print({k: v for k, v in lp.items() if not hasattr(v, "shape")})
