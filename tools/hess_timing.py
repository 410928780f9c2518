"""development aid: FD-of-FD Hessian block timing on the 2Q design (kernel time from HIP events).
   python tools/hess_timing.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsti_amd import modelpacks
from pygsti_amd.layout import HipCOPALayout

pack = modelpacks.smq2Q_XYICNOT
model = pack.target_model().depolarize(op_noise=0.01, spam_noise=0.01)
circuits = pack.create_gst_circuits(256, lite=True)
layout = HipCOPALayout(circuits, model, num_atoms=1, devices=[0], rank=0, size=1)
plan = layout.atoms[0].plan()
plan.set_model(*layout.model_arrays(model)); plan.set_param_map(*layout.param_map(model))
nE = layout.num_elements
st = plan.stats()
print("design: %d circuits, nE=%d, applies/pass %d, tasks %d" % (len(circuits), nE, st["applies_per_pass"], st["n_tasks"]))
rng = np.random.default_rng(0)
pb = plan.fill_probs()
d_c = plan.device_malloc(nE * 8); d_N = plan.device_malloc(nE * 8)
plan.memcpy_h2d(d_c, rng.binomial(1000, np.clip(pb, 0, 1)).astype(np.float64)); plan.memcpy_h2d(d_N, np.full(nE, 1000.0))
for n1, n2 in ((16, 1616),):
    i1 = np.arange(80, 80 + n1); i2 = np.arange(0, n2)
    plan.objective_hessian_block("logl", d_c, d_N, i1, i2)
    t0 = time.perf_counter()
    Hb = plan.objective_hessian_block("logl", d_c, d_N, i1, i2)
    t = time.perf_counter() - t0
    print("objective Hessian block %d x %d contracted on the device: %.1f ms in all (%d numbers come back)" % (n1, n2, 1e3 * t, n1 * n2))
    from pygsti_amd import _lib
    plan.objective_hessian_block("logl", d_c, d_N, i1, i2, mode=_lib.DERIV_ANALYTIC)
    t0 = time.perf_counter()
    Ha = plan.objective_hessian_block("logl", d_c, d_N, i1, i2, mode=_lib.DERIV_ANALYTIC)
    t = time.perf_counter() - t0
    print("   same block with EXACT hprobs / dprobs (analytic mode): %.1f ms; max |FD-of-FD - exact| / max|exact| = %.2e" % (
        1e3 * t, np.abs(Hb - Ha).max() / np.abs(Ha).max()))
for n1, n2 in ((4, 256), (8, 512), (16, 1616)):
    i1 = np.arange(80, 80 + n1); i2 = np.arange(0, n2) if n2 == 1616 else np.arange(80, 80 + n2)
    H = plan.fill_hprobs(idx1=i1, idx2=i2, eps=1e-5)
    t0 = time.perf_counter()
    H = plan.fill_hprobs(idx1=i1, idx2=i2, eps=1e-5)
    t = time.perf_counter() - t0
    s = plan.stats()
    passes = n1 * n2
    flops = passes * 2.0 * 256 * st["applies_per_pass"]
    print("block %3d x %4d: total %.1f ms (incl. D2H of %.2f GB), S=2 kernel %.2f ms -> %.3g Hessian-el/s, %.1f TFLOP/s algorithmic" % (
        n1, n2, 1e3 * t, nE * n1 * n2 * 8 / 1e9, s["last_kernel_ms"], nE * n1 * n2 / (s["last_kernel_ms"] * 1e-3), flops / (s["last_kernel_ms"] * 1e-3) / 1e12))
