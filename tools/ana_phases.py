"""Development aid: where a wavefront of analytic_mfma_kernel (two-circuit items, block stream) spends its cycles on the
bench design.  Needs the timing build of gst_kernels_analytic.hip (-DGST_ANA_TIMING=1, linked as
tools/bin/libgstfwd_anatiming.so: `make -C pygsti_amd/csrc ana-timing`)."""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GST_LIBGSTFWD"] = os.path.join(ROOT, "tools", "bin", "libgstfwd_anatiming.so")
sys.path.insert(0, ROOT)
import bench
from pygsti_amd import _lib
pack, model, circuits, layout = bench.build_workload("full", 1024, 1, 0, 0, 0, 0, "strong")
plan = layout.atoms[0].plan()
plan.set_model(*layout.model_arrays(model)); plan.set_param_map(*layout.param_map(model))
nE, nP = layout.num_elements, model.num_params
d = plan.device_malloc(nE * nP * 8, tracked=True); dp = plan.device_malloc(nE * 8)
pidx = np.arange(nP, dtype=np.int64)
L = _lib.lib()
L.gst_debug_ana_phases.argtypes = [C.c_void_p, C.c_int]
for rep in range(3):
    L.gst_debug_ana_phases(None, 1)
    t0 = time.perf_counter()
    plan.fill_dprobs_dev(d, nP, pidx, None, 1e-7, dp, _lib.DERIV_ANALYTIC); plan.sync()
    dt = time.perf_counter() - t0
    out = (C.c_ulonglong * 16)(); L.gst_debug_ana_phases(out, 0)
    v = np.array(list(out), float)
st = plan.stats()
print("fill %.2f ms (kernel %.2f ms, zeros resident %d); items %d, stream blocks %d (%.1f per item), flushes %d (%.1f per item)" % (
    dt * 1e3, st.get("last_kernel_ms", 0.0), st["last_zeros_resident"], v[9], v[7], v[7] / v[9], v[8], v[8] / v[9]))
tot = v[5]
names = ["item fetch (group fetch, two barriers)", "outcome tables + SPAM columns", "block stream incl. flushes", "  of which flushes (stores, zeroing)", "seek / set-up before the stream"]
print("wavefront cycles in the kernel: %.3g total (%d wavefronts x %.0f)" % (tot, 256 * 12, tot / (256 * 12)))
for nm, x in zip(names, v[:5]):
    print("  %-42s %6.1f %%   %8.0f cycles per item" % (nm, 100 * x / tot, x / v[9]))
print("  %-42s %6.1f %%" % ("unaccounted (other item kinds, tail, exit)", 100 * (tot - v[0] - v[1] - v[2] - v[4]) / tot))
print("  per stream block (8 MFMAs = 512 matrix-pipe cycles): %.0f cycles excl. flushes; per flush (32 stores): %.0f cycles" % (
    (v[2] - v[3]) / v[7], v[3] / max(v[8], 1)))
