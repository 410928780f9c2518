"""Development aid: where a wavefront of jtj_mfma_lds_kernel (generic one-panel-ahead loop) spends its cycles.  Needs the
timing build (-DGST_JTJ_TIMING=1 of gst_kernels_normal.hip, linked as tools/bin/libgstfwd_jtjtiming.so) and GST_JTJ_FAST=0."""
import os, sys, ctypes as C, numpy as np
os.environ["GST_LIBGSTFWD"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libgstfwd_jtjtiming.so")
os.environ.setdefault("GST_JTJ_FAST", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsti_amd import _lib
nE, nP = 545100, 1616
n = 4
ptr = np.arange(n + 1, dtype=np.int64)
pl = _lib.Plan.from_circuits(16, 1, 1, 1, n, np.zeros(n, np.int32), ptr * 0, np.zeros(0, np.int32), ptr, np.zeros(n, np.int32), np.arange(n, dtype=np.int32))
d_J = pl.device_malloc(nE * nP * 8); d_jtj = pl.device_malloc(nP * nP * 8)
blk = np.random.default_rng(0).standard_normal((5451, nP))
for i in range(100):
    pl.memcpy_h2d(d_J, blk, offset_bytes=i * blk.nbytes)
L = _lib.lib()
L.gst_debug_jtj_phases.argtypes = [C.c_void_p, C.c_int]
pl.fill_jtj_dev(d_J, nE, nP, nP, d_jtj); pl.sync()
L.gst_debug_jtj_phases(None, 1)
import time
t = time.perf_counter(); pl.fill_jtj_dev(d_J, nE, nP, nP, d_jtj); pl.sync(); dt = time.perf_counter() - t
out = (C.c_ulonglong * 8)(); L.gst_debug_jtj_phases(out, 0)
v = np.array(list(out), float)
names = ["next_live + fetch issue", "ds_read + MFMA issue", "wait vmcnt(0)", "stash + lgkmcnt(0)", "barrier"] if os.environ["GST_JTJ_FAST"] == "0" else \
    ["fetch issue", "ds_read + MFMA issue", "next_live", "wait + stash + lgkmcnt(0)", "barrier"]
tot = v[:5].sum()
print("launch %.2f ms; wave-panels %d; cycles per wave-panel %.0f" % (dt * 1e3, v[5], tot / v[5]))
for nm, x in zip(names, v[:5]):
    print("  %-26s %7.0f cycles per panel  %5.1f %%" % (nm, x / v[5], 100 * x / tot))
