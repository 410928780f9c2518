"""Dev tool: time gst_fill_jtj_dev / gst_fill_jtf_dev on a synthetic resident matrix (no simulation)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from pygsti_amd import _lib
nE, nP = 545100, 1616
n = 4
ptr = np.arange(n + 1, dtype=np.int64)
pl = _lib.Plan.from_circuits(16, 1, 1, 1, n, np.zeros(n, np.int32), ptr * 0, np.zeros(0, np.int32), ptr, np.zeros(n, np.int32), np.arange(n, dtype=np.int32))
d_J = pl.device_malloc(nE * nP * 8); d_jtj = pl.device_malloc(nP * nP * 8)
rng = np.random.default_rng(0)
blk = rng.standard_normal((5451, nP))
for i in range(100):
    pl.memcpy_h2d(d_J, blk, offset_bytes=i * blk.nbytes)
pl.fill_jtj_dev(d_J, nE, nP, nP, d_jtj); pl.sync()
t = time.perf_counter()
for _ in range(5): pl.fill_jtj_dev(d_J, nE, nP, nP, d_jtj)
pl.sync(); dt = (time.perf_counter() - t) / 5
print("jtj %.2f ms  %.1f TFLOP/s (triangle flops)" % (dt * 1e3, nE * nP * nP / dt / 1e12))
C = pl.memcpy_d2h(np.empty((nP, nP)), d_jtj)
ref = 100 * (blk.T @ blk)
print("rel err", np.abs(C - ref).max() / np.abs(ref).max())
