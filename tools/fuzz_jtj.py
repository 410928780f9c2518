"""Randomised check of the normal-equation kernels (gst_fill_normal_eqs_dev / gst_fill_jtj_dev / gst_fill_jtf_dev) through the
C ABI against numpy: random shapes (both loop forms: n_cols % 8 == 0 with even ld, and everything else), leading dimensions,
block-sparse patterns with long dead stretches, zero / non-zero weights.  usage: python tools/fuzz_jtj.py [n_cases] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygsti_amd import _lib

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
n = 4
ptr = np.arange(n + 1, dtype=np.int64)
pl = _lib.Plan.from_circuits(16, 1, 1, 1, n, np.zeros(n, np.int32), ptr * 0, np.zeros(0, np.int32), ptr, np.zeros(n, np.int32), np.arange(n, dtype=np.int32))
bad = 0
t0 = time.time()
for case in range(n_cases):
    kind = rng.integers(0, 4)
    n_rows = int(rng.integers(1, 200)) if kind == 0 else int(rng.integers(16384, 60000)) if kind in (1, 2) else int(rng.integers(200, 16384))
    n_cols = int(rng.integers(1, 260)) * 8 if rng.random() < 0.6 else int(rng.integers(1, 2000))
    n_cols = min(n_cols, 2048)
    pad = int(rng.choice([0, 0, 2, 8, 1, 3]))
    ld = n_cols + pad
    if n_rows * ld > 60e6: n_rows = int(60e6 // ld)
    Jp = rng.standard_normal((n_rows, ld)) * np.exp(rng.uniform(-2, 2, size=(1, ld)))
    if kind in (1, 2):                       # block sparsity: column bands zeroed over row runs, long dead stretches
        r = 0
        while r < n_rows:
            m = int(rng.integers(3, 400))
            c0 = int(rng.integers(0, n_cols)); c1 = int(rng.integers(c0, n_cols + 1))
            if rng.random() < 0.7: Jp[r:r + m, c0:c1] = 0.0
            if rng.random() < 0.05: Jp[r:r + 8 * m, :] = 0.0
            r += m
    w = rng.random(n_rows) + 0.5
    w[rng.random(n_rows) < 0.1] = 0.0
    f = rng.standard_normal(n_rows)
    use_w = rng.random() < 0.8
    d_J = pl.device_malloc(Jp.nbytes); d_K = pl.device_malloc(Jp.nbytes)
    d_a = pl.device_malloc(n_cols * n_cols * 8); d_b = pl.device_malloc(n_cols * n_cols * 8)
    d_ya = pl.device_malloc(n_cols * 8); d_yb = pl.device_malloc(n_cols * 8)
    d_w = pl.device_malloc(n_rows * 8); d_f = pl.device_malloc(n_rows * 8)
    pl.memcpy_h2d(d_J, Jp); pl.memcpy_h2d(d_K, Jp); pl.memcpy_h2d(d_w, w); pl.memcpy_h2d(d_f, f)
    pl.fill_jtj_dev(d_K, n_rows, n_cols, ld, d_a, d_w if use_w else None); pl.fill_jtf_dev(d_K, n_rows, n_cols, ld, d_f, d_ya)
    pl.fill_normal_eqs_dev(d_J, n_rows, n_cols, ld, d_w if use_w else None, d_f, d_b, d_yb)
    a = pl.memcpy_d2h(np.empty((n_cols, n_cols)), d_a); b = pl.memcpy_d2h(np.empty((n_cols, n_cols)), d_b)
    ya = pl.memcpy_d2h(np.empty(n_cols), d_ya); yb = pl.memcpy_d2h(np.empty(n_cols), d_yb)
    back = pl.memcpy_d2h(np.empty((n_rows, ld)), d_J)
    Js = Jp[:, :n_cols] * (w[:, None] if use_w else 1.0)
    want = Js.T @ Js; wy = Js.T @ f
    scale = np.sqrt(np.outer(np.diag(want), np.diag(want))) + 1e-300
    ok = (np.array_equal(a, b) and np.array_equal(back, Jp) and np.array_equal(b, b.T)
          and (np.abs(b - want) <= 2e-12 * scale).all()
          and np.abs(yb - wy).max() <= 1e-11 * max(np.abs(wy).max(), 1e-300) and np.abs(ya - yb).max() <= 1e-12 * max(np.abs(ya).max(), 1e-300))
    if not ok:
        bad += 1
        print("FAIL case %d: rows %d cols %d ld %d kind %d w %s: jtj bits %s, J untouched %s, sym %s, jtj err %.2e, jtf err %.2e / %.2e" % (
            case, n_rows, n_cols, ld, kind, use_w, np.array_equal(a, b), np.array_equal(back, Jp), np.array_equal(b, b.T),
            np.abs((b - want) / scale).max(), np.abs(yb - wy).max() / max(np.abs(wy).max(), 1e-300), np.abs(ya - yb).max() / max(np.abs(ya).max(), 1e-300)))
    for d in (d_J, d_K, d_a, d_b, d_ya, d_yb, d_w, d_f): pl.device_free(d)
print("fuzz_jtj: %d cases, %d failures, %.0f s (seed %d)" % (n_cases, bad, time.time() - t0, seed))
sys.exit(1 if bad else 0)
