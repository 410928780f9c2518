"""Development aid: two GST_FD_TRACE files of the same atom (A = base pass inside the launch, B = separate base pass):
which pairs got slower, and where -- on the SIMDs that hosted a chain or everywhere."""
import sys
import numpy as np


def load(path):
    raw = np.fromfile(path, dtype=np.uint64)
    n = int(raw[0]); rec = raw[1:1 + 4 * n].reshape(n, 4)
    hw = rec[:, 3]
    hwid = (hw & np.uint64(0xffffffff)).astype(np.int64); xcc = (hw >> np.uint64(32)).astype(np.int64)
    simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 15; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    chain = (rec[:, 0] & np.uint64(0x40000000)) != 0
    tm = rec[:, 1].astype(np.int64).min()
    return dict(pair=(rec[:, 0] & np.uint64(0xffffffff)).astype(np.int64), t0=(rec[:, 1].astype(np.int64) - tm) / 100.0, t1=(rec[:, 2].astype(np.int64) - tm) / 100.0,
                key=key, chain=chain)


A, B = load(sys.argv[1]), load(sys.argv[2])
chain_simds = set(A["key"][A["chain"]].tolist())
chain_end = {k: e for k, e in zip(A["key"][A["chain"]], A["t1"][A["chain"]])}
print("chains: %d on %d SIMDs" % (A["chain"].sum(), len(chain_simds)))


def by_pair(T):
    d = {}
    m = ~T["chain"]
    for p, t0, t1, k in zip(T["pair"][m], T["t0"][m], T["t1"][m], T["key"][m]):
        d.setdefault(int(p), []).append((t0, t1, int(k)))
    return {p: sorted(v) for p, v in d.items()}


a, b = by_pair(A), by_pair(B)
rows = []
for p, va in a.items():
    vb = b.get(p)
    if not vb or len(vb) != len(va):
        continue
    for (a0, a1, ka), (b0, b1, kb) in zip(va, vb):
        rows.append((a1 - a0, b1 - b0, a0, b0, ka in chain_simds, a1, b1))
r = np.array(rows, float)
big = r[:, 1] > 500
print("matched segments: %d (%d longer than 500 us in B)" % (len(r), big.sum()))
for nm, m in (("on chain SIMDs", r[:, 4] == 1), ("elsewhere", r[:, 4] == 0)):
    mm = m & big
    print(" %-15s n=%5d  sum A %.1f ms  sum B %.1f ms  ratio %.3f   median ratio %.3f   mean start A %.0f us B %.0f us   mean end A %.0f B %.0f" % (
        nm, mm.sum(), r[mm, 0].sum() / 1e3, r[mm, 1].sum() / 1e3, r[mm, 0].sum() / r[mm, 1].sum(), np.median(r[mm, 0] / r[mm, 1]),
        r[mm, 2].mean(), r[mm, 3].mean(), r[mm, 5].mean(), r[mm, 6].mean()))
# per-SIMD finish
for nm, T in (("A", A), ("B", B)):
    ends = {}
    m = ~T["chain"]
    for k, e in zip(T["key"][m], T["t1"][m]):
        ends[int(k)] = max(ends.get(int(k), 0.0), e)
    on = np.array([e for k, e in ends.items() if k in chain_simds]); off = np.array([e for k, e in ends.items() if k not in chain_simds])
    print(" %s per-SIMD finish: chain SIMDs(of A) median %.0f max %.0f | others median %.0f max %.0f" % (nm, np.median(on), on.max(), np.median(off), off.max()))
# slowdown of long segments by start-time bucket (A)
for lo, hi in ((0, 50), (50, 600), (600, 1500), (1500, 2500), (2500, 9999)):
    m = big & (r[:, 2] >= lo) & (r[:, 2] < hi)
    if m.sum() > 5:
        print(" segments starting in [%4d, %4d) us of A: n=%4d ratio A/B %.3f" % (lo, hi, m.sum(), r[m, 0].sum() / r[m, 1].sum()))
# slow-path statistics (an instrumented build may pack poll counts into the upper half of the record's first word; zeros otherwise)
rawA = np.fromfile(sys.argv[1], dtype=np.uint64)
nA = int(rawA[0]); w0 = rawA[1:1 + 4 * nA].reshape(nA, 4)[:, 0]
w0 = w0[(w0 & np.uint64(0x40000000)) == 0]
polls = ((w0 >> np.uint64(32)) & np.uint64(0xffff)).astype(np.int64); wait_us = ((w0 >> np.uint64(48)) & np.uint64(0xffff)).astype(np.int64)
if polls.sum() > 0:
    print(" slow-path entries (polls) per segment: total %d, mean %.1f, max %d; waiting time total %.1f ms, per segment mean %.1f us max %d us" % (
        polls.sum(), polls.mean(), polls.max(), wait_us.sum() / 1e3, wait_us.mean(), wait_us.max()))
