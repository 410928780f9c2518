"""Development aid: digest of the GST_FD_TRACE records (pair, t0, t1, hw_id) -- per-SIMD busy time, longest pairs,
measured duration against the work table's estimate."""
import sys
import numpy as np

for path in sys.argv[1:]:
    raw = np.fromfile(path, dtype=np.uint64)
    n = int(raw[0]); rec = raw[1:1 + 4 * n].reshape(n, 4)
    chain = (rec[:, 0] & np.uint64(0x40000000)) != 0        # base-pass chains walked inside the launch (overlap form)
    if chain.any():
        c0 = rec[chain, 1].astype(np.int64); c1 = rec[chain, 2].astype(np.int64); tm = rec[:, 1].astype(np.int64).min()
        print("== %s: %d chains inside the launch: start (us) min/max %.1f %.1f, end 50/90/100 %.1f %.1f %.1f, duration 50/100 %.1f %.1f" % (
            (path, int(chain.sum()), (c0.min() - tm) / 100.0, (c0.max() - tm) / 100.0) + tuple(np.percentile((c1 - tm) / 100.0, [50, 90, 100])) +
            tuple(np.percentile((c1 - c0) / 100.0, [50, 100]))))
        rec = rec[~chain]; n = len(rec)
    pair = rec[:, 0].astype(np.int64); t0 = rec[:, 1].astype(np.int64); t1 = rec[:, 2].astype(np.int64)
    hw = rec[:, 3]
    hwid = (hw & np.uint64(0xffffffff)).astype(np.int64); xcc = (hw >> np.uint64(32)).astype(np.int64)
    simd = (hwid >> 4) & 3; cu = (hwid >> 8) & 15; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    tmin = t0.min(); span = (t1.max() - tmin) / 100.0      # wall_clock64: 100 MHz -> us
    dur = (t1 - t0) / 100.0
    print("== %s: %d pairs, span %.1f us, sum of durations %.1f ms, %d distinct SIMDs" % (path, n, span, dur.sum() / 1e3, len(np.unique(key))))
    order = np.argsort(-dur)
    print(" longest pairs (us):", np.round(dur[order[:8]], 1), " start at (us):", np.round((t0[order[:8]] - tmin) / 100.0, 1))
    print(" duration percentiles 50/90/99/100: %.1f %.1f %.1f %.1f us" % tuple(np.percentile(dur, [50, 90, 99, 100])))
    # per-SIMD: last end time and number of pairs
    ends = {}; cnt = {}
    for k, e in zip(key, t1):
        ends[k] = max(ends.get(k, 0), e); cnt[k] = cnt.get(k, 0) + 1
    e = (np.array(list(ends.values())) - tmin) / 100.0
    print(" per-SIMD finish time (us) min/median/max: %.1f %.1f %.1f ; pairs per SIMD min/max %d %d" % (e.min(), np.median(e), e.max(), min(cnt.values()), max(cnt.values())))
    segs = {}
    for i in np.argsort(t0):
        segs.setdefault(int(pair[i]), []).append(i)
    gaps = [(t0[v[k + 1]] - t1[v[k]]) / 100.0 for v in segs.values() for k in range(len(v) - 1)]
    if gaps:
        print(" hand-overs: %d; wait between post and pick-up (us) 50/90/100: %.1f %.1f %.1f" % ((len(gaps),) + tuple(np.percentile(gaps, [50, 90, 100]))))
    try:
        c = np.fromfile(path + ".cost", dtype=np.int64).reshape(-1, 2)
        est = dict(zip(c[:, 0], c[:, 1]))
        ev = np.array([est.get(int(p), 0) for p in pair], float)
        big = ev > np.percentile(ev, 75)
        k = np.polyfit(ev[big], dur[big], 1)
        print(" duration ~ %.4f us * estimate + %.1f us on the top quartile; rel. residual std %.3f; corr %.3f" % (
            k[0], k[1], np.std(dur[big] - np.polyval(k, ev[big])) / dur[big].mean(), np.corrcoef(ev, dur)[0, 1]))
        # solo speed: pairs that started in the last 15 % of the span vs the first 15 %
        early = (t0 - tmin) < 0.15 * (t1.max() - tmin)
        late = (t0 - tmin) > 0.6 * (t1.max() - tmin)
        for nm, m in (("early", early & (ev > 50)), ("late", late & (ev > 50))):
            if m.sum() > 10:
                print("  %s pairs: %d, us per estimated unit: %.4f" % (nm, m.sum(), (dur[m] / ev[m]).mean()))
    except FileNotFoundError:
        pass
    # pace of the long pairs against how crowded their CU was while they ran (round 5): the average number of traced pairs
    # alive on the same CU during the pair's life
    try:
        cukey = key // 4
        longm = ev > 800
        if longm.sum() > 5:
            idx = np.nonzero(longm)[0]
            crowd = []
            for i in idx:
                same = cukey == cukey[i]
                ov = np.clip(np.minimum(t1[same], t1[i]) - np.maximum(t0[same], t0[i]), 0, None).sum() / max(t1[i] - t0[i], 1)
                crowd.append(ov)
            crowd = np.array(crowd); pace = dur[idx] / ev[idx]
            for lo, hi in ((0, 4), (4, 6), (6, 8), (8, 10), (10, 13)):
                m = (crowd >= lo) & (crowd < hi)
                if m.sum():
                    print("  long pairs with %d-%d pairs alive on their CU: %d, us per application %.3f (min %.3f max %.3f)" % (lo, hi, m.sum(), pace[m].mean(), pace[m].min(), pace[m].max()))
            print("  long pairs: %d, estimate 50/100: %.0f %.0f; duration 50/100: %.0f %.0f us" % ((longm.sum(),) + tuple(np.percentile(ev[longm], [50, 100])) + tuple(np.percentile(dur[longm], [50, 100]))))
            print("  all pairs: estimate sum %.0f over %d SIMDs = %.0f each; estimate histogram (<100, <400, <800, <1200, >=1200): %s" % (
                ev.sum(), len(np.unique(key)), ev.sum() / len(np.unique(key)), [int(((ev >= a) & (ev < b)).sum()) for a, b in ((0, 100), (100, 400), (400, 800), (800, 1200), (1200, 1e9))]))
    except Exception as e:
        print("  (crowding analysis failed: %s)" % e)
