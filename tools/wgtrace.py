"""Per-workgroup lifetimes of the last sampler launch (variant build with the trace patch):
python tools/wgtrace.py FILE"""
import sys, collections
import numpy as np
rows = [tuple(int(x) for x in l.split()) for l in open(sys.argv[1])]
a = np.array(rows, dtype=np.int64)
w, t0, t1, meta, tg = a[:, 0], a[:, 1], a[:, 2], a[:, 3], a[:, 4]
ln = meta >> 32; hw = meta & 0xFFFF; xcc = (meta >> 16) & 0xF
base = t0.min()
dur = (t1 - t0) / 100.0; gram = (tg - t0) / 100.0       # us (100 MHz)
print("workgroups %d, launch span %.1f us" % (len(w), (t1.max() - base) / 100.0))
print("start offsets: p50 %.1f p90 %.1f max %.1f us" % tuple(np.percentile((t0 - base) / 100.0, [50, 90, 100])))
for lo, hi in ((0, 0), (1, 32), (33, 100), (101, 250), (251, 500), (501, 5000)):
    m = (ln >= lo) & (ln <= hi)
    if m.sum():
        print("len %4d..%4d: n=%5d  life p50 %.1f p90 %.1f us   gram-phase p50 %.1f us   start p50 %.1f us" % (lo, hi, m.sum(), np.percentile(dur[m], 50), np.percentile(dur[m], 90), np.percentile(gram[m], 50), np.percentile((t0[m] - base) / 100.0, 50)))
simd = (xcc << 16) | (hw & 0xFFF0)    # xcc, se, sh, cu, simd (drop wave id)
busy = collections.defaultdict(float); last = collections.defaultdict(float); cnt = collections.Counter()
for s, a0, a1 in zip(simd, t0, t1):
    busy[s] += (a1 - a0) / 100.0; last[s] = max(last[s], (a1 - base) / 100.0); cnt[s] += 1
v = np.array(list(last.values())); c = np.array(list(cnt.values()))
print("SIMDs %d: items per SIMD min/median/max %d/%d/%d; last-finish p10 %.1f p50 %.1f max %.1f us" % (len(v), c.min(), np.median(c), c.max(), np.percentile(v, 10), np.percentile(v, 50), v.max()))
