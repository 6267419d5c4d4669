#!/usr/bin/env python
"""Averages of the PMC counters rocprofv3 collected for the sampler kernel (rocpd sqlite db).
For the finish probe only the dispatches of the largest problem are kept."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tag = sys.argv[2] if len(sys.argv) > 2 else ""
pat = sys.argv[3] if len(sys.argv) > 3 else "%k_sample%"
rows = list(db.execute("select dispatch_id, kernel_name, counter_name, value, start, end from counters_collection "
                       "where kernel_name like ?", (pat,)))
if not rows:
    print(tag, "no rows"); sys.exit(0)
# keep the dispatches in the last third (bench: steady state; probe: biggest problem = last launches)
disp = sorted({r[0] for r in rows})
keep = set(disp[-max(3, len(disp) // 6):])
acc = {}
for d, k, c, v, s, e in rows:
    if d in keep:
        acc.setdefault(c, []).append(v)
for c in sorted(acc):
    v = acc[c]
    print("%-10s %-28s avg=%16.1f  n=%d" % (tag, c, sum(v) / len(v), len(v)))
# Per launch SHAPE (argv[4] = "bygrid"): the two sides of a half-iteration run the same kernel with different grids -- the
# memory-side counters of each (strong_10Mx1M: the users side streams 10M columns and gathers from the 256 MB items factor, the
# items side the other way round).  Lines "pmc[grid=N] COUNTER avg=..".  Needs a grid column in the rocpd view; silently
# skipped where there is none.
if len(sys.argv) > 4 and sys.argv[4] == "bygrid":
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    gcol = next((c for c in ("grid_size_x", "grid_size", "grid_x") if c in cols), None)
    if gcol:
        by = {}
        for d, g, c, v in db.execute("select dispatch_id, %s, counter_name, value from counters_collection where kernel_name like ?" % gcol, (pat,)):
            if d in keep:
                by.setdefault((g, c), {}).setdefault(d, 0.0)
                by[(g, c)][d] += v
        for (g, c) in sorted(by):
            v = list(by[(g, c)].values())
            print("%-10s %-28s avg=%16.1f  n=%d" % ("%s[grid=%s]" % (tag, g), c, sum(v) / len(v), len(v)))
