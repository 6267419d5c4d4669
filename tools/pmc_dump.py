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
