#!/usr/bin/env python
"""Per-KERNEL averages of the PMC counters of a rocprofv3 --pmc run (rocpd sqlite db): python tools/pmc_by_kernel.py DB [pattern]
(steady state: the last third of every kernel's dispatches)."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else "%k_%"
rows = list(db.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection where kernel_name like ?", (pat,)))
by = collections.defaultdict(lambda: collections.defaultdict(dict))
for d, k, c, v in rows:
    by[k][c][d] = by[k][c].get(d, 0.0) + v
for k in sorted(by):
    print(k[:100])
    for c in sorted(by[k]):
        ds = sorted(by[k][c])
        keep = ds[-max(3, len(ds) // 3):]
        v = [by[k][c][d] for d in keep]
        print("    %-28s avg=%16.1f  n=%d" % (c, sum(v) / len(v), len(v)))
