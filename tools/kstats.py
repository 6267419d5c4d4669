#!/usr/bin/env python
"""Per-kernel duration summary from a rocprofv3 rocpd sqlite database (the default
output of `rocprofv3 --kernel-trace`): name, calls, avg/min/max microseconds."""
import sqlite3
import sys


def main(path, out=sys.stdout):
    db = sqlite3.connect(path)
    rows = list(db.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                           "from kernels group by name order by sum(end-start) desc"))
    out.write("%-64s %7s %10s %10s %10s %10s\n" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_ms"))
    for name, n, avg, mn, mx, tot in rows:
        out.write("%-64s %7d %10.2f %10.2f %10.2f %10.3f\n" % (name[:64], n, avg / 1e3, mn / 1e3, mx / 1e3, tot / 1e6))


if __name__ == "__main__":
    main(sys.argv[1])
