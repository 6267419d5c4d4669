#!/bin/bash
# per-kernel timeline of a few iterations (rocprofv3 kernel trace): tools/trace_timeline.sh <workload>
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
W=${1:-ml1m_k128}
rm -rf /tmp/prof_kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_kt -o kt -- python bench.py --workload $W --no-cpu-baseline --no-strong --steps 30 --warmup 10 --repeats 1 --prewarm-ms 0 > /dev/null 2> /tmp/prof_kt.err
python - <<PY
import csv, glob
f = glob.glob('/tmp/prof_kt/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows); mid = rows[n // 2: n // 2 + 40]
t0 = int(mid[0]['Start_Timestamp'])
for r in mid:
    print("%10.1f %10.1f  q%-3s %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:60]))
PY
