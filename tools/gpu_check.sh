#!/bin/bash
# usage: tools/gpu_check.sh [tests] [bench] [prof]   -- runs on the GPU box, writes gpurun_out/check/
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/check; mkdir -p $O
for what in "$@"; do
case $what in
tests) timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log ;;
bench) timeout 600 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-2500 ;;
benchfast) timeout 600 python bench.py --no-cpu-baseline > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-2500 ;;
prof) rm -rf $O/prof; rocprofv3 --kernel-trace -d $O/prof -o r -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/prof.log 2>&1
      DB=$(find $O/prof -name "*.db" | head -1); python tools/kstats.py $DB | head -12
      python - "$DB" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
mid = len(rows) // 2
while 'stage' not in rows[mid][0]: mid += 1
t0 = rows[mid][1]
for r in rows[mid:mid + 24]: print("%-50s start=%9.1f us dur=%8.1f us" % (r[0][:50], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3))
PY
      rm -rf $O/prof ;;
esac
done
