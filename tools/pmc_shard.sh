#!/bin/bash
# HBM-side traffic counters of the sampler on one rank's share of the 10M x 1M configuration
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/pmcshard; rm -rf $O; mkdir -p $O
R=${1:-7}
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/ps; rocprofv3 --pmc $c --kernel-trace -d /tmp/ps -o p -- python tools/shard_bench.py 8 $R > $O/log_$R.txt 2>&1
  python - "$(find /tmp/ps -name '*.db' | head -1)" <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select dispatch_id, kernel_name, counter_name, value, start, end from counters_collection where kernel_name like '%k_sample1%' order by dispatch_id"))
per = collections.defaultdict(dict)
for d, k, c, v, s, e in rows: per[d][c] = v; per[d]['dur_ms'] = (e - s) / 1e6
for d in sorted(per): print("dispatch", d, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in per[d].items()})
PY
done > $O/pmc_$R.txt 2>&1
cat $O/pmc_$R.txt | grep dispatch | tail -12
