#!/bin/bash
# usage: tools/sweep.sh <workload> "ENV=.. ENV=.." "ENV=.." ...   -- the bench of one workload under several environments (two rounds, interleaved)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
W=$1; shift
for rep in 1 2; do for cfg in "$@"; do
  env $cfg python bench.py --workload $W --no-cpu-baseline --no-strong ${SWEEP_FLAGS:---steps 200} 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W', '$cfg', round(j['value']/1e6,2), round(j['ms_per_step'],4), {k: round(v*1e3,1) for k,v in j['roofline']['launch_ms_per_side'].items()})" || echo "$W $cfg failed"
done; done
