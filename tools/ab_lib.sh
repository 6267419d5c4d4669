#!/bin/bash
# usage: tools/ab_lib.sh <workload> <steps> libA.so libB.so ...  -- one workload under different builds of the library, interleaved
cd "$GRAFT_REPO_ROOT"; W=$1; ST=$2; shift 2
for r in 1 2 3; do for so in "$@"; do
  BPMF_HIP_LIBRARY=$PWD/$so timeout 300 python bench.py --workload $W --no-cpu-baseline --no-strong --no-bpmf-exe --steps $ST --warmup 10 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$so  ms/step %.4f ' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['roofline']['launch_ms_per_side'].items()})" || true
done; done
