#!/bin/bash
# usage: tools/abmode.sh "ENV1=.. ENV2=.." "ENV.." ...   -- the bench under different environments, interleaved
cd "$GRAFT_REPO_ROOT"
for r in 1 2 3; do
  for e in "$@"; do
    env $e timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 50 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-60s ms/step %.4f  sampler %.4f' % ('$e', d['ms_per_step'], d['roofline']['launch_ms']))" || echo "$e failed"
  done
done
