#!/bin/bash
# usage: tools/abk128.sh libA.so libB.so ...  -- the K = 128 fp32 bench with different builds, interleaved
cd "$GRAFT_REPO_ROOT"
for r in 1 2; do
  for so in "$@"; do
    BPMF_HIP_LIBRARY=$PWD/$so timeout 300 python bench.py --K 128 --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$so  ms/step %.4f  sampler %.4f' % (d['ms_per_step'], d['roofline']['launch_ms']))" || true
  done
done
