#!/bin/bash
# usage: tools/abso.sh libA.so libB.so ...   -- the bench with different builds of libbpmf_hip.so,
# interleaved in one GPU session (differences of ~1 us per launch are below the box-to-box noise)
cd "$GRAFT_REPO_ROOT"
for r in 1 2 3; do
  for so in "$@"; do
    BPMF_HIP_LIBRARY=$PWD/$so timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$so  ms/step %.4f  sampler %.4f' % (d['ms_per_step'], d['roofline']['launch_ms']))" || true
  done
done
