#!/bin/bash
# usage: tools/abk64.sh libA.so libB.so ...  -- K = 64: ML-1M shape and the ChEMBL shape with different builds, interleaved
cd "$GRAFT_REPO_ROOT"
export SHAPE_PIPELINED=1
for r in 1 2; do
  for so in "$@"; do
    echo "$so ML-1M K=64: $(BPMF_HIP_LIBRARY=$PWD/$so timeout 300 python bench.py --K 64 --no-cpu-baseline --steps 60 --warmup 10 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.4f' % d['ms_per_step'])")   ChEMBL: $(BPMF_HIP_LIBRARY=$PWD/$so timeout 300 python tools/shape_bench.py 64 483500 5775 1023952 20 real 2>&1 | tail -1 | cut -c7-22)"
  done
done
