#!/bin/bash
# A/B of sampler variants through environment switches: tools/ab.sh "VAR=1 ..." "VAR=0 ..."
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for cfg in "$@"; do
  echo "=== $cfg"
  env $cfg python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q 2>&1 | tail -2
  for rep in 1 2; do
    env $cfg python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('   ms/step %.4f  samples/s %.2fM  sampler launch %.1f us  colstats %.1f us' % (d['ms_per_step'], d['value']/1e6, d['roofline']['launch_ms']*1e3, d['roofline']['colstats_ms']*1e3))"
  done
done
