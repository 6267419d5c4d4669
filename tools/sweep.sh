#!/bin/bash
# sampler kernel time for chunk size / grid size variants (ML-1M-shaped bench workload)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for chunk in 64 128 192 256 384 512; do
 for grid in 2048 4096; do
  r=$(BPMF_HIP_CHUNK=$chunk BPMF_HIP_GRID=$grid python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms/step %.4f  sampler launch %.1f us' % (d['ms_per_step'], d['roofline']['launch_ms']*1e3))")
  echo "chunk=$chunk grid=$grid  $r"
 done
done
