#!/bin/bash
# the Gram phase without its MFMAs (BPMF_HIP_ABLATE 1 + 8), with hot rows (+ 4), and the launch floor (3): K = 32 / 64 workloads
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
line() { grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  ms/step %.4f ' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['launch_ms_per_side'].items()})"; }
for wl in ml1m; do
  for ab in 0 1 9 13 3; do
    timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 200 --warmup 20 --ablate $ab 2>/dev/null | line "$wl ablate=$ab"
  done
done
