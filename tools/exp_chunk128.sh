#!/bin/bash
# K = 128: chunk length of the heavy columns (BPMF_HIP_CHUNK) after the staged gathers changed the cost of a rating
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
line() { grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  ms/step %.4f ' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['launch_ms_per_side'].items()})"; }
for wl in ml1m_k128 ml1m_k128_f64; do
  for c in 0 256 384 512 768 1024 1536; do
    E=""; [ $c != 0 ] && E="BPMF_HIP_CHUNK=$c"
    env $E timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 60 --warmup 10 2>/dev/null | line "$wl chunk=$c"
  done
done
