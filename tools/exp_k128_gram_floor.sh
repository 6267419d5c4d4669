#!/bin/bash
# K = 128: what the Gram phase costs without its MFMAs (BPMF_HIP_ABLATE 1 = Gram only, + 8 = operands loaded and summed, no MFMA, + 4 = hot rows)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
line() { grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  ms/step %.4f ' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['launch_ms_per_side'].items()})"; }
for wl in ml1m_k128 ml1m_k128_f64; do
  for ab in 1 9 13 3; do
    BPMF_HIP_F32_RIDERS=0 timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 50 --warmup 10 --ablate $ab 2>/dev/null | line "$wl ablate=$ab"
  done
done
