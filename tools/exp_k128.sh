#!/bin/bash
# K = 128 experiments (round 5): ablations of the fp32 / fp64 workgroup form and LDS-occupancy variants, interleaved
cd "$GRAFT_REPO_ROOT"
line() { grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1  ms/step %.4f ' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['roofline']['launch_ms_per_side'].items()})"; }
for w in ml1m_k128 ml1m_k128_f64; do
  for ab in 0 1 2 3; do
    if [ $ab = 0 ]; then python bench.py --workload $w --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 50 --warmup 10 2>/dev/null | line "$w full"
    else BPMF_HIP_F32_RIDERS=0 python bench.py --workload $w --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 50 --warmup 10 --ablate $ab 2>/dev/null | line "$w ablate=$ab"; fi
  done
done
for r in 1 2; do for v in "" lds3 lds2 lds1; do
  if [ -z "$v" ]; then python bench.py --workload ml1m_k128 --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 50 --warmup 10 2>/dev/null | line "base(4/CU)"
  else BPMF_HIP_LIBRARY=$PWD/bpmf_amd/csrc/variants/$v.so python bench.py --workload ml1m_k128 --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 50 --warmup 10 2>/dev/null | line "$v"; fi
done; done
