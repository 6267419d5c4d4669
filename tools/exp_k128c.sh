#!/bin/bash
cd "$GRAFT_REPO_ROOT"
line() { grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  ms/step %.4f ' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['launch_ms_per_side'].items()}, [(k['kernel'][:24], k['workgroups_per_cu'], k['vgprs'], k['lds_bytes_per_workgroup']) for k in r['lds']['per_side']['movs']])"; }
for lib in "" base; do
  for ab in 0 1 2; do
    E=""; [ -n "$lib" ] && E="BPMF_HIP_LIBRARY=$PWD/bpmf_amd/csrc/variants/$lib.so"
    if [ $ab = 0 ]; then env $E python bench.py --workload ml1m_k128 --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 50 --warmup 10 2>/dev/null | line "${lib:-new} full"
    else env $E BPMF_HIP_F32_RIDERS=0 python bench.py --workload ml1m_k128 --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 50 --warmup 10 --ablate $ab 2>/dev/null | line "${lib:-new} ablate=$ab"; fi
  done
done
