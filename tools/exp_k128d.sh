#!/bin/bash
cd "$GRAFT_REPO_ROOT"
line() { grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  ms/step %.4f ' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['launch_ms_per_side'].items()})"; }
for w in ml1m_k128 ml1m_k128_f64 ml1m_k64; do for lib in "" nob; do
    E=""; [ -n "$lib" ] && E="BPMF_HIP_LIBRARY=$PWD/bpmf_amd/csrc/variants/$lib.so"
    env $E python bench.py --workload $w --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 50 --warmup 10 2>/dev/null | line "$w ${lib:-base} full"
    env $E BPMF_HIP_F32_RIDERS=0 python bench.py --workload $w --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 50 --warmup 10 --ablate 2 2>/dev/null | line "$w ${lib:-base} fact-only"
done; done
