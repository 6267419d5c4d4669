#!/bin/bash
# round 5: K = 128 with every wave loading 1 / NW of its operands (TIMING HACK, wrong sums: only the no-MFMA and Gram-only ablations run);
# the variant: a one-line patch of gather() in the round-4 form of kernels_wg2.h (tools/patches/wg2_round4_form.patch first)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
line() { grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  ms/step %.4f ' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['launch_ms_per_side'].items()})"; }
for wl in ml1m_k128 ml1m_k128_f64; do
  for ab in 1 9; do
   for lib in tree halfload; do
    E=""; [ $lib = halfload ] && E="BPMF_HIP_LIBRARY=$PWD/bpmf_amd/csrc/variants/halfload.so"
    env $E BPMF_HIP_F32_RIDERS=0 timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 50 --warmup 10 --ablate $ab 2>/dev/null | line "$wl $lib ablate=$ab"
   done
  done
done
