#!/bin/bash
# round 5: K = 128 fp64 Gram on v_mfma_f64_4x4x4_4b_f64 (tree) against the 16x16x4 form (bpmf_amd/csrc/variants/base.so)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "== parity tests K = 128"
timeout 900 python -m pytest tests -m gpu -x -q -k "128 or f32 or fp32 or padded or latent or heavy or chunk" 2>&1 | tail -4
line() { grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  ms/step %.4f ' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['launch_ms_per_side'].items()}, 'frac %.3f' % r['frac'])"; }
for WL in ml1m_k128 ml1m_k128_f64; do for r in 1; do
  for ab in 0 1; do
    for lib in base tree; do
      E=""; [ $lib = base ] && E="BPMF_HIP_LIBRARY=$PWD/bpmf_amd/csrc/variants/base.so"
      env $E timeout 300 python bench.py --workload $WL --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 60 --warmup 10 --ablate $ab 2>/dev/null | line "$WL $lib ablate=$ab"
    done
  done
done; done
