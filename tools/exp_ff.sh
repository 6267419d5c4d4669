#!/bin/bash
# round 5: fraction-free pivot blocks (factor_block44) + look-ahead (k_sample_wg2) against the tree
# (BPMF_PATCH=tools/patches/wg2_lookahead.patch:tools/patches/factor_block44_fraction_free.patch bash tools/build_variant.sh ff tools/patches/apply.py)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "== parity tests K = 64 / 128"
BPMF_HIP_LIBRARY=$PWD/bpmf_amd/csrc/variants/ff.so timeout 1200 python -m pytest tests -m gpu -x -q -k "128 or f32 or fp32 or 64 or chembl or slab or low_rank" 2>&1 | tail -5
line() { grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  ms/step %.4f ' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['launch_ms_per_side'].items()}, 'frac %.3f' % r['frac'])"; }
for r in 1 2; do
  for wl in ml1m_k64 chembl ml1m_k128 ml1m_k128_f64; do
    for lib in tree ff; do
      E=""; [ $lib = ff ] && E="BPMF_HIP_LIBRARY=$PWD/bpmf_amd/csrc/variants/ff.so"
      env $E timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 60 --warmup 10 2>/dev/null | line "$wl $lib"
    done
  done
done
