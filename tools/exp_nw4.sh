#!/bin/bash
# round 5: K = 128 fp32 with FOUR waves per workgroup (variants nw4_3 / nw4_4: launch bounds for 3 / 4 workgroups per CU) against the tree's two;
# riders off in all runs (the host sizes the rider grid for two waves per workgroup)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
line() { grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  ms/step %.4f ' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['launch_ms_per_side'].items()}, 'frac %.3f' % r['frac'], [(k['workgroups_per_cu'], k['vgprs']) for k in r['lds']['per_side']['movs']])"; }
for r in 1 2; do
  for lib in tree nw4_3 nw4_4; do
    E=""; [ $lib != tree ] && E="BPMF_HIP_LIBRARY=$PWD/bpmf_amd/csrc/variants/$lib.so"
    env $E BPMF_HIP_F32_RIDERS=0 timeout 300 python bench.py --workload ml1m_k128 --no-cpu-baseline --no-strong --no-bpmf-exe --steps 60 --warmup 10 2>&1 | line "ml1m_k128 $lib"
  done
done
