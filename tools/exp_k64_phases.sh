#!/bin/bash
# K = 64 slab sampler: full launch, Gram only (--ablate 1), everything but the Gram (--ablate 2), interleaved: committed reference library
# (bpmf_amd/csrc/variants/head.so, whose kernels carry the run-time phase switches) against the tree's profiling build (make prof)
cd "$GRAFT_REPO_ROOT"
line() { grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  ms/step %.4f ' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['launch_ms_per_side'].items()})"; }
for r in 1 2; do for wl in ${1:-ml1m_k64}; do for lib in head tree; do for ab in 0 1 2; do
  L=$PWD/bpmf_amd/libbpmf_hip_prof.so; [ $lib = head ] && L=$PWD/bpmf_amd/csrc/variants/head.so
  if [ $ab = 0 ]; then BPMF_HIP_LIBRARY=$L timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --no-configs --window-s 0.5 --steps 60 --warmup 10 2>/dev/null | line "$wl $lib full    "
  else BPMF_HIP_LIBRARY=$L timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --no-configs --window-s 0.5 --steps 60 --warmup 10 --ablate $ab 2>/dev/null | line "$wl $lib ablate=$ab"; fi
done; done; done; done
