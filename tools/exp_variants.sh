#!/bin/bash
# usage: tools/exp_variants.sh "WORKLOAD ..." ROUNDS lib ...   (lib = a name under bpmf_amd/csrc/variants/, or `tree`) -- interleaved benches
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
WLS=$1; ROUNDS=$2; shift 2
line() { grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1  ms/step %.4f ' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in r['launch_ms_per_side'].items()}, 'frac %.3f' % r['frac'])"; }
for r in $(seq $ROUNDS); do
  for wl in $WLS; do
    for lib in "$@"; do
      E=""; [ $lib != tree ] && E="BPMF_HIP_LIBRARY=$PWD/bpmf_amd/csrc/variants/$lib.so"
      env $E timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --no-configs --window-s ${WINDOW_S:-0.6} --steps 60 --warmup 10 2>/dev/null | line "$wl $lib"
    done
  done
done
