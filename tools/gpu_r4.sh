#!/bin/bash
# round-4 GPU session stages (one gpurun call = a list of stages): tools/gpu_r4.sh <stage> ...
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; mkdir -p gpurun_out
for stage in "$@"; do
  echo "=== stage $stage"
  case $stage in
    prof_*)  w=${stage#prof_}; bash tools/profile_round.sh r04 $w 1 > gpurun_out/r4_prof_$w.log 2>&1; tail -4 gpurun_out/r4_prof_$w.log | cut -c1-300 ;;
    bench)   python bench.py > gpurun_out/profiles/r04_bench.json 2> gpurun_out/profiles/r04_bench.err; tail -c 400 gpurun_out/profiles/r04_bench.json ;;
    bench20) python bench.py --steps 20 --warmup 5 > gpurun_out/profiles/r04_bench_20steps.json 2>/dev/null; cut -c1-260 gpurun_out/profiles/r04_bench_20steps.json ;;
    gpu)     timeout 2900 python -m pytest tests -q -m gpu 2>&1 | tail -8 ;;
    smoke)   python -c "import __graft_entry__ as g; g.smoke()" ;;
    cli)     bash tools/cli_ml1m.sh 400 6 ;;
  esac
done
