#!/bin/bash
# per-kernel times for the full sampler and with phases ablated / different chunk sizes
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/abl
python -m pytest tests -m gpu -x -q > gpurun_out/abl/pytest.log 2>&1; tail -3 gpurun_out/abl/pytest.log
run() { # name, env...
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace -d gpurun_out/abl/$name -o r -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/abl/$name.log 2>&1
  echo "== $name"; python tools/kstats.py $(find gpurun_out/abl/$name -name "*.db" | head -1) | head -8
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/abl/$name.log
  rm -rf gpurun_out/abl/$name
}
run full BPMF_HIP_ABLATE=0
run nofinish BPMF_HIP_ABLATE=1
run nogram BPMF_HIP_ABLATE=2
run neither BPMF_HIP_ABLATE=3
for c in 64 256 512 1024 4096; do run chunk$c BPMF_HIP_CHUNK=$c; done
python bench.py --steps 50 --warmup 5 > gpurun_out/abl/bench_plain.log 2>&1; tail -1 gpurun_out/abl/bench_plain.log | cut -c1-2000
