#!/bin/bash
# first contact with the GPU: device info, smoke, GPU tests, a short bench
set -x
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rocminfo 2>/dev/null | grep -E "Name:|Compute Unit|Max Clock" | head -12 > gpurun_out/rocminfo.txt
nproc > gpurun_out/host.txt; lscpu | grep -E "Model name|Socket|Core|Thread" >> gpurun_out/host.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_first.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_first.log
tail -5 gpurun_out/smoke.log; tail -30 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/bench_first.log
