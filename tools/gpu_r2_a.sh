#!/bin/bash
# round 2, first contact: new full-size parity + bounded-wait tests, then the whole GPU suite, then a bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r2a
nproc > gpurun_out/r2a/host.txt; lscpu | grep -E "Model name|Socket|Core|Thread" >> gpurun_out/r2a/host.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_waits.py -x -q -s > gpurun_out/r2a/new_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r2a/new_tests.log
tail -15 gpurun_out/r2a/new_tests.log
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r2a/pytest_gpu.log
tail -8 gpurun_out/r2a/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench_20.log 2>&1; tail -c 1500 gpurun_out/r2a/bench_20.log
