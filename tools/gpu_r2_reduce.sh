#!/bin/bash
# BPMF_REDUCE formulation: parity tests + time of the two kernels on the ML-1M shape
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_reduce.py tests/test_cli.py -m gpu -x -q 2>&1 | tail -15
timeout 300 python tools/reduce_time.py 2>&1 | tail -12
