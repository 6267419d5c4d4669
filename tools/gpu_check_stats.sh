#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reduce.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not sampler_mode or persistent or auto" 2>&1 | tail -3
bash tools/gpu_r2_status.sh
BPMF_HIP_MODE=0 timeout 300 python bench.py --workload ml1m --no-cpu-baseline --no-strong | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ml1m MODE=0 (persistent)', round(j['ms_per_step'],4), j['roofline']['launch_ms_per_side'])"
timeout 300 python tools/reduce_time.py 2>&1 | tail -4
