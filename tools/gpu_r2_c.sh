#!/bin/bash
# round 2: slab form (K = 64 / 128) -- parity, then same-session A/B against the previous forms
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r2c; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_f32.py -x -q -s > $O/tests_full.log 2>&1; echo "rc=$?" >> $O/tests_full.log; tail -12 $O/tests_full.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "64 or low_rank or chunk or heavy" > $O/tests_parity64.log 2>&1; echo "rc=$?" >> $O/tests_parity64.log; tail -6 $O/tests_parity64.log
run() { # tag, workload, env...
  tag=$1; w=$2; shift 2
  env "$@" timeout 600 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-strong > $O/bench_${w}_$tag.json 2> $O/bench_${w}_$tag.err
  python -c "
import json; j=json.loads(open('$O/bench_${w}_$tag.json').read().strip().splitlines()[-1]); print('$w $tag', round(j['value']/1e6,2), 'M/s', round(j['ms_per_step'],4), 'ms/iter launch', j['roofline']['launch_ms_per_side'], 'frac', round(j['roofline']['frac'],3), 'rmse', j['rmse'])"
}
run slab ml1m_k64 BPMF_HIP_MODE=4
run old ml1m_k64 BPMF_HIP_MODE=1
run slab chembl BPMF_HIP_MODE=4
run old chembl BPMF_HIP_MODE=1
run slab ml1m_k128 BPMF_HIP_MODE=4
run old ml1m_k128 BPMF_HIP_MODE=2
