#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r2k; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_f32.py tests/test_gpu_fullsize.py tests/test_cli.py -x -q -s -k "f32 or k128 or g1" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -8 $O/tests.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "parts" > $O/tests_parts.log 2>&1; tail -3 $O/tests_parts.log
run() { # tag, workload, env...
  tag=$1; w=$2; shift 2
  env "$@" timeout 600 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-strong > $O/bench_${w}_$tag.json 2> $O/bench_${w}_$tag.err
  python -c "
import json; j=json.loads(open('$O/bench_${w}_$tag.json').read().strip().splitlines()[-1]); print('$w $tag', round(j['value']/1e6,2), 'M/s', round(j['ms_per_step'],4), 'ms/iter launch', {k: round(v,4) for k,v in j['roofline']['launch_ms_per_side'].items()}, 'frac', round(j['roofline']['frac'],3), 'rmse', j['rmse'])"
}
run wg2 ml1m_k128 BPMF_HIP_MODE=5
run wg2_4w ml1m_k128 BPMF_HIP_MODE=5 BPMF_HIP_WG_WAVES=4
run old ml1m_k128 BPMF_HIP_MODE=2
run wg2_c1024 ml1m_k128 BPMF_HIP_CHUNK=1024
run wg2_c4096 ml1m_k128 BPMF_HIP_CHUNK=4096
run wg2_c256 ml1m_k128 BPMF_HIP_CHUNK=256
