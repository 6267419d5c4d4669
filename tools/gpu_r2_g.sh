#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r2g; mkdir -p $O
BPMF_HIP_SLAB32=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -k "not 64 and not low_rank and not bench and not sharded and not blocking" > $O/tests_slab32.log 2>&1; echo "rc=$?" >> $O/tests_slab32.log; tail -6 $O/tests_slab32.log
run() { # tag, workload, env...
  tag=$1; w=$2; shift 2
  env "$@" timeout 600 python bench.py --workload $w --no-cpu-baseline --no-strong > $O/bench_${w}_$tag.json 2> $O/bench_${w}_$tag.err
  python -c "
import json; j=json.loads(open('$O/bench_${w}_$tag.json').read().strip().splitlines()[-1]); print('$w $tag', round(j['value']/1e6,2), 'M/s', round(j['ms_per_step'],4), 'ms/iter launch', {k: round(v,4) for k,v in j['roofline']['launch_ms_per_side'].items()}, 'frac', round(j['roofline']['frac'],3), 'rmse', j['rmse'])"
}
run old ml1m BPMF_HIP_SLAB32=0
run slab ml1m BPMF_HIP_SLAB32=1
run old2 ml1m BPMF_HIP_SLAB32=0
run slab2 ml1m BPMF_HIP_SLAB32=1
run slab_c384 ml1m BPMF_HIP_SLAB32=1 BPMF_HIP_CHUNK=384
run slab_c1024 ml1m BPMF_HIP_SLAB32=1 BPMF_HIP_CHUNK=1024
