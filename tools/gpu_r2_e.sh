#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r2e; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -x -q -s -k "k64" > $O/tests_full.log 2>&1; echo "rc=$?" >> $O/tests_full.log; tail -6 $O/tests_full.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "64 or low_rank or chunk or heavy" > $O/tests_parity64.log 2>&1; echo "rc=$?" >> $O/tests_parity64.log; tail -4 $O/tests_parity64.log
run() { # tag, workload, env...
  tag=$1; w=$2; shift 2
  env "$@" timeout 600 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline --no-strong > $O/bench_${w}_$tag.json 2> $O/bench_${w}_$tag.err
  python -c "
import json; j=json.loads(open('$O/bench_${w}_$tag.json').read().strip().splitlines()[-1]); print('$w $tag', round(j['value']/1e6,2), 'M/s', round(j['ms_per_step'],4), 'ms/iter launch', {k: round(v,4) for k,v in j['roofline']['launch_ms_per_side'].items()}, 'frac', round(j['roofline']['frac'],3))"
}
run full ml1m_k64 BPMF_HIP_ABLATE=0
run gramonly ml1m_k64 BPMF_HIP_ABLATE=1
run finishonly ml1m_k64 BPMF_HIP_ABLATE=2
run neither ml1m_k64 BPMF_HIP_ABLATE=3
run chunk256 ml1m_k64 BPMF_HIP_CHUNK=256
run chunk384 ml1m_k64 BPMF_HIP_CHUNK=384
run full chembl BPMF_HIP_ABLATE=0
run full ml1m BPMF_HIP_ABLATE=0
