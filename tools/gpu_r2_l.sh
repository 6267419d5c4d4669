#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r2l; mkdir -p $O
run() { # tag, workload, env...
  tag=$1; w=$2; shift 2
  env "$@" timeout 600 python bench.py --workload $w --steps 60 --warmup 10 --no-cpu-baseline --no-strong > $O/bench_${w}_$tag.json 2> $O/bench_${w}_$tag.err
  python -c "
import json; j=json.loads(open('$O/bench_${w}_$tag.json').read().strip().splitlines()[-1]); print('$w $tag', round(j['ms_per_step'],4), 'ms/iter launch', {k: round(v,4) for k,v in j['roofline']['launch_ms_per_side'].items()})"
}
run full ml1m_k128 BPMF_HIP_ABLATE=0
run gramonly ml1m_k128 BPMF_HIP_ABLATE=1
run finishonly ml1m_k128 BPMF_HIP_ABLATE=2
run neither ml1m_k128 BPMF_HIP_ABLATE=3
run gramonly4w ml1m_k128 BPMF_HIP_ABLATE=1 BPMF_HIP_WG_WAVES=4
run finishonly4w ml1m_k128 BPMF_HIP_ABLATE=2 BPMF_HIP_WG_WAVES=4
