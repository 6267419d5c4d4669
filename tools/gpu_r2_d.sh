#!/bin/bash
# slab form: where does the time go?  Gram only / factorisation only (BPMF_HIP_ABLATE) per workload
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r2d; mkdir -p $O
run() { # tag, workload, env...
  tag=$1; w=$2; shift 2
  env "$@" timeout 600 python bench.py --workload $w --steps 60 --warmup 10 --no-cpu-baseline --no-strong > $O/bench_${w}_$tag.json 2> $O/bench_${w}_$tag.err
  python -c "
import json; j=json.loads(open('$O/bench_${w}_$tag.json').read().strip().splitlines()[-1]); print('$w $tag', round(j['ms_per_step'],4), 'ms/iter launch', {k: round(v,4) for k,v in j['roofline']['launch_ms_per_side'].items()})"
}
for w in ml1m_k64 ml1m_k128; do
  run full $w BPMF_HIP_ABLATE=0
  run gramonly $w BPMF_HIP_ABLATE=1
  run finishonly $w BPMF_HIP_ABLATE=2
  run neither $w BPMF_HIP_ABLATE=3
done
run chunk256 ml1m_k128 BPMF_HIP_CHUNK=256
run chunk1024 ml1m_k128 BPMF_HIP_CHUNK=1024
run chunk256 ml1m_k64 BPMF_HIP_CHUNK=256
run chunk1024 ml1m_k64 BPMF_HIP_CHUNK=1024
