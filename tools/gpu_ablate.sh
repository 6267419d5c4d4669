#!/bin/bash
# BPMF_HIP_ABLATE sweep of one workload: tools/gpu_ablate.sh <workload> "<list>"
cd "$GRAFT_REPO_ROOT"; W=${1:-ml1m}; O=gpurun_out/ablate; mkdir -p $O
for ab in ${2:-0 1 2 3}; do
  BPMF_HIP_ABLATE=$ab timeout 300 python bench.py --workload $W --no-cpu-baseline --no-strong --repeats 5 > $O/${W}_$ab.json 2> $O/${W}_$ab.err
  python -c "
import json; j=json.loads(open('$O/${W}_$ab.json').read().strip().splitlines()[-1]); print('$W ablate $ab', round(j['ms_per_step'],4), 'ms/iter launch', {k: round(v,4) for k,v in j['roofline']['launch_ms_per_side'].items()})" || tail -3 $O/${W}_$ab.err
done
