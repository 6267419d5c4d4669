#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/k128; mkdir -p $O
for ab in ${ABL:-1 5 9 13}; do for c in 384; do
  BPMF_HIP_ABLATE=$ab BPMF_HIP_CHUNK=$c timeout 300 python bench.py --workload ml1m_k128 --no-cpu-baseline --no-strong --steps 100 --repeats 3 > $O/a_$ab.json 2> $O/a_$ab.err
  python -c "
import json; j=json.loads(open('$O/a_$ab.json').read().strip().splitlines()[-1]); print('ablate $ab chunk $c', round(j['ms_per_step'],4), 'ms/iter launch', {k: round(v,4) for k,v in j['roofline']['launch_ms_per_side'].items()})" || tail -3 $O/a_$ab.err
done; done
