#!/bin/bash
# A/B of the pair launch (BPMF_HIP_PAIR=1|0) on the headline workload, interleaved: tools/ab_pair.sh [steps...]
cd "$GRAFT_REPO_ROOT"
for st in ${@:-20 500}; do for rep in 1 2; do for p in 1 0; do BPMF_HIP_PAIR=$p timeout 200 python bench.py --steps $st --warmup 5 --no-strong --no-cpu-baseline --no-bpmf-exe 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']
print('pair=$p steps=$st', round(j['value']/1e6,2), 'M  ms/step', round(j['ms_per_step'],5), 'min', round(j['ms_per_step_min'],5), 'launch', {k: round(v, 5) for k, v in r['launch_ms_per_side'].items()}, r['kernel'], 'rmse', round(j['rmse'], 6))"; done; done; done
