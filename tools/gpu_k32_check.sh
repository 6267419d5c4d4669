#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/k32; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
  timeout 300 python bench.py --workload ml1m --no-cpu-baseline --no-strong > $O/b_$i.json 2> $O/b_$i.err
  python -c "
import json; j=json.loads(open('$O/b_$i.json').read().strip().splitlines()[-1]); print('ml1m', round(j['value']/1e6,2), 'M/s', round(j['ms_per_step'],4), 'ms/iter launch', {k: round(v,4) for k,v in j['roofline']['launch_ms_per_side'].items()}, 'frac', round(j['roofline']['frac'],3))"
done
