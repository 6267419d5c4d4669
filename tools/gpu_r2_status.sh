#!/bin/bash
# one line per single-GPU workload (same session): samples/s, ms per iteration, sampler launch ms per side
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/status; mkdir -p $O
for w in ml1m ml1m_k64 chembl ml1m_k128; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --no-strong > $O/bench_$w.json 2> $O/bench_$w.err
  python -c "
import json; j=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', round(j['value']/1e6,2), 'M/s', round(j['ms_per_step'],4), 'ms/iter launch', {k: round(v,4) for k,v in j['roofline']['launch_ms_per_side'].items()}, 'frac', round(j['roofline']['frac'],3))"
done
