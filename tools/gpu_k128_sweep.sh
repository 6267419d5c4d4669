#!/bin/bash
# K = 128 fp32: parity, then launch times over chunk sizes (+ stamps at the default)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/k128; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_f32.py -m gpu -x -q 2>&1 | tail -4
for c in ${CHUNKS:-0 256 384 512}; do
  BPMF_HIP_CHUNK=$c timeout 300 python bench.py --workload ml1m_k128 --no-cpu-baseline --no-strong > $O/b_$c.json 2> $O/b_$c.err
  python -c "
import json; j=json.loads(open('$O/b_$c.json').read().strip().splitlines()[-1]); print('chunk $c', round(j['value']/1e6,2), 'M/s', round(j['ms_per_step'],4), 'ms/iter launch', {k: round(v,4) for k,v in j['roofline']['launch_ms_per_side'].items()}, 'frac', round(j['roofline']['frac'],3))"
done
BPMF_HIP_STAMPS=1 timeout 300 python bench.py --workload ml1m_k128 --no-cpu-baseline --no-strong --steps 20 --warmup 5 --repeats 1 > $O/stamps.json 2> $O/stamps.err
grep -i "stamp" $O/stamps.err | head
