#!/bin/bash
# K = 128: host timeline of the asynchronous path
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/k128; mkdir -p $O
BPMF_HIP_CHUNK=${CHUNK:-384} BPMF_HIP_TRACE=1 timeout 600 python bench.py --workload ml1m_k128 --no-cpu-baseline --no-strong --steps 20 --warmup 5 --repeats 1 --prewarm-ms 0 > $O/trace.json 2> $O/trace.err
tail -75 $O/trace.err | head -70
