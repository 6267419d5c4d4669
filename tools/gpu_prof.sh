#!/bin/bash
# kernel-trace profile of bench.py + CPU baseline thread-scaling probe
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof1 -o r1 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof1/bench.log 2>&1
find gpurun_out/prof1 -name "*stats*" | head
python - <<'PY' > gpurun_out/cpu_scaling.txt 2>&1
import sys, os, time
sys.path.insert(0, '.')
import numpy as np
from oracle import oracle as orc
from bpmf_amd import synth
orc.build(native=True)
o = orc.Oracle(fast=True)
M, Mt, T, Tt, nu, nm = synth.ml1m_shaped(seed=42)
for nt in (1, 8, 16, 32, 64, 128, 256):
    o.gibbs(32, M, Mt, T, Tt, nsims=1, burnin=0, nthreads=nt)
    n = 2 if nt == 1 else 6
    r = o.gibbs(32, M, Mt, T, Tt, nsims=n, burnin=0, nthreads=nt)
    print(nt, 'threads', float(np.mean(r['secs'][1:]))*1e3, 'ms/iter', flush=True)
PY
cat gpurun_out/cpu_scaling.txt
