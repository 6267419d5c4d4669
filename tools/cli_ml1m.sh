#!/bin/bash
# the `bpmf` executable on the ML-1M-shaped synthetic matrix (written as .sdm files first)
cd "$GRAFT_REPO_ROOT"
python - <<PY
import sys, os
sys.path.insert(0, ".")
from bpmf_amd import synth, io
M, Mt, T, Tt, nu, nm = synth.ml1m_shaped(seed=42)
os.makedirs("/tmp/ml1m", exist_ok=True)
io.write_sparse("/tmp/ml1m/train.sdm", nu, nm, M)      # rows = users, one column per movie
io.write_sparse("/tmp/ml1m/test.sdm", nu, nm, T)
PY
bpmf_amd/bpmf -n /tmp/ml1m/train.sdm -p /tmp/ml1m/test.sdm -i ${1:-12} -b 5 -d 32 2>&1 | tail -${2:-18}
