#!/bin/bash
# the `bpmf` executable on the ChEMBL-shaped synthetic matrix at K = 64 (written as .sdm files first)
cd "$GRAFT_REPO_ROOT"
python - <<PY
import sys, os
sys.path.insert(0, ".")
from bpmf_amd import synth, io
M, Mt, T, Tt, nu, nm = synth.ratings(483500, 5775, 1023952, seed=42, real_valued=True)
os.makedirs("/tmp/chembl", exist_ok=True)
io.write_sparse("/tmp/chembl/train.sdm", nu, nm, M)
io.write_sparse("/tmp/chembl/test.sdm", nu, nm, T)
PY
bpmf_amd/bpmf -n /tmp/chembl/train.sdm -p /tmp/chembl/test.sdm -i ${1:-12} -b 5 -d 64 2>&1 | tail -${2:-8}
