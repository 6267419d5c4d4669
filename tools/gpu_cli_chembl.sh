#!/bin/bash
# the `bpmf` executable on the ChEMBL-shaped matrix (big side: workgroup statistics + head start), -g 1 as well
cd "$GRAFT_REPO_ROOT"; D=/tmp/chembl_cli; mkdir -p $D
python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from bpmf_amd import synth, io as bio
M, Mt, T, Tt, nu, nm = synth.ratings(483500, 5775, 1023952, seed=7, real_valued=True)
bio.write_sparse('/tmp/chembl_cli/train.sdm', nu, nm, M)
bio.write_sparse('/tmp/chembl_cli/test.sdm', nu, nm, T)
print("written", nu, nm, len(M[2]), len(T[2]))
PY
cd $D
timeout 600 $GRAFT_REPO_ROOT/bpmf_amd/bpmf -i 8 -b 2 -d 64 -n train.sdm -p test.sdm 2>&1 | tail -12
mkdir -p g; cd g
timeout 600 $GRAFT_REPO_ROOT/bpmf_amd/bpmf -i 8 -b 2 -d 64 -g 1 -n ../train.sdm -p ../test.sdm 2>&1 | tail -3; tail -4 bpmf_0.out
