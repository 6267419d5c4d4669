"""The BPMF_REDUCE formulation (Sys::preComputeMuLambda, c++/sample.cpp:234-246; c++/mpi_reduce.h) on
the GPU against the oracle's restatement of that build -- see tests/_reduce_worker.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("K", [8, 32, 64])
def test_reduce_formulation_matches_oracle(K):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_reduce_worker.py"), str(K)], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and "REDUCE-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_reduce_rejects_fp32():
    import bpmf_amd
    import numpy as np
    eng = bpmf_amd.HipEngine(128, dtype="f32")
    cp = np.zeros(5, np.int64)
    a = eng.side_create(4, 4, cp, np.zeros(0, np.int32), np.zeros(0), 0.0)
    b = eng.side_create(4, 4, cp, np.zeros(0, np.int32), np.zeros(0), 0.0)
    with pytest.raises(Exception, match="fp64"):
        eng.sys_set_reduce(a, b, True)
    eng.side_destroy(a); eng.side_destroy(b); eng.close()
