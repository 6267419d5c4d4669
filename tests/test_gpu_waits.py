"""Every in-kernel wait is bounded and a wait that gives up is an ERROR, not a silent continuation:
  * the stateful loop with a quarter of the CUs (HSA_CU_MASK / ROC_GLOBAL_CU_MASK): the gate workgroup,
    the waiting item workgroups and the statistics waves must still all make progress -- same chain as the oracle;
  * a host worker that stalls longer than BPMF_HIP_WAIT_TIMEOUT_MS (test hook BPMF_HIP_TEST_STALL_WORKER_MS):
    the gate gives up, the sampler runs on stale parameters, and the library reports
    BPMF_HIP_ENODEV "device wait timed out" instead of handing out that half-iteration.
Both run in a subprocess (the limits are read once per process)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHAIN = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import bpmf_amd
from oracle.oracle import Oracle
from tests import util
K = int(sys.argv[1]); nsims = int(sys.argv[2])
M, Mt, T, Tt, nu, nm = util.ml100k()
eng = bpmf_amd.HipEngine(K)
res = bpmf_amd.gibbs(eng, M, Mt, T, nu, nm, nsims=nsims, burnin=2)
ref = Oracle().gibbs(K, M, Mt, T, Tt, nsims=nsims, burnin=2)
err = max(np.abs(res["U"] - ref["U"]).max(), np.abs(res["V"] - ref["V"]).max())
assert err < 1e-6 * max(1.0, np.abs(ref["U"]).max()), err
assert np.allclose(res["rmse"], ref["rmse"], atol=1e-6)
eng.close()
print("chain ok", err)
""" % ROOT


def _run(code, args, env_extra, timeout=600):
    env = dict(os.environ); env.update(env_extra)
    return subprocess.run([sys.executable, "-c", code] + [str(a) for a in args], env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("K", [32, 64])
def test_stateful_loop_with_a_quarter_of_the_cus(K):
    mask = "0x" + "f" * 16                                       # 64 of 256 CUs
    r = _run(CHAIN, [K, 6], {"HSA_CU_MASK": "0:0-63", "ROC_GLOBAL_CU_MASK": mask, "BPMF_HIP_WAIT_TIMEOUT_MS": "20000"})
    assert r.returncode == 0 and "chain ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


STALL = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import bpmf_amd
from tests import util
M, Mt, T, Tt, nu, nm = util.ml100k()
from bpmf_amd.sys import Sys
eng = bpmf_amd.HipEngine(32)
movies = Sys("movs", eng, M, nm, nu, T=T); users = Sys("users", eng, Mt, nu, nm)
try:
    # the pipelined loop of bench.py / the bpmf executable: the next half-iteration of a side is enqueued
    # while its host worker still owes the hyper-parameters -- the sampler's gate workgroup waits for them
    for i in range(4):
        movies.sample(users); users.sample(movies)
    movies.refresh(); users.refresh()
except RuntimeError as e:
    print("raised:", e)
    sys.exit(0 if "device wait timed out" in str(e) else 3)
sys.exit(4)
""" % ROOT


def test_stalled_host_worker_is_an_error_not_a_stale_chain():
    r = _run(STALL, [], {"BPMF_HIP_WAIT_TIMEOUT_MS": "150", "BPMF_HIP_TEST_STALL_WORKER_MS": "1500"})
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
