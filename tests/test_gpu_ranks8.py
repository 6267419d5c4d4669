"""Eight ranks without eight GPUs (BASELINE configs[3] and the north star say 8; the boxes have one device).

Everything the 8-rank job does above the nccl* entry points runs for real -- 7-peer grouped send / recv meshes, 8-way
work-balanced ranges, parts = 4 on 8 ranks, the 8-rank all-reduces over both communicators, bench.py's preflight ladder,
watchdog and per-rank record at `--gpus 8` -- with the eight processes sharing cuda:0 through the tests' RCCL double
(tests/rccl_double, MAXR = 8).  Reference pattern being replaced: c++/mpi_isendirecv.h:222-260 (items travel while others
are sampled), c++/mpi_common.h:44-50 (all-reduce of the sums).  Time-boxed: each test well under 90 s on the box."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse as sp

from tests import util
from tests.conftest import ROOT
from tests.test_gpu_multirank import DOUBLE, rel_err, run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["sync", "async"])
def test_eight_ranks_strong_scaling_shape_matches_the_oracle(oracle, tmp_path, mode):
    """bench.py::strong_10Mx1M's set-up at scale 0.002 (20 000 users x 2 000 items x 200 per user = 4 M ratings, K = 32): rank r
    of 8 holds user chunk r and item range r, every exchange cut into 4 parts (BPMF_HIP_OVERLAP=4).  Every replica must hold
    the same bits, the chain must be the oracle's single-process chain, and the communication library must count 8 ranks."""
    nsims, burnin, K = 3, 1, 32
    env = {"BPMF_HIP_OVERLAP": "4"}
    if mode == "async":
        env.update({"BPMF_RCCL_DOUBLE_ASYNC": "1", "GPU_MAX_HW_QUEUES": "24"})      # (see tests/test_gpu_multirank.py::double_mode)
    res = run_ranks(tmp_path, 8, "big", "0.002", K, nsims, burnin, env)
    z = np.load(str(tmp_path / "res_big.matrix.npz"))
    nu, nm = int(z["shape"][0]), int(z["shape"][1])
    M = (z["m0"], z["m1"], z["m2"]); T = (z["t0"], z["t1"], z["t2"])
    assert int(M[0][-1]) == nu * 200
    csc = lambda a, shape: sp.csc_matrix((a[2], a[1], a[0]), shape=shape)
    Mt = util.csc_arrays(csc(M, (nu, nm)).T.tocsc()); Tt = util.csc_arrays(csc(T, (nu, nm)).T.tocsc())
    M = util.csc_arrays(csc(M, (nu, nm))); T = util.csc_arrays(csc(T, (nu, nm)))
    ref = oracle.gibbs(K, M, Mt, T, Tt, nsims=nsims, burnin=burnin, nthreads=max(1, min(os.cpu_count() or 1, 16)))
    doms_u, doms_m = set(), set()
    for r in res:
        assert int(r["nranks"]) == 8
        assert list(r["parts"]) == [4, 4], "parts = 4 was asked for on both sides: %r" % (r["parts"],)
        assert np.allclose(r["rmse"], ref["rmse"], atol=1e-7) and np.allclose(r["rmse_avg"], ref["rmse_avg"], atol=1e-7)
        assert np.allclose(r["norm_u"], ref["norm_u"], rtol=1e-7) and np.allclose(r["norm_m"], ref["norm_m"], rtol=1e-7)
        assert rel_err(r["U"], ref["U"]) < 1e-7 and rel_err(r["V"], ref["V"]) < 1e-7
        assert np.array_equal(r["U"], res[0]["U"]) and np.array_equal(r["V"], res[0]["V"])       # every replica holds the same bits
        doms_u.add(tuple(r["dom_u"])); doms_m.add(tuple(r["dom_m"]))
    # eight disjoint ranges that tile both sides
    for doms, n in ((doms_u, nu), (doms_m, nm)):
        d = sorted(doms)
        assert len(d) == 8 and d[0][0] == 0 and d[-1][1] == n and all(a[1] == b[0] for a, b in zip(d, d[1:]))


def _bench_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith('{"metric"')]
    return json.loads(lines[-1]) if lines else None


def test_bench_gpus8_preflight_ladder_and_per_rank_record():
    """`bench.py --gpus 8` in the declared test set-up (BPMF_BENCH_SHARED_GPU=1: all ranks on device 0, the double as the
    communication library -- the explicit test-only switch that relaxes the one-rank-per-device check).  Rank 5's trial of the
    first rung hangs (test hook): it is killed after BPMF_BENCH_PREFLIGHT_TIMEOUT_S, the eight ranks agree, the second rung
    runs on all eight, and the line carries n_gpus 8, rccl_nranks 8, the ladder with the reason, and eight per-rank records."""
    env = dict(os.environ, BPMF_BENCH_SHARED_GPU="1", BPMF_HIP_RCCL_LIBRARY=DOUBLE, BPMF_BENCH_TEST_HANG_RUNG="mesh+parts+2comms:5",
               BPMF_BENCH_PREFLIGHT_TIMEOUT_S="16", BPMF_RCCL_DOUBLE_TIMEOUT_S="8")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--repeats", "1", "--prewarm-ms", "0",
                        "--no-strong"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    j = _bench_line(r.stdout)
    assert r.returncode == 0 and j is not None and j["value"] and not j.get("error"), (r.stdout[-800:], r.stderr[-3000:])
    assert j["n_gpus"] == 8 and j["rccl_nranks"] == 8 and j["launcher"] == "self"
    x = j["exchange_config"]
    assert x["chosen"] == "mesh+1comm"
    assert [l["config"] for l in x["ladder"]] == ["mesh+parts+2comms", "mesh+1comm"] and not x["ladder"][0]["ok"] and x["ladder"][1]["ok"]
    assert "rank 5" in x["ladder"][0]["why"]
    pr = j["per_rank"]
    assert [p["rank"] for p in pr] == list(range(8))
    assert sum(p["columns"]["movs"] for p in pr) == 3706 and sum(p["columns"]["users"] for p in pr) == 6040 * 8      # weak scaling: 8 x the users
    assert all(set(p["launch_ms_per_side"]) == {"movs", "users"} and "exchange_and_rest_ms" in p and p["device"] == 0 for p in pr)
    assert j["parity"]["value"] is None and "N > 1" in j["parity"]["reason"]
