"""The N > 1 path on CPU: world_size 2, gloo backend, the oracle as column sampler (test-only
engine).  Checks that sharding + exchange + all-reduce of [prod | sum | norm] reproduce the
single-process run: same factors on every rank, same RMSE trace (to summation order)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import util
from tests.conftest import ROOT


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def run_job(tmp_path, dataset, K, nsims, burnin, world=2):
    port = free_port()
    out = str(tmp_path / "res")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), dataset, str(K),
                                       str(nsims), str(burnin), out], env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=600) == 0
    return [np.load(out + ".rank%d.npz" % r) for r in range(world)]


@pytest.mark.parametrize("dataset,K,nsims,burnin,world", [("tiny", 8, 4, 1, 2), ("ml100k", 8, 3, 1, 2), ("ml100k", 8, 2, 0, 3),
                                                           ("ml100k", 8, 2, 0, 8)])        # BASELINE configs[3]'s rank count
def test_sharded_equals_single_process(oracle, tmp_path, dataset, K, nsims, burnin, world):
    res = run_job(tmp_path, dataset, K, nsims, burnin, world)
    M, Mt, T, Tt, nu, nm = util.tiny() if dataset == "tiny" else util.ml100k()
    ref = oracle.gibbs(K, M, Mt, T, Tt, nsims=nsims, burnin=burnin)
    for r in res:
        # every rank ends with the full, identical factor matrices
        assert np.array_equal(r["U"], res[0]["U"]) and np.array_equal(r["V"], res[0]["V"])
        assert np.allclose(r["U"], ref["U"], rtol=1e-9, atol=1e-11) and np.allclose(r["V"], ref["V"], rtol=1e-9, atol=1e-11)
        assert np.allclose(r["rmse"], ref["rmse"], atol=1e-9) and np.allclose(r["rmse_avg"], ref["rmse_avg"], atol=1e-9)
        assert np.allclose(r["norm_u"], ref["norm_u"], rtol=1e-10) and np.allclose(r["norm_m"], ref["norm_m"], rtol=1e-10)
        assert abs(float(r["final"]) - ref["final_rmse_avg"]) < 1e-9


def test_connectivity_aware_exchange(oracle, tmp_path):
    """SURVEY 8f rank 2 (c++/assign.cpp:204-241): on a two-community matrix a column only travels to
    the rank that reads it.  Every rank must still own exact samples and report the global RMSE;
    replica columns nobody on the rank reads are left untouched (zero, the initial state)."""
    from bpmf_amd import synth
    from bpmf_amd.dist import connectivity, conn_lists
    K, nsims, burnin, world = 8, 3, 1, 2
    res = run_job(tmp_path, "blocks", K, nsims, burnin, world)
    M, Mt, T, Tt, nu, nm = util.blocks()
    ref = oracle.gibbs(K, M, Mt, T, Tt, nsims=nsims, burnin=burnin)
    bm = synth.balanced_ranges(M[0], world); bu = synth.balanced_ranges(Mt[0], world)
    need_u = connectivity(M, bm, T); need_m = connectivity(Mt, bu)
    for rank, r in enumerate(res):
        assert r["conn_used"].all()
        assert list(r["dom_m"]) == [bm[rank], bm[rank + 1]] and list(r["dom_u"]) == [bu[rank], bu[rank + 1]]
        assert np.allclose(r["rmse"], ref["rmse"], atol=1e-9) and np.allclose(r["rmse_avg"], ref["rmse_avg"], atol=1e-9)
        assert np.allclose(r["norm_u"], ref["norm_u"], rtol=1e-10) and np.allclose(r["norm_m"], ref["norm_m"], rtol=1e-10)
        for X, Xref, dom, need in ((r["U_replica"], ref["U"], r["dom_u"], need_u[rank]), (r["V_replica"], ref["V"], r["dom_m"], need_m[rank])):
            held = np.zeros(len(X), bool); held[dom[0]:dom[1]] = True; held[need] = True
            assert np.allclose(X[held], Xref[held], rtol=1e-9, atol=1e-11)          # owned or read here: current
            assert (~held).sum() > len(X) // 4 and np.all(X[~held] == 0.0)          # never shipped during the loop
        # what is handed out is the FULL factor on every rank (users.bcast() / movies.bcast(), c++/bpmf.cpp:217-218)
        assert np.allclose(r["U"], ref["U"], rtol=1e-9, atol=1e-11) and np.allclose(r["V"], ref["V"], rtol=1e-9, atol=1e-11)
        assert np.array_equal(r["U"], res[0]["U"]) and np.array_equal(r["V"], res[0]["V"])
    # the lists mirror each other: what q sends to r is what r expects from q, in the same order
    for need, bounds in ((need_u, bu), (need_m, bm)):
        L = [conn_lists(need, bounds, r) for r in range(world)]
        for r in range(world):
            assert L[r][0][r + 1] == L[r][0][r] and L[r][2][r + 1] == L[r][2][r]      # nothing to self
            for q in range(world):
                assert np.array_equal(L[q][1][L[q][0][r]:L[q][0][r + 1]], L[r][3][L[r][2][q]:L[r][2][q + 1]])


def test_balanced_ranges_cover_and_balance():
    from bpmf_amd import synth
    M, Mt, T, Tt, nu, nm = util.ml100k()
    for parts in (1, 2, 3, 8):
        b = synth.balanced_ranges(M[0], parts)
        assert b[0] == 0 and b[-1] == nm and all(x <= y for x, y in zip(b, b[1:]))
        work = [(M[0][hi] - M[0][lo]) + 10 * (hi - lo) for lo, hi in zip(b, b[1:])]
        assert max(work) < 1.25 * (sum(work) / parts) + 600
        lo, hi = b[0], b[1]
        sl = synth.slice_cols(M, lo, hi)
        assert sl[0][0] == 0 and sl[0][-1] == M[0][hi] - M[0][lo] and len(sl[1]) == sl[0][-1]
