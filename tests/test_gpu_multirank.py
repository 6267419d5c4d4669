"""The library's OWN multi-rank path with nranks >= 2 on the one GPU there is.

RCCL refuses two ranks per device, so the ranks load the tests' RCCL double (tests/rccl_double: the nccl* entry points
for processes / threads that share a GPU, a shared-memory rendez-vous underneath) through BPMF_HIP_RCCL_LIBRARY.
Everything above those entry points is the product: bpmf_hip_ctx_comm_init, the ranges, the mesh all-gather-v of
launch_impl.h (never executed with a peer before round 3), parts, the second communicator, the connectivity-aware
lists, BPMF_REDUCE's grouped reduce, the fp32 context, `bpmf -g 2` and `bench.py --gpus 2`.  Reference behaviour being
matched: c++/mpi_isendirecv.h:222-260 (items travel while others are sampled), c++/mpi_common.h:44-50 (all-reduce of
the sums), c++/mpi_reduce.h:24-47.  Every run is compared with the ORACLE's single-process chain: the samples do not
depend on the rank count beyond the order of the all-reduced sums."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tests import util
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

DOUBLE = os.path.join(ROOT, "tests", "rccl_double", "librccl_double.so")


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def run_ranks(tmp_path, nranks, case, dataset, K, nsims, burnin, env_extra=None):
    assert os.path.exists(DOUBLE), "tests/rccl_double/librccl_double.so is missing: __graft_entry__.build() makes it"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / ("res_%s" % case))
    procs = []
    for rank in range(nranks):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(nranks), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   BPMF_HIP_RCCL_LIBRARY=DOUBLE, BPMF_RCCL_DOUBLE_TIMEOUT_S="120")
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_mr_worker.py"), case, dataset, str(K), str(nsims),
                                       str(burnin), out], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    errs = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        errs.append((p.returncode, so[-500:], se[-3000:]))
    for rc, so, se in errs:
        assert rc == 0 and "MR-OK" in so, (so, se)
    return [np.load(out + ".rank%d.npz" % r) for r in range(nranks)]


def check_against_oracle(oracle, res, dataset, K, nsims, burnin, tol=1e-7, owned_only=False):
    data = {"ml100k": util.ml100k, "blocks": util.blocks, "heavy": lambda: util.synthetic(700, 500, 30000, seed=3, heavy=(7, 650))}[dataset]
    M, Mt, T, Tt, nu, nm = data()
    ref = oracle.gibbs(K, M, Mt, T, Tt, nsims=nsims, burnin=burnin)
    for r in res:
        assert int(r["nranks"]) == len(res)
        assert np.allclose(r["rmse"], ref["rmse"], atol=tol) and np.allclose(r["rmse_avg"], ref["rmse_avg"], atol=tol)
        assert np.allclose(r["norm_u"], ref["norm_u"], rtol=10 * tol) and np.allclose(r["norm_m"], ref["norm_m"], rtol=10 * tol)
        if owned_only:
            for X, Xref, dom in ((r["U"], ref["U"], r["dom_u"]), (r["V"], ref["V"], r["dom_m"])):
                assert rel_err(X[dom[0]:dom[1]], Xref[dom[0]:dom[1]]) < tol
        else:
            assert rel_err(r["U"], ref["U"]) < tol and rel_err(r["V"], ref["V"]) < tol
            assert np.array_equal(r["U"], res[0]["U"]) and np.array_equal(r["V"], res[0]["V"])       # every replica holds the same bits
    return ref


@pytest.mark.parametrize("nranks,K,env", [
    (2, 32, {}),                                           # the default: mesh of grouped send / recv, second communicator
    (3, 16, {}),                                           # three uneven ranges
    (2, 32, {"BPMF_HIP_EXCHANGE": "bcast"}),              # one broadcast per owner
    (2, 32, {"BPMF_HIP_COMM_STREAMS": "1"}),              # one communicator: statistics all-reduce on the main stream
    (2, 64, {}),                                           # slab form, unfused (sharded) launch
])
def test_mesh_exchange_between_two_ranks(oracle, tmp_path, nranks, K, env):
    nsims, burnin = 4, 1
    res = run_ranks(tmp_path, nranks, "mesh", "ml100k", K, nsims, burnin, env)
    check_against_oracle(oracle, res, "ml100k", K, nsims, burnin)


@pytest.mark.parametrize("nranks,parts", [(2, 2), (2, 4), (3, 3)])
def test_parts_overlap_between_ranks(oracle, tmp_path, nranks, parts):
    """bpmf_hip_side_set_overlap with a peer: the per-part sub-ranges of every rank are all-gathered once, part c travels
    on the exchange stream while part c + 1 is sampled; `heavy`: a 650-rating column that is cut into chunks."""
    nsims, burnin = 4, 1
    res = run_ranks(tmp_path, nranks, "parts", "heavy", 32, nsims, burnin, {"BPMF_HIP_OVERLAP": str(parts)})
    check_against_oracle(oracle, res, "heavy", 32, nsims, burnin)


def test_auto_overlap_decision_is_rank_invariant(oracle, tmp_path):
    """the automatic switch to parts must be taken from data every rank holds (a rank-local width near the
    threshold made some ranks enter the collective set_overlap and others not: a hang).  Threshold lowered so that the
    narrowest-range rule flips on for ml100k's users (3 uneven ranges) and stays off for its movies."""
    nsims, burnin = 3, 1
    res = run_ranks(tmp_path, 3, "auto", "ml100k", 16, nsims, burnin, {"BPMF_HIP_OVERLAP_MIN_KB": "100"})
    check_against_oracle(oracle, res, "ml100k", 16, nsims, burnin)


def test_connectivity_lists_between_ranks(oracle, tmp_path):
    """k_pack_cols -> grouped send / recv per peer -> k_unpack_cols between DIFFERENT ranks (c++/assign.cpp:204-241)."""
    nsims, burnin = 4, 1
    res = run_ranks(tmp_path, 2, "conn", "blocks", 32, nsims, burnin)
    for r in res:
        assert r["conn_used"].all()
    check_against_oracle(oracle, res, "blocks", 32, nsims, burnin, owned_only=True)


def test_fp32_context_between_ranks(tmp_path, hip_engine_factory):
    """K = 128 fp32 sharded over two ranks == the same context on one rank, up to the order of the all-reduced sums."""
    import bpmf_amd
    nsims, burnin = 3, 1
    res = run_ranks(tmp_path, 2, "f32", "ml100k", 128, nsims, burnin)
    M, Mt, T, Tt, nu, nm = util.ml100k()
    eng = bpmf_amd.HipEngine(128, dtype="f32")
    one = bpmf_amd.gibbs(eng, M, Mt, T, nu, nm, nsims=nsims, burnin=burnin)
    for r in res:
        assert np.allclose(r["rmse"], one["rmse"], atol=1e-4)
        assert rel_err(r["U"], one["U"]) < 2e-3 and rel_err(r["V"], one["V"]) < 2e-3
        assert np.array_equal(r["U"], res[0]["U"])
    eng.close()


@pytest.mark.parametrize("K", [16, 64])
def test_reduce_formulation_between_ranks(oracle, tmp_path, K):
    """BPMF_REDUCE with two ranks: every rank precomputes the parts of ALL columns of the other side from its own fresh
    columns, the grouped ncclReduce sums them onto the owners (c++/mpi_reduce.h:24-47): against the oracle's restatement
    of that build with two simulated ranks and the same ranges."""
    from bpmf_amd import synth
    nsims, burnin = 3, 1
    res = run_ranks(tmp_path, 2, "reduce", "heavy", K, nsims, burnin)
    M, Mt, T, Tt, nu, nm = util.synthetic(700, 500, 30000, seed=3, heavy=(7, 650))
    bm, bu = synth.balanced_ranges(M[0], 2), synth.balanced_ranges(Mt[0], 2)
    ref = oracle.gibbs_reduce(K, M, Mt, T, alpha=2.0, nsims=nsims, burnin=burnin, bounds_m=bm, bounds_u=bu)
    for r in res:
        assert int(r["dom_m"][0]) in bm and int(r["dom_u"][0]) in bu
        assert rel_err(r["U"], ref["U"]) < 1e-8 and rel_err(r["V"], ref["V"]) < 1e-8
        assert np.allclose(r["rmse"], ref["rmse"], atol=1e-9)
        assert np.array_equal(r["U"], res[0]["U"]) and np.array_equal(r["V"], res[0]["V"])


@pytest.mark.parametrize("k,parts,tol", [(2, 3, 1e-2), (1, 2, 6e-2)])
def test_bounded_staleness_exchange_is_a_mild_relaxation(oracle, tmp_path, k, parts, tol):
    """SURVEY 8 f4, third variant (c++/bpmf_gaspi.h:91-104 send throttling, c++/mpi_allreduce.h:134-175 stale blocks):
    BPMF_HIP_STALE=k lets a part of a side travel every (k + 1)-th half-iteration only.  Property test, as the reference's
    own relaxations have no exact answer: the chain still converges -- final averaged RMSE of a 40-iteration run on
    MovieLens-100K (20 burn-in) close to the exact chain's 0.9397 (measured: 0.9415 with k = 2 over three parts, 0.980
    with k = 1 over two: staleness costs accuracy per iteration, as the reference's authors found) and far below the
    mean predictor's 1.1537 -- it is a different chain (the switch did something), and after the closing full
    exchange every replica holds the same bits."""
    nsims, burnin, K = 40, 20, 16
    res = run_ranks(tmp_path, 2, "stale", "ml100k", K, nsims, burnin, {"BPMF_HIP_STALE": str(k), "BPMF_HIP_OVERLAP": str(parts)})
    M, Mt, T, Tt, nu, nm = util.ml100k()
    ref = oracle.gibbs(K, M, Mt, T, Tt, nsims=nsims, burnin=burnin)
    for r in res:
        assert abs(float(r["final"]) - ref["final_rmse_avg"]) < tol
        assert float(r["final"]) < 1.0
        assert rel_err(r["U"], ref["U"]) > 1e-6                     # not the exact chain
        assert np.array_equal(r["U"], res[0]["U"]) and np.array_equal(r["V"], res[0]["V"])
    # k = 0 through the same switch is the exact chain
    res0 = run_ranks(tmp_path, 2, "stale0", "ml100k", K, 4, 1, {"BPMF_HIP_STALE": "0", "BPMF_HIP_OVERLAP": str(parts)})
    check_against_oracle(oracle, res0, "ml100k", K, 4, 1)


def test_bpmf_g2_rank_threads_share_the_gpu(tmp_path):
    """`bpmf -g 2`: two rank THREADS of one process, both on device 0 (BPMF_HIP_DEVICES=0,0), the double as the
    communication library.  With BPMF_ASSIGN=contiguous the column ids -- hence the RNG streams -- are those of the
    single-GPU run: every sample, Pavg / Pm2 and the posterior files must agree with plain `bpmf` up to the order of the
    all-reduced sums; both rank logs carry the same RMSE lines and `nprocs: 2`."""
    from bpmf_amd import io as bio
    exe = os.path.join(ROOT, "bpmf_amd", "bpmf")
    train, test = os.path.join(util.GOLDEN, "ml100k-train.mtx.gz"), os.path.join(util.GOLDEN, "ml100k-test.mtx.gz")
    runs = {}
    for name, extra_args, extra_env in (("plain", [], {}), ("g2", ["-g", "2"], {"BPMF_HIP_DEVICES": "0,0", "BPMF_HIP_RCCL_LIBRARY": DOUBLE,
                                                                                 "BPMF_ASSIGN": "contiguous"})):
        d = tmp_path / name
        (d / "out").mkdir(parents=True)
        r = subprocess.run([exe, "-n", train, "-p", test, "-i", "6", "-b", "2", "-d", "16", "-v", "-o", str(d / "out")] + extra_args,
                           cwd=str(d), env=dict(os.environ, **extra_env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
        runs[name] = d
        if name == "plain":
            plain_stdout = r.stdout
    log0, log1 = open(runs["g2"] / "bpmf_0.out").read(), open(runs["g2"] / "bpmf_1.out").read()
    assert "nprocs: 2" in log0 and "nprocs: 2" in log1
    import re
    pick = lambda text: re.findall(r"RMSE: ([0-9.]+)\s+avg RMSE: ([0-9.]+)", text)
    assert pick(log0) == pick(log1) and len(pick(log0)) == 6
    assert pick(log0) == pick(plain_stdout)                          # (4 printed decimals)
    for f in ["U-%d.ddm" % i for i in range(6)] + ["V-%d.ddm" % i for i in range(6)] + ["U-mu.ddm", "V-mu.ddm"]:
        a = bio.read_dense(str(runs["plain"] / "out" / f)); b = bio.read_dense(str(runs["g2"] / "out" / f))
        assert a.shape == b.shape and rel_err(b, a) < 1e-8, f
    a = bio.read_sparse(str(runs["plain"] / "out" / "Pavg.sdm"))[2]; b = bio.read_sparse(str(runs["g2"] / "out" / "Pavg.sdm"))[2]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.allclose(a[2], b[2], atol=1e-8)


def test_bench_self_launches_two_ranks_on_the_shared_gpu():
    """bench.py --gpus 2 without a launcher: it starts the two ranks itself; on this one-GPU box that is only allowed in
    the declared test set-up (BPMF_BENCH_SHARED_GPU=1 + the double); the line says n_gpus 2, rccl_nranks 2."""
    env = dict(os.environ, BPMF_BENCH_SHARED_GPU="1", BPMF_HIP_RCCL_LIBRARY=DOUBLE, BPMF_BENCH_STRONG_SCALE="0.01")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--repeats", "1", "--prewarm-ms", "0",
                        "--strong-steps", "8"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert j["n_gpus"] == 2 and j["rccl_nranks"] == 2 and j["launcher"] == "self" and len(j["per_rank"]) == 2
    assert j["env"].get("BPMF_BENCH_SHARED_GPU") == "1"
    s = j["strong_10Mx1M"]
    assert s["n_gpus"] == 2 and s["rccl_nranks"] == 2 and s["spot_check"]["ok"], s
    # and the same matrices on one rank: same RMSE after the same number of iterations (the chain does not depend on N)
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--repeats", "1", "--prewarm-ms", "0",
                         "--no-cpu-baseline", "--strong-steps", "8"], cwd=ROOT, env=dict(env, BPMF_BENCH_SHARED_GPU="0"), stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, text=True, timeout=1200)
    assert r1.returncode == 0, r1.stderr[-3000:]
    j1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert j1["n_gpus"] == 1 and j1["rccl_nranks"] == 1 and j1["launcher"] == "none"
    assert abs(j1["strong_10Mx1M"]["rmse"] - s["rmse"]) < 1e-6 and j1["strong_10Mx1M"]["spot_check"]["ok"]


def test_bench_refuses_more_ranks_than_devices():
    """`bench.py --gpus 2` on a one-GPU box exits non-zero with a clear message instead of printing n_gpus: 1."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices present")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BPMF_BENCH_SHARED_GPU"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0 and "--gpus 2" in r.stderr and "device" in r.stderr
    assert '{"metric"' not in r.stdout


def test_bench_refuses_ablate_in_the_environment():
    env = dict(os.environ, BPMF_HIP_ABLATE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--no-strong", "--no-cpu-baseline"], cwd=ROOT,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0 and "BPMF_HIP_ABLATE" in r.stderr and '{"metric"' not in r.stdout
