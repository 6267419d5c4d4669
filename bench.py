#!/usr/bin/env python
"""bench.py -- user+item column samples/sec per Gibbs iteration on MI355X.

One "step" = one full Gibbs iteration of the reference's main loop
(c++/bpmf.cpp:182-195): movies.sample(users); users.sample(movies);
movies.predict(users) -- both host hyper-parameter draws, the device->host
reductions and the RMSE evaluation are inside the timed region, exactly what the
reference's `items/sec` covers.  value = (N_users + N_movies) * steps / seconds.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ml1m|ml1m_k64|chembl|ml1m_k128]

Workloads (BASELINE.json configs; synthetic stand-ins, the reference ships only ML-100K):
  ml1m        configs[1]  6040 x 3706, 1 000 209 ratings (90/10 split), K = 32 fp64   <- the headline / default
  ml1m_k64                the same matrix, K = 64 fp64
  chembl      configs[2]  483 500 x 5 775, 1 023 952 real-valued activities, K = 64 fp64
  ml1m_k128   configs[4]  the ML-1M shape, K = 128, fp32 factors (mixed-precision path)
N > 1 (one rank per GPU, torch.distributed launcher, RCCL inside the library): weak scaling of the
selected workload (N times the users and ratings) AND, next to it, the north star's strong-scaling
experiment as the sub-record `strong_10Mx1M` (configs[3]: the SAME device-generated 10M x 1M x 200
matrix at every N, N = 1 included; --no-strong skips it).

Timing: W warm-up steps, then untimed steps until >= 50 ms have gone by since start-up (`prewarm_ms`:
the clocks ramp for the first ~20 ms after an idle period), then R blocks of EXACTLY K steps, each
bracketed by barrier + synchronize; `ms_per_step` / `value` are the MEDIAN block (min / max beside it).

Prints ONE JSON line (rank 0) with `roofline` (the dominant kernel = the per-column sampler, timed
with HIP events on its own stream inside the library) and `cpu_baseline` (oracle/cpu_baseline.py: the
oracle's OpenMP build on this box's host cores, in a process of its own).
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

# peaks (/opt/skills/guides/MI355X_MICROARCH.md; fp64 from the MI355X data sheet: vector = matrix = 78.6 TF)
HBM_PEAK_GBS = 8000.0
FP64_PEAK_TFLOPS = 78.6
FP32_PEAK_TFLOPS = 157.3
LDS_PER_CU = 160 * 1024

WORKLOADS = {
    #            K    dtype  dominant kernel            LDS bytes / workgroup, workgroups resident per CU (launch bounds, LDS)
    "ml1m":      (32, "f64", "k_sample1<32>",           (32 * 34 + 4 * 32 + 2) * 8, 12),
    "ml1m_k64":  (64, "f64", "k_sample1<64>",           None, None),
    "chembl":    (64, "f64", "k_sample_pf<64,NB> + k_sample1s<64>", None, None),
    "ml1m_k128": (128, "f32", "k_sample_wg<128,float>", None, None),
}


def algorithmic_bytes(nnz, ncols, K, s=8):
    """SURVEY 8(d), one half-iteration (= one launch of the sampler): per rating a row index,
    a value and one K-vector; per column the K-vector written back + its column pointer."""
    return nnz * (4 + s + K * s) + ncols * (K * s + 8)


def algorithmic_flops(nnz, ncols, K):
    return nnz * (K * (K + 1) + 2 * K) + ncols * (K ** 3 / 3.0 + 4 * K * K + 3 * K)


def profiled(workload):
    """Per-launch PMC figures of the sampler from the committed rocprofv3 passes of this same command
    (profiles/r*_pmc_<workload>.txt, newest round; separate --pmc passes): HBM-side bytes and the LDS
    bank-conflict rate.  Not a live measurement: counters need rocprofv3.  HBM correction as
    MI355X_MICROARCH.md prescribes, calibrated on this kernel's own access pattern
    (tools/probes/fetch_calib.hip: 1 GiB read once with the sampler's 16-byte gathers reports
    0.50 GiB, 1 GiB written reports 1.00 GiB): traffic = 2 * FETCH_SIZE(KB) + WRITE_SIZE(KB)."""
    names = {"ml1m": "sampler"}
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.txt" % workload)) +
                   glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_%s.txt" % names.get(workload, "-"))))
    if not files:
        return None, None, None
    vals = {}
    for line in open(files[-1]):
        f = line.replace("avg=", "avg= ").split()
        if len(f) >= 4 and f[2] == "avg=":
            try:
                vals[f[1]] = float(f[3])
            except ValueError:
                pass
    traffic = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0 if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals else None
    conflict = None
    if vals.get("SQ_LDS_IDX_ACTIVE") or vals.get("SQ_ACTIVE_INST_LDS"):
        conflict = vals.get("SQ_LDS_BANK_CONFLICT", 0.0) / (vals.get("SQ_LDS_IDX_ACTIVE") or vals.get("SQ_ACTIVE_INST_LDS"))
    return traffic, conflict, os.path.basename(files[-1])


def cpu_baseline(M, Mt, T, Tt, K, nusers, nmovies, budget_s=12.0):
    """oracle/cpu_baseline.py in a process of its own (placement: one thread per physical core, spread)."""
    fd, path = tempfile.mkstemp(suffix=".npz", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    os.close(fd)
    try:
        arrs = {}
        for name, m in (("M", M), ("Mt", Mt), ("T", T), ("Tt", Tt)):
            for i in range(3):
                arrs["%s%d" % (name, i)] = m[i]
        np.savez(path, shape=np.array([nusers, nmovies]), **arrs)
        env = dict(os.environ)
        env.update({"OMP_PLACES": "cores", "OMP_PROC_BIND": "spread", "OMP_WAIT_POLICY": "active"})
        env.pop("OMP_NUM_THREADS", None)
        try:
            usable = len(os.sched_getaffinity(0))
        except AttributeError:
            usable = os.cpu_count() or 1
        r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--matrix", path, "--K", str(K),
                            "--budget", str(budget_s), "--usable", str(usable)], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            raise RuntimeError("cpu_baseline.py rc=%d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:]))
        return json.loads(line[-1])
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass


def timed_blocks(step_block, fence, steps, dist_max, min_blocks=5, max_blocks=25, budget_s=8.0):
    """R blocks of exactly `steps` steps; returns the block times (max over ranks each)."""
    times = []
    t_all = time.perf_counter()
    while True:
        fence()
        t0 = time.perf_counter()
        step_block(steps)
        fence()
        times.append(dist_max(time.perf_counter() - t0))
        done = len(times) >= max_blocks or (len(times) >= min_blocks and time.perf_counter() - t_all > budget_s)
        if dist_max(1.0 if done else 0.0) > 0.5:          # (every rank takes the same decision)
            return times


def strong_10Mx1M(world, rank, local_rank, steps, scale=1.0):
    """The north star's strong-scaling experiment (configs[3]): 10M x 1M x 200 per user, K = 32, the same
    matrix whatever N; rank r of N holds user chunks / item ranges [8r/N, 8(r+1)/N)."""
    import torch
    import bpmf_amd
    from bpmf_amd.synth_dev import BigMatrix
    from bpmf_amd.sys import Sys
    K, G = 32, 8
    if G % world:
        return {"skipped": "needs a rank count that divides %d" % G}
    t_gen = time.perf_counter()
    dev = torch.device("cuda", local_rank)
    big = BigMatrix(dev, nusers=int(10_000_000 * scale), nitems=int(1_000_000 * scale), groups=G)
    parts = list(range(rank * G // world, (rank + 1) * G // world))
    bnd = big.item_bounds()
    bm = [bnd[r * G // world] for r in range(world)] + [big.NI]
    bu = [big.chunk_range(r * G // world)[0] for r in range(world)] + [big.NU]
    ucp, uri, uva, u0, u1 = big.users_csc(parts)
    mcp, mri, mva, i0, i1 = big.items_csc(parts)
    tcsc = big.test_csc(i0, i1)
    torch.cuda.synchronize()
    gen_s = time.perf_counter() - t_gen
    eng = bpmf_amd.HipEngine(K, device=local_rank)
    comm = None
    if world > 1:
        from bpmf_amd.dist import NativeComm
        comm = NativeComm(eng)
    Sys.nsims, Sys.burnin, Sys.alpha = 10 ** 6, 5, 2.0
    movies = Sys("movs", eng, (mcp, mri, mva), big.NI, big.NU, T=tcsc, dom=(i0, i1), mean_rating=big.mean_rating, comm=comm)
    users = Sys("users", eng, (ucp, uri, uva), big.NU, big.NI, dom=(u0, u1), mean_rating=big.mean_rating, comm=comm)
    if comm is not None:
        comm.register(movies, bm); comm.register(users, bu)

    def fence():
        eng.sync(); torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier(); torch.cuda.synchronize()

    def dist_max(x):
        if world == 1:
            return x
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def block(n):
        for i in range(n):
            movies.sample(users); users.sample(movies)
            if i > 0:
                movies.predict_finish()
            movies.predict_launch(users)
        movies.predict_finish()

    block(2)
    fence()
    base = {sd.name: eng.kernel_ms_sum(sd.side) for sd in (movies, users)}
    t0 = time.perf_counter()
    block(steps)
    fence()
    dt = dist_max(time.perf_counter() - t0)
    out = {"workload": "device-generated %d users x %d items, %d ratings per user, K=32 fp64 (BASELINE configs[3]); the same matrix at every N"
                       % (big.NU, big.NI, big.PER),
           "n_gpus": world, "steps": steps, "ms_per_step": dt / steps * 1e3, "value": (big.NU + big.NI) * steps / dt, "unit": "samples/s",
           "scaling": "strong", "rmse": movies.rmse, "generate_s": gen_s}
    kern = {}
    for sd in (movies, users):
        a1 = eng.kernel_ms_sum(sd.side); a0 = base[sd.name]
        nl = a1[2] - a0[2]
        kern[sd.name] = (a1[0] - a0[0]) / nl if nl > 0 else None
    if kern["movs"] and kern["users"]:
        byt = algorithmic_bytes(movies.local_nnz, i1 - i0, K) + algorithmic_bytes(users.local_nnz, u1 - u0, K)
        ks = (kern["movs"] + kern["users"]) * 1e-3
        out.update({"sampler_ms": {"items_side": kern["movs"], "users_side": kern["users"]},
                    "hbm_achieved_gbs": byt / ks / 1e9, "hbm_frac": byt / ks / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_bytes_per_iteration_this_rank": byt,
                    # what one Gibbs iteration spends outside this rank's two sampler launches: exchange
                    # (all-gather of the fresh ranges + all-reduce of the sums), statistics, host draws
                    "exchange_and_rest_ms": dt / steps * 1e3 - (kern["movs"] + kern["users"])})
    eng.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--K", type=int, default=None, help="shorthand: ML-1M shape with this K (32, 64, 128)")
    ap.add_argument("--repeats", type=int, default=0, help="timed blocks of --steps steps (0 = auto: 5..25 within ~8 s)")
    ap.add_argument("--prewarm-ms", type=float, default=50.0, help="untimed steps until this much time has passed (0: the W warm-up steps only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong_10Mx1M sub-record")
    ap.add_argument("--strong-steps", type=int, default=8, help="timed steps of the strong_10Mx1M record (>= 8: the library times every 8th launch of a side)")
    ap.add_argument("--strong-scale", type=float, default=float(os.environ.get("BPMF_BENCH_STRONG_SCALE", "1.0")))
    args = ap.parse_args()
    wl = args.workload or {None: "ml1m", 32: "ml1m", 64: "ml1m_k64", 128: "ml1m_k128"}.get(args.K)
    if wl is None:
        raise SystemExit("bench.py: --K must be 32, 64 or 128 (or use --workload)")
    K, dtype, kernel_name, lds_wg, wg_per_cu = WORKLOADS[wl]
    t_process = time.perf_counter()

    import torch
    import bpmf_amd
    from bpmf_amd import synth
    from bpmf_amd.sys import Sys

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (bpmf_amd has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    comm = None
    force_dist = os.environ.get("BPMF_BENCH_FORCE_DIST") == "1"        # test hook: run the sharded path with 1 rank
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    mult = world if (world > 1 or force_dist) else 1
    if wl == "chembl":
        M, Mt, T, Tt, nusers, nmovies = synth.ratings(483500 * mult, 5775, 1_023_952 * mult, seed=42, real_valued=True)
        shape_note = "ChEMBL-shaped synthetic R (%d compounds x %d targets, %d real-valued activities, 90/10 split)"
    elif mult == 1:
        M, Mt, T, Tt, nusers, nmovies = synth.ml1m_shaped(seed=42)
        shape_note = "ML-1M-shaped synthetic R (%d users x %d movies, %d ratings, 90/10 split)"
    else:
        M, Mt, T, Tt, nusers, nmovies = synth.ratings(6040 * mult, 3706, 1_000_209 * mult, seed=42)
        shape_note = "ML-1M-shaped synthetic R (%d users x %d movies, %d ratings, 90/10 split)"
    nnz = int(M[0][-1])
    mean = float(np.sum(M[2])) / nnz

    esz = 4 if dtype == "f32" else 8
    eng = bpmf_amd.HipEngine(K, device=local_rank, dtype=dtype)
    if world > 1 or force_dist:
        from bpmf_amd.dist import NativeComm, TorchComm
        # default: RCCL inside the library (exchange + all-reduce behind the sampling call);
        # BPMF_DIST=torch keeps the collectives in torch.distributed (same results, slower host path)
        comm = TorchComm(torch.device("cuda", local_rank)) if os.environ.get("BPMF_DIST") == "torch" else NativeComm(eng)
    Sys.nsims, Sys.burnin, Sys.alpha = 10 ** 6, 5, 2.0
    if comm is None:
        movies = Sys("movs", eng, M, nmovies, nusers, T=T, mean_rating=mean)
        users = Sys("users", eng, Mt, nusers, nmovies, mean_rating=mean)
        dom_m, dom_u = (0, nmovies), (0, nusers)
    else:
        from bpmf_amd.dist import build_sharded
        movies, users = build_sharded(eng, comm, M, Mt, T, nusers, nmovies, mean_rating=mean)
        dom_m, dom_u = movies.dom, users.dom

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    def dist_max(x):
        if world == 1:
            return x
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    pipelined = comm is None or getattr(comm, "native", False)

    def step_block(n):
        if not pipelined:
            for _ in range(n):
                movies.sample(users); users.sample(movies); movies.predict(users)
            return
        # the same n iterations, software-pipelined the way the `bpmf` executable runs them: the RMSE
        # of iteration i is collected after iteration i+1 has been enqueued (the evaluation runs on
        # its own stream beside those samplers, which write the other copy of the factors)
        for i in range(n):
            movies.sample(users)
            users.sample(movies)
            if i > 0:
                movies.predict_finish()
            movies.predict_launch(users)
        movies.predict_finish()

    # warm-up: W steps, then by TIME -- a 20-step timed region straight after start-up otherwise sits
    # on the clock ramp (round 1: 0.122 ms per step measured by the driver against 0.102 steady state)
    t_warm = time.perf_counter()
    if args.warmup > 0:
        step_block(args.warmup)
    fence()
    extra = 0
    while dist_max(time.perf_counter() - t_warm) < args.prewarm_ms * 1e-3:
        step_block(max(1, min(args.steps, 50))); extra += max(1, min(args.steps, 50))
        fence()
    prewarm_ms = (time.perf_counter() - t_warm) * 1e3
    base = {sd.name: eng.kernel_ms_sum(sd.side) for sd in (movies, users)}
    if args.repeats > 0:
        times = timed_blocks(step_block, fence, args.steps, dist_max, min_blocks=args.repeats, max_blocks=args.repeats)
    else:
        times = timed_blocks(step_block, fence, args.steps, dist_max)
    fence()
    dt = float(np.median(times))

    # roofline of the dominant kernel (the sampler), per launch, this rank's shard
    nnz_m = movies.local_nnz; nnz_u = users.local_nnz
    bytes_launch = 0.5 * (algorithmic_bytes(nnz_m, dom_m[1] - dom_m[0], K, esz) + algorithmic_bytes(nnz_u, dom_u[1] - dom_u[0], K, esz))
    flops_launch = 0.5 * (algorithmic_flops(nnz_m, dom_m[1] - dom_m[0], K) + algorithmic_flops(nnz_u, dom_u[1] - dom_u[0], K))
    # HIP-event times of the sampler / statistics kernels on their streams, summed by the library
    # over the timed steps (events ride on every 8th launch of a side: BPMF_HIP_TIMING_EVERY)
    kern_ms, red_ms, nl, per_side = 0.0, 0.0, 0, {}
    for sd in (movies, users):
        a1 = eng.kernel_ms_sum(sd.side); a0 = base[sd.name]
        if a1[2] > a0[2]:
            kern_ms += a1[0] - a0[0]; red_ms += a1[1] - a0[1]; nl += a1[2] - a0[2]
            per_side[sd.name] = (a1[0] - a0[0]) / (a1[2] - a0[2])
        else:
            a, b = eng.last_kernel_ms(sd.side); kern_ms += a; red_ms += b; nl += 1
            per_side[sd.name] = a
    launch_s = kern_ms / max(nl, 1) * 1e-3
    hbm_gbs = bytes_launch / launch_s / 1e9 if launch_s > 0 else 0.0
    tflops = flops_launch / launch_s / 1e12 if launch_s > 0 else 0.0
    flop_peak = FP32_PEAK_TFLOPS if dtype == "f32" else FP64_PEAK_TFLOPS
    traffic, conflict, pmc_file = profiled(wl) if world == 1 else (None, None, None)

    movies.predict(users, True)
    # Which resource binds?  K = 32 on this matrix: the factors (1.5 + 0.95 MB) live in L2 / MALL -- HBM-side
    # traffic is ~0.2 x the algorithmic bytes -- and the launch is bound by instruction issue: the fp64 MFMA
    # Gram and the VALU factorisation share the SIMD.  K >= 64: the dense contraction / factorisation.
    roofline = {"bound": "mfma", "bound_detail": ("fp64 issue: the MFMA Gram and the VALU/MFMA factorisation share the SIMD; "
                                                  "the factor matrices sit in L2/MALL at this size, HBM is secondary")
                if dtype == "f64" else "fp32 MFMA Gram + blocked factorisation",
                "achieved": tflops, "peak": flop_peak, "unit": "TFLOP/s", "frac": tflops / flop_peak,
                "traffic": traffic, "kernel": kernel_name, "launch_ms": launch_s * 1e3,
                "launch_ms_per_side": per_side, "algorithmic_flops_per_launch": flops_launch,
                "algorithmic_bytes_per_launch": bytes_launch,
                "hbm_achieved_gbs": hbm_gbs, "hbm_frac": hbm_gbs / HBM_PEAK_GBS,
                "hbm_traffic_over_algorithmic": (traffic / bytes_launch) if traffic else None,
                "colstats_ms": red_ms / max(nl, 1), "pmc_source": pmc_file}
    if lds_wg:
        roofline["lds"] = {"bytes_per_workgroup": lds_wg, "workgroups_per_cu": wg_per_cu, "occupancy": lds_wg * wg_per_cu / LDS_PER_CU,
                           "bank_conflict_rate": conflict}
    out = {
        "metric": "user+item column samples/sec per Gibbs iter; test RMSE vs reference",
        "value": (nusers + nmovies) * args.steps / dt,
        "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": dtype,
        "data": "synthetic",
        "config": {"workload": (shape_note + ", K=%d, alpha=2, full Gibbs iteration incl. host Normal-Wishart draws and RMSE")
                               % (nusers, nmovies, nnz + int(T[0][-1]), K),
                   "name": wl, "nnz_train": nnz, "nnz_test": int(T[0][-1]), "K": K,
                   "parallelism": "columns of U and V sharded over %d GPU(s)" % world},
        "repeats": len(times), "prewarm_ms": prewarm_ms, "prewarm_extra_steps": extra,
        "ms_per_step_median": dt / args.steps * 1e3, "ms_per_step_min": min(times) / args.steps * 1e3,
        "ms_per_step_max": max(times) / args.steps * 1e3, "ms_per_step_first_block": times[0] / args.steps * 1e3,
        "roofline": roofline,
        "rmse": movies.rmse, "rmse_avg": movies.rmse_avg,
        # secondary figures of SURVEY 8(d): the reference's ratings/s (nnz / t_iter, bpmf.cpp:195) and
        # the sampling-only rate (columns of both sides / the two sampler launches of one iteration)
        "ratings_per_s": nnz * args.steps / dt,
        "sampling_only_samples_per_s": (nusers + nmovies) / (2.0 * launch_s) if (launch_s > 0 and world == 1) else None,
    }
    try:
        eng.close()                      # sides, collector threads, streams, (RCCL communicator)
    except Exception:
        pass
    del movies, users

    if not args.no_strong and wl == "ml1m":
        try:
            out["strong_10Mx1M"] = strong_10Mx1M(world, rank, local_rank, args.strong_steps, args.strong_scale)
        except Exception as e:               # the headline must still be reported
            out["strong_10Mx1M"] = {"error": repr(e)[:400]}
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(M, Mt, T, Tt, K, nusers, nmovies)
            except Exception as e:  # the GPU number must still be reported
                out["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        out["wall_s"] = time.perf_counter() - t_process
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
