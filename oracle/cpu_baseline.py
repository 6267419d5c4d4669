#!/usr/bin/env python
"""CPU baseline leg of bench.py -- TEST / MEASUREMENT INFRASTRUCTURE (never the product path).

Times the oracle's -O3 -march=native -fopenmp build (a restatement of /root/reference
c++/sample.cpp:341-385 + c++/bpmf.cpp:180-198: the reference itself needs Eigen3 + Random123 and
cannot be built on the GPU box) on this box's host cores, in a process of its own so that the
OpenMP runtime sees the placement bench.py asks for (OMP_PLACES=cores OMP_PROC_BIND=spread: one
thread per physical core, spread over both sockets, first-touch placement of the factors).

    python oracle/cpu_baseline.py --matrix /dev/shm/x.npz --K 32 --budget 12

The matrix (CSC triples of M, Mt, T, Tt as saved by bench.py) is the SAME workload the GPU ran.
Prints one JSON object: the rate at ALL physical cores, the best of the sweep, the
sweep itself, CPU model, compiler flags.
"""
import argparse
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != HERE]      # (else `oracle` would resolve to oracle/oracle.py)
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np  # noqa: E402


def cpu_info():
    model, pairs = "unknown", set()
    phys = core = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    physical = min(len(pairs), usable) if pairs else usable
    cpu_info.pairs_total = len(pairs) if pairs else usable
    return model, max(1, physical), usable


def cpu_quota():
    """CPU-time quota of this container in cores (cgroup v2 cpu.max / v1 cfs_quota), or None.  Threads beyond it
    are throttled: on the MI355X boxes of this pool the quota is 16 cores of a 2 x 64-core host, and a 64-thread
    run that looks fine for three iterations (3.6 ms) settles at 27 ms once the quota bites."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // per)
    except (OSError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrix", required=True)
    ap.add_argument("--K", type=int, default=32)
    ap.add_argument("--budget", type=float, default=12.0, help="seconds of CPU work for the final measurement")
    ap.add_argument("--usable", type=int, default=0, help="hardware threads the launching process may use (with OMP_PROC_BIND "
                    "the OpenMP runtime pins THIS process's initial thread to one core as soon as it loads, so sched_getaffinity here says 2)")
    ap.add_argument("--parity-nsims", type=int, default=0, help="also run the CHECKER build's chain of this many iterations (bpmf -i) "
                    "on the same matrix and write its traces + factors to --parity-out: what bench.py's `parity` object is compared with")
    ap.add_argument("--parity-burnin", type=int, default=5)
    ap.add_argument("--parity-out", default=None)
    ap.add_argument("--parity-only", action="store_true", help="the parity chain only, no timing (bench.py's `configs` legs: their CPU figure is not asked for)")
    args = ap.parse_args()

    from oracle import oracle as orc
    flags = "gcc -O3 -march=native -fopenmp"
    try:
        orc.build(native=True)
    except Exception:
        flags = "gcc -O3 -march=x86-64-v3 -fopenmp"
    o = orc.Oracle(fast=True)
    z = np.load(args.matrix)
    M, Mt, T, Tt = [tuple(z["%s%d" % (n, i)] for i in range(3)) for n in ("M", "Mt", "T", "Tt")]
    nusers, nmovies = int(z["shape"][0]), int(z["shape"][1])
    K = args.K
    model, physical, usable = cpu_info()
    if args.usable > 0:
        physical = max(physical, min(cpu_info.pairs_total, args.usable)); usable = args.usable

    quota = cpu_quota()
    limit = min(usable, quota) if quota else usable

    parity = None
    if args.parity_nsims > 0 and args.parity_out:
        # "test RMSE vs reference": the oracle proper (libbpmf_oracle.so, -O2 -ffp-contract=off: the build the parity tests
        # check against, not the -O3 -march=native one that is timed below) runs main()'s loop (c++/bpmf.cpp:180-253) on the
        # matrix the GPU ran, identical seeds; bench.py diffs the traces and the factors with those of its own chain
        t0 = time.time()
        r = orc.Oracle().gibbs(K, M, Mt, T, Tt, nsims=args.parity_nsims, burnin=args.parity_burnin, nthreads=limit)
        np.savez(args.parity_out, rmse=r["rmse"], rmse_avg=r["rmse_avg"], norm_u=r["norm_u"], norm_m=r["norm_m"],
                 final=np.array([r["final_rmse_avg"], float(r["num_predict"])]), U=r["U"], V=r["V"])
        parity = {"iterations": args.parity_nsims, "burnin": args.parity_burnin, "threads": limit, "seconds": time.time() - t0,
                  "build": "oracle/libbpmf_oracle.so (gcc -O2 -ffp-contract=off -fopenmp)"}

    if args.parity_only:
        print(json.dumps({"parity_chain": parity, "value": None, "unit": "samples/s", "cores": limit, "kind": "port",
                          "sample": "not timed (--parity-only): the oracle's chain for the `parity` object only"}))
        return

    def per_iter(nt, n):
        o.gibbs(K, M, Mt, T, Tt, nsims=1, burnin=0, nthreads=nt)               # first touch + warm-up
        r = o.gibbs(K, M, Mt, T, Tt, nsims=n + 1, burnin=0, nthreads=nt)
        return float(np.median(r["secs"][1:]))

    # thread sweep up to what the container may use: all physical cores when there is no quota, else the quota
    # (one point beyond it is kept in the record to show the throttling)
    sweep = {}
    t_start = time.time()
    cand = sorted({t for t in (4, 8, 16, 32, 64, physical, limit) if t <= limit})
    if quota and 2 * limit <= usable:
        cand.append(2 * limit)
    for nt in cand:
        sweep[nt] = per_iter(nt, 20 if nt <= limit else 40)
        if time.time() - t_start > 60.0:                                        # (the sweep itself stays bounded)
            break
    over = {k: v for k, v in sweep.items() if k > limit}
    sweep_in = {k: v for k, v in sweep.items() if k <= limit}
    physical = min(physical, limit)
    best = min(sweep_in, key=sweep_in.get)
    n = int(max(3, min(400, args.budget / max(sweep_in[best], 1e-4))))
    t_best = per_iter(best, n)
    t_phys = sweep_in.get(physical)
    nsamp = nusers + nmovies
    print(json.dumps({
        "parity_chain": parity,
        "value": nsamp / t_best, "unit": "samples/s", "cores": best, "kind": "port",
        "all_usable_cores": {"cores": physical, "value": (nsamp / t_phys) if t_phys else None,
                             "ms_per_iter": t_phys * 1e3 if t_phys else None},
        "ms_per_iter": t_best * 1e3, "cpu_model": model, "physical_cores": cpu_info.pairs_total, "hardware_threads": usable,
        "cpu_quota_cores": quota, "beyond_quota_ms_per_iter": {str(k): v * 1e3 for k, v in over.items()},
        "flags": flags, "placement": "OMP_PLACES=%s OMP_PROC_BIND=%s" % (os.environ.get("OMP_PLACES", "-"), os.environ.get("OMP_PROC_BIND", "-")),
        "sweep_ms_per_iter": {str(k): v * 1e3 for k, v in sweep.items()},
        "sample": "%d full Gibbs iterations (both sides, host hyper draws, both predicts) of the same matrix the GPU ran, K=%d, "
                  "oracle restatement of c++/sample.cpp (omp parallel for schedule(guided) proc_bind(spread)), "
                  "%d threads = best of a sweep up to the %s; median of the iterations" % (
                      n, K, best, ("container's CPU quota of %d cores (host: %d physical cores)" % (quota, cpu_info.pairs_total)) if quota
                      else "%d physical cores" % physical)}))


if __name__ == "__main__":
    main()
