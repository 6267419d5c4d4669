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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrix", required=True)
    ap.add_argument("--K", type=int, default=32)
    ap.add_argument("--budget", type=float, default=12.0, help="seconds of CPU work for the final measurement")
    ap.add_argument("--usable", type=int, default=0, help="hardware threads the launching process may use (with OMP_PROC_BIND "
                    "the OpenMP runtime pins THIS process's initial thread to one core as soon as it loads, so sched_getaffinity here says 2)")
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

    def per_iter(nt, n):
        o.gibbs(K, M, Mt, T, Tt, nsims=1, burnin=0, nthreads=nt)               # first touch + warm-up
        r = o.gibbs(K, M, Mt, T, Tt, nsims=n + 1, burnin=0, nthreads=nt)
        return float(np.mean(r["secs"][1:]))

    sweep = {}
    t_start = time.time()
    for nt in sorted({t for t in (8, 16, 32, 64, physical, usable) if t <= usable}):
        sweep[nt] = per_iter(nt, 3)
        if time.time() - t_start > 60.0:                                        # (the sweep itself stays bounded)
            break
    best = min(sweep, key=sweep.get)
    n = int(max(3, min(400, args.budget / max(sweep[best], 1e-4))))
    t_best = per_iter(best, n)
    t_phys = sweep.get(physical)
    nsamp = nusers + nmovies
    print(json.dumps({
        "value": nsamp / t_best, "unit": "samples/s", "cores": best, "kind": "port",
        "all_physical_cores": {"cores": physical, "value": (nsamp / t_phys) if t_phys else None,
                               "ms_per_iter": t_phys * 1e3 if t_phys else None},
        "ms_per_iter": t_best * 1e3, "cpu_model": model, "physical_cores": physical, "hardware_threads": usable,
        "flags": flags, "placement": "OMP_PLACES=%s OMP_PROC_BIND=%s" % (os.environ.get("OMP_PLACES", "-"), os.environ.get("OMP_PROC_BIND", "-")),
        "sweep_ms_per_iter": {str(k): v * 1e3 for k, v in sweep.items()},
        "sample": "%d full Gibbs iterations (both sides, host hyper draws, both predicts) of the same matrix the GPU ran, K=%d, "
                  "oracle restatement of c++/sample.cpp (omp parallel for schedule(guided) proc_bind(spread)), "
                  "%d threads = best of a sweep; all %d physical cores: %s ms/iter" % (
                      n, K, best, physical, ("%.2f" % (t_phys * 1e3)) if t_phys else "n/a")}))


if __name__ == "__main__":
    main()
