"""Worker of test_gpu_reduce.py: the BPMF_REDUCE formulation (bpmf_hip_sys_set_reduce) of the stateful path, in a
process of its own (one-rank RCCL communicator) --
  * single GPU, no communicator: the chain against the oracle's restatement of the reference's BPMF_REDUCE build
    (oracle.gibbs_reduce: preComputeMuLambda / sample from precMu + precLambda, c++/sample.cpp:234-246,289-291,375-377);
  * the same over a one-rank communicator + ranges (the grouped ncclReduce onto the owners runs, as the identity):
    bit for bit the chain without a communicator;
  * switching the formulation off again returns to the gather form."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bpmf_amd
    from bpmf_amd import synth
    from bpmf_amd.sys import Sys
    from oracle import oracle as orc
    K = int(sys.argv[1])
    alpha = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0        # (-a F, c++/bpmf.cpp:91: tests/test_gpu_alpha.py)
    nsims = 4
    # one heavy movie (650 ratings), empty columns on both sides
    M, Mt, T, Tt, nu, nm = synth.ratings(700, 500, 30000, seed=3, heavy=(7, 650))
    mean = float(np.sum(M[2])) / len(M[2])
    ref = orc.Oracle().gibbs_reduce(K, M, Mt, T, alpha=alpha, nsims=nsims, burnin=1)

    def run(comm, reduce, nocov=False):
        eng = bpmf_amd.HipEngine(K)
        if comm:
            eng.comm_init(1, 0, eng.comm_unique_id())
        Sys.nsims, Sys.burnin, Sys.alpha = nsims, 1, alpha
        movies = Sys("movs", eng, M, nm, nu, T=T, mean_rating=mean)
        users = Sys("users", eng, Mt, nu, nm, mean_rating=mean)
        if comm:
            eng.side_set_ranges(movies.side, [0, nm]); eng.side_set_ranges(users.side, [0, nu])
        if reduce:
            eng.sys_set_reduce(movies.side, users.side, True)
        rm = []
        for i in range(nsims):
            movies.sample(users); users.sample(movies)
            movies.predict(users)
            rm.append(movies.rmse)
        movies.refresh(); users.refresh()
        out = (np.asarray(rm), users.items().copy(), movies.items().copy())
        if reduce:                                               # off again: one more iteration of the gather form runs
            eng.sys_set_reduce(movies.side, users.side, False)
            movies.sample(users); users.sample(movies); movies.refresh()
            assert np.all(np.isfinite(movies.items()))
        eng.close()
        return out

    got = run(False, True)
    scale = max(np.abs(ref["U"]).max(), np.abs(ref["V"]).max())
    eu = np.abs(got[1] - ref["U"]).max() / scale; ev = np.abs(got[2] - ref["V"]).max() / scale
    er = np.abs(got[0] - ref["rmse"]).max()
    # fp64: summation order of the Gram, 1/sqrt vs divide, log / sqrt ulps, over `nsims` iterations of the chain
    assert eu < 1e-8 and ev < 1e-8 and er < 1e-9, (eu, ev, er)
    plain = run(False, False)                                    # the default formulation: same chain up to rounding
    assert np.abs(plain[1] - got[1]).max() / scale < 1e-8
    shard = run(True, True)
    for a, b in zip(got, shard):
        assert np.array_equal(a, b), "BPMF_REDUCE over a one-rank communicator differs from the plain one"
    print("REDUCE-OK K=%d alpha=%g factors %.1e / %.1e rmse %.1e" % (K, alpha, eu, ev, er))


if __name__ == "__main__":
    main()
