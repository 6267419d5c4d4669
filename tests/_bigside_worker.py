"""Worker of test_gpu_parity.py::test_sharded_big_side_single_rank: a side with more than 100 000 columns takes the
workgroup form of the statistics pass (k_colstats_wg); over a one-rank RCCL communicator + ranges its sums go through
the device blob and the all-reduce instead of straight to the host -- the chain must be the plain one bit for bit.
(The 10M x 1M strong-scaling record of bench.py runs this combination on every rank.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bpmf_amd
    from bpmf_amd import synth
    from bpmf_amd.sys import Sys
    K = int(sys.argv[1])
    M, Mt, T, Tt, nu, nm = synth.ratings(120000, 300, 500000, seed=11)
    mean = float(np.sum(M[2])) / len(M[2])

    def run(comm):
        eng = bpmf_amd.HipEngine(K)
        if comm:
            eng.comm_init(1, 0, eng.comm_unique_id())
        Sys.nsims, Sys.burnin, Sys.alpha = 3, 1, 2.0
        movies = Sys("movs", eng, M, nm, nu, T=T, mean_rating=mean)
        users = Sys("users", eng, Mt, nu, nm, mean_rating=mean)
        if comm:
            eng.side_set_ranges(movies.side, [0, nm]); eng.side_set_ranges(users.side, [0, nu])
        rm = []
        for i in range(3):
            movies.sample(users); users.sample(movies)
            movies.predict(users)
            rm.append(movies.rmse)
        movies.refresh(); users.refresh()
        out = (np.asarray(rm), users.items().copy(), movies.items().copy(), users.norm, movies.norm)
        eng.close()
        return out

    base = run(False)
    assert np.all(np.isfinite(base[1])) and np.all(np.isfinite(base[2]))
    got = run(True)
    for a, b in zip(base, got):
        assert np.array_equal(np.asarray(a), np.asarray(b)), "sharded chain of a big side differs from the plain one"
    # and against the oracle: first half-iterations of the same matrix (the statistics feed the next hyper-parameter draw)
    from oracle import oracle as orc
    ref = orc.Oracle().gibbs(K, M, Mt, T, Tt, alpha=2.0, nsims=3, burnin=1, nthreads=8)
    scale = max(np.abs(ref["U"]).max(), np.abs(ref["V"]).max())
    eu = np.abs(base[1] - ref["U"]).max() / scale; ev = np.abs(base[2] - ref["V"]).max() / scale
    assert eu < 1e-8 and ev < 1e-8, (eu, ev)
    print("BIGSIDE-OK K=%d %.1e %.1e" % (K, eu, ev))


if __name__ == "__main__":
    main()
