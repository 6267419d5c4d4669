"""Worker of test_gpu_parity.py::test_sharded_parts_single_rank: the sharded stateful path over a ONE-rank RCCL
communicator (own process: the communicator is per process) --
  * sharded (communicator + ranges) == plain NO_COMM chain, for fp64 and for the fp32 K = 128 context
    (whose exchange / statistics / evaluation went through "single-GPU for now" until round 2);
  * bpmf_hip_side_set_overlap: a side cut into 2 / 3 / 4 / 8 parts (part c exchanged on its own stream while part
    c + 1 is sampled) gives the chain of the uncut side bit for bit -- with one rank the exchange itself has no
    peer, what is exercised is the per-part item windows, the launch sequence and the stream / event hand-offs."""
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
    dtype = "f32" if K == 128 else "f64"
    M, Mt, T, Tt, nu, nm = synth.ratings(700, 500, 30000, seed=3, heavy=(7, 650))      # (one heavy movie: chunked column)
    mean = float(np.sum(M[2])) / len(M[2])

    def run(comm, parts):
        eng = bpmf_amd.HipEngine(K, dtype=dtype)
        if comm:
            eng.comm_init(1, 0, eng.comm_unique_id())
        Sys.nsims, Sys.burnin, Sys.alpha = 5, 1, 2.0
        movies = Sys("movs", eng, M, nm, nu, T=T, mean_rating=mean)
        users = Sys("users", eng, Mt, nu, nm, mean_rating=mean)
        if comm:
            eng.side_set_ranges(movies.side, [0, nm]); eng.side_set_ranges(users.side, [0, nu])
            if parts > 1:
                eng.side_set_overlap(movies.side, parts); eng.side_set_overlap(users.side, parts)
        tr = []
        for i in range(5):
            movies.sample(users); users.sample(movies)
            if i > 0:
                movies.predict_finish()
                tr.append((movies.rmse, movies.rmse_avg))
            movies.predict_launch(users)
        movies.predict_finish(); tr.append((movies.rmse, movies.rmse_avg))
        movies.refresh(); users.refresh()
        out = (np.asarray(tr), users.items().copy(), movies.items().copy(), movies.norm, users.norm)
        eng.close()
        return out

    base = run(False, 1)
    assert np.all(np.isfinite(base[1])) and np.all(np.isfinite(base[2]))
    for parts in (1, 3, 8):                                          # (2 and 4 parts: tests/_rccl1_worker.py, tests/test_gpu_multirank.py)
        got = run(True, parts)
        for a, b in zip(base, got):
            assert np.array_equal(np.asarray(a), np.asarray(b)), "sharded chain with %d part(s) differs from the plain one" % parts
    print("PARTS-OK")


if __name__ == "__main__":
    main()
