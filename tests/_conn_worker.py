"""Worker of test_gpu_parity.py::test_connectivity_exchange_loopback: the packed exchange of
bpmf_hip_side_set_conn over a ONE-rank RCCL communicator.  With the only peer being the rank
itself, a send list A and a receive list B of equal length turn the exchange into "copy columns
A onto columns B": pack kernel, grouped ncclSend / ncclRecv, scatter kernel are all exercised."""
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
    eng = bpmf_amd.HipEngine(K)
    eng.comm_init(1, 0, eng.comm_unique_id())
    M, Mt, T, Tt, nu, nm = synth.ratings(300, 200, 6000, seed=3)
    mean = float(np.sum(M[2])) / len(M[2])

    # (1) the copy semantics of the packed exchange
    side = eng.side_create(nm, nu, M[0], M[1], M[2], mean)
    eng.side_set_ranges(side, [0, nm])
    rng = np.random.default_rng(5)
    X = rng.standard_normal((nm, K))
    eng.set_items(side, X)
    A = rng.choice(nm, size=70, replace=False).astype(np.int32)
    B = rng.choice(nm, size=70, replace=False).astype(np.int32)
    eng.side_set_conn(side, [0, 70], A, [0, 70], B)
    eng.side_exchange(side)
    want = X.copy(); want[B] = X[A]
    got = eng.get_items(side)
    assert np.array_equal(got, want), "packed exchange did not move the listed columns"
    # empty lists: nothing moves
    eng.side_set_conn(side, [0, 0], np.empty(0, np.int32), [0, 0], np.empty(0, np.int32))
    eng.side_exchange(side)
    assert np.array_equal(eng.get_items(side), want)
    # a list may not name columns outside the owner's range, offsets must be sane
    i32 = lambda v: np.asarray(v, np.int32)
    for bad in (([0, 1], i32([nm]), [0, 1], i32([0])),          # send list: not a column of this rank
                ([0, 1], i32([0]), [0, 1], i32([-1])),          # receive list: not a column of the sender
                ([1, 1], i32([0]), [0, 0], i32([])),            # offsets must start at 0
                ([0, 2], i32([0, 1]), [0, -1], i32([]))):       # offsets must be monotone
        try:
            eng.side_set_conn(side, *bad)
        except bpmf_amd.BpmfHipError as e:
            assert e.code == -1, e
        else:
            raise AssertionError("bad connectivity list accepted: %r" % (bad,))
    eng.side_set_conn(side)                                     # back to the all-gather form
    eng.side_exchange(side)
    assert np.array_equal(eng.get_items(side), want)
    eng.side_destroy(side)

    # (2) inside the pipelined Gibbs loop: identity lists (every column "sent to itself") must not change a run
    def run(conn):
        Sys.nsims, Sys.burnin, Sys.alpha = 4, 1, 2.0
        movies = Sys("movs", eng, M, nm, nu, T=T, mean_rating=mean)
        users = Sys("users", eng, Mt, nu, nm, mean_rating=mean)
        eng.side_set_ranges(movies.side, [0, nm]); eng.side_set_ranges(users.side, [0, nu])
        if conn:
            eng.side_set_conn(movies.side, [0, nm], np.arange(nm, dtype=np.int32), [0, nm], np.arange(nm, dtype=np.int32))
            eng.side_set_conn(users.side, [0, nu], np.arange(nu, dtype=np.int32), [0, nu], np.arange(nu, dtype=np.int32))
        tr = []
        for _ in range(4):
            movies.sample(users); users.sample(movies); movies.predict(users, True)
            tr.append((movies.rmse, movies.rmse_avg))
        U, V = users.items().copy(), movies.items().copy()
        return np.asarray(tr), U, V

    t0, U0, V0 = run(False)
    t1, U1, V1 = run(True)
    assert np.array_equal(t0, t1) and np.array_equal(U0, U1) and np.array_equal(V0, V1)
    print("CONN-OK")
    eng.close()


if __name__ == "__main__":
    main()
