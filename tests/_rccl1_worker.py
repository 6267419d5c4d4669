"""Worker of test_gpu_parity.py::test_real_rccl_split_parts_and_packed_lists_single_rank: everything of the library's RCCL path
that can run on the REAL librccl with one rank and one GPU, TOGETHER (the multi-rank tests go through tests/rccl_double,
because RCCL refuses two ranks per device) --
  * the second communicator is there and came from ncclCommSplit (bpmf_hip_ctx_comm_streams == 2; with
    BPMF_HIP_COMM_STREAMS=1 it is 1), ncclCommCount == 1 on the first;
  * every exchange cut into FOUR parts (bpmf_hip_side_set_overlap: per-part windows, exchange stream, events) AND going
    through the packed connectivity lists (bpmf_hip_side_set_conn: pack kernel, grouped ncclSend / ncclRecv to the only
    peer -- the rank itself --, scatter kernel) at the same time, inside the pipelined Gibbs loop with the twin evaluation:
    the factors must be the NO_COMM chain's bit for bit (identity lists move every column onto itself), the RMSE trace to 1e-13;
  * statistics all-reduce + RMSE count all-reduce over the second communicator (identity at one rank).
What replaces: MPI_Init / MPI_Comm_size (c++/mpi_common.h:44-50), the chunked MPI_Isend progress of c++/mpi_isendirecv.h:222-260."""
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
    want_streams = 1 if os.environ.get("BPMF_HIP_COMM_STREAMS") == "1" else 2
    M, Mt, T, Tt, nu, nm = synth.ratings(900, 600, 40000, seed=11, heavy=(5, 700))
    mean = float(np.sum(M[2])) / len(M[2])

    def run(comm, parts, conn):
        eng = bpmf_amd.HipEngine(K)
        assert eng.comm_streams() == 0 and eng.comm_nranks() == 1
        if comm:
            eng.comm_init(1, 0, eng.comm_unique_id())
            assert eng.comm_nranks() == 1, eng.comm_nranks()
            assert eng.comm_streams() == want_streams, "second communicator: %d, expected %d" % (eng.comm_streams(), want_streams)
        Sys.nsims, Sys.burnin, Sys.alpha = 6, 2, 2.0
        movies = Sys("movs", eng, M, nm, nu, T=T, mean_rating=mean)
        users = Sys("users", eng, Mt, nu, nm, T=Tt, mean_rating=mean)
        if comm:
            eng.side_set_ranges(movies.side, [0, nm]); eng.side_set_ranges(users.side, [0, nu])
            if parts > 1:
                eng.side_set_overlap(movies.side, parts); eng.side_set_overlap(users.side, parts)
                assert eng.schedule_info(movies.side)["parts"] == parts
            if conn:
                eng.side_set_conn(movies.side, [0, nm], np.arange(nm, dtype=np.int32), [0, nm], np.arange(nm, dtype=np.int32))
                eng.side_set_conn(users.side, [0, nu], np.arange(nu, dtype=np.int32), [0, nu], np.arange(nu, dtype=np.int32))
        movies.set_twin(users)
        tr = []
        for i in range(6):
            movies.sample(users); users.sample(movies)
            if i > 0:
                movies.predict_finish(); users.predict_finish()
                tr.append((movies.rmse, movies.rmse_avg, users.rmse))
            movies.predict_launch(users)
        movies.predict_finish(); users.predict_finish()
        tr.append((movies.rmse, movies.rmse_avg, users.rmse))
        movies.refresh(); users.refresh()
        out = (np.asarray(tr), users.items().copy(), movies.items().copy(), movies.norm, users.norm)
        eng.close()
        return out

    base = run(False, 1, False)
    assert np.all(np.isfinite(base[0])) and np.all(np.isfinite(base[1]))
    for parts, conn in ((4, True), (4, False), (1, True)):
        got = run(True, parts, conn)
        # factors and norms bit for bit; the RMSE trace to the last ulp or two (with a communicator the squared-error sums of the
        # evaluation -- the twin's included -- come back through an all-reduce: another summation path, not another chain)
        assert np.allclose(base[0], got[0], rtol=0, atol=1e-13), "parts=%d conn=%s: RMSE trace differs from the NO_COMM chain" % (parts, conn)
        for a, b in zip(base[1:], got[1:]):
            assert np.array_equal(np.asarray(a), np.asarray(b)), "parts=%d conn=%s: differs from the NO_COMM chain" % (parts, conn)
    print("RCCL1-OK streams=%d" % want_streams)


if __name__ == "__main__":
    main()
