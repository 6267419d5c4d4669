"""Worker of test_gpu_multirank.py: ONE RANK of an N-process job in which every rank drives the HIP kernels on cuda:0
and the exchange runs INSIDE libbpmf_hip.so (NativeComm -> bpmf_hip_ctx_comm_init ...) over the tests' RCCL double
(BPMF_HIP_RCCL_LIBRARY = tests/rccl_double/librccl_double.so: RCCL itself refuses two ranks per device).  This is the
code path `bench.py --gpus N` and `bpmf -g N` take, with N >= 2: the mesh of grouped ncclSend / ncclRecv, the parts
exchanged on their own stream, the second communicator, the packed connectivity-aware lists between different ranks,
the grouped ncclReduce of the BPMF_REDUCE formulation, the fp32 context.  torch.distributed (gloo) only carries the
128-byte id.  Switches of the library come through the environment (set by the test); argv: case dataset K nsims burnin out."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import bpmf_amd
    from bpmf_amd.dist import NativeComm, gibbs_sharded, build_sharded
    from bpmf_amd.sys import Sys
    from tests import util

    case, dataset, K, nsims, burnin, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    assert os.environ.get("BPMF_HIP_RCCL_LIBRARY"), "the test sets BPMF_HIP_RCCL_LIBRARY"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    data = {"ml100k": util.ml100k, "blocks": util.blocks,
            "heavy": lambda: util.synthetic(700, 500, 30000, seed=3, heavy=(7, 650))}[dataset]
    M, Mt, T, Tt, nu, nm = data()
    eng = bpmf_amd.HipEngine(K, device=0, dtype="f32" if K == 128 else "f64")
    comm = NativeComm(eng)
    assert eng.comm_nranks() == dist.get_world_size() >= 2
    extra = {}
    if case == "reduce":
        # the reference's BPMF_REDUCE build over N ranks: parts precomputed per rank, grouped ncclReduce onto the owners
        Sys.nsims, Sys.burnin, Sys.alpha = nsims, burnin, 2.0
        movies, users = build_sharded(eng, comm, M, Mt, T, nu, nm, mean_rating=None, conn=False)
        eng.sys_set_reduce(movies.side, users.side, True)
        rm = []
        for _ in range(nsims):
            movies.sample(users); users.sample(movies)
            movies.predict(users, True)
            rm.append(movies.rmse)
        res = dict(U=users.items(), V=movies.items(), rmse=rm, rmse_avg=rm, norm_u=[0.0], norm_m=[0.0], final_rmse_avg=movies.rmse_avg,
                   conn_used=(False, False), dom_m=movies.dom, dom_u=users.dom)
    elif case == "stale_age":
        # bounded staleness, observed directly: after every iteration this rank's replicas of both factor matrices
        Sys.nsims, Sys.burnin, Sys.alpha = nsims, burnin, 2.0
        movies, users = build_sharded(eng, comm, M, Mt, T, nu, nm, mean_rating=None, conn=False)
        k = int(os.environ["BPMF_TEST_STALE_K"])
        eng.side_set_staleness(movies.side, k); eng.side_set_staleness(users.side, k)
        snaps_u, snaps_v, rm = [], [], []
        for _ in range(nsims):
            movies.sample(users); users.sample(movies)
            movies.predict(users, True)
            rm.append(movies.rmse)
            snaps_u.append(users.items()); snaps_v.append(movies.items())
        extra = dict(snaps_u=np.stack(snaps_u), snaps_v=np.stack(snaps_v))
        res = dict(U=users.items(), V=movies.items(), rmse=rm, rmse_avg=rm, norm_u=[0.0], norm_m=[0.0], final_rmse_avg=movies.rmse_avg,
                   conn_used=(False, False), dom_m=movies.dom, dom_u=users.dom)
    else:
        res = gibbs_sharded(eng, comm, M, Mt, T, nu, nm, nsims=nsims, burnin=burnin, conn=(case == "conn"))
    np.savez(out + ".rank%d.npz" % comm.rank, U=res["U"], V=res["V"], rmse=res["rmse"], rmse_avg=res["rmse_avg"],
             norm_u=res["norm_u"], norm_m=res["norm_m"], final=res["final_rmse_avg"], conn_used=np.asarray(res["conn_used"]),
             dom_m=np.asarray(res["dom_m"]), dom_u=np.asarray(res["dom_u"]), nranks=eng.comm_nranks(), **extra)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()
    print("MR-OK rank %d" % comm.rank)


if __name__ == "__main__":
    main()
