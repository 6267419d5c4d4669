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
    if case == "big":
        return big(dataset, K, nsims, burnin, out)
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


def big(scale, K, nsims, burnin, out):
    """One rank of the north star's strong-scaling set-up (BASELINE configs[3], bench.py::strong_10Mx1M) at a reduced size: the
    device-generated users x items x 200-per-user matrix cut into 8 user chunks / 8 nnz-balanced item ranges, rank r of N
    holding [8r/N, 8(r+1)/N), parts forced through BPMF_HIP_OVERLAP.  Rank 0 also writes the WHOLE matrix for the oracle."""
    import torch
    import torch.distributed as dist
    import bpmf_amd
    from bpmf_amd.dist import NativeComm
    from bpmf_amd.synth_dev import BigMatrix
    from bpmf_amd.sys import Sys
    scale = float(scale)
    world, rank = dist.get_world_size(), dist.get_rank()
    G = 8
    dev = torch.device("cuda", 0)
    bm_ = BigMatrix(dev, nusers=int(10_000_000 * scale), nitems=int(1_000_000 * scale), groups=G)
    parts = list(range(rank * G // world, (rank + 1) * G // world))
    bnd = bm_.item_bounds()
    bm = [bnd[r * G // world] for r in range(world)] + [bm_.NI]
    bu = [bm_.chunk_range(r * G // world)[0] for r in range(world)] + [bm_.NU]
    ucp, uri, uva, u0, u1 = bm_.users_csc(parts)
    mcp, mri, mva, i0, i1 = bm_.items_csc(parts)
    tcsc = bm_.test_csc(i0, i1)
    box = [None]
    if rank == 0:                                                   # the whole matrix, by item (= M of gibbs()), and the whole test set
        fcp, fri, fva, _, _ = bm_.items_csc(list(range(G)))
        tcp, tri, tva = bm_.test_csc(0, bm_.NI)
        Mv = fva.cpu().numpy()
        np.savez(out + ".matrix.npz", m0=fcp, m1=fri.cpu().numpy(), m2=Mv, t0=tcp, t1=tri, t2=tva, shape=np.array([bm_.NU, bm_.NI]))
        box[0] = float(np.sum(Mv)) / len(Mv)                        # Sys::init's mean_rating (c++/sample.cpp:183), the oracle's expression
        del fri, fva
    dist.broadcast_object_list(box, src=0)
    mean = box[0]
    eng = bpmf_amd.HipEngine(K, device=0)
    comm = NativeComm(eng)
    Sys.nsims, Sys.burnin, Sys.alpha = nsims, burnin, 2.0
    movies = Sys("movs", eng, (mcp, mri, mva), bm_.NI, bm_.NU, T=tcsc, dom=(i0, i1), mean_rating=mean, comm=comm)
    users = Sys("users", eng, (ucp, uri, uva), bm_.NU, bm_.NI, dom=(u0, u1), mean_rating=mean, comm=comm)
    comm.register(movies, bm); comm.register(users, bu)
    rm, rma, nu_, nm_ = [], [], [], []
    for _ in range(nsims):
        movies.sample(users); users.sample(movies)
        movies.predict(users, True)
        movies.refresh(); users.refresh()
        rm.append(movies.rmse); rma.append(movies.rmse_avg); nu_.append(float(np.sqrt(users.norm))); nm_.append(float(np.sqrt(movies.norm)))
    movies.predict(users, True)
    info_m, info_u = eng.schedule_info(movies.side), eng.schedule_info(users.side)
    np.savez(out + ".rank%d.npz" % rank, U=users.items(), V=movies.items(), rmse=rm, rmse_avg=rma, norm_u=nu_, norm_m=nm_,
             final=movies.rmse_avg, conn_used=np.asarray((False, False)), dom_m=np.asarray(movies.dom), dom_u=np.asarray(users.dom),
             nranks=eng.comm_nranks(), parts=np.asarray([info_m["parts"], info_u["parts"]]))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()
    print("MR-OK rank %d" % rank)


if __name__ == "__main__":
    main()
