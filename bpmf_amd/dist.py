"""Multi-GPU exchange for the sharded sampler: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

What the reference's MPI/GASPI back-ends do per half-iteration (send_item of every
fresh K-vector to the ranks that need it + all-reduce of sum/cov/norm,
c++/mpi_common.h:44-50, c++/mpi_bcast.h:21-30) becomes:
  * an all-gather of the freshly sampled column range of the factor matrix
    (disjoint ownership, so every rank contributes its own slice), and
  * one small all-reduce of [prod | sum | norm] (K*K+K+1 doubles); cov is then
    formed once from the GLOBAL sums, so results do not depend on the GPU count
    beyond summation order.
Ranges are contiguous and nnz-balanced (synth.balanced_ranges), hence uneven: the
all-gather is issued as one broadcast per owner on views of the bound factor tensor.
"""
import numpy as np
import torch
import torch.distributed as dist


class NativeComm:
    """The exchange runs inside libbpmf_hip.so over RCCL (bpmf_hip_ctx_comm_init): the fresh column
    range of every rank is broadcast in place and sum | prod | norm are all-reduced on the device,
    behind the same C-ABI call that samples.  torch.distributed is only the launcher here: it
    ships rank 0's 128-byte RCCL id to the other ranks."""

    native = True

    def __init__(self, engine):
        self.rank = dist.get_rank()
        self.size = dist.get_world_size()
        box = [engine.comm_unique_id() if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        engine.comm_init(self.size, self.rank, box[0])
        self.engine = engine

    def register(self, sys, bounds):
        self.engine.side_set_ranges(sys.side, bounds)


class TorchComm:
    def __init__(self, device):
        self.device = torch.device(device)
        self.rank = dist.get_rank()
        self.size = dist.get_world_size()
        self._items = {}      # id(sys) -> (tensor [ncols, K], bounds)

    def register(self, sys, bounds):
        """Binds the factor matrix of `sys` to a torch tensor the collectives can use."""
        t = sys.engine.items_tensor(sys.side, self.device)
        self._items[id(sys)] = (t, list(bounds))

    def exchange_items(self, sys):
        t, bounds = self._items[id(sys)]
        works = []
        for r in range(self.size):
            lo, hi = bounds[r], bounds[r + 1]
            if hi > lo:
                works.append(dist.broadcast(t[lo:hi], src=r, async_op=True))
        for w in works:
            w.wait()
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()

    def allreduce(self, arr):
        t = torch.as_tensor(np.ascontiguousarray(arr, np.float64)).to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()


def build_sharded(engine, comm, M, Mt, T, nusers, nmovies, mean_rating=None):
    """Creates the two `Sys` of this rank: contiguous nnz-balanced column ranges of both sides
    (the reference's assign(), c++/assign.cpp:52-58,109-120, without the permutation), CSC
    slices of exactly those ranges, full factor replicas bound to tensors the collectives use."""
    from . import synth
    from .sys import Sys
    world, rank = comm.size, comm.rank
    if mean_rating is None:
        mean_rating = float(np.sum(M[2])) / len(M[2])
    bm = synth.balanced_ranges(M[0], world)
    bu = synth.balanced_ranges(Mt[0], world)
    dom_m, dom_u = (bm[rank], bm[rank + 1]), (bu[rank], bu[rank + 1])
    movies = Sys("movs", engine, synth.slice_cols(M, *dom_m), nmovies, nusers,
                 T=synth.slice_cols(T, *dom_m) if T is not None else None, dom=dom_m, mean_rating=mean_rating, comm=comm)
    users = Sys("users", engine, synth.slice_cols(Mt, *dom_u), nusers, nmovies, dom=dom_u, mean_rating=mean_rating, comm=comm)
    comm.register(movies, bm)
    comm.register(users, bu)
    return movies, users


def gibbs_sharded(engine, comm, M, Mt, T, nusers, nmovies, nsims=20, burnin=5, alpha=2.0):
    """main()'s loop (c++/bpmf.cpp:180-253) with the columns sharded over comm.size ranks."""
    from .sys import Sys
    Sys.nsims, Sys.burnin, Sys.alpha = nsims, burnin, alpha
    movies, users = build_sharded(engine, comm, M, Mt, T, nusers, nmovies)
    res = dict(rmse=[], rmse_avg=[], norm_u=[], norm_m=[])
    for _ in range(nsims):
        movies.sample(users)
        users.sample(movies)
        movies.predict(users, True)          # all-reduced partial sums: every rank reports the global RMSE
        res["rmse"].append(movies.rmse); res["rmse_avg"].append(movies.rmse_avg)
        res["norm_u"].append(float(np.sqrt(users.norm))); res["norm_m"].append(float(np.sqrt(movies.norm)))
    movies.predict(users, True)
    res["final_rmse_avg"] = movies.rmse_avg
    res["U"] = users.items(); res["V"] = movies.items()
    return res
