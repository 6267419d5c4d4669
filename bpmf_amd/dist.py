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


def connectivity(M, col_bounds, T=None):
    """need[r] = sorted unique ROW ids referenced by the stored ratings (and test entries) of the
    columns col_bounds[r]..col_bounds[r+1] of the CSC matrix M = (colptr, rowidx, vals): the columns
    of the OTHER side's factor matrix rank r reads.  (Sys::update_conn, c++/assign.cpp:204-241, sets
    the same bits from the other end: bm.set(other.proc(it.row())) over M and Pavg.)"""
    need = []
    for r in range(len(col_bounds) - 1):
        lo, hi = col_bounds[r], col_bounds[r + 1]
        rows = [np.asarray(M[1][M[0][lo]:M[0][hi]], np.int64)]
        if T is not None:
            rows.append(np.asarray(T[1][T[0][lo]:T[0][hi]], np.int64))
        need.append(np.unique(np.concatenate(rows)))
    return need


def conn_lists(need, bounds, rank):
    """Send / receive lists of `rank` for a side whose columns are owned in ranges `bounds` and read
    as `need[r]` by rank r: (send_ptr, send_cols, recv_ptr, recv_cols), global column ids, ascending
    per peer, nothing to self."""
    n = len(bounds) - 1
    lo, hi = bounds[rank], bounds[rank + 1]
    send, recv = [], []
    for r in range(n):
        if r == rank:
            send.append(np.empty(0, np.int64)); recv.append(np.empty(0, np.int64)); continue
        a = need[r]
        send.append(a[(a >= lo) & (a < hi)])                                  # mine, read by r
        b = need[rank]
        recv.append(b[(b >= bounds[r]) & (b < bounds[r + 1])])               # r's, read by me
    ptr = lambda parts: np.concatenate([[0], np.cumsum([len(x) for x in parts])]).astype(np.int64)
    cat = lambda parts: np.concatenate(parts).astype(np.int32) if parts else np.empty(0, np.int32)
    return ptr(send), cat(send), ptr(recv), cat(recv)


def conn_pays(lists, bounds, rank, threshold=0.5):
    """The packed exchange replaces an all-gather that delivers every foreign column to this rank:
    use it when it moves less than `threshold` of that (pack / scatter kernels and per-peer
    messages are not free).  BPMF_DIST_CONN=1 / 0 forces the choice."""
    import os
    force = os.environ.get("BPMF_DIST_CONN")
    if force is not None and force != "":
        return force != "0"
    foreign = bounds[-1] - (bounds[rank + 1] - bounds[rank])
    return foreign > 0 and lists[2][-1] < threshold * foreign


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

    def register(self, sys, bounds, conn=None):
        self.engine.side_set_ranges(sys.side, bounds)
        if conn is not None:
            self.engine.side_set_conn(sys.side, *conn)

    def full_gather(self, sys):
        """users.bcast() / movies.bcast() of the reference (c++/bpmf.cpp:217-218, 268-273): after a run with the
        connectivity-aware exchange a replica only holds the columns its rank reads; before the factors are handed
        out (outputs, -v dumps) every rank's range travels to everyone once, in the all-gather form."""
        self.engine.side_set_conn(sys.side)
        self.engine.side_exchange(sys.side)


class TorchComm:
    def __init__(self, device):
        self.device = torch.device(device)
        self.rank = dist.get_rank()
        self.size = dist.get_world_size()
        self._items = {}      # id(sys) -> (tensor [ncols, K], bounds)

    def register(self, sys, bounds, conn=None):
        """Binds the factor matrix of `sys` to a torch tensor the collectives can use."""
        t = sys.engine.items_tensor(sys.side, self.device)
        if conn is not None:
            conn = tuple(torch.as_tensor(np.asarray(a, np.int64)).to(self.device) for a in conn)
        self._items[id(sys)] = (t, list(bounds), conn)

    def exchange_items(self, sys):
        t, bounds, conn = self._items[id(sys)]
        if conn is not None:
            # connectivity-aware form: one packed message per peer that reads any of my columns
            sp, sc, rp, rc = conn
            sbuf = t.index_select(0, sc) if len(sc) else t[:0]
            rbuf = torch.empty((len(rc), t.shape[1]), dtype=t.dtype, device=t.device)
            ops = []
            for r in range(self.size):
                if sp[r + 1] > sp[r]:
                    ops.append(dist.P2POp(dist.isend, sbuf[sp[r]:sp[r + 1]].contiguous(), r))
                if rp[r + 1] > rp[r]:
                    ops.append(dist.P2POp(dist.irecv, rbuf[rp[r]:rp[r + 1]], r))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            if len(rc):
                t.index_copy_(0, rc, rbuf)
            if self.device.type == "cuda":
                torch.cuda.current_stream(self.device).synchronize()
            return
        works = []
        for r in range(self.size):
            lo, hi = bounds[r], bounds[r + 1]
            if hi > lo:
                works.append(dist.broadcast(t[lo:hi], src=r, async_op=True))
        for w in works:
            w.wait()
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()

    def full_gather(self, sys):
        t, bounds, conn = self._items[id(sys)]
        self._items[id(sys)] = (t, bounds, None)
        self.exchange_items(sys)

    def allreduce(self, arr):
        t = torch.as_tensor(np.ascontiguousarray(arr, np.float64)).to(self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.cpu().numpy()


def build_sharded(engine, comm, M, Mt, T, nusers, nmovies, mean_rating=None, conn=True, Tt=None):
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
    users = Sys("users", engine, synth.slice_cols(Mt, *dom_u), nusers, nmovies,
                T=synth.slice_cols(Tt, *dom_u) if Tt is not None else None, dom=dom_u, mean_rating=mean_rating, comm=comm)
    # who reads what (c++/assign.cpp:204-241): sampling the movies of rank r reads the users rated in
    # them, predict() of rank r reads the users of its test entries; sampling users reads movies
    conn_u = conn_lists(connectivity(M, bm, T), bu, rank)            # exchange of the USERS' columns
    conn_m = conn_lists(connectivity(Mt, bu, Tt), bm, rank)          # exchange of the MOVIES' columns (users.predict(movies) reads them too)
    movies.conn_used = bool(conn and conn_pays(conn_m, bm, rank))
    users.conn_used = bool(conn and conn_pays(conn_u, bu, rank))
    movies.conn_lists, users.conn_lists = conn_m, conn_u
    comm.register(movies, bm, conn_m if movies.conn_used else None)
    comm.register(users, bu, conn_u if users.conn_used else None)
    return movies, users


def gibbs_sharded(engine, comm, M, Mt, T, nusers, nmovies, nsims=20, burnin=5, alpha=2.0, conn=True):
    """main()'s loop (c++/bpmf.cpp:180-253) with the columns sharded over comm.size ranks."""
    from .sys import Sys
    Sys.nsims, Sys.burnin, Sys.alpha = nsims, burnin, alpha
    movies, users = build_sharded(engine, comm, M, Mt, T, nusers, nmovies, conn=conn)
    res = dict(rmse=[], rmse_avg=[], norm_u=[], norm_m=[])
    for _ in range(nsims):
        movies.sample(users)
        users.sample(movies)
        movies.predict(users, True)          # all-reduced partial sums: every rank reports the global RMSE
        movies.refresh(); users.refresh()    # (exchange inside the library: norm / cov / hp live there)
        res["rmse"].append(movies.rmse); res["rmse_avg"].append(movies.rmse_avg)
        res["norm_u"].append(float(np.sqrt(users.norm))); res["norm_m"].append(float(np.sqrt(movies.norm)))
    movies.predict(users, True)
    res["final_rmse_avg"] = movies.rmse_avg
    # replicas as the loop leaves them (with the connectivity-aware exchange: only the columns this rank reads are current)
    res["U_replica"] = users.items(); res["V_replica"] = movies.items()
    import os
    stale = int(os.environ.get("BPMF_HIP_STALE", "0") or 0) > 0       # bounded-staleness exchange: replicas may lag behind
    if movies.conn_used or stale:
        comm.full_gather(movies)
    if users.conn_used or stale:
        comm.full_gather(users)
    res["U"] = users.items(); res["V"] = movies.items()
    res["conn_used"] = (movies.conn_used, users.conn_used)
    res["dom_m"], res["dom_u"] = movies.dom, users.dom
    return res
