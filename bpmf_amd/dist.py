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
