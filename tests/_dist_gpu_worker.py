"""Worker of test_gpu_parity.py::test_two_ranks_share_one_gpu: one rank of a two-process gloo job in
which BOTH ranks drive the HIP kernels on cuda:0 (RCCL refuses two ranks on one device; gloo moves
the shards through the host).  Exercises what a second GPU would: column ranges, CSC slices,
shard-local sampling with the real kernels, exchange of the fresh ranges, all-reduced sums / RMSE."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    import bpmf_amd
    from bpmf_amd.dist import TorchComm, gibbs_sharded
    from tests import util

    dataset, K, nsims, burnin, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    M, Mt, T, Tt, nu, nm = {"ml100k": util.ml100k, "blocks": util.blocks}[dataset]()
    comm = TorchComm(torch.device("cuda", 0))
    eng = bpmf_amd.HipEngine(K, device=0)
    res = gibbs_sharded(eng, comm, M, Mt, T, nu, nm, nsims=nsims, burnin=burnin)
    # the tensors TorchComm bound (HipEngine.items_tensor: [ncols, ld()]): with a padded num_latent (K = 20 on the K = 32 kernels) the
    # rows K .. ld()-1 of every column must still be zero after the run -- samplers, exchanges and every writer that went through
    # factors_view() leave them alone (writing randn(t.shape) into the whole tensor fed non-zero padding to the kernels)
    for t, _, _ in comm._items.values():
        assert tuple(t.shape)[1] == eng.ld() and tuple(eng.factors_view(t).shape)[1] == K
        assert not bool(t[:, K:].any()), "padding rows of a bound factor tensor were written"
        assert bool(eng.factors_view(t).any())
    np.savez(out + ".rank%d.npz" % comm.rank, U=res["U"], V=res["V"], rmse=res["rmse"], rmse_avg=res["rmse_avg"],
             norm_u=res["norm_u"], norm_m=res["norm_m"], final=res["final_rmse_avg"], conn_used=np.asarray(res["conn_used"]),
             dom_m=np.asarray(res["dom_m"]), dom_u=np.asarray(res["dom_u"]))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
