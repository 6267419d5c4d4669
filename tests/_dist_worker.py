"""Worker of tests/test_dist_gloo.py: one rank of a gloo job running the sharded Gibbs loop on CPU."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    from bpmf_amd.dist import TorchComm, gibbs_sharded
    from tests import util
    from tests.oracle_engine import OracleEngine

    dataset, K, nsims, burnin, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    dist.init_process_group("gloo")
    M, Mt, T, Tt, nu, nm = {"tiny": util.tiny, "ml100k": util.ml100k, "blocks": util.blocks}[dataset]()
    comm = TorchComm("cpu")
    res = gibbs_sharded(OracleEngine(K), comm, M, Mt, T, nu, nm, nsims=nsims, burnin=burnin)
    np.savez(out + ".rank%d.npz" % comm.rank, U=res["U"], V=res["V"], U_replica=res["U_replica"], V_replica=res["V_replica"], rmse=res["rmse"], rmse_avg=res["rmse_avg"],
             norm_u=res["norm_u"], norm_m=res["norm_m"], final=res["final_rmse_avg"], conn_used=np.asarray(res["conn_used"]),
             dom_m=np.asarray(res["dom_m"]), dom_u=np.asarray(res["dom_u"]))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
