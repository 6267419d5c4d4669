"""pair launch vs two launches: plain chain, bit for bit (ML-100K)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bpmf_amd
from tests import util
K = int(sys.argv[1]) if len(sys.argv) > 1 else 16
M, Mt, T, Tt, nu, nm = util.ml100k()
res = {}
for pair in ("0", "1"):
    os.environ["BPMF_HIP_PAIR"] = pair
    eng = bpmf_amd.HipEngine(K)
    r = bpmf_amd.gibbs(eng, M, Mt, T, nu, nm, nsims=6, burnin=2)
    res[pair] = r
    print("pair", pair, "kernel", eng.kernel_name(r["users"].side), "rmse", r["rmse"])
    eng.close()
a, b = res["0"], res["1"]
print("U equal", np.array_equal(a["U"], b["U"]), np.abs(a["U"] - b["U"]).max(), "V equal", np.array_equal(a["V"], b["V"]), np.abs(a["V"] - b["V"]).max())
