import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, bpmf_amd
from bpmf_amd import synth
M, Mt, T, Tt, nu, nm = synth.ml1m_shaped(seed=42)
for dt in ("f64", "f32"):
    eng = bpmf_amd.HipEngine(128, dtype=dt)
    t0 = time.time()
    res = bpmf_amd.gibbs(eng, M, Mt, T, nu, nm, nsims=300, burnin=20, Tt=Tt, pipelined=True)
    print(dt, "300 iterations in %.2f s, rmse first %.4f last %.4f avg %.4f, finite %s" % (time.time() - t0, res["rmse"][0], res["rmse"][-1], res["final_rmse_avg"], bool(np.all(np.isfinite(res["U"])) and np.all(np.isfinite(res["V"])))))
    eng.close()
