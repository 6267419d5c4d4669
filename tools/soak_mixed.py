import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bpmf_amd
from bpmf_amd import synth
from bpmf_amd.sys import Sys
M, Mt, T, Tt, nu, nm = synth.ratings(3000, 1200, 120000, seed=5)
Ms, Mts, Ts, Tts, nus, nms = synth.ratings(30000, 300, 60000, seed=6, real_valued=True)   # sparse columns: low-rank path at K=64
t0 = time.time()
for rep in range(24):
    K = (8, 16, 32, 64, 128, 64)[rep % 6]
    eng = bpmf_amd.HipEngine(K, dtype="f32" if K == 128 else "f64")
    Sys.nsims, Sys.burnin, Sys.alpha = 12, 3, 2.0
    data = (Ms, Mts, Ts, nus, nms) if rep % 6 == 5 else (M, Mt, T, nu, nm)
    movies = Sys("movs", eng, data[0], data[4], data[3], T=data[2]); users = Sys("users", eng, data[1], data[3], data[4])
    for i in range(12):
        movies.sample(users); users.sample(movies)
        if i > 0: movies.predict_finish()
        movies.predict_launch(users)
        if i == 7: eng.sys_state(users.side)
    movies.predict_finish()
    assert np.isfinite(movies.rmse) and np.all(np.isfinite(users.items()))
    eng.close()
print("24 engines (K = 8..128, both sampler families, low-rank side) x 12 pipelined iterations ok in %.1f s" % (time.time() - t0))
