"""Time of one Gibbs iteration in the BPMF_REDUCE formulation against the default one (ML-1M shape, K = 32 / 64)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bpmf_amd                       # noqa: E402
from bpmf_amd import synth            # noqa: E402
from bpmf_amd.sys import Sys          # noqa: E402

M, Mt, T, Tt, nu, nm = synth.ml1m_shaped()
mean = float(np.sum(M[2])) / len(M[2])
for K in (32, 64):
    for reduce in (False, True):
        eng = bpmf_amd.HipEngine(K)
        Sys.nsims, Sys.burnin, Sys.alpha = 100, 5, 2.0
        movies = Sys("movs", eng, M, nm, nu, T=T, mean_rating=mean)
        users = Sys("users", eng, Mt, nu, nm, mean_rating=mean)
        if reduce:
            eng.sys_set_reduce(movies.side, users.side, True)
        for i in range(20):
            movies.sample(users); users.sample(movies)
        movies.refresh()
        t = time.perf_counter()
        n = 100
        for i in range(n):
            movies.sample(users); users.sample(movies)
        movies.refresh(); users.refresh()
        dt = (time.perf_counter() - t) / n
        movies.predict(users)
        ms = eng.kernel_ms_sum(movies.side), eng.kernel_ms_sum(users.side)
        print("K=%d reduce=%d: %.3f ms per iteration, rmse %.4f, sampler ms per launch movs %.3f users %.3f" % (
            K, reduce, dt * 1e3, movies.rmse, ms[0][0] / max(ms[0][2], 1), ms[1][0] / max(ms[1][2], 1)))
        eng.close()
