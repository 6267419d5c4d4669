import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bpmf_amd
from bpmf_amd import synth
from bpmf_amd.sys import Sys
M, Mt, T, Tt, nu, nm = synth.ml1m_shaped(seed=42)
t0 = time.time()
for rep in range(60):
    eng = bpmf_amd.HipEngine(32 if rep % 3 else 16)
    Sys.nsims, Sys.burnin, Sys.alpha = 30, 5, 2.0
    movies = Sys("movs", eng, M, nm, nu, T=T); users = Sys("users", eng, Mt, nu, nm)
    for i in range(30):
        movies.sample(users)
        users.sample(movies)
        if i > 0: movies.predict_finish()
        movies.predict_launch(users)
    movies.predict_finish()
    eng.close()
print("60 engines x 30 pipelined iterations ok in %.1f s" % (time.time() - t0))
eng = bpmf_amd.HipEngine(32)
movies = Sys("movs", eng, M, nm, nu, T=T); users = Sys("users", eng, Mt, nu, nm)
t0 = time.time()
n = 30000
for i in range(n):
    movies.sample(users)
    users.sample(movies)
    if i > 0: movies.predict_finish()
    movies.predict_launch(users)
movies.predict_finish(); eng.sync()
dt = time.time() - t0
print("%d pipelined iterations in %.2f s = %.4f ms/iter, rmse %.4f" % (n, dt, dt / n * 1e3, movies.rmse))
import resource
print("max RSS %.0f MB" % (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024))
eng.close()
