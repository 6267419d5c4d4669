"""Gibbs iteration time on a synthetic rating matrix of a given shape:
   python tools/shape_bench.py K nusers nmovies nnz [steps] [real]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bpmf_amd
from bpmf_amd import synth
from bpmf_amd.sys import Sys

K, nu, nm, nnz = (int(x) for x in sys.argv[1:5])
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
real = len(sys.argv) > 6
t0 = time.time()
M, Mt, T, Tt, nu, nm = synth.ratings(nu, nm, nnz, seed=42, real_valued=real)
print("generated %d x %d, %d train / %d test ratings in %.1f s" % (nu, nm, M[0][-1], T[0][-1], time.time() - t0), flush=True)
eng = bpmf_amd.HipEngine(K, dtype="f32" if K == 128 else "f64")
movies = Sys("movs", eng, M, nm, nu, T=T); users = Sys("users", eng, Mt, nu, nm)
for _ in range(3):
    movies.sample(users); users.sample(movies); movies.predict(users)
eng.sync()
km = ku = 0.0
t0 = time.perf_counter()
if os.environ.get("SHAPE_PIPELINED") == "1":
    # the loop bench.py times: nothing is read back between the half-iterations
    nopred = os.environ.get("SHAPE_NO_PREDICT") == "1"      # (upper bound of what the evaluation costs the samplers it runs beside)
    for i in range(steps):
        movies.sample(users); users.sample(movies)
        if nopred: continue
        if i > 0: movies.predict_finish()
        movies.predict_launch(users)
    if not nopred: movies.predict_finish()
else:
    for _ in range(steps):
        movies.sample(users); km += eng.last_kernel_ms(movies.side)[0]
        users.sample(movies); ku += eng.last_kernel_ms(users.side)[0]
        movies.predict(users)
eng.sync()
dt = (time.perf_counter() - t0) / steps
flops = 2 * M[0][-1] * (K * (K + 1) + 2 * K) + (nu + nm) * (K ** 3 / 3 + 4 * K * K + 3 * K)
print("K=%d: %.3f ms/iter, %.2f M samples/s; sampler movies %.1f us users %.1f us; %.1f TF fp64; rmse %.4f" % (
    K, dt * 1e3, (nu + nm) / dt / 1e6, km / steps * 1e3, ku / steps * 1e3, flops / dt / 1e12, movies.rmse))
