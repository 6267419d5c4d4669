"""Run-to-run reproducibility of the K = 128 kernels (barrier / LDS hand-over bugs show up as differing bits): the same `-i 4 -b 1` chain
N times in fresh engines, sha-1 of U and V must be the same every time.   python tools/repro_k128.py [N=20]"""
import sys, os, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np, bpmf_amd
from bpmf_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M, Mt, T, Tt, nu, nm = synth.ml1m_shaped(seed=42)
for dt in ("f32", "f64"):
    seen = {}
    for rep in range(N):
        eng = bpmf_amd.HipEngine(128, dtype=dt)
        res = bpmf_amd.gibbs(eng, M, Mt, T, nu, nm, nsims=4, burnin=1, Tt=Tt, pipelined=True)
        h = hashlib.sha1(np.ascontiguousarray(res["U"]).tobytes() + np.ascontiguousarray(res["V"]).tobytes()).hexdigest()[:16]
        seen[h] = seen.get(h, 0) + 1
        eng.close()
    print(dt, seen)
    assert len(seen) == 1, "K = 128 %s chain is not reproducible run to run: %s" % (dt, seen)
print("reproducible")
