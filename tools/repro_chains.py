"""Run-to-run bit-reproducibility of the pipelined chains on BASELINE's shapes (races between the fused launches, riders, gate, double
buffers, collector threads would show as differing bits):  python tools/repro_chains.py [N=10]"""
import sys, os, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np, bpmf_amd
from bpmf_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for shape, K, dt, it in (("ml1m", 32, "f64", 12), ("ml1m", 64, "f64", 6), ("chembl", 64, "f64", 4), ("ml1m", 16, "f64", 8), ("ml1m", 8, "f64", 8)):
    data = synth.ml1m_shaped(seed=42) if shape == "ml1m" else synth.ratings(483500, 5775, 1_023_952, seed=42, real_valued=True)
    M, Mt, T, Tt, nu, nm = data
    seen = {}
    for rep in range(N):
        eng = bpmf_amd.HipEngine(K, dtype=dt)
        res = bpmf_amd.gibbs(eng, M, Mt, T, nu, nm, nsims=it, burnin=1, Tt=Tt, pipelined=True)
        h = hashlib.sha1(np.ascontiguousarray(res["U"]).tobytes() + np.ascontiguousarray(res["V"]).tobytes() + np.asarray(res["rmse"]).tobytes()).hexdigest()[:16]
        seen[h] = seen.get(h, 0) + 1
        eng.close()
    print(shape, K, dt, seen, flush=True)
    assert len(seen) == 1
print("reproducible")
