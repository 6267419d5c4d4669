"""Where the one-time hand-over of the host matrices goes (bench.py's `handover`): python tools/handover_breakdown.py [ml1m|chembl]"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bpmf_amd
from bpmf_amd import synth
wl = sys.argv[1] if len(sys.argv) > 1 else "ml1m"
K = 32 if wl == "ml1m" else 64
M, Mt, T, Tt, nu, nm = synth.ml1m_shaped(seed=42) if wl == "ml1m" else synth.ratings(483500, 5775, 1_023_952, seed=42, real_valued=True)
mean = float(np.sum(M[2])) / len(M[2])
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
for rep in range(3):
    eng = bpmf_amd.HipEngine(K)
    t = [time.perf_counter()]
    def lap(): eng.sync(); torch.cuda.synchronize(); t.append(time.perf_counter())
    sm = eng.side_create(nm, nu, M[0], M[1], M[2], mean); lap()
    su = eng.side_create(nu, nm, Mt[0], Mt[1], Mt[2], mean); lap()
    tm = eng.test_create(sm, *T); lap()
    tu = eng.test_create(su, *Tt); lap()
    eng.test_set_twin(tm, tu); lap()
    eng.sys_sample(sm, su, 2.0); lap()
    eng.sys_sample(su, sm, 2.0); lap()
    d = np.diff(t) * 1e3
    print("%s rep %d ms: side(movs) %.2f  side(users) %.2f  test(movs) %.2f  test(users) %.2f  twin %.2f  first sample movs %.2f users %.2f" % ((wl, rep) + tuple(d)))
    eng.close()
