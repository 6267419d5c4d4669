#!/bin/bash
# K = 128: host timeline of the asynchronous path + in-kernel phase stamps + the time of hyper_finish on this host
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/k128; mkdir -p $O
BPMF_HIP_TRACE=1 timeout 600 python bench.py --workload ml1m_k128 --no-cpu-baseline --no-strong --steps 20 --warmup 5 --repeats 1 --prewarm-ms 0 > $O/trace.json 2> $O/trace.err
tail -130 $O/trace.err | head -120
BPMF_HIP_STAMPS=1 timeout 600 python bench.py --workload ml1m_k128 --no-cpu-baseline --no-strong --steps 20 --warmup 5 --repeats 1 > $O/stamps.json 2> $O/stamps.err
grep -i "stamp" $O/stamps.err | head -60
python - <<'PY'
import ctypes as C, numpy as np, time, os
lib = C.CDLL(os.path.join("bpmf_amd", "libbpmf_hip.so"))
K = 128; N = 6040
rng = np.random.default_rng(0)
X = rng.normal(size=(N, K)); cov = np.cov(X.T)
au = np.zeros((K, K)); z = np.zeros(K); mu = np.zeros(K); LU = np.zeros((K, K)); LF = np.zeros((K, K))
p = lambda a: a.ctypes.data_as(C.c_void_p)
lib.bpmf_hyper_draws(K, C.c_int64(N), 3, p(au), p(z))
for rep in range(3):
    t = time.perf_counter()
    for i in range(50): lib.bpmf_hyper_finish(K, C.c_int64(N), p(cov), None, p(au), p(z), p(mu), p(LU), p(LF))
    print("hyper_finish K=128: %.1f us" % ((time.perf_counter() - t) / 50 * 1e6))
t = time.perf_counter()
for i in range(20): lib.bpmf_hyper_draws(K, C.c_int64(N), 3 + i, p(au), p(z))
print("hyper_draws K=128: %.1f us" % ((time.perf_counter() - t) / 20 * 1e6))
PY
