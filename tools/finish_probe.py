"""Latency vs throughput of the per-column finish (Cholesky + solves + RNG): columns with a
single rating make the Gram negligible.  Prints kernel ms per launch for growing column counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bpmf_amd

K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nnz_per = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng = bpmf_amd.HipEngine(K)
nrows = 1000
rng = np.random.default_rng(0)
U = rng.standard_normal((nrows, K))
ot = None
for n in (1, 64, 256, 1024, 2048, 4096, 8192, 16384, 65536):
    colptr = (np.arange(n + 1) * nnz_per).astype(np.int64)
    rowidx = rng.integers(0, nrows, n * nnz_per).astype(np.int32)
    rowidx = np.sort(rowidx.reshape(n, nnz_per), axis=1).ravel()
    vals = rng.integers(1, 6, n * nnz_per).astype(np.float64)
    me = eng.side_create(n, nrows, colptr, rowidx, vals, 3.0)
    ot = eng.side_create(nrows, n, np.zeros(nrows + 1, np.int64), np.zeros(0, np.int32), np.zeros(0), 0.0)
    eng.set_items(ot, U)
    mu = np.zeros(K); LF = np.eye(K) * 2.0
    ts = []
    for it in range(6):
        eng.sample_side(me, ot, it, 2.0, mu, LF)
        ts.append(eng.last_kernel_ms(me)[0])
    t = float(np.median(ts[1:]))
    print("K=%d nnz/col=%d cols=%6d  kernel %.1f us   %.2f ns/col   (%.0f cycles@2.4GHz per col per SIMD-slot)" % (
        K, nnz_per, n, t * 1e3, t * 1e6 / n, t * 1e-3 * 2.4e9 / max(1.0, n / 4096.0) if n >= 4096 else t * 1e-3 * 2.4e9))
    eng.side_destroy(me); eng.side_destroy(ot)
