#!/usr/bin/env python
"""Independent numpy / pure-Python restatement of the reference's Gibbs path, used ONLY to
generate the golden vectors in this directory (run in the authoring container; the .npz files
are committed, this script documents how they were made and can regenerate them).

It is deliberately written against the reference's source (c++/sample.cpp, c++/mvnormal.cpp,
c++/bpmf.h, libstdc++ <random>) and NOT against oracle/bpmf_oracle.c: Philox runs on Python
integers, the dense algebra goes through LAPACK (numpy.linalg / scipy.linalg), so an error in
the C oracle's hand-written loops or in its reading of the stream layout shows up as a mismatch
in tests/test_oracle_golden.py.  Agreement is to rounding (1e-10), not bitwise: LAPACK orders
its sums differently.

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.npz
"""
import math
import os
import sys

import numpy as np
import scipy.io
import scipy.linalg as sla
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
M32 = 0xFFFFFFFF


# ---- Philox4x32-10 + MicroURNG (c++/mvnormal.cpp:18-23,34-39) ---------------------------------
def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for r in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


class MicroURNG:
    """r123::MicroURNG<Philox4x32>: counter {c,0,0,n}, key {42,0}, words handed out last to first."""

    def __init__(self, c):
        self.reset(c)

    def reset(self, c):
        self.c, self.n, self.buf = c & M32, 0, []

    def __call__(self):
        if not self.buf:
            self.buf = list(philox4x32_10((self.c, 0, 0, self.n & M32), (42, 0)))   # pop() takes w3 first
            self.n += 1
        return self.buf.pop()


def canonical(rng):
    """std::generate_canonical<double,53> with a 32-bit engine (two calls, first = low word)."""
    lo = rng(); hi = rng()
    s = float(lo) + float(hi) * 4294967296.0
    r = s / 18446744073709551616.0
    return r if r < 1.0 else math.nextafter(1.0, 0.0)


class Normal:
    """std::normal_distribution<double> (polar method with a saved second variate)."""

    def __init__(self):
        self.saved = None

    def __call__(self, rng):
        if self.saved is not None:
            r, self.saved = self.saved, None
            return r
        while True:
            x = 2.0 * canonical(rng) - 1.0
            y = 2.0 * canonical(rng) - 1.0
            r2 = x * x + y * y
            if not (r2 > 1.0 or r2 == 0.0):
                break
        mult = math.sqrt(-2 * math.log(r2) / r2)
        self.saved = x * mult
        return y * mult


def randn(rng):
    return Normal()(rng)            # a temporary distribution per call: the saved variate is dropped


def gamma(rng, alpha):
    """std::gamma_distribution<double>(alpha, 1) constructed fresh (Marsaglia-Tsang)."""
    malpha = alpha + 1.0 if alpha < 1.0 else alpha
    a1 = malpha - 1.0 / 3.0
    a2 = 1.0 / math.sqrt(9.0 * a1)
    nd = Normal()
    while True:
        while True:
            n = nd(rng)
            v = 1.0 + a2 * n
            if v > 0.0:
                break
        v = v * v * v
        u = canonical(rng)
        if not (u > 1.0 - 0.0331 * n * n * n * n and math.log(u) > 0.5 * n * n + a1 * (1.0 - v + math.log(v))):
            break
    if alpha == malpha:
        return a1 * v
    while True:
        u = canonical(rng)
        if u != 0.0:
            break
    return math.pow(u, 1.0 / alpha) * a1 * v


# ---- hyper parameters (c++/mvnormal.cpp:56-135, c++/bpmf.h:78-104) -----------------------------
def hyper_sample(K, N, cov, counter):
    rng = MicroURNG(counter)
    kappa, nu = 2.0, K
    Um = np.zeros(K)                                   # the reference's member `sum` stays 0 (Q1)
    mu_m = -Um
    mu_c = (kappa * 0 + N * Um) / (kappa + N)
    kappa_c = kappa + N
    kappa_m = kappa * N / (kappa + N)
    X = np.eye(K) + N * cov + kappa_m * np.outer(mu_m, mu_m)
    T_c = np.linalg.inv(X)
    nu_c = nu + N
    R = np.linalg.cholesky(np.tril(T_c) + np.tril(T_c, -1).T).T        # chol.matrixU() (LLT reads the lower triangle)
    au = np.zeros((K, K))
    for i in range(K):
        au[i, i] = math.sqrt(2.0 * gamma(rng, 0.5 * (nu_c - i)))
        for _ in range(K - i - 1):
            randn(rng)                                  # VectorXd r = nrandn(K-i-1): discarded (Q5)
        for j in range(i + 1, K):
            au[i, j] = randn(rng)
    U = au @ R
    r = np.array([randn(rng) for _ in range(K)])
    x = sla.solve_triangular(U, r, lower=False)
    mu = x / math.sqrt(kappa_c) + mu_c
    LambdaF = np.triu(U).T @ U
    return mu, U, LambdaF


# ---- column update (c++/sample.cpp:248-336) ------------------------------------------------------
def sample_side(K, csc, mean, alpha, other, it, mu, LF):
    colptr, rowidx, vals = csc
    n = len(colptr) - 1
    out = np.zeros((n, K))
    Lmu = LF @ mu
    for idx in range(n):
        rng = MicroURNG(((idx + 1) * K * (it + 1)) & M32)
        rows = rowidx[colptr[idx]:colptr[idx + 1]]
        Y = other[rows]                                 # nnz x K
        rr = Lmu + Y.T @ ((vals[colptr[idx]:colptr[idx + 1]] - mean) * alpha)
        MM = LF + alpha * (Y.T @ Y)
        L = np.linalg.cholesky(MM)
        y = sla.solve_triangular(L, rr, lower=True)
        y = y + np.array([randn(rng) for _ in range(K)])
        out[idx] = sla.solve_triangular(L.T, y, lower=False)
    return out


def predict(T, items, other, mean, n, Pavg, Pm2):
    colptr, rowidx, vals = T
    se = se_avg = 0.0
    for k in range(len(colptr) - 1):
        for p in range(colptr[k], colptr[k + 1]):
            pred = float(items[k] @ other[rowidx[p]]) + mean
            se += (vals[p] - pred) ** 2
            avg = Pavg[p]
            delta = pred - avg
            avg = pred if n == 0 else avg + delta / n
            Pavg[p] = avg
            Pm2[p] = 0.0 if n == 0 else Pm2[p] + delta * (pred - avg)
            se_avg += (vals[p] - avg) ** 2
    cnt = int(colptr[-1])
    return math.sqrt(se / cnt), math.sqrt(se_avg / cnt)


def csc(m):
    m = m.tocsc(); m.sum_duplicates(); m.sort_indices()
    return m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data.astype(np.float64)


def load(train, test):
    m = scipy.io.mmread(os.path.join(HERE, train)).tocoo(); t = scipy.io.mmread(os.path.join(HERE, test)).tocoo()
    nr = max(m.shape[0], t.shape[0]); nc = max(m.shape[1], t.shape[1])
    M = sp.coo_matrix((m.data.astype(float), (m.row, m.col)), shape=(nr, nc)).tocsc()
    T = sp.coo_matrix((t.data.astype(float), (t.row, t.col)), shape=(nr, nc)).tocsc()
    return csc(M), csc(M.T), csc(T), nr, nc


def gibbs(K, M, Mt, T, nu, nm, nsims, burnin, alpha=2.0, keep=None):
    """main() loop, NO_COMM (c++/bpmf.cpp:180-253).  Returns the trace."""
    mean = M[2].sum() / len(M[2])
    U = np.zeros((nu, K)); V = np.zeros((nm, K))
    cov_m = np.zeros((K, K)); cov_u = np.zeros((K, K))
    Pavg = T[2].copy(); Pm2 = T[2].copy()
    tr = dict(rmse=[], rmse_avg=[], norm_u=[], norm_m=[], mu_m=[], LF_m=[], mu_u=[], LF_u=[], U=[], V=[])
    for it in range(nsims):
        mu, LU, LF = hyper_sample(K, nm, cov_m, it)
        tr["mu_m"].append(mu); tr["LF_m"].append(LF)
        V = sample_side(K, M, mean, alpha, U, it, mu, LF)
        s = V.sum(0); cov_m = (V.T @ V - np.outer(s, s) / nm) / (nm - 1)
        mu, LU, LF = hyper_sample(K, nu, cov_u, it)
        tr["mu_u"].append(mu); tr["LF_u"].append(LF)
        U = sample_side(K, Mt, mean, alpha, V, it, mu, LF)
        s = U.sum(0); cov_u = (U.T @ U - np.outer(s, s) / nu) / (nu - 1)
        n = 0 if it < burnin else it - burnin
        r, ra = predict(T, V, U, mean, n, Pavg, Pm2)
        tr["rmse"].append(r); tr["rmse_avg"].append(ra)
        tr["norm_u"].append(math.sqrt((U * U).sum())); tr["norm_m"].append(math.sqrt((V * V).sum()))
        if keep is None:
            tr["U"].append(U.copy()); tr["V"].append(V.copy())
        else:
            tr["U"].append(U[keep[0]].copy()); tr["V"].append(V[keep[1]].copy())
    n = 0 if nsims - 1 < burnin else nsims - 1 - burnin
    r, ra = predict(T, V, U, mean, n, Pavg, Pm2)            # the extra predict before "Final Avg RMSE" (Q6)
    tr["final_rmse_avg"] = ra
    tr["Pavg"] = Pavg; tr["Pm2"] = Pm2
    return {k: np.asarray(v) for k, v in tr.items()}


def main():
    out = {}
    # 1. RNG layer: first 64 normals of a few streams; words; gamma draws
    for c in (0, 1, 32, 2 ** 32 - 1):
        rng = MicroURNG(c)
        out["randn_%d" % c] = np.array([randn(rng) for _ in range(64)])
    rng = MicroURNG(7)
    out["words_7"] = np.array([rng() for _ in range(16)], dtype=np.uint32)
    rng = MicroURNG(3)
    alphas = np.array([0.5 * k for k in range(1, 41)] + [0.25, 471.5, 3024.0])
    out["gamma_alphas"] = alphas
    out["gamma_3"] = np.array([gamma(rng, a) for a in alphas])
    np.savez_compressed(os.path.join(HERE, "rng.npz"), **out)

    # 2. hyper-parameter draws
    out = {}
    gen = np.random.default_rng(12345)
    for K, N in ((8, 4), (16, 50), (32, 943)):
        A = gen.standard_normal((K, 3 * K)); cov = A @ A.T / (3 * K)
        for counter in (0, 5):
            mu, LU, LF = hyper_sample(K, N, cov, counter)
            out["hyper_K%d_N%d_c%d_cov" % (K, N, counter)] = cov
            out["hyper_K%d_N%d_c%d_mu" % (K, N, counter)] = mu
            out["hyper_K%d_N%d_c%d_LU" % (K, N, counter)] = LU
            out["hyper_K%d_N%d_c%d_LF" % (K, N, counter)] = LF
    np.savez_compressed(os.path.join(HERE, "hyper.npz"), **out)

    # 3. tiny, K = 8, the reference's run_test.sh settings (-i 9 -b 0)
    M, Mt, T, nu, nm = load("tiny-train.mtx", "tiny-test.mtx")
    tr = gibbs(8, M, Mt, T, nu, nm, nsims=9, burnin=0)
    np.savez_compressed(os.path.join(HERE, "tiny_k8.npz"), **tr)
    print("tiny K=8: Final Avg RMSE", tr["final_rmse_avg"])

    # 4. MovieLens-100K, K = 32, first 3 iterations (burnin 1 so that the averaging branch runs);
    #    16 selected columns per side are kept
    M, Mt, T, nu, nm = load("ml100k-train.mtx.gz", "ml100k-test.mtx.gz")
    nnz_m = np.diff(M[0]); nnz_u = np.diff(Mt[0])
    keep_m = np.unique(np.concatenate([np.argsort(nnz_m)[:4], np.argsort(nnz_m)[-4:], np.arange(0, nm, nm // 8)[:8]]))
    keep_u = np.unique(np.concatenate([np.argsort(nnz_u)[:4], np.argsort(nnz_u)[-4:], np.arange(0, nu, nu // 8)[:8]]))
    tr = gibbs(32, M, Mt, T, nu, nm, nsims=3, burnin=1, keep=(keep_u, keep_m))
    tr["keep_u"] = keep_u; tr["keep_m"] = keep_m
    del tr["Pavg"], tr["Pm2"]
    np.savez_compressed(os.path.join(HERE, "ml100k_k32.npz"), **tr)
    print("ML-100K K=32, 3 iterations: rmse", tr["rmse"])


if __name__ == "__main__":
    sys.exit(main())
