"""ctypes wrapper around oracle/libbpmf_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (and only as the checker / the reported CPU baseline).  The product
package bpmf_amd never does.  See the header of oracle/bpmf_oracle.c for what
the oracle restates and how it is pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def build(native=False):
    """(Re)build the oracle libraries with gcc.  `native=True` rebuilds the
    timed variant with -march=native on the machine it will be timed on."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    if native:
        subprocess.check_call(["make", "-s", "-C", _HERE, "native"])


def _load(name):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    return C.CDLL(path)


class Oracle:
    """One loaded oracle library (`fast=True`: the -O3/OpenMP CPU-baseline build)."""

    def __init__(self, fast=False):
        self.lib = lib = _load("libbpmf_oracle_fast.so" if fast else "libbpmf_oracle.so")
        lib.bpmf_oracle_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
        lib.bpmf_oracle_philox4x32_10.restype = None
        lib.bpmf_oracle_randn_stream.argtypes = [C.c_uint32, C.c_int, _f64p]
        lib.bpmf_oracle_words_stream.argtypes = [C.c_uint32, C.c_int, _u32p]
        lib.bpmf_oracle_gamma_stream.argtypes = [C.c_uint32, C.c_int, _f64p, _f64p, _f64p]
        lib.bpmf_oracle_hyper_sample.argtypes = [C.c_int, C.c_int, _f64p, C.c_void_p, C.c_uint32, _f64p, _f64p, _f64p]
        lib.bpmf_oracle_hyper_sample.restype = C.c_int
        lib.bpmf_oracle_sample_side.argtypes = [
            C.c_int, C.c_int64, C.c_int64, _i64p, _i32p, _f64p, C.c_double, C.c_double, _f64p, _f64p,
            C.c_int, _f64p, _f64p, _f64p, _f64p, _f64p, C.c_int]
        lib.bpmf_oracle_sample_side.restype = C.c_int64
        lib.bpmf_oracle_sample_column.argtypes = [C.c_int, C.c_int64, C.c_int64, _i32p, _f64p, C.c_double, C.c_double, _f64p, C.c_int,
                                                  _f64p, _f64p, _f64p]
        lib.bpmf_oracle_sample_column.restype = C.c_int
        lib.bpmf_oracle_sample_side_prop.argtypes = [
            C.c_int, C.c_int64, C.c_int64, _i64p, _i32p, _f64p, C.c_double, C.c_double, _f64p, _f64p,
            C.c_int, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p, C.c_int]
        lib.bpmf_oracle_sample_side_prop.restype = C.c_int64
        lib.bpmf_oracle_sample_side_nocov.argtypes = lib.bpmf_oracle_sample_side.argtypes
        lib.bpmf_oracle_sample_side_nocov.restype = C.c_int64
        lib.bpmf_oracle_cov.argtypes = [C.c_int, C.c_int64, _f64p, _f64p, _f64p]
        lib.bpmf_oracle_predict.argtypes = [
            C.c_int, C.c_int64, C.c_int64, _i64p, _i32p, _f64p, _f64p, _f64p, C.c_double, C.c_int,
            _f64p, _f64p, _f64p, _f64p, _i64p, C.c_int]
        lib.bpmf_oracle_gibbs.argtypes = (
            [C.c_int, C.c_int64, C.c_int64] + [_i64p, _i32p, _f64p] * 4 +
            [C.c_double, C.c_int, C.c_int, C.c_int] + [_f64p] * 4 + [_f64p] * 5 + [_f64p, C.c_void_p])
        lib.bpmf_oracle_gibbs.restype = C.c_int64

    # -- RNG layer ---------------------------------------------------------
    def philox(self, ctr, key):
        out = np.zeros(4, np.uint32)
        self.lib.bpmf_oracle_philox4x32_10(np.asarray(ctr, np.uint32), np.asarray(key, np.uint32), out)
        return out

    def words(self, counter, n):
        out = np.zeros(n, np.uint32)
        self.lib.bpmf_oracle_words_stream(int(counter) & 0xFFFFFFFF, n, out)
        return out

    def randn(self, counter, n):
        out = np.zeros(n, np.float64)
        self.lib.bpmf_oracle_randn_stream(int(counter) & 0xFFFFFFFF, n, out)
        return out

    def gamma_stream(self, counter, alphas):
        alphas = np.ascontiguousarray(alphas, np.float64)
        g = np.zeros(len(alphas)); z = np.zeros(len(alphas))
        self.lib.bpmf_oracle_gamma_stream(int(counter) & 0xFFFFFFFF, len(alphas), alphas, g, z)
        return g, z

    # -- hyper parameters ---------------------------------------------------
    def hyper_sample(self, K, N, cov, counter, Um=None):
        """HyperParams::sample; returns (mu[K], LambdaU[K,K], LambdaF[K,K]) with
        the matrices as numpy arrays indexed [row, col]."""
        cov_cm = np.asfortranarray(cov, np.float64)
        mu = np.zeros(K); LU = np.zeros((K, K), order="F"); LF = np.zeros((K, K), order="F")
        um = None
        if Um is not None:
            um_arr = np.ascontiguousarray(Um, np.float64)
            um = um_arr.ctypes.data_as(C.c_void_p)
        rc = self.lib.bpmf_oracle_hyper_sample(K, int(N), cov_cm.T, um, int(counter) & 0xFFFFFFFF, mu, LU.T, LF.T)
        if rc:
            raise RuntimeError("oracle hyper_sample failed rc=%d" % rc)
        return mu, LU, LF

    # -- sampling ------------------------------------------------------------
    def sample_side(self, K, csc, mean_rating, alpha, other_items, items, it, mu, LambdaF,
                    from_=0, to=None, nthreads=1, prop_lambda=None, no_covariance=False):
        """Samples columns [from_,to) of `items` in place ([N,K] C-order arrays =
        column-major K x N); returns (sum[K], prod[K,K], norm).  prop_lambda: [N, K, K] array of
        per-column prior precisions (the propagated posterior of -m / -l), each stored transposed,
        i.e. prop_lambda[i].T is the matrix (column-major K x K per column, like *-Lambda.ddm)."""
        colptr, rowidx, vals = csc
        n = len(colptr) - 1
        to = n if to is None else to
        s = np.zeros(K); prod = np.zeros((K, K), order="F"); nrm = np.zeros(1)
        LF = np.asfortranarray(LambdaF, np.float64)
        if prop_lambda is not None:
            pl = np.ascontiguousarray(prop_lambda, np.float64)
            rc = self.lib.bpmf_oracle_sample_side_prop(
                K, from_, to, colptr, rowidx, vals, float(mean_rating), float(alpha), other_items, items,
                int(it), np.ascontiguousarray(mu, np.float64), LF.T, pl, s, prod.T, nrm, nthreads)
            if rc:
                raise RuntimeError("Cholesky failed in column %d" % (-rc - 1))
            return s, prod, float(nrm[0])
        fn = self.lib.bpmf_oracle_sample_side_nocov if no_covariance else self.lib.bpmf_oracle_sample_side
        rc = fn(K, from_, to, colptr, rowidx, vals, float(mean_rating), float(alpha), other_items, items,
                int(it), np.ascontiguousarray(mu, np.float64), LF.T, s, prod.T, nrm, nthreads)
        if rc:
            raise RuntimeError("Cholesky failed in column %d" % (-rc - 1))
        return s, prod, float(nrm[0])

    def sample_column(self, K, idx, rowidx, vals, mean_rating, alpha, other_items, it, mu, LambdaF):
        """One column (global id `idx`) from its own ratings; rowidx index the rows of `other_items` ([n_rows, K])."""
        out = np.zeros(K)
        LF = np.asfortranarray(LambdaF, np.float64)
        rc = self.lib.bpmf_oracle_sample_column(K, int(idx), len(rowidx), np.ascontiguousarray(rowidx, np.int32),
                                                np.ascontiguousarray(vals, np.float64), float(mean_rating), float(alpha),
                                                np.ascontiguousarray(other_items, np.float64), int(it),
                                                np.ascontiguousarray(mu, np.float64), LF.T, out)
        if rc:
            raise RuntimeError("Cholesky failed in column %d" % idx)
        return out

    # -- BPMF_REDUCE build (c++/sample.cpp:234-246, 289-291, 375-377; c++/mpi_reduce.h) -----------------
    def precompute(self, K, csc, mean_rating, alpha, other_items, other_from=0, other_to=None):
        """preComputeMuLambda of the Sys whose matrix is `csc`: (precMu [N, K], precLambda [N, K, K] -- each [i].T the
        column-major matrix, upper triangle) from the rows [other_from, other_to) of `other_items`."""
        colptr, rowidx, vals = csc
        n = len(colptr) - 1
        other_to = len(other_items) if other_to is None else other_to
        pmu = np.zeros((n, K)); plam = np.zeros((n, K, K))
        self.lib.bpmf_oracle_precompute.restype = None
        self.lib.bpmf_oracle_precompute(K, C.c_int64(n), colptr.ctypes.data_as(C.c_void_p), rowidx.ctypes.data_as(C.c_void_p),
                                        vals.ctypes.data_as(C.c_void_p), C.c_double(mean_rating), C.c_double(alpha),
                                        other_items.ctypes.data_as(C.c_void_p), C.c_int64(other_from), C.c_int64(other_to),
                                        pmu.ctypes.data_as(C.c_void_p), plam.ctypes.data_as(C.c_void_p))
        return pmu, plam

    def sample_side_prec(self, K, alpha, prec, items, it, mu, LambdaF, from_=0, to=None, nthreads=1, no_covariance=False):
        """Sys::sample(Sys&) of the BPMF_REDUCE build over columns [from_, to): prec = (precMu, precLambda) already
        summed over the ranks."""
        pmu, plam = prec
        to = len(items) if to is None else to
        s = np.zeros(K); prod = np.zeros((K, K), order="F"); nrm = np.zeros(1)
        LF = np.asfortranarray(LambdaF, np.float64)
        f = self.lib.bpmf_oracle_sample_side_prec
        f.restype = C.c_int64
        rc = f(K, C.c_int64(from_), C.c_int64(to), C.c_double(alpha), pmu.ctypes.data_as(C.c_void_p), plam.ctypes.data_as(C.c_void_p),
               items.ctypes.data_as(C.c_void_p), int(it), np.ascontiguousarray(mu, np.float64).ctypes.data_as(C.c_void_p),
               LF.T.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p), prod.T.ctypes.data_as(C.c_void_p),
               nrm.ctypes.data_as(C.c_void_p), int(nthreads), 1 if no_covariance else 0)
        if rc:
            raise RuntimeError("Cholesky failed in column %d" % (-rc - 1))
        return s, prod, float(nrm[0])

    def gibbs_reduce(self, K, M, Mt, T, alpha=2.0, nsims=4, burnin=0, bounds_m=None, bounds_u=None):
        """The Gibbs loop of the BPMF_REDUCE build for `nranks` simulated ranks (bounds_*: column ranges per rank; None: one
        rank): per half-iteration the parts of all ranks are summed in rank order (MPI_Reduce), the owners sample their
        ranges, every rank precomputes the other side's parts from its own fresh columns.  Returns U, V, rmse per iteration."""
        nmovies, nusers = len(M[0]) - 1, len(Mt[0]) - 1
        bounds_m = [0, nmovies] if bounds_m is None else list(bounds_m)
        bounds_u = [0, nusers] if bounds_u is None else list(bounds_u)
        nr = len(bounds_m) - 1
        mean_m = float(M[2].sum() / len(M[2])); mean_u = float(Mt[2].sum() / len(Mt[2]))
        U = np.zeros((nusers, K)); V = np.zeros((nmovies, K))
        cov_m = np.zeros((K, K)); cov_u = np.zeros((K, K))
        zero = lambda n: [(np.zeros((n, K)), np.zeros((n, K, K))) for _ in range(nr)]      # Sys::init, :192-195
        prec_m, prec_u = zero(nmovies), zero(nusers)
        Pavg = T[2].copy(); Pm2 = T[2].copy()
        rmse = []
        for it in range(nsims):
            for (n, csc_o, mean_o, X, cov, prec, prec_o, b, who) in (
                    (nmovies, Mt, mean_u, V, cov_m, prec_m, prec_u, bounds_m, "m"), (nusers, M, mean_m, U, cov_u, prec_u, prec_m, bounds_u, "u")):
                mu, LU, LF = self.hyper_sample(K, n, cov, it)
                tot = (sum(p[0] for p in prec), sum(p[1] for p in prec))                   # the reduction onto the owners
                s = np.zeros(K); prod = np.zeros((K, K))
                for r in range(nr):
                    sr, pr, _ = self.sample_side_prec(K, alpha, tot, X, it, mu, LF, b[r], b[r + 1])
                    s += sr; prod += pr
                for r in range(nr):                                                         # other.preComputeMuLambda(*this)
                    prec_o[r] = self.precompute(K, csc_o, mean_o, alpha, X, b[r], b[r + 1])
                cov[:] = self.cov(K, n, s, prod)
            nn = 0 if it < burnin else it - burnin
            se, _, nump = self.predict(K, T, V, U, mean_m, nn, Pavg, Pm2)
            rmse.append(float(np.sqrt(se / max(nump, 1))))
        return dict(U=U, V=V, rmse=np.array(rmse))

    def cov(self, K, N, s, prod):
        c = np.zeros((K, K), order="F")
        self.lib.bpmf_oracle_cov(K, N, np.ascontiguousarray(s), np.asfortranarray(prod).T, c.T)
        return c

    def predict(self, K, tcsc, items, other_items, mean_rating, n, Pavg, Pm2, from_=0, to=None, nthreads=1):
        colptr, rowidx, vals = tcsc
        to = (len(colptr) - 1) if to is None else to
        se = np.zeros(1); sea = np.zeros(1); nump = np.zeros(1, np.int64)
        self.lib.bpmf_oracle_predict(K, from_, to, colptr, rowidx, vals, items, other_items, float(mean_rating),
                                     int(n), Pavg, Pm2, se, sea, nump, nthreads)
        return float(se[0]), float(sea[0]), int(nump[0])

    def gibbs(self, K, M, Mt, T, Tt, alpha=2.0, nsims=20, burnin=5, nthreads=1, trace=False):
        """Full NO_COMM run.  M/T: CSC with one column per movie, Mt/Tt their
        transposes.  Returns a dict with U, V ([N,K]), per-iteration rmse etc."""
        nmovies = len(M[0]) - 1
        nusers = len(Mt[0]) - 1
        U = np.zeros((nusers, K)); V = np.zeros((nmovies, K))
        nnzt = int(T[0][-1])
        Pavg = np.zeros(max(nnzt, 1)); Pm2 = np.zeros(max(nnzt, 1))
        arrs = [np.zeros(max(nsims, 1)) for _ in range(5)]
        final = np.zeros(2)
        tr = np.zeros(nsims * 2 * (K + K * K)) if trace else None
        rc = self.lib.bpmf_oracle_gibbs(
            K, nusers, nmovies, *M, *Mt, *T, *Tt, float(alpha), nsims, burnin, nthreads,
            U, V, Pavg, Pm2, *arrs, final, tr.ctypes.data_as(C.c_void_p) if trace else None)
        if rc:
            raise RuntimeError("oracle gibbs failed rc=%d" % rc)
        out = dict(U=U, V=V, Pavg=Pavg[:nnzt], Pm2=Pm2[:nnzt], rmse=arrs[0][:nsims], rmse_avg=arrs[1][:nsims],
                   norm_u=arrs[2][:nsims], norm_m=arrs[3][:nsims], secs=arrs[4][:nsims],
                   final_rmse_avg=float(final[0]), num_predict=int(final[1]))
        if trace:
            out["trace"] = tr.reshape(nsims, 2, K + K * K)
        return out


class PinLibstdcxx:
    """The real libstdc++ <random> distributions on the oracle's Philox stream."""

    def __init__(self):
        self.lib = lib = _load("libpin_libstdcxx.so")
        lib.pin_randn_stream.argtypes = [C.c_uint32, C.c_int, _f64p]
        lib.pin_gamma_stream.argtypes = [C.c_uint32, C.c_int, _f64p, _f64p, _f64p]

    def randn(self, counter, n):
        out = np.zeros(n)
        self.lib.pin_randn_stream(int(counter) & 0xFFFFFFFF, n, out)
        return out

    def gamma_stream(self, counter, alphas):
        alphas = np.ascontiguousarray(alphas, np.float64)
        g = np.zeros(len(alphas)); z = np.zeros(len(alphas))
        self.lib.pin_gamma_stream(int(counter) & 0xFFFFFFFF, len(alphas), alphas, g, z)
        return g, z


# ---- helpers shared by tests / bench (pure numpy, no product code) -----------
def csc_from_coo(rows, cols, vals, nrows, ncols):
    """Sorted CSC with duplicates summed (Eigen setFromTriplets semantics,
    c++/io.cpp:521): returns (colptr int64[ncols+1], rowidx int32[nnz], vals f64[nnz])."""
    import scipy.sparse as sp
    m = sp.coo_matrix((np.asarray(vals, np.float64), (np.asarray(rows), np.asarray(cols))), shape=(nrows, ncols)).tocsc()
    m.sum_duplicates()
    m.sort_indices()
    return (np.ascontiguousarray(m.indptr, np.int64), np.ascontiguousarray(m.indices, np.int32),
            np.ascontiguousarray(m.data, np.float64))


def transpose_csc(csc, nrows):
    import scipy.sparse as sp
    colptr, rowidx, vals = csc
    m = sp.csc_matrix((vals, rowidx, colptr), shape=(nrows, len(colptr) - 1)).T.tocsc()
    m.sort_indices()
    return (np.ascontiguousarray(m.indptr, np.int64), np.ascontiguousarray(m.indices, np.int32),
            np.ascontiguousarray(m.data, np.float64))
