"""Host-side mirror of the reference's `struct Sys` / `HyperParams` (c++/bpmf.h:78-239).

Same names, same argument meaning, same quirks, so that the
parity tests read like the reference: `movies.sample(users); users.sample(movies);
movies.predict(users)`.  All column work happens in the engine (HipEngine ->
libbpmf_hip.so); this file only sequences the calls the way `Sys::sample(Sys&)`
(c++/sample.cpp:341-385) and `main` (c++/bpmf.cpp:180-253) do.
"""
import math
import sys as _sys
import time

import numpy as np

from . import engine as _engine


class HyperParams:
    """c++/bpmf.h:78-104: fixed prior b0=2, df=K, mu0=0, WI=I; sampled mu, LambdaU, LambdaF."""

    def __init__(self, K):
        self.K = K
        self.b0 = 2
        self.df = K
        self.mu0 = np.zeros(K)
        self.WI = np.eye(K)
        self.mu = np.zeros(K)
        self.LambdaF = np.zeros((K, K))
        self.LambdaU = np.zeros((K, K))
        self.LambdaL = np.zeros((K, K))

    def sample(self, N, sum_, cov, counter):
        """std::tie(mu, LambdaU) = CondNormalWishart(N, cov, sum / N, mu0, b0, WI, df) after
        rng_set_pos(counter) (c++/sample.cpp:349-350)."""
        um = None if not np.any(sum_) else np.asarray(sum_) / N
        self.mu, self.LambdaU, self.LambdaF = _engine.hyper_sample(self.K, N, cov, counter, um)
        self.LambdaL = self.LambdaU.T


class Sys:
    """One factor ("movs" or "users").  M is CSC with one column per item of this
    side; rows index the other side.  `dom` = (col_from, col_to) is the column
    range this rank samples (Sys::from()/to(), c++/bpmf.h:170-172) and M / T are
    the slices of exactly those columns; `num` is the global number of columns."""

    # static members of the reference (c++/sample.cpp:26-34)
    alpha = 2.0
    burnin = 5
    nsims = 20
    procid = 0
    nprocs = 1

    def __init__(self, name, engine, M, num, nrows, T=None, dom=None, mean_rating=None, comm=None):
        self.name = name
        self.engine = engine
        self.K = engine.K
        self.iter = -1                                    # c++/sample.cpp:113
        self._num = int(num)
        self.nrows = int(nrows)
        self.dom = (0, self._num) if dom is None else (int(dom[0]), int(dom[1]))
        self.comm = comm
        colptr, rowidx, vals = M
        self.local_nnz = int(colptr[-1])
        on_device = hasattr(rowidx, "data_ptr")           # torch tensors on the GPU (a matrix generated there): adopted, not copied
        if mean_rating is None:                           # Sys::init, c++/sample.cpp:183 (single rank)
            mean_rating = float(vals.sum()) / max(self.local_nnz, 1)
        self.mean_rating = float(mean_rating)
        if on_device:
            self.side = engine.side_create_dev(self._num, self.nrows, colptr, rowidx.data_ptr(), vals.data_ptr(), self.mean_rating,
                                               self.dom[0], self.dom[1], keep=(rowidx, vals))
        else:
            self.side = engine.side_create(self._num, self.nrows, colptr, rowidx, vals, self.mean_rating,
                                           self.dom[0], self.dom[1])
        self.test = None
        self.T_nnz = 0
        if T is not None:
            self.test = engine.test_create(self.side, *T)
            self.T_nnz = int(T[0][-1])
        self.hp = HyperParams(self.K)
        self.sum = np.zeros(self.K)                       # never updated: SURVEY Q1 (c++/sample.cpp:187,379)
        self.cov = np.zeros((self.K, self.K))
        self.norm = 0.0
        self.rmse = float("nan")
        self.rmse_avg = float("nan")
        self.num_predict = 0
        self.sample_ms = 0.0

    def refresh(self):
        """Pulls norm / cov / hp back from the library after sys_sample calls."""
        if getattr(self, "_stale", False):
            it, self.norm, self.cov, self.hp.mu, self.hp.LambdaF, self.hp.LambdaU = self.engine.sys_state(self.side)
            assert it == self.iter
            self._stale = False

    # -- accessors with the reference's names ---------------------------------
    def num(self):
        return self._num

    def from_(self):
        return self.dom[0]

    def to(self):
        return self.dom[1]

    def items(self):
        """K x num() factor matrix as an [num, K] array (row = one column of the reference's items())."""
        return self.engine.get_items(self.side)

    # -- Sys::sample(Sys&), c++/sample.cpp:341-385 ------------------------------
    def sample(self, other):
        if (self.comm is None or getattr(self.comm, "native", False)) and hasattr(self.engine, "sys_sample"):
            # NO_COMM: the whole of Sys::sample(Sys&) (iter++, hyper draw, column loop, cov) runs
            # behind one C-ABI call, which also overlaps the host draws with the kernels
            self.engine.sys_sample(self.side, other.side, Sys.alpha)
            self.iter += 1
            self._stale = True
            return
        self.iter += 1
        self.hp.sample(self.num(), self.sum, self.cov, self.iter)          # :349-350
        t0 = time.perf_counter()
        s, prod, norm = self.engine.sample_side(self.side, other.side, self.iter, Sys.alpha, self.hp.mu, self.hp.LambdaF)
        if self.comm is not None:
            # exchange the fresh columns (send_item / bcast of the MPI back-ends) and
            # all-reduce sum | prod | norm, then form cov once from the global sums
            self.comm.exchange_items(self)
            red = self.comm.allreduce(np.concatenate([np.asarray(prod, order="F").ravel(order="F"), s, [norm]]))
            K = self.K
            prod = red[:K * K].reshape((K, K), order="F"); s = red[K * K:K * K + K]; norm = float(red[-1])
        self.sample_ms = (time.perf_counter() - t0) * 1e3
        self.norm = norm                                                     # :381
        N = self.num()
        self.cov = _engine.cov_from_sums(self.K, N, s, prod)                 # :383-384
        self.last_sum = s

    # -- Sys::predict, c++/sample.cpp:48-96 -------------------------------------
    def predict(self, other, all=False):
        n = 0 if self.iter < Sys.burnin else self.iter - Sys.burnin          # :50
        if self.test is None:
            return
        if getattr(self, "_twin_of", None) is not None:                    # evaluated with its owner: only the sums are collected here
            self.predict_finish()
            return
        se, se_avg, nump = self.engine.predict(self.test, self.side, other.side, n)
        if self.comm is not None and not getattr(self.comm, "native", False) and all:
            red = self.comm.allreduce(np.array([se, se_avg, float(nump)]))
            se, se_avg, nump = float(red[0]), float(red[1]), int(round(red[2]))
        self.num_predict = nump
        self.rmse = math.sqrt(se / nump) if nump else float("nan")
        self.rmse_avg = math.sqrt(se_avg / nump) if nump else float("nan")

    def set_twin(self, other):
        """`other.predict(self)` of the reference's loop (c++/bpmf.cpp:190) rides with every self.predict(other): `other`
        must hold the transposed test entries (its own T).  Its sums: other.predict_finish() after self's."""
        self.engine.test_set_twin(self.test, other.test)
        self._twin = other
        other._twin_of = self

    def predict_launch(self, other):
        """First half of predict(): enqueue the evaluation behind the samplers.  The caller may start
        the next half-iteration before predict_finish() (software pipelining of main()'s loop)."""
        n = 0 if self.iter < Sys.burnin else self.iter - Sys.burnin
        if self.test is not None:
            self.engine.predict_launch(self.test, self.side, other.side, n)

    def predict_finish(self):
        if self.test is None:
            return
        se, se_avg, nump = self.engine.predict_finish(self.test)
        self.num_predict = nump
        self.rmse = math.sqrt(se / nump) if nump else float("nan")
        self.rmse_avg = math.sqrt(se_avg / nump) if nump else float("nan")

    # -- Sys::print, c++/sample.cpp:101-107 --------------------------------------
    def format_line(self, items_per_sec, ratings_per_sec, norm_u, norm_m):
        phase = "Burnin" if self.iter < Sys.burnin else "Sampling"
        return "%d: %s iteration %d:\t RMSE: %3.4f\tavg RMSE: %3.4f\tFU(%6.2f)\tFM(%6.2f)\titems/sec: %6.2f\tratings/sec: %6.2fM\n" % (
            Sys.procid, phase, self.iter, self.rmse, self.rmse_avg, norm_u, norm_m, items_per_sec, ratings_per_sec / 1e6)


def gibbs(engine, M, Mt, T, nusers, nmovies, nsims=20, burnin=5, alpha=2.0, out=None, keep_samples=False, Tt=None, pipelined=False):
    """The loop of main() (c++/bpmf.cpp:131-253) in NO_COMM mode.  M / T: CSC
    with one column per movie (rows = users); Mt its transpose.  Returns a dict
    with the per-iteration trace; `out` (a file object) receives the reference's
    stdout lines.

    pipelined=True: the same iterations the way the `bpmf` executable (bpmf_main.cpp) and bench.py's timed
    region run them -- the line of iteration i - 1 (RMSE sums, norms) is collected after iteration i has been
    enqueued, the evaluation of i - 1 runs beside the samplers of i (which write the other copy of the factors).
    Same chain, same numbers; `secs` is then the time between two collected lines."""
    Sys.nsims, Sys.burnin, Sys.alpha = nsims, burnin, alpha
    movies = Sys("movs", engine, M, nmovies, nusers, T=T)
    users = Sys("users", engine, Mt, nusers, nmovies, T=Tt)
    if Tt is not None:
        movies.set_twin(users)                       # users.predict(movies) rides with movies.predict(users)
    res = dict(rmse=[], rmse_avg=[], norm_u=[], norm_m=[], secs=[], samples=[])
    nnz = movies.local_nnz

    def line(it, secs, norm_u, norm_m):
        ips = (users.num() + movies.num()) / secs
        if out is not None:
            saved, movies.iter = movies.iter, it
            out.write(movies.format_line(ips, nnz / secs, math.sqrt(norm_u), math.sqrt(norm_m)))
            movies.iter = saved
        res["rmse"].append(movies.rmse); res["rmse_avg"].append(movies.rmse_avg)
        res["norm_u"].append(math.sqrt(norm_u)); res["norm_m"].append(math.sqrt(norm_m))
        res["secs"].append(secs)

    if pipelined and not keep_samples and hasattr(engine, "sys_norm"):
        mark = time.perf_counter()
        for i in range(nsims):
            movies.sample(users)
            users.sample(movies)
            if i > 0:
                norm_m = engine.sys_norm(movies.side, i - 1)
                norm_u = engine.sys_norm(users.side, i - 1)
                movies.predict_finish()
                if Tt is not None:
                    users.predict_finish()
                now = time.perf_counter()
                line(i - 1, now - mark, norm_u, norm_m)
                mark = now
            movies.predict_launch(users)
        if nsims > 0:
            movies.predict_finish()
            if Tt is not None:
                users.predict_finish()
            movies.refresh(); users.refresh()
            line(nsims - 1, time.perf_counter() - mark, users.norm, movies.norm)
    else:
        for i in range(nsims):
            start = time.perf_counter()
            movies.sample(users)
            users.sample(movies)
            movies.predict(users)
            if Tt is not None:
                users.predict(movies)                # c++/bpmf.cpp:190 (nothing reads its results; Tt = None leaves it out)
            stop = time.perf_counter()
            movies.refresh(); users.refresh()
            line(i, stop - start, users.norm, movies.norm)
            if keep_samples:
                res["samples"].append((users.items(), movies.items()))
    movies.predict(users, True)                      # c++/bpmf.cpp:242 (the extra call of Q6)
    if Tt is not None:
        users.predict(movies)                        # (the twin was evaluated with it: collect its sums)
    res["final_rmse_avg"] = movies.rmse_avg
    res["num_predict"] = movies.num_predict
    res["U"] = users.items(); res["V"] = movies.items()
    res["movies"], res["users"] = movies, users
    if out is not None:
        out.write("Final Avg RMSE: %g\n" % movies.rmse_avg)
    return res


if __name__ == "__main__":
    _sys.exit("use the `bpmf` executable (bpmf_amd/csrc) or bench.py")
