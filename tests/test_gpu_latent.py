"""Every num_latent the reference ships a binary for, in the reference's fp64.

`/root/reference/ci/multilatent.sh:5` builds `bpmf-K` for K in 8 16 32 64 128 10 20 30 ... 100 (BPMF_NUMLATENT,
c++/bpmf.h:22-24,53; all arithmetic double, :55-58).  The kernels here are instantiated for 8, 16, 32, 64, 128; any
other K runs on the next of those with zero factor rows / an identity block of the prior precision in the extra
dimensions.  What must NOT follow the kernel size: the per-column stream id (idx + 1) * K * (iter + 1) and the number
of normals a column draws (c++/sample.cpp:266,322), the hyper-parameter draw (host, at the true K), and the size of
everything that crosses the C ABI.  The oracle runs at the TRUE K -- nothing here knows about the padding.

K = 128 (and 65 .. 127 on top of it) in fp64 is k_sample_wg2<128, 4, double> (kernels_wg2.h): same tolerances as every
other fp64 size -- 1e-9 of max|U| per half-iteration, 1e-6 on RMSE traces over 20 iterations."""
import os

import numpy as np
import pytest

from tests import util
from tests.test_gpu_parity import RTOL, check_half_iteration, half_iteration_pair, rel_err

pytestmark = pytest.mark.gpu

NT = max(1, min(os.cpu_count() or 1, 16))


def _cov(K, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((K, 3 * K))
    return A @ A.T / (3 * K)


@pytest.mark.parametrize("K", [1, 3, 10, 20, 30, 50, 70, 100, 128])
def test_tiny_half_iterations_at_any_num_latent(oracle, hip_engine_factory, K):
    M, Mt, T, Tt, nu, nm = util.tiny()
    eng = hip_engine_factory(K)
    rng = np.random.default_rng(K)
    for it, U in ((0, np.zeros((nu, K))), (3, rng.standard_normal((nu, K)))):      # iteration 0: Sys::init's zero factors
        check_half_iteration(*half_iteration_pair(oracle, eng, K, M, nu, U, it))
    check_half_iteration(*half_iteration_pair(oracle, eng, K, Mt, nm, rng.standard_normal((nm, K)), 2, cov=_cov(K, 5)))


@pytest.mark.parametrize("K", [10, 40, 50, 100, 128])
def test_ml100k_half_iterations_at_any_num_latent(oracle, hip_engine_factory, K):
    """Both sides of MovieLens-100K (32 empty movie columns, median 21 ratings, users up to 737) from random factors and a
    random positive definite cov; K = 40 / 50 run on the K = 64 forms (slab), 100 / 128 on k_sample_wg2<128, 4, double>."""
    M, Mt, T, Tt, nu, nm = util.ml100k()
    eng = hip_engine_factory(K)
    rng = np.random.default_rng(100 + K)
    U = 0.3 * rng.standard_normal((nu, K)); V = 0.3 * rng.standard_normal((nm, K))
    check_half_iteration(*half_iteration_pair(oracle, eng, K, M, nu, U, 5, cov=_cov(K, K)))
    check_half_iteration(*half_iteration_pair(oracle, eng, K, Mt, nm, V, 7, cov=_cov(K, K + 1)))


@pytest.mark.parametrize("K", [50, 100, 128])
def test_light_and_heavy_columns_at_padded_sizes(oracle, hip_engine_factory, K):
    """A ChEMBL-like side (thousands of columns with 0 .. 12 ratings: at K = 50 the product / low-rank forms of the K = 64
    kernels fire, whose shared factor of LambdaF is the padded one) and a side with a 5 000-rating column (cut into chunks;
    the last workgroup to arrive adds the partial tiles)."""
    rng = np.random.default_rng(K)
    ncols, nrows = 3000, 150
    counts = rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 40], size=ncols,
                        p=[0.06, 0.2, 0.2, 0.1, 0.08, 0.05, 0.05, 0.04, 0.04, 0.03, 0.03, 0.03, 0.03, 0.03, 0.03])
    colptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    rowidx = np.concatenate([np.sort(rng.choice(nrows, size=c, replace=False)) for c in counts]).astype(np.int32)
    vals = rng.normal(6.0, 1.3, size=len(rowidx))
    eng = hip_engine_factory(K)
    check_half_iteration(*half_iteration_pair(oracle, eng, K, (colptr, rowidx, vals), nrows, 0.4 * rng.standard_normal((nrows, K)), 4, cov=_cov(K, 3)))
    M, Mt, T, Tt, nu, nm = util.synthetic(6000, 300, 60000, seed=3, heavy=(7, 5000))
    assert np.diff(M[0]).max() >= 5000
    me = eng.side_create(nm, nu, *M, util.mean_rating(M))
    assert eng.schedule_info(me)["chunked_columns"] >= 1
    eng.side_destroy(me)
    check_half_iteration(*half_iteration_pair(oracle, eng, K, M, nu, 0.25 * rng.standard_normal((nu, K)), 2, cov=_cov(K, 4)))


@pytest.mark.parametrize("K", [10, 100, 128])
def test_variants_at_padded_sizes(oracle, hip_engine_factory, K):
    """BPMF_NO_COVARIANCE (c++/sample.cpp:300-304) and propagated-posterior priors (-m / -l, :272-277, Q2): the per-column
    prior precisions are K x K of the TRUE size on the interface, identity-padded behind it."""
    M, Mt, T, Tt, nu, nm = util.ml100k()
    eng = hip_engine_factory(K)
    rng = np.random.default_rng(7 * K)
    U = 0.3 * rng.standard_normal((nu, K))
    mean = util.mean_rating(M)
    mu, LU, LF = oracle.hyper_sample(K, nm, _cov(K, 9), 3)
    # per-column priors
    B = rng.standard_normal((nm, K, K)) * 0.2
    lam = np.einsum("nij,nkj->nik", B, B) + np.eye(K)[None] * 1.5              # [n, K, K] symmetric
    ref = np.zeros((nm, K))
    s_ref, p_ref, n_ref = oracle.sample_side(K, M, mean, 2.0, U, ref, 3, mu, LF, prop_lambda=lam.reshape(nm, K * K), nthreads=NT)
    me = eng.side_create(nm, nu, *M, mean)
    ot = eng.side_create(nu, nm, np.zeros(nu + 1, np.int64), np.zeros(0, np.int32), np.zeros(0), 0.0)
    eng.set_items(ot, U)
    eng.set_prop_posterior(me, lam.reshape(nm, K * K))
    s, p, n = eng.sample_side(me, ot, 3, 2.0, mu, LF)
    check_half_iteration((eng.get_items(me), s, p, n), (ref, s_ref, p_ref, n_ref))
    eng.set_prop_posterior(me, None)
    # diagonal precision
    eng.set_no_covariance(True)
    try:
        ref = np.zeros((nm, K))
        s_ref, p_ref, n_ref = oracle.sample_side(K, M, mean, 2.0, U, ref, 4, mu, LF, no_covariance=True, nthreads=NT)
        s, p, n = eng.sample_side(me, ot, 4, 2.0, mu, LF)
        check_half_iteration((eng.get_items(me), s, p, n), (ref, s_ref, p_ref, n_ref))
    finally:
        eng.set_no_covariance(False)
    eng.side_destroy(me); eng.side_destroy(ot)


@pytest.mark.parametrize("K", [10, 50, 100, 128])
def test_full_run_ml100k_matches_the_oracle_at_any_num_latent(oracle, hip_engine_factory, K):
    """-i 20 -b 5 on the shipped MovieLens-100K split through the stateful path (bpmf_hip_sys_sample: the library's own
    hyper-parameter draws at the true K, cov from the padded sums, predict): RMSE traces to 1e-6, factors to 1e-6 of max|U|
    after 20 coupled iterations, like tests/test_gpu_parity.py::test_full_run_ml100k_matches_oracle does for K = 32."""
    import bpmf_amd
    M, Mt, T, Tt, nu, nm = util.ml100k()
    eng = hip_engine_factory(K)
    res = bpmf_amd.gibbs(eng, M, Mt, T, nu, nm, nsims=20, burnin=5)
    ref = oracle.gibbs(K, M, Mt, T, Tt, nsims=20, burnin=5, nthreads=NT)
    assert res["U"].shape == (nu, K) and res["V"].shape == (nm, K)
    assert abs(res["rmse"][0] - 1.153676) < 2e-3
    assert np.allclose(res["rmse"], ref["rmse"], atol=1e-6) and np.allclose(res["rmse_avg"], ref["rmse_avg"], atol=1e-6)
    assert abs(res["final_rmse_avg"] - ref["final_rmse_avg"]) < 1e-6
    assert np.allclose(res["norm_u"], ref["norm_u"], rtol=1e-7) and np.allclose(res["norm_m"], ref["norm_m"], rtol=1e-7)
    assert rel_err(res["U"], ref["U"]) < 1e-6 and rel_err(res["V"], ref["V"]) < 1e-6


def test_state_and_raw_layout_of_a_padded_context(oracle, hip_engine_factory):
    """The Sys state has the caller's sizes; the raw device matrix has the kernel's leading dimension with zero rows behind
    the caller's (bpmf_hip_ctx_ld), which is what a caller binding its own storage must provide."""
    from bpmf_amd.sys import Sys
    K = 20
    M, Mt, T, Tt, nu, nm = util.tiny()
    eng = hip_engine_factory(K)
    lib = eng.lib
    assert lib.bpmf_hip_ctx_num_latent(eng.ctx) == 20 and lib.bpmf_hip_ctx_ld(eng.ctx) == 32 and lib.bpmf_hip_kernel_k(20, 0) == 32
    Sys.nsims, Sys.burnin, Sys.alpha = 3, 0, 2.0
    movies = Sys("movs", eng, M, nm, nu, T=T); users = Sys("users", eng, Mt, nu, nm)
    for _ in range(3):
        movies.sample(users); users.sample(movies); movies.predict(users)
    it, nrm, cov, mu, LF, LU = eng.sys_state(users.side)
    ref = oracle.gibbs(K, M, Mt, T, Tt, nsims=3, burnin=0)
    assert it == 2 and cov.shape == (K, K) and mu.shape == (K,) and abs(np.sqrt(nrm) - ref["norm_u"][-1]) < 1e-8 * max(1.0, ref["norm_u"][-1])
    assert np.allclose(LF, LU.T @ LU, rtol=1e-10, atol=1e-12)
    X = users.items()
    assert X.shape == (nu, K) and rel_err(X, ref["U"]) < 1e-8
    ptr = eng.items_dev_ptr(users.side)
    assert ptr
    raw = np.empty((nu, 32))
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(raw.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), ctypes.c_size_t(raw.nbytes), 2) == 0   # hipMemcpyDeviceToHost
    assert np.array_equal(raw[:, :K], X) and not raw[:, K:].any()
    eng.side_destroy(movies.side); eng.side_destroy(users.side)


def test_bind_items_refuses_storage_with_the_callers_leading_dimension():
    """bpmf_hip_side_bind_items states what was allocated: num_latent 20 runs with 32 rows per column on the device, so a
    [ncols, 20] buffer (what a caller thinking in num_latent would allocate) is refused instead of being written past; with
    [ncols, ld] the samplers stay inside it (guard words behind the storage).  Raw hipMalloc through ctypes: the HIP runtime
    the library itself is linked against (torch brings its own copy, which must be the first one a process loads)."""
    import ctypes as C
    import bpmf_amd
    from bpmf_amd import _lib
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    eng = bpmf_amd.HipEngine(20)
    dev = C.c_void_p()
    try:
        assert eng.ld() == 32
        n, guard = 300, 64
        side = eng.side_create(n, 50, np.zeros(n + 1, np.int64), np.zeros(0, np.int32), np.zeros(0), 0.0)
        other = eng.side_create(50, n, np.zeros(51, np.int64), np.zeros(0, np.int32), np.zeros(0), 0.0)
        host = np.zeros(n * 32 + guard); host[n * 32:] = 7.0
        assert hip.hipMalloc(C.byref(dev), host.nbytes) == 0
        assert hip.hipMemcpy(dev, host.ctypes.data_as(C.c_void_p), host.nbytes, 1) == 0
        with pytest.raises(_lib.BpmfHipError) as e:
            eng.bind_items(side, dev.value)                                              # ld = num_latent = 20
        assert "leading dimension" in str(e.value)
        with pytest.raises(_lib.BpmfHipError):
            eng.bind_items(side, dev.value, ld=32, nbytes=n * 20 * 8)                    # right ld, too few bytes
        eng.bind_items(side, dev.value, ld=32, nbytes=n * 32 * 8)
        eng.sample_side(side, other, 0, 2.0, np.zeros(20), np.eye(20))
        eng.sync()
        assert hip.hipMemcpy(host.ctypes.data_as(C.c_void_p), dev, host.nbytes, 2) == 0
        assert (host[n * 32:] == 7.0).all()
        X = host[:n * 32].reshape(n, 32)
        assert (X[:, 20:] == 0).all() and (X[:, :20] != 0).any()
        assert np.array_equal(eng.get_items(side), X[:, :20])
        eng.side_destroy(side)
    finally:
        eng.close()
        if dev.value:
            hip.hipFree(dev)

