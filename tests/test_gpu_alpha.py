"""The reference's `-a F` flag (c++/bpmf.cpp:91 -> Sys::alpha; used at c++/sample.cpp:256 as the weight (v - mean) * alpha of a
rating and at :298 as Lambda* = LambdaF + alpha * G) is a run-time input of the hot path.  Every other parity test runs the
reference's default alpha = 2, a power of two, for which scaling by alpha is exact -- so a kernel that reorders the arithmetic
around alpha (pre-scaled priors, sqrt(alpha) in the product form, alpha folded into the weights) would pass them all and still
differ from the reference for `-a 1.5`.  Here every kernel family runs alpha in {0.5, 1.5, 3, 10} against the oracle, at the
same tolerances as everywhere else (1e-9 of max|U| per half-iteration in fp64, 2e-3 for the fp32 opt-in, 1e-6 on RMSE traces):

  family                                                   test
  k_sample1<8|16|32>, k_sample4<8|16|32> (forced)          test_alpha_small_latent
  k_sample1s<64> / k_sample_slab<64> (+ chunked column)    test_alpha_k64_slab
  k_pf_prepare<64> + k_sample_pf<64, 3|6|16>               test_alpha_k64_product_form   (sqrt(alpha) enters p = R0^-T sqrt(alpha) u)
  k_sample_wg2<128, 4, double>: whole + chunked, K = 100   test_alpha_k128_fp64          (the Lambda_F / alpha pre-fill is gated on
                                                                                          alpha being a power of two: kernels_wg2.h)
  k_sample_wg2<128, 2, float>: whole + chunked             test_alpha_k128_fp32
  BPMF_NO_COVARIANCE, propagated posterior                 test_alpha_variants
  BPMF_REDUCE formulation                                  test_alpha_reduce_formulation
  the whole chain through the timed pipeline, `bpmf -a`    test_alpha_chain_pipelined, test_alpha_cli_end_to_end
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from tests import util
from tests.conftest import ROOT
from tests.test_gpu_parity import RTOL, check_half_iteration, rel_err, sampler_mode, _one_form_only  # noqa: F401 (fixture)

pytestmark = pytest.mark.gpu

ALPHAS = [0.5, 1.5, 3.0, 10.0]
NT = max(1, min(os.cpu_count() or 1, 16))


def _cov(K, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((K, 3 * K))
    return A @ A.T / (3 * K)


def _side_pair(eng, M, nrows):
    ncols = len(M[0]) - 1
    me = eng.side_create(ncols, nrows, *M, util.mean_rating(M))
    ot = eng.side_create(nrows, ncols, np.zeros(nrows + 1, np.int64), np.zeros(0, np.int32), np.zeros(0), 0.0)
    return me, ot


def _sweep(oracle, eng, K, M, nrows, other, cov_seed, tol=RTOL, stat_tol=1e-8, expect_kernel=None, **okw):
    """One side, one set of factors; a half-iteration per alpha (fresh hyper-parameters each) against the oracle."""
    ncols = len(M[0]) - 1
    mean = util.mean_rating(M)
    me, ot = _side_pair(eng, M, nrows)
    if expect_kernel is not None:
        assert re.search(expect_kernel, eng.kernel_name(me)), eng.kernel_name(me)
    eng.set_items(ot, other)
    worst = 0.0
    for it, alpha in enumerate(ALPHAS, start=2):
        mu, LU, LF = oracle.hyper_sample(K, ncols, _cov(K, cov_seed + it), it)
        ref = np.zeros((ncols, K))
        s_ref, p_ref, n_ref = oracle.sample_side(K, M, mean, alpha, other, ref, it, mu, LF, nthreads=NT, **okw)
        s, p, n = eng.sample_side(me, ot, it, alpha, mu, LF)
        items = eng.get_items(me)
        assert np.all(np.isfinite(items)), alpha
        err = rel_err(items, ref)
        assert err < tol, (alpha, err)
        assert rel_err(s, s_ref) < stat_tol and rel_err(p, p_ref) < stat_tol and abs(n - n_ref) <= stat_tol * abs(n_ref), alpha
        worst = max(worst, err)
    info = eng.schedule_info(me)
    eng.side_destroy(me); eng.side_destroy(ot)
    return worst, info


def _heavy():
    M, Mt, T, Tt, nu, nm = util.synthetic(6000, 300, 60000, seed=3, heavy=(7, 5000))
    assert np.diff(M[0]).max() >= 5000
    return M, nu, nm


@pytest.mark.parametrize("K", [8, 16, 32])
def test_alpha_small_latent(oracle, hip_engine_factory, K, sampler_mode):
    """k_sample1<K> (per-item) and k_sample4<K> (four columns per wave; forced, the matrices are small): MovieLens-100K's movie
    side (32 empty columns, median 21 ratings) and, at K = 32, a side with a 5 000-rating column cut into chunks."""
    M, Mt, T, Tt, nu, nm = util.ml100k()
    eng = hip_engine_factory(K)
    rng = np.random.default_rng(900 + K)
    name = {None: r"k_sample", 1: r"k_sample1", 3: r"k_sample4"}[sampler_mode]
    _sweep(oracle, eng, K, M, nu, 0.3 * rng.standard_normal((nu, K)), 11, expect_kernel=name)
    if K == 32:
        H, hu, hm = _heavy()
        worst, info = _sweep(oracle, eng, K, H, hu, 0.25 * rng.standard_normal((hu, K)), 13)
        assert info["chunked_columns"] >= 1


def test_alpha_k64_slab(oracle, hip_engine_factory):
    """k_sample1s<64> / k_sample_slab<64>: MovieLens-100K users side (20 .. 737 ratings) + the chunked heavy column."""
    K = 64
    M, Mt, T, Tt, nu, nm = util.ml100k()
    eng = hip_engine_factory(K)
    rng = np.random.default_rng(964)
    _sweep(oracle, eng, K, Mt, nm, 0.3 * rng.standard_normal((nm, K)), 21, expect_kernel=r"k_sample1s|k_sample_slab")
    H, hu, hm = _heavy()
    worst, info = _sweep(oracle, eng, K, H, hu, 0.25 * rng.standard_normal((hu, K)), 23)
    assert info["chunked_columns"] >= 1


def test_alpha_k64_product_form(oracle, hip_engine_factory):
    """The three product-form classes (<= 3, <= 6, <= 16 ratings per column; ragged last passes) + the slab launch of the
    heavier columns of a ChEMBL-shaped side.  Here alpha enters as x_m = sqrt(alpha) u_m (kernels_lr.h): the SQUARE of the
    rounded root stands in for alpha in Lambda* = R0^T (I + sum p p^T) R0, while the right-hand side takes alpha itself
    (c++/sample.cpp:256) -- for no alpha is that bit-exact, for every alpha it must stay within 1e-9."""
    K = 64
    rng = np.random.default_rng(641)
    nrows = 400
    counts = np.concatenate([np.full(301, 0), np.full(203, 1), np.full(97, 2), np.full(250, 3),
                             np.full(333, 4), np.full(334, 5), np.full(335, 6)] +
                            [np.full(33, n) for n in range(7, 17)] + [np.full(9, 30), np.full(3, 300)])
    rng.shuffle(counts)
    colptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    rowidx = np.concatenate([np.sort(rng.choice(nrows, size=c, replace=False)) for c in counts]).astype(np.int32)
    vals = rng.normal(5.0, 1.1, size=len(rowidx))
    eng = hip_engine_factory(K)
    worst, info = _sweep(oracle, eng, K, (colptr, rowidx, vals), nrows, 0.35 * rng.standard_normal((nrows, K)), 31,
                         expect_kernel=r"k_sample_pf")
    assert info["pf_le3"] > 0 and info["pf_4to6"] > 0 and info["pf_7to16"] > 0, info


@pytest.mark.parametrize("K", [128, 100])
def test_alpha_k128_fp64(oracle, hip_engine_factory, K):
    """k_sample_wg2<128, 4, double>: whole columns (MovieLens-100K users side) and a chunked 5 000-rating column; K = 100 runs
    padded on the same kernel.  alpha = 1.5 / 3 / 10 take the prior after the Gram (fma(alpha, G, LambdaF)); 0.5 is a power of
    two and takes the pre-filled accumulators, like the default 2 -- both must be the reference's sum to 1e-9."""
    M, Mt, T, Tt, nu, nm = util.ml100k()
    eng = hip_engine_factory(K)
    rng = np.random.default_rng(9000 + K)
    _sweep(oracle, eng, K, Mt, nm, 0.3 * rng.standard_normal((nm, K)), 41, expect_kernel=r"k_sample_wg2")
    H, hu, hm = _heavy()
    worst, info = _sweep(oracle, eng, K, H, hu, 0.25 * rng.standard_normal((hu, K)), 43)
    assert info["chunked_columns"] >= 1


def test_alpha_k128_whole_and_chunked_columns_agree(oracle, hip_engine_factory, monkeypatch):
    """The same heavy column as ONE work item and cut into chunks (BPMF_HIP_CHUNK moves the threshold): for an alpha that is not
    a power of two both take fma(alpha, G, LambdaF), so they differ by the Gram's summation order only (<= 1e-12 of max|U|) --
    round 5's pre-fill would have made the rounding of a column depend on whether the schedule cut it."""
    K = 128
    M, Mt, T, Tt, nu, nm = util.synthetic(3000, 40, 12000, seed=5, heavy=(3, 2000))
    eng = hip_engine_factory(K)
    rng = np.random.default_rng(77)
    U = 0.2 * rng.standard_normal((nu, K))
    mean = util.mean_rating(M)
    mu, LU, LF = oracle.hyper_sample(K, nm, _cov(K, 5), 3)
    out = {}
    for tag, chunk in (("chunked", "256"), ("whole", "1000000")):
        monkeypatch.setenv("BPMF_HIP_CHUNK", chunk)
        me, ot = _side_pair(eng, M, nu)
        info = eng.schedule_info(me)
        assert (info["chunked_columns"] >= 1) == (tag == "chunked"), info
        eng.set_items(ot, U)
        eng.sample_side(me, ot, 3, 1.5, mu, LF)
        out[tag] = eng.get_items(me).copy()
        eng.side_destroy(me); eng.side_destroy(ot)
    monkeypatch.delenv("BPMF_HIP_CHUNK")
    ref = np.zeros((nm, K))
    oracle.sample_side(K, M, mean, 1.5, U, ref, 3, mu, LF, nthreads=NT)
    assert rel_err(out["whole"], ref) < RTOL and rel_err(out["chunked"], ref) < RTOL
    assert rel_err(out["whole"], out["chunked"]) < 1e-11


def test_alpha_k128_fp32(oracle, hip_engine_factory):
    """k_sample_wg2<128, 2, float> (the fp32 opt-in): 2e-3 of max|U| against the fp64 oracle fed the fp32-rounded factors."""
    K = 128
    M, Mt, T, Tt, nu, nm = util.ml100k()
    eng = hip_engine_factory(K, "f32")
    rng = np.random.default_rng(9128)
    V = (0.3 * rng.standard_normal((nm, K))).astype(np.float32).astype(np.float64)
    _sweep(oracle, eng, K, Mt, nm, V, 51, tol=2e-3, stat_tol=1e-3, expect_kernel=r"k_sample_wg2")
    H, hu, hm = _heavy()
    U = (0.25 * rng.standard_normal((hu, K))).astype(np.float32).astype(np.float64)
    worst, info = _sweep(oracle, eng, K, H, hu, U, 53, tol=2e-3, stat_tol=1e-3)
    assert info["chunked_columns"] >= 1


@pytest.mark.parametrize("K", [32, 64, 128])
def test_alpha_variants(oracle, hip_engine_factory, K):
    """BPMF_NO_COVARIANCE (c++/sample.cpp:300-304: the off-diagonal of LambdaF + alpha G dropped) and per-column propagated
    priors (-m / -l, :272-277) at alpha = 1.5 and 3."""
    M, Mt, T, Tt, nu, nm = util.ml100k()
    eng = hip_engine_factory(K)
    rng = np.random.default_rng(70 + K)
    U = 0.3 * rng.standard_normal((nu, K))
    mean = util.mean_rating(M)
    mu, LU, LF = oracle.hyper_sample(K, nm, _cov(K, 9), 3)
    me, ot = _side_pair(eng, M, nu)
    eng.set_items(ot, U)
    B = rng.standard_normal((nm, K, K)) * 0.2
    lam = (np.einsum("nij,nkj->nik", B, B) + np.eye(K)[None] * 1.5).reshape(nm, K * K)
    eng.set_prop_posterior(me, lam)
    for alpha in (1.5, 3.0):
        ref = np.zeros((nm, K))
        sr = oracle.sample_side(K, M, mean, alpha, U, ref, 3, mu, LF, prop_lambda=lam, nthreads=NT)
        s = eng.sample_side(me, ot, 3, alpha, mu, LF)
        check_half_iteration((eng.get_items(me),) + tuple(s), (ref,) + tuple(sr))
    eng.set_prop_posterior(me, None)
    eng.set_no_covariance(True)
    try:
        for alpha in (1.5, 3.0):
            ref = np.zeros((nm, K))
            sr = oracle.sample_side(K, M, mean, alpha, U, ref, 4, mu, LF, no_covariance=True, nthreads=NT)
            s = eng.sample_side(me, ot, 4, alpha, mu, LF)
            check_half_iteration((eng.get_items(me),) + tuple(s), (ref,) + tuple(sr))
    finally:
        eng.set_no_covariance(False)
    eng.side_destroy(me); eng.side_destroy(ot)


def test_alpha_reduce_formulation():
    """BPMF_REDUCE (c++/sample.cpp:234-246: alpha enters in preComputeMuLambda on the producer side) at alpha = 1.5, K = 32."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_reduce_worker.py"), "32", "1.5"], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and "REDUCE-OK K=32 alpha=1.5" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("K,alpha,dtype", [(32, 1.5, "f64"), (64, 3.0, "f64"), (128, 10.0, "f64"), (128, 1.5, "f32")])
def test_alpha_chain_pipelined(oracle, hip_engine_factory, K, alpha, dtype):
    """`-i 6 -b 2` on MovieLens-100K through the pipeline bench.py times (fused launches, riders, twin evaluation), alpha != 2."""
    import bpmf_amd
    M, Mt, T, Tt, nu, nm = util.ml100k()
    eng = hip_engine_factory(K, dtype)
    res = bpmf_amd.gibbs(eng, M, Mt, T, nu, nm, nsims=6, burnin=2, alpha=alpha, pipelined=True)
    ref = oracle.gibbs(K, M, Mt, T, Tt, alpha=alpha, nsims=6, burnin=2, nthreads=NT)
    tol_r, tol_f = (1e-6, 1e-6) if dtype == "f64" else (1e-3, 2e-3)
    assert np.abs(np.asarray(res["rmse"]) - np.asarray(ref["rmse"])).max() < tol_r
    assert abs(res["final_rmse_avg"] - ref["final_rmse_avg"]) < tol_r
    scale = max(np.abs(ref["U"]).max(), np.abs(ref["V"]).max())
    assert np.abs(res["U"] - ref["U"]).max() < tol_f * scale and np.abs(res["V"] - ref["V"]).max() < tol_f * scale


def test_alpha_cli_end_to_end(oracle, tmp_path):
    """`bpmf -a 1.5 -i 6 -b 2 -d 32` on MovieLens-100K: the stdout contract carries `alpha: 1.5` (c++/bpmf.cpp:172) and the RMSE
    of every iteration + Final Avg RMSE are the oracle's chain at alpha = 1.5 (and NOT the default chain's)."""
    G = util.GOLDEN
    exe = os.path.join(ROOT, "bpmf_amd", "bpmf")
    r = subprocess.run([exe, "-a", "1.5", "-i", "6", "-b", "2", "-d", "32", "-n", os.path.join(G, "ml100k-train.mtx.gz"),
                        "-p", os.path.join(G, "ml100k-test.mtx.gz")], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert re.search(r"^alpha: 1\.5$", r.stdout, re.M), r.stdout
    M, Mt, T, Tt, nu, nm = util.ml100k()
    ref = oracle.gibbs(32, M, Mt, T, Tt, alpha=1.5, nsims=6, burnin=2, nthreads=NT)
    ref2 = oracle.gibbs(32, M, Mt, T, Tt, alpha=2.0, nsims=6, burnin=2, nthreads=NT)
    final = float(re.search(r"Final Avg RMSE: (\S+)", r.stdout).group(1))
    assert abs(final - ref["final_rmse_avg"]) < 2e-6 * max(1.0, final)          # (printed with 6 significant digits)
    assert abs(ref["final_rmse_avg"] - ref2["final_rmse_avg"]) > 1e-4           # the flag does change the chain
