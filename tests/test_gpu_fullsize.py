"""Oracle parity at BASELINE.json's FULL single-GPU sizes: every column of both sides, alpha = 2,
random non-zero factors and a random (positive definite) hyper-parameter draw, so that the data
term -- the Gram of every column, the rank-n updates of the K = 64 low-rank forms, the K = 128 fp32
Gram -- is exercised on real numbers at the size the bench is quoted on
(Sys::sample(long, Sys&), /root/reference c++/sample.cpp:263-336).

  * configs[1]  exactly bench.py's `synth.ml1m_shaped(seed=42)` (6040 x 3706, 1 000 209 ratings), K = 32 fp64
  * configs[2]  ChEMBL-shaped 483 500 x 5 775 x 1 023 952 real-valued activities, K = 64 fp64, default
                schedule (k_sample_pf<64,2|6|12> for the light compounds, k_sample1<64> for the rest)
  * configs[4]  ML-1M shape, K = 128, fp32 (k_sample_wg2<128,2,float>), tolerance 2e-3 of max|U|
  * the same shape at K = 128 and K = 100 in the reference's fp64 (k_sample_wg2<128,4,double>; what `bpmf-128` / `bpmf-100` of
                ci/multilatent.sh:5 compute), fp64 tolerance
  * configs[3]  one rank's share of 10M x 1M x 200 per user, K = 32: test_gpu_shard.py

The oracle runs its OpenMP column loop on the box's host cores (an ML-1M iteration is ~0.1 s on one
thread, ChEMBL K = 64 a few seconds on 32): "too slow" does not hold, so these are not property checks.
fp64 tolerance: 1e-9 of max|U| per half-iteration (as tests/test_gpu_parity.py), sums 1e-8.
"""
import os

import numpy as np
import pytest

from tests import util
from tests.test_gpu_parity import rel_err

pytestmark = pytest.mark.gpu

NT = max(1, min(os.cpu_count() or 1, 32))


def _hyper(oracle, K, N, seed, it):
    """A non-trivial Normal-Wishart draw: cov of random factors, counter = iteration."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((K, 3 * K))
    return oracle.hyper_sample(K, N, A @ A.T / (3 * K) * 0.5, it)


def _both_sides(oracle, eng, K, data, tol, sum_tol, seed, f32=False):
    M, Mt, T, Tt, nu, nm = data
    rng = np.random.default_rng(seed)
    U = 0.3 * rng.standard_normal((nu, K)); V = 0.3 * rng.standard_normal((nm, K))
    if f32:                                          # the oracle sees the factors the device sees
        U = U.astype(np.float32).astype(np.float64); V = V.astype(np.float32).astype(np.float64)
    mean = util.mean_rating(M)
    worst = {}
    for name, mat, ncols, nrows, other, it in (("movies", M, nm, nu, U, 3), ("users", Mt, nu, nm, V, 4)):
        mu, LU, LF = _hyper(oracle, K, ncols, seed + it, it)
        ref = np.zeros((ncols, K))
        s_ref, p_ref, n_ref = oracle.sample_side(K, mat, mean, 2.0, other, ref, it, mu, LF, nthreads=NT)
        me = eng.side_create(ncols, nrows, *mat, mean)
        ot = eng.side_create(nrows, ncols, np.zeros(nrows + 1, np.int64), np.zeros(0, np.int32), np.zeros(0), 0.0)
        eng.set_items(ot, other)
        s, p, n = eng.sample_side(me, ot, it, 2.0, mu, LF)
        X = eng.get_items(me)
        eng.side_destroy(me); eng.side_destroy(ot)
        assert np.all(np.isfinite(X))
        scale = np.abs(ref).max()
        err_col = np.abs(X - ref).max(axis=1) / scale           # EVERY column
        bad = int(np.argmax(err_col))
        assert err_col[bad] < tol, "%s: column %d (%d ratings) off by %.3e" % (name, bad, mat[0][bad + 1] - mat[0][bad], err_col[bad])
        assert rel_err(s, s_ref) < sum_tol and rel_err(p, p_ref) < sum_tol and abs(n - n_ref) <= sum_tol * abs(n_ref)
        worst[name] = float(err_col[bad])
    return worst


def test_ml1m_k32_every_column_matches_the_oracle(oracle, hip_engine_factory):
    from bpmf_amd import synth
    data = synth.ml1m_shaped(seed=42)                 # the matrix bench.py times
    assert data[4:] == (6040, 3706) and int(data[0][0][-1]) + int(data[2][0][-1]) == 1_000_209
    w = _both_sides(oracle, hip_engine_factory(32), 32, data, 1e-9, 1e-8, seed=11)
    print("ML-1M K=32 worst column error / max|U|:", w)


def test_ml1m_k64_every_column_matches_the_oracle(oracle, hip_engine_factory):
    from bpmf_amd import synth
    w = _both_sides(oracle, hip_engine_factory(64), 64, synth.ml1m_shaped(seed=42), 1e-9, 1e-8, seed=12)
    print("ML-1M K=64 worst column error / max|U|:", w)


def test_chembl_k64_every_column_matches_the_oracle(oracle, hip_engine_factory):
    """Default schedule: the compounds side splits into k_sample_pf<64,NB> (<= 16 activities) and the
    regular form; the targets side is all regular (heavy columns chunked)."""
    from bpmf_amd import synth
    data = synth.ratings(483500, 5775, 1_023_952, seed=42, real_valued=True)
    nnzc = np.diff(data[1][0])
    assert (nnzc <= 16).mean() > 0.5 and (nnzc == 0).any() and (nnzc > 16).any()
    w = _both_sides(oracle, hip_engine_factory(64), 64, data, 1e-9, 1e-8, seed=13)
    print("ChEMBL-shaped K=64 worst column error / max|U|:", w)


def test_ml1m_k128_f32_every_column_within_2e3_of_the_fp64_oracle(oracle, hip_engine_factory):
    from bpmf_amd import synth
    w = _both_sides(oracle, hip_engine_factory(128, "f32"), 128, synth.ml1m_shaped(seed=42), 2e-3, 1e-3, seed=14, f32=True)
    print("ML-1M K=128 fp32 worst column error / max|U|:", w)


@pytest.mark.parametrize("K", [128, 100])
def test_ml1m_k128_fp64_every_column_matches_the_oracle(oracle, hip_engine_factory, K):
    """num_latent 128 (and 100: the K = 128 kernels with 28 padded dimensions, RNG streams of K = 100) in fp64, the
    arithmetic the reference's bpmf-128 / bpmf-100 use (c++/bpmf.h:55-58): every column of both sides at the ML-1M size."""
    from bpmf_amd import synth
    w = _both_sides(oracle, hip_engine_factory(K), K, synth.ml1m_shaped(seed=42), 1e-9, 1e-8, seed=15)
    print("ML-1M K=%d fp64 worst column error / max|U|:" % K, w)
