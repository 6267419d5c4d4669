"""Size-independent properties at BASELINE.json sizes (the oracle would take too long
to be the checker for every column): ML-1M-shaped synthetic R, K = 32."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ml1m():
    return util.synthetic(6040, 3706, 1_150_000, seed=42)


def test_ml1m_shape_half_iteration_properties(oracle, hip_engine_factory, ml1m):
    K = 32
    M, Mt, T, Tt, nu, nm = ml1m
    eng = hip_engine_factory(K)
    rng = np.random.default_rng(0)
    U = 0.3 * rng.standard_normal((nu, K))
    mean = util.mean_rating(M)
    mu, LU, LF = oracle.hyper_sample(K, nm, np.eye(K) * 0.1, 2)
    me = eng.side_create(nm, nu, *M, mean)
    ot = eng.side_create(nu, nm, np.zeros(nu + 1, np.int64), np.zeros(0, np.int32), np.zeros(0), 0.0)
    eng.set_items(ot, U)
    s, p, n = eng.sample_side(me, ot, 2, 2.0, mu, LF)
    X = eng.get_items(me)
    # (1) reductions are consistent with the sampled factors (checksum of checksums)
    assert np.allclose(s, X.sum(0), rtol=1e-10, atol=1e-9)
    assert np.allclose(p, X.T @ X, rtol=1e-10, atol=1e-8)
    assert abs(n - (X * X).sum()) < 1e-9 * n
    # (2) bit-reproducible: a second call with the same seeds returns identical bits
    s2, p2, n2 = eng.sample_side(me, ot, 2, 2.0, mu, LF)
    assert np.array_equal(X, eng.get_items(me)) and np.array_equal(p, p2) and np.array_equal(s, s2)
    # (3) spot-check 64 columns (heaviest, lightest, random) against the oracle
    nnzc = np.diff(M[0])
    order = np.argsort(nnzc)
    pick = np.unique(np.concatenate([order[:16], order[-16:], rng.choice(nm, 32, replace=False)]))
    sub_ptr = np.concatenate([[0], np.cumsum(nnzc[pick])]).astype(np.int64)
    sub_idx = np.concatenate([M[1][M[0][c]:M[0][c + 1]] for c in pick]).astype(np.int32)
    sub_val = np.concatenate([M[2][M[0][c]:M[0][c + 1]] for c in pick])
    # the oracle keys the stream on the column id, so sample column c as column c of a wide matrix
    for c, a, b in zip(pick, sub_ptr[:-1], sub_ptr[1:]):
        cp = np.zeros(nm + 1, np.int64); cp[c + 1:] = b - a
        ref = np.zeros((nm, K))
        oracle.sample_side(K, (cp, sub_idx[a:b].copy(), sub_val[a:b].copy()), mean, 2.0, U, ref, 2, mu, LF, from_=int(c), to=int(c) + 1)
        assert np.abs(X[c] - ref[c]).max() < 1e-9 * max(1.0, np.abs(ref[c]).max()), c
    # (4) independence of the schedule: sampling a sub-range gives the same columns
    lo, hi = 1000, 1500
    sub = (np.ascontiguousarray(M[0][lo:hi + 1] - M[0][lo]), M[1][M[0][lo]:M[0][hi]].copy(), M[2][M[0][lo]:M[0][hi]].copy())
    part = eng.side_create(nm, nu, *sub, mean, col_from=lo, col_to=hi)
    eng.sample_side(part, ot, 2, 2.0, mu, LF)
    Xp = eng.get_items(part)
    assert np.allclose(Xp[lo:hi], X[lo:hi], rtol=1e-11, atol=1e-13) and not np.any(Xp[:lo]) and not np.any(Xp[hi:])
    for sd in (me, ot, part):
        eng.side_destroy(sd)


def _prior_only_check(eng, K, M, nrows, it, tol, nsample=256, seed=0):
    """alpha = 0 removes the data term: Lambda* = LambdaF, b = LambdaF mu, hence
    x_i = mu + L^-T z_i with L = chol(LambdaF) and z_i the K normals of stream (i+1)*K*(it+1) mod 2^32
    -- a closed form for EVERY column that needs no oracle: (x_i - mu)^T L must be z_i."""
    import bpmf_amd
    rng = np.random.default_rng(seed)
    ncols = len(M[0]) - 1
    A = rng.standard_normal((K, 2 * K)); LF = A @ A.T / (2 * K) + np.eye(K)
    mu = rng.standard_normal(K)
    other = 0.3 * rng.standard_normal((nrows, K))
    me = eng.side_create(ncols, nrows, *M, 3.0)
    ot = eng.side_create(nrows, ncols, np.zeros(nrows + 1, np.int64), np.zeros(0, np.int32), np.zeros(0), 0.0)
    eng.set_items(ot, other)
    s, p, n = eng.sample_side(me, ot, it, 0.0, mu, LF)
    X = eng.get_items(me)
    eng.side_destroy(me); eng.side_destroy(ot)
    assert np.all(np.isfinite(X))
    Z = (X - mu) @ np.linalg.cholesky(LF)
    # every column: standard normals (the whole matrix is ncols*K draws)
    assert abs(Z.mean()) < 5.0 / np.sqrt(Z.size) + tol and abs(Z.var() - 1.0) < 0.02
    # sampled columns: exactly the reference's stream (host implementation of the same Philox / polar draw)
    pick = np.unique(np.concatenate([[0, 1, ncols - 1], rng.choice(ncols, nsample, replace=False)]))
    for c in pick:
        z = bpmf_amd.engine.randn_host(((int(c) + 1) * K * (it + 1)) % 2 ** 32, K)
        assert np.abs(Z[c] - z).max() < tol, (c, np.abs(Z[c] - z).max())
    # the reductions belong to these columns
    assert np.allclose(s, X.sum(0), rtol=1e-9, atol=1e-6 * max(1.0, tol * 1e6))
    return X


def test_chembl_shape_k64_prior_only_closed_form(hip_engine_factory):
    """BASELINE configs[2] size (483 500 x 5 775, ~1.02 M activities, K = 64): both sides."""
    K = 64
    M, Mt, T, Tt, nu, nm = util.synthetic(483500, 5775, 1_023_952, seed=42)
    eng = hip_engine_factory(K)
    _prior_only_check(eng, K, Mt, nm, 3, 1e-9)          # 483 500 compound columns, 1-3 activities each
    _prior_only_check(eng, K, M, nu, 4, 1e-9, nsample=64)


def test_ml1m_shape_k128_f32_prior_only_closed_form(hip_engine_factory, ml1m):
    """BASELINE configs[4] size (ML-1M shape, K = 128, fp32): the closed form within fp32 tolerance."""
    K = 128
    M, Mt, T, Tt, nu, nm = ml1m
    eng = hip_engine_factory(K, "f32")
    _prior_only_check(eng, K, Mt, nm, 2, 2e-3, nsample=64)


def test_large_side_k32_prior_only_closed_form(hip_engine_factory):
    """10^6 columns on one side (the per-rank column count of the 10M x 1M config is 1.25 M): the
    per-item sampler with a seven-digit grid."""
    M, Mt, T, Tt, nu, nm = util.synthetic(1_000_000, 20_000, 6_000_000, seed=7)
    for K in (32, 16, 8):                                # >= 20 000 columns: four columns per wave (k_sample4)
        eng = hip_engine_factory(K)
        _prior_only_check(eng, K, Mt, nm, 1, 1e-9, nsample=128)
        _prior_only_check(eng, K, M, nu, 2, 1e-9, nsample=32)
