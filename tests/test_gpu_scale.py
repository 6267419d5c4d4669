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
