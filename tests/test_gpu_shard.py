"""BASELINE.json configs[3] -- "Synthetic R 10M users x 1M items, 200 nnz/row, K=32, 8 x MI355X sharded": one rank's
share of the REAL matrix (bpmf_amd/synth_dev.py: device-generated, the same matrix whatever the rank count; rank 7 of 8
= user chunk 7 and the item range with the most columns), both half-iterations on one GPU with random factors:

  * spot-checked against the oracle: the heaviest, the lightest and random columns of both sides (a column's ratings
    are pulled off the device, the rows it reads gathered into a small factor; oracle.sample_column keys the RNG
    stream on the GLOBAL column id), 1e-9 of max|U| as everywhere;
  * size-independent properties of the whole shard: the returned sums are the sums of the sampled columns
    (checksum of checksums), bit-reproducible when run twice, columns outside the rank's range untouched.
The factors of this configuration live in HBM (U: 2.56 GB), not in L2 -- the regime the ML-1M tests cannot reach.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K = 32


def _spot_check(oracle, big_part, X, other_t, mean, it, mu, LF, col0, rng, tol=1e-9):
    colptr, rowidx, vals = big_part
    nnzc = np.diff(colptr)
    order = np.argsort(nnzc)
    pick = np.unique(np.concatenate([order[:6], order[-6:], rng.choice(len(nnzc), 40, replace=False)]))
    worst = 0.0
    for c in pick:
        a, b = int(colptr[c]), int(colptr[c + 1])
        rows = rowidx[a:b].cpu().numpy().astype(np.int64)
        v = vals[a:b].cpu().numpy()
        uq, inv = np.unique(rows, return_inverse=True)
        import torch
        small = other_t[torch.as_tensor(uq, device=other_t.device)].cpu().numpy()
        ref = oracle.sample_column(K, col0 + int(c), inv.astype(np.int32), v, mean, 2.0, small, it, mu, LF)
        got = X[col0 + int(c)].cpu().numpy()
        worst = max(worst, float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max())))
        assert worst < tol, (int(c), b - a, worst)
    return worst


def test_one_rank_of_the_10Mx1M_configuration(oracle, hip_engine_factory):
    import torch
    from bpmf_amd.synth_dev import BigMatrix
    dev = torch.device("cuda", 0)
    big = BigMatrix(dev)                                             # 10M x 1M x 200 per user, 8 groups
    rank, G = 7, 8
    bnd = big.item_bounds()
    assert bnd[0] == 0 and bnd[-1] == big.NI and all(x < y for x, y in zip(bnd, bnd[1:]))
    ucp, uri, uva, u0, u1 = big.users_csc([rank])
    mcp, mri, mva, i0, i1 = big.items_csc([rank])
    assert u1 - u0 == big.NU // G and int(ucp[-1]) == (u1 - u0) * big.PER
    assert abs(int(mcp[-1]) - big.NU * big.PER // G) < 0.02 * big.NU * big.PER // G        # nnz-balanced item ranges
    # exactly 200 distinct, ascending items per user
    r = uri[:200 * 1000].reshape(1000, 200).to(torch.int64)
    assert bool((r[:, 1:] > r[:, :-1]).all()) and int(r.min()) >= 0 and int(r.max()) < big.NI

    eng = hip_engine_factory(K)
    users = eng.side_create_dev(big.NU, big.NI, ucp, uri.data_ptr(), uva.data_ptr(), big.mean_rating, col_from=u0, col_to=u1, keep=(uri, uva))
    movies = eng.side_create_dev(big.NI, big.NU, mcp, mri.data_ptr(), mva.data_ptr(), big.mean_rating, col_from=i0, col_to=i1, keep=(mri, mva))
    U = eng.items_tensor(users, dev); V = eng.items_tensor(movies, dev)
    g = torch.Generator(device=dev); g.manual_seed(7)
    U.copy_(0.3 * torch.randn(U.shape, generator=g, device=dev, dtype=torch.float64))
    V.copy_(0.3 * torch.randn(V.shape, generator=g, device=dev, dtype=torch.float64))
    U0 = U[:u0].clone() if u0 > 0 else None
    rng = np.random.default_rng(3)
    A = rng.standard_normal((K, 3 * K)); cov = A @ A.T / (3 * K) * 0.5
    worst = {}
    for name, me, ot, X, other_t, part, col0, ncols, it in (("users", users, movies, U, V, (ucp, uri, uva), u0, big.NU, 2),
                                                              ("items", movies, users, V, U, (mcp, mri, mva), i0, big.NI, 3)):
        mu, LU, LF = oracle.hyper_sample(K, ncols, cov, it)
        snapshot = other_t.clone()
        s, p, n = eng.sample_side(me, ot, it, 2.0, mu, LF)
        lo, hi = (u0, u1) if name == "users" else (i0, i1)
        mine = X[lo:hi]
        # checksum of checksums: the sums the library hands back belong to the columns it wrote
        assert np.allclose(s, mine.sum(0).cpu().numpy(), rtol=1e-9, atol=1e-6)
        assert np.allclose(p, (mine.T @ mine).cpu().numpy(), rtol=1e-9, atol=1e-5)
        assert bool(torch.isfinite(mine).all())
        worst[name] = _spot_check(oracle, part, X, snapshot, big.mean_rating, it, mu, LF, lo, rng)
        # bit-reproducible; the other side's factor is only read
        first = mine.clone()
        s2, p2, n2 = eng.sample_side(me, ot, it, 2.0, mu, LF)
        assert torch.equal(X[lo:hi], first) and np.array_equal(p, p2) and torch.equal(other_t, snapshot)
        del first, snapshot
    if U0 is not None:
        assert torch.equal(U[:u0], U0)                                # columns of other ranks' ranges: untouched
    print("10M x 1M shard, worst spot-check error / max|U|:", worst)
    eng.side_destroy(users); eng.side_destroy(movies)
