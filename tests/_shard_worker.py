"""Worker of tests/test_gpu_shard.py (a process of its own: torch and the library each bring up the HIP runtime;
behind a long pytest session with many contexts torch's lazy initialisation reported "No HIP GPUs are available")."""
import numpy as np
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

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


def main():
    import torch
    import bpmf_amd
    from oracle.oracle import Oracle
    oracle = Oracle()
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

    eng = bpmf_amd.HipEngine(K)
    users = eng.side_create_dev(big.NU, big.NI, ucp, uri.data_ptr(), uva.data_ptr(), big.mean_rating, col_from=u0, col_to=u1, keep=(uri, uva))
    movies = eng.side_create_dev(big.NI, big.NU, mcp, mri.data_ptr(), mva.data_ptr(), big.mean_rating, col_from=i0, col_to=i1, keep=(mri, mva))
    U = eng.items_tensor(users, dev); V = eng.items_tensor(movies, dev)
    g = torch.Generator(device=dev); g.manual_seed(7)
    eng.factors_view(U).copy_(0.3 * torch.randn((U.shape[0], K), generator=g, device=dev, dtype=torch.float64))      # (never the padding rows)
    eng.factors_view(V).copy_(0.3 * torch.randn((V.shape[0], K), generator=g, device=dev, dtype=torch.float64))
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
    eng.close()
    print("SHARD-OK")


if __name__ == "__main__":
    main()
