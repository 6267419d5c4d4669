"""Seeded synthetic rating matrices of the shapes BASELINE.json names (there is no
network for MovieLens-1M / ChEMBL-20, and the reference ships only ML-100K).
Host-side numpy; returns CSC triples (colptr int64, rowidx int32, vals f64)."""
import numpy as np
import scipy.sparse as sp


def csc_arrays(m):
    m = m.tocsc()
    m.sum_duplicates()
    m.sort_indices()
    return (np.ascontiguousarray(m.indptr, np.int64), np.ascontiguousarray(m.indices, np.int32),
            np.ascontiguousarray(m.data, np.float64))


def ratings(nusers, nmovies, nnz, seed=42, test_frac=0.1, heavy=None, rating_levels=5, real_valued=False):
    """Item popularity ~ Zipf(0.9), user activity ~ log-normal, integer ratings 1..levels
    (or N(6,1.3^2) when real_valued, ChEMBL-like).  `heavy` = (movie, count) forces one
    movie to have `count` ratings.  Returns (M, Mt, T, Tt, nusers, nmovies): M/T are CSC
    with one column per movie (rows = users), Mt/Tt the transposes."""
    rng = np.random.default_rng(seed)
    pu = rng.lognormal(0.0, 1.0, nusers); pu /= pu.sum()
    pm = 1.0 / (np.arange(1, nmovies + 1) + 10.0) ** 0.9; pm = pm[rng.permutation(nmovies)]; pm /= pm.sum()
    # draw until `nnz` distinct cells exist (popular items saturate), keep the first nnz of them
    key = np.zeros(0, np.int64)
    need = nnz
    while True:
        n_draw = int(need * 1.3) + 16
        k = rng.choice(nusers, size=n_draw, p=pu).astype(np.int64) * nmovies + rng.choice(nmovies, size=n_draw, p=pm)
        key = np.concatenate([key, k])
        _, first = np.unique(key, return_index=True)
        if len(first) >= nnz:
            break
        need = nnz - len(first)
    first = np.sort(first)[:nnz]
    key = key[first]
    rows, cols = key // nmovies, key % nmovies
    if heavy is not None:
        hc, cnt = heavy
        extra = rng.choice(nusers, size=min(cnt, nusers), replace=False)
        rows = np.concatenate([rows[cols != hc], extra]); cols = np.concatenate([cols[cols != hc], np.full(len(extra), hc)])
    if real_valued:
        vals = rng.normal(6.0, 1.3, size=len(rows))
    else:
        vals = rng.integers(1, rating_levels + 1, size=len(rows)).astype(np.float64)
    is_test = rng.random(len(rows)) < test_frac
    if heavy is not None:
        is_test &= cols != heavy[0]
    M = sp.coo_matrix((vals[~is_test], (rows[~is_test], cols[~is_test])), shape=(nusers, nmovies)).tocsc()
    T = sp.coo_matrix((vals[is_test], (rows[is_test], cols[is_test])), shape=(nusers, nmovies)).tocsc()
    return csc_arrays(M), csc_arrays(M.T), csc_arrays(T), csc_arrays(T.T), nusers, nmovies


def ml1m_shaped(seed=42):
    """6040 users x 3706 movies, 1 000 209 ratings, 90/10 train/test split."""
    return ratings(6040, 3706, 1_000_209, seed=seed)


def slice_cols(csc, lo, hi):
    """CSC slice of the columns [lo, hi) with colptr rebased to 0."""
    colptr, rowidx, vals = csc
    a, b = int(colptr[lo]), int(colptr[hi])
    return (np.ascontiguousarray(colptr[lo:hi + 1] - colptr[lo]), np.ascontiguousarray(rowidx[a:b]),
            np.ascontiguousarray(vals[a:b]))


def balanced_ranges(colptr, nparts, fixed_cost=10):
    """Contiguous column ranges balanced on work = fixed_cost + nnz per column (the
    reference's assign() uses 10 + nnz, c++/assign.cpp:109-120).  Returns nparts+1 bounds."""
    n = len(colptr) - 1
    work = np.diff(colptr).astype(np.float64) + fixed_cost
    cum = np.concatenate([[0.0], np.cumsum(work)])
    bounds = [0]
    for p in range(1, nparts):
        bounds.append(int(np.searchsorted(cum, cum[-1] * p / nparts)))
    bounds.append(n)
    for i in range(1, len(bounds)):
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds
