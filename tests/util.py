"""Shared helpers of the test-suite: data loading and synthetic rating matrices."""
import os

import numpy as np
import scipy.io
import scipy.sparse as sp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def csc_arrays(m):
    m = m.tocsc()
    m.sum_duplicates()
    m.sort_indices()
    return (np.ascontiguousarray(m.indptr, np.int64), np.ascontiguousarray(m.indices, np.int32),
            np.ascontiguousarray(m.data, np.float64))


def load_pair(train, test):
    """Reads train/test MatrixMarket files like Sys::Sys (c++/sample.cpp:112-127):
    both are resized to max(rows) x max(cols).  Returns (M, Mt, T, Tt, nusers, nmovies)
    with M/T = CSC by movie (rows = users) and Mt/Tt the transposes."""
    m = scipy.io.mmread(os.path.join(GOLDEN, train)).tocoo()
    t = scipy.io.mmread(os.path.join(GOLDEN, test)).tocoo()
    nr = max(m.shape[0], t.shape[0]); nc = max(m.shape[1], t.shape[1])
    M = sp.coo_matrix((m.data.astype(np.float64), (m.row, m.col)), shape=(nr, nc)).tocsc()
    T = sp.coo_matrix((t.data.astype(np.float64), (t.row, t.col)), shape=(nr, nc)).tocsc()
    return csc_arrays(M), csc_arrays(M.T), csc_arrays(T), csc_arrays(T.T), nr, nc


def tiny():
    return load_pair("tiny-train.mtx", "tiny-test.mtx")


def ml100k():
    return load_pair("ml100k-train.mtx.gz", "ml100k-test.mtx.gz")


def synthetic(nusers, nmovies, nnz, seed=42, test_frac=0.1, heavy=None, rating_levels=5):
    from bpmf_amd import synth
    return synth.ratings(nusers, nmovies, nnz, seed=seed, test_frac=test_frac, heavy=heavy, rating_levels=rating_levels)


def mean_rating(M):
    return float(np.sum(M[2])) / len(M[2])


def blocks(nusers=240, nmovies=160, per_user=12, cross=6, seed=7):
    """Two communities: the first half of the users rates the first half of the movies, the second
    half the second, plus `cross` ratings across -- the shape where a column is read by few ranks
    (c++/assign.cpp:204-241).  Same return layout as synth.ratings()."""
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    hu, hm = nusers // 2, nmovies // 2
    for u in range(nusers):
        base = 0 if u < hu else hm
        for m in rng.choice(hm if u < hu else nmovies - hm, size=per_user, replace=False):
            rows.append(u); cols.append(base + int(m))
    for _ in range(cross):
        u = int(rng.integers(0, nusers)); m = int(rng.integers(0, nmovies))
        rows.append(u); cols.append(m)
    key = np.unique(np.asarray(rows, np.int64) * nmovies + np.asarray(cols, np.int64))
    rows, cols = key // nmovies, key % nmovies
    vals = rng.integers(1, 6, size=len(rows)).astype(np.float64)
    is_test = rng.random(len(rows)) < 0.1
    M = sp.coo_matrix((vals[~is_test], (rows[~is_test], cols[~is_test])), shape=(nusers, nmovies)).tocsc()
    T = sp.coo_matrix((vals[is_test], (rows[is_test], cols[is_test])), shape=(nusers, nmovies)).tocsc()
    return csc_arrays(M), csc_arrays(M.T), csc_arrays(T), csc_arrays(T.T), nusers, nmovies
