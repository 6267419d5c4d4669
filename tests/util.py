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
