"""The matrix IO of the `bpmf` surface (include/bpmf_io.h) against scipy's independent readers
and through write/read round trips: .mtx (coordinate + array, comments, tabs), .gz, .sdm, .sbm,
.ddm, .csv; duplicates summed; error behaviour (c++/io.cpp:117: "File '...' not found")."""
import gzip
import os
import struct

import numpy as np
import pytest
import scipy.io
import scipy.sparse as sp

from bpmf_amd import io as bio
from tests import util


def as_scipy(nr, nc, csc):
    return sp.csc_matrix((csc[2], csc[1], csc[0]), shape=(nr, nc))


@pytest.mark.parametrize("name", ["tiny-train.mtx", "tiny-test.mtx", "ml100k-train.mtx.gz", "ml100k-test.mtx.gz"])
def test_mtx_matches_scipy(name):
    path = os.path.join(util.GOLDEN, name)
    nr, nc, csc = bio.read_sparse(path)
    ref = scipy.io.mmread(path).tocsc(); ref.sort_indices()
    assert (nr, nc) == ref.shape
    assert np.array_equal(csc[0], ref.indptr) and np.array_equal(csc[1], ref.indices) and np.array_equal(csc[2], ref.data)
    for c in range(nc):                                  # ascending rows per column
        r = csc[1][csc[0][c]:csc[0][c + 1]]
        assert np.all(np.diff(r) > 0)


def test_mtx_duplicates_comments_pattern(tmp_path):
    p = tmp_path / "d.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n% a comment\n\n3 2 5\n1 1 1.5\n3\t2   2\n% mid comment\n1 1 2.5\n2 2 0\n3 1 -1e-3\n")
    nr, nc, csc = bio.read_sparse(p)
    m = as_scipy(nr, nc, csc).toarray()
    assert (nr, nc) == (3, 2) and csc[0][-1] == 4                 # duplicates summed, explicit zero kept
    assert np.allclose(m, [[4.0, 0], [0, 0], [-1e-3, 2.0]])
    q = tmp_path / "p.mtx"
    q.write_text("%%MatrixMarket matrix coordinate pattern general\n2 2 2\n1 2\n2 1\n")
    nr, nc, csc = bio.read_sparse(q)
    assert np.allclose(as_scipy(nr, nc, csc).toarray(), [[0, 1], [1, 0]])


@pytest.mark.parametrize("ext", [".sdm", ".sdm.gz", ".mtx", ".mtx.gz", ".mm"])
def test_sparse_round_trip(tmp_path, ext):
    rng = np.random.default_rng(0)
    m = sp.random(40, 17, density=0.2, random_state=3, format="csc")
    m.data = np.round(rng.standard_normal(len(m.data)), 3); m.sort_indices()
    csc = (m.indptr.astype(np.int64), m.indices.astype(np.int32), m.data)
    p = tmp_path / ("m" + ext)
    bio.write_sparse(p, 40, 17, csc)
    nr, nc, back = bio.read_sparse(p)
    assert (nr, nc) == (40, 17) and np.array_equal(back[0], csc[0]) and np.array_equal(back[1], csc[1])
    assert np.allclose(back[2], csc[2], rtol=1e-5 if "m" in ext[1:3] else 0, atol=0)
    if ext == ".sdm":                                      # byte layout: u64 x3, u32 rows, u32 cols (1-based), f64 vals
        raw = p.read_bytes()
        nrow, ncol, nnz = struct.unpack("<3Q", raw[:24])
        assert (nrow, ncol, nnz) == (40, 17, len(m.data)) and len(raw) == 24 + nnz * 16
        rows = np.frombuffer(raw, np.uint32, nnz, 24); cols = np.frombuffer(raw, np.uint32, nnz, 24 + 4 * nnz)
        assert rows.min() >= 1 and cols[0] == 1 and np.all(np.diff(cols.astype(int)) >= 0)
        assert np.array_equal(np.frombuffer(raw, np.float64, nnz, 24 + 8 * nnz), m.data)


def test_sbm_is_pattern(tmp_path):
    csc = (np.array([0, 2, 3], np.int64), np.array([0, 2, 1], np.int32), np.array([2.0, -1.0, 5.0]))
    p = tmp_path / "b.sbm"
    bio.write_sparse(p, 3, 2, csc)                          # entries with value <= 0 are dropped (c++/io.cpp:666)
    nr, nc, back = bio.read_sparse(p)
    assert np.allclose(as_scipy(nr, nc, back).toarray(), [[1, 0], [0, 1], [0, 0]])


@pytest.mark.parametrize("ext", [".ddm", ".ddm.gz", ".csv", ".mtx", ".csv.gz"])
def test_dense_round_trip(tmp_path, ext):
    a = np.round(np.random.default_rng(1).standard_normal((5, 7)), 4)
    p = tmp_path / ("d" + ext)
    bio.write_dense(p, a)
    b = bio.read_dense(p)
    assert b.shape == a.shape and np.allclose(a, b, rtol=1e-5, atol=1e-12)
    if ext == ".ddm":                                       # u64 nrow, ncol, then column-major doubles
        raw = p.read_bytes()
        assert struct.unpack("<2Q", raw[:16]) == (5, 7)
        assert np.array_equal(np.frombuffer(raw, np.float64, 35, 16).reshape(7, 5).T, a)
    if ext == ".mtx":
        assert np.allclose(scipy.io.mmread(str(p)), a, rtol=1e-5)


def test_errors(tmp_path):
    with pytest.raises(bio.BpmfIoError, match="File '.*nope.mtx' not found"):
        bio.read_sparse(tmp_path / "nope.mtx")
    with pytest.raises(bio.BpmfIoError, match="Unknown matrix type"):
        bio.read_sparse(tmp_path / "x.bin")
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket matrix coordinate real symmetric\n2 2 1\n1 1 1\n")
    with pytest.raises(bio.BpmfIoError, match="symmetry"):
        bio.read_sparse(bad)
    short = tmp_path / "short.mtx"
    short.write_text("%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1\n")
    with pytest.raises(bio.BpmfIoError, match="fewer entries"):
        bio.read_sparse(short)
    with pytest.raises(bio.BpmfIoError, match="Invalid matrix type"):
        bio.read_sparse(tmp_path / "x.ddm")
