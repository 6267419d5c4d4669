"""Assignment of columns to ranks (bpmf_amd/csrc/assign.cpp behind include/bpmf_io.h): the reference's greedy,
permuting Sys::assign (c++/assign.cpp:52-201) against an independent Python restatement of the same lines, and the
contiguous work-balanced cuts.  GPU part: `bpmf` with the renumbering of 3 ranks applied on one GPU gives the chain
of the oracle on the renumbered matrices, and writes every output in the ORIGINAL numbering."""
import os
import re
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

from bpmf_amd import _lib, io as bio
from tests import util
from tests.conftest import ROOT

G = util.GOLDEN


def assign_py(nnz, nprocs):
    """c++/assign.cpp:66-201 with other.assigned == false (the comm term has weight 0 anyway, :158)."""
    n = len(nnz)
    work = [0.0] * nprocs; owner = [-1] * n
    total = 0.01
    for _ in range(3):
        for i in range(n):
            if owner[i] >= 0:
                work[owner[i]] -= 7.1 + nnz[i]; total -= 7.1 + nnz[i]; owner[i] = -1
            best, mn = -1, 1e9
            for p in range(nprocs):
                cost = 10000 * (work[p] / total) + 0 * 0.0
                if cost > mn:
                    continue
                best, mn = p, cost
            owner[i] = best; work[best] += 10.0 + nnz[i]; total += 10.0 + nnz[i]
    order = [i for p in range(nprocs) for i in range(n) if owner[i] == p]
    counts = [sum(1 for o in owner if o == p) for p in range(nprocs)]
    return np.array(order), np.concatenate([[0], np.cumsum(counts)])


def greedy(colptr, parts):
    lib = _lib.load_library()
    n = len(colptr) - 1
    order = np.zeros(n, np.int64); dom = np.zeros(parts + 1, np.int64)
    cp = np.ascontiguousarray(colptr, np.int64)
    assert lib.bpmf_assign_greedy(n, cp.ctypes.data, parts, order.ctypes.data, dom.ctypes.data) == 0
    return order, dom


@pytest.mark.parametrize("parts", [2, 3, 8])
def test_greedy_assignment_is_the_reference_algorithm(parts):
    M, Mt, T, Tt, nu, nm = util.ml100k()
    for csc in (M, Mt):
        order, dom = greedy(csc[0], parts)
        ref_order, ref_dom = assign_py(np.diff(csc[0]).tolist(), parts)
        assert np.array_equal(order, ref_order) and np.array_equal(dom, ref_dom)
        nnz = np.diff(csc[0])
        load = [nnz[order[a:b]].sum() for a, b in zip(dom, dom[1:])]
        assert max(load) - min(load) < 0.02 * sum(load) / parts + 700          # nnz AND column counts balanced
        assert max(np.diff(dom)) - min(np.diff(dom)) < 0.25 * len(nnz) / parts + 2


def test_contiguous_cuts_balance_work():
    lib = _lib.load_library()
    M, Mt, T, Tt, nu, nm = util.ml100k()
    for parts in (1, 2, 5):
        dom = np.zeros(parts + 1, np.int64)
        assert lib.bpmf_assign_contiguous(nm, M[0].ctypes.data, parts, 64.0, dom.ctypes.data) == 0
        assert dom[0] == 0 and dom[-1] == nm and np.all(np.diff(dom) >= 0)
        work = [(M[0][b] - M[0][a]) + 64 * (b - a) for a, b in zip(dom, dom[1:])]
        assert max(work) < 1.15 * sum(work) / parts + 700


@pytest.mark.gpu
def test_renumbered_run_equals_the_oracle_on_the_renumbered_matrix(oracle, tmp_path):
    K, parts, nsims, burnin = 8, 3, 6, 2
    M, Mt, T, Tt, nu, nm = util.ml100k()
    # the renumbering bpmf applies: movies, users, movies, users (c++/bpmf.cpp:140-143)
    pm, pu = np.arange(nm), np.arange(nu)
    for _ in range(2):
        for side in (0, 1):
            perm, csc = (pm, M) if side == 0 else (pu, Mt)
            cp = np.concatenate([[0], np.cumsum(np.diff(csc[0])[perm])])
            order, _ = greedy(cp, parts)
            if side == 0:
                pm = perm[order]
            else:
                pu = perm[order]
    A = sp.csc_matrix((M[2], M[1], M[0]), shape=(nu, nm)); B = sp.csc_matrix((T[2], T[1], T[0]), shape=(nu, nm))
    Ap = A[pu][:, pm].tocsc(); Bp = B[pu][:, pm].tocsc()
    Mp, Tp = util.csc_arrays(Ap), util.csc_arrays(Bp)
    ref = oracle.gibbs(K, Mp, util.csc_arrays(Ap.T), Tp, util.csc_arrays(Bp.T), nsims=nsims, burnin=burnin)
    (tmp_path / "o").mkdir()
    bpmf = os.path.join(ROOT, "bpmf_amd", "bpmf")
    r = subprocess.run([bpmf, "-i", str(nsims), "-b", str(burnin), "-d", str(K), "-v", "-o", "o/", "-n", os.path.join(G, "ml100k-train.mtx.gz"),
                        "-p", os.path.join(G, "ml100k-test.mtx.gz")], cwd=tmp_path, env=dict(os.environ, BPMF_TEST_ASSIGN_PARTS=str(parts)),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "assignment: greedy" in r.stdout
    rm = [float(m.group(1)) for m in re.finditer(r"\t RMSE: (\S+)", r.stdout)]
    assert np.allclose(rm, ref["rmse"], atol=1e-4)
    # last sample, written in the ORIGINAL numbering: row pu[j] of the file = row j of the oracle's (renumbered) factor
    U = bio.read_dense(tmp_path / "o" / ("U-%d.ddm" % (nsims - 1))).T; V = bio.read_dense(tmp_path / "o" / ("V-%d.ddm" % (nsims - 1))).T
    assert np.allclose(U[pu], ref["U"], rtol=1e-8, atol=1e-10) and np.allclose(V[pm], ref["V"], rtol=1e-8, atol=1e-10)
    nr, nc, pavg = bio.read_sparse(tmp_path / "o" / "Pavg.sdm")
    P = sp.csc_matrix((pavg[2], pavg[1], pavg[0]), shape=(nu, nm))
    Pref = sp.csc_matrix((ref["Pavg"], Tp[1], Tp[0]), shape=(nu, nm))
    assert abs(P[pu][:, pm] - Pref).max() < 1e-8
    samples = np.stack([bio.read_dense(tmp_path / "o" / ("U-%d.ddm" % i)) for i in range(burnin, nsims)])
    assert np.allclose(bio.read_dense(tmp_path / "o" / "U-mu.ddm"), samples.mean(0), rtol=1e-10, atol=1e-12)
