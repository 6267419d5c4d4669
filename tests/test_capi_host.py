"""CPU-only checks of the shipped library: it loads, exports every symbol the header
declares, fails loudly without a GPU, and its host-side pieces (hyper-parameter
draw, RNG stream, cov) agree with the oracle.  No device compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import bpmf_amd
from bpmf_amd import _lib, engine
from tests.conftest import ROOT


def header_symbols():
    names = set()
    for h in ("bpmf_hip.h", "bpmf_io.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        names |= set(re.findall(r"^BPMF(?:_IO)?_API[^;(]*?\b(bpmf_\w+)\s*\(", text, re.M))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(bpmf_amd.library_path())
    names = header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    # and the binding covers exactly the header
    assert sorted(_lib.exported_signatures()) == names


def test_abi_version_and_k_support():
    lib = bpmf_amd.load_library()
    assert lib.bpmf_hip_abi_version() == 1
    # every num_latent the reference ships a binary for (ci/multilatent.sh:5: 8 16 32 64 128 10 20 ... 100) runs in fp64,
    # on the next instantiated size; fp32 is an explicit opt-in for the large sizes only
    assert [k for k in (0, 1, 4, 8, 10, 16, 32, 50, 64, 100, 128, 129, 256) if lib.bpmf_hip_supports_k(k)] == [1, 4, 8, 10, 16, 32, 50, 64, 100, 128]
    assert [lib.bpmf_hip_kernel_k(k, 0) for k in (1, 8, 9, 10, 16, 20, 32, 40, 50, 64, 70, 100, 128, 129)] == [8, 8, 16, 16, 16, 32, 32, 64, 64, 64, 128, 128, 128, 0]
    assert [lib.bpmf_hip_kernel_k(k, 1) for k in (32, 64, 65, 100, 128)] == [0, 0, 128, 128, 128]
    assert lib.bpmf_hip_supports(128, 0) == 1 and lib.bpmf_hip_supports(128, 1) == 1 and lib.bpmf_hip_supports(32, 1) == 0


def test_no_silent_cpu_fallback():
    """Without a HIP device the product must raise, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(bpmf_amd.BpmfHipError) as e:
        bpmf_amd.HipEngine(32)
    assert e.value.code == -2


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "bpmf_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("the oracle", "").lower() or f == "Makefile", (dirpath, f)


def test_host_randn_stream_equals_oracle(oracle):
    for c in (0, 1, 32 * 944 * 20, 2 ** 32 - 1):
        assert np.array_equal(engine.randn_host(c, 500), oracle.randn(c, 500))


@pytest.mark.parametrize("K,N", [(8, 2), (8, 4), (16, 50), (32, 943), (32, 1682), (64, 300), (5, 30), (13, 100), (35, 500), (128, 6040), (128, 3706)])   # (odd K: the ragged ends of the four-at-a-time loops; K = 128: the inverse-free factor)
def test_hyper_sample_matches_oracle(oracle, K, N):
    rng = np.random.default_rng(K * 1000 + N)
    A = rng.standard_normal((K, max(N, K + 3)))
    cov = A @ A.T / A.shape[1]
    for counter in (0, 1, 7):
        mu, LU, LF = engine.hyper_sample(K, N, cov, counter)
        mu2, LU2, LF2 = oracle.hyper_sample(K, N, cov, counter)
        scale = np.abs(LF2).max()
        assert np.allclose(mu, mu2, rtol=1e-10, atol=1e-12)
        assert np.allclose(LU, LU2, rtol=1e-10, atol=1e-12 * scale)
        assert np.allclose(LF, LF2, rtol=1e-10, atol=1e-12 * scale)
        assert np.allclose(LF, LU.T @ LU, rtol=1e-12, atol=1e-12 * scale)
        assert np.all(np.tril(LU, -1) == 0)
    # first call of a run: cov = 0 (Sys::init) -> Lambda ~ Wishart(I, K+N)
    mu, LU, LF = engine.hyper_sample(K, N, np.zeros((K, K)), 0)
    mu2, LU2, LF2 = oracle.hyper_sample(K, N, np.zeros((K, K)), 0)
    assert np.allclose(LF, LF2, rtol=1e-11) and np.allclose(mu, mu2, rtol=1e-11, atol=1e-14)


def test_hyper_sample_with_mean_statistic(oracle):
    # the "fixed" variant (Um != 0) is not what the reference runs (Q1) but the code path exists
    K, N = 16, 40
    rng = np.random.default_rng(1)
    A = rng.standard_normal((K, 60)); cov = A @ A.T / 60; um = rng.standard_normal(K)
    a = engine.hyper_sample(K, N, cov, 3, um); b = oracle.hyper_sample(K, N, cov, 3, um)
    for x, y in zip(a, b):
        assert np.allclose(x, y, rtol=1e-10, atol=1e-12)


def test_cov_from_sums(oracle):
    K, N = 8, 17
    rng = np.random.default_rng(0)
    X = rng.standard_normal((N, K))
    s = X.sum(0); prod = X.T @ X
    c = engine.cov_from_sums(K, N, s, prod)
    assert np.allclose(c, np.cov(X.T), rtol=1e-12, atol=1e-14)
    assert np.allclose(c, oracle.cov(K, N, s, prod), rtol=0, atol=0)


def test_hyper_sample_rejects_bad_arguments():
    lib = bpmf_amd.load_library()
    assert lib.bpmf_hyper_sample(8, 0, None, None, 0, None, None, None) == -1
    assert b"bad argument" in lib.bpmf_hip_last_error()


def test_rccl_double_exports_what_the_library_resolves():
    """tests/rccl_double/librccl_double.so (test infrastructure for ranks that share a GPU, built by build()) must offer
    every nccl* entry point rccl() (capi_context.hip) looks up; loading it needs no GPU."""
    path = os.path.join(ROOT, "tests", "rccl_double", "librccl_double.so")
    assert os.path.exists(path), "run __graft_entry__.build()"
    lib = C.CDLL(path)
    src = open(os.path.join(ROOT, "bpmf_amd", "csrc", "capi_context.hip")).read()
    wanted = set(re.findall(r"BPMF_SYM\((\w+)\)", src)) - {"f"}
    assert {"GetUniqueId", "CommInitRank", "Send", "Recv", "AllReduce", "Reduce", "CommSplit", "CommCount"} <= wanted
    for name in wanted:
        assert hasattr(lib, "nccl" + name), "the double lacks nccl" + name
