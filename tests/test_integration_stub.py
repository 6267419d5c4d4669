"""INTEGRATION.md section 2 shows the back-end header (`c++/hip_sys.h`) a maintainer of the reference would add.  The
reference's own headers need Eigen3, which this image lacks, so the stub cannot be compiled into the reference here.
What IS done: the stub is extracted from INTEGRATION.md, compiled and LINKED against libbpmf_hip.so together with
tests/integration/ref_shim.h (an Eigen-free stand-in for exactly the members of `struct Sys` a back-end header touches:
/root/reference c++/bpmf.h:112-239) and tests/integration/ref_main.cpp (main()'s NO_COMM loop, c++/bpmf.cpp:180-253) -- and,
on the GPU box, RUN on data/tiny the way the reference's data/tiny/run_test.sh runs `bpmf`, with every output compared
with the oracle.  The shim lives only under tests/; it is not a build of the reference."""
import os
import re
import subprocess

import numpy as np
import pytest

from tests import util
from tests.conftest import ROOT

G = util.GOLDEN


def _build(tmp_path, K):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```cpp\n(.*?)```", text, re.S)
    stub = [b for b in blocks if "struct HIP_Sys" in b]
    assert len(stub) == 1
    (tmp_path / "hip_sys.h").write_text(stub[0])
    exe = str(tmp_path / "bpmf_ref_hip")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unused", "-DBPMF_NUMLATENT=%d" % K, "-I", os.path.join(ROOT, "include"),
                        "-I", os.path.join(ROOT, "tests", "integration"), "-I", str(tmp_path),
                        os.path.join(ROOT, "tests", "integration", "ref_main.cpp"), "-o", exe,
                        "-L", os.path.join(ROOT, "bpmf_amd"), "-lbpmf_hip", "-Wl,-rpath," + os.path.join(ROOT, "bpmf_amd")],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_hip_sys_stub_compiles_and_links_against_the_c_abi(tmp_path):
    """Valid C++, every C-ABI call with the right arguments, every symbol resolved by libbpmf_hip.so."""
    import bpmf_amd
    bpmf_amd.load_library()                                           # (built)
    exe = _build(tmp_path, 8)
    assert os.path.exists(exe)
    stub = (tmp_path / "hip_sys.h").read_text()
    for call in ("bpmf_hip_side_aggr_add", "bpmf_hip_side_aggr_finalize", "bpmf_hip_sys_sample", "bpmf_hip_predict", "bpmf_hip_test_get"):
        assert call in stub, call


@pytest.mark.gpu
def test_hip_sys_stub_runs_tiny_like_the_reference(oracle, tmp_path):
    """data/tiny/run_test.sh through the maintainer's binding: -i 9 -b 0 -v -o output/, K = 8."""
    from bpmf_amd import io as bio
    K = 8
    exe = _build(tmp_path, K)
    (tmp_path / "output").mkdir()
    r = subprocess.run([exe, "-i", "9", "-b", "0", "-v", "-n", os.path.join(G, "tiny-train.mtx"), "-p", os.path.join(G, "tiny-test.mtx"),
                        "-o", "output"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    out = r.stdout
    M, Mt, T, Tt, nu, nm = util.tiny()
    ref = oracle.gibbs(K, M, Mt, T, Tt, nsims=9, burnin=0)
    final = float(re.search(r"Final Avg RMSE: (\S+)", out).group(1))
    assert final < 3.0 and abs(final - ref["final_rmse_avg"]) < 1e-4           # (run_test.sh:15; printed with 6 significant digits)
    lines = [l for l in out.splitlines() if " iteration " in l]
    assert len(lines) == 9 and lines[0].startswith("0: Sampling iteration 0:")
    rm = [float(re.search(r"\t RMSE: (\S+)", l).group(1)) for l in lines]
    rma = [float(re.search(r"avg RMSE: (\S+)", l).group(1)) for l in lines]
    assert np.allclose(rm, ref["rmse"], atol=1e-4) and np.allclose(rma, ref["rmse_avg"], atol=1e-4)
    fu = [float(re.search(r"FU\(\s*(\S+)\)", l).group(1)) for l in lines]
    assert np.allclose(fu, ref["norm_u"], atol=6e-3)                            # (printed with 2 decimals)
    assert "mean rating: 3.66667" in out and "num movs: 2" in out and "num users: 4" in out and "num_latent: 8" in out
    assert "computed on 2 items (100% of total items in test set)" in out
    # -v: the host's items() after every iteration; the last ones are the oracle's final factors
    U8 = bio.read_dense(tmp_path / "output" / "U-8.ddm"); V8 = bio.read_dense(tmp_path / "output" / "V-8.ddm")
    assert U8.shape == (K, nu) and V8.shape == (K, nm)
    assert np.allclose(U8.T, ref["U"], rtol=1e-8, atol=1e-10) and np.allclose(V8.T, ref["V"], rtol=1e-8, atol=1e-10)
    # -o: predictions, and the posterior the round-4 stub lost (aggrMu / aggrLambda were never updated)
    nr, nc, pavg = bio.read_sparse(tmp_path / "output" / "Pavg.sdm")
    assert (nr, nc) == (nu, nm) and np.array_equal(pavg[1], T[1]) and np.allclose(pavg[2], ref["Pavg"], rtol=1e-9)
    nr, nc, pm2 = bio.read_sparse(tmp_path / "output" / "Pm2.sdm")
    assert np.allclose(pm2[2], ref["Pm2"], rtol=1e-7, atol=1e-9)
    for side, n in (("U", nu), ("V", nm)):
        samples = np.stack([bio.read_dense(tmp_path / "output" / ("%s-%d.ddm" % (side, i))) for i in range(9)])
        mu = bio.read_dense(tmp_path / "output" / ("%s-mu.ddm" % side))
        lam = bio.read_dense(tmp_path / "output" / ("%s-Lambda.ddm" % side))
        assert mu.shape == (K, n) and lam.shape == (K * K, n)
        assert np.all(np.isfinite(mu)) and np.all(np.isfinite(lam)) and np.abs(mu).max() > 0
        assert np.allclose(mu, samples.mean(0), rtol=1e-10, atol=1e-12)
        for c in range(n):                                                       # 9 samples of an 8-vector: invertible
            assert np.allclose(lam[:, c].reshape(K, K, order="F"), np.linalg.inv(np.cov(samples[:, :, c].T)), rtol=1e-5, atol=1e-7)
