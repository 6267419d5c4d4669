"""The C oracle against the committed golden vectors (tests/golden/*.npz), which were produced
by an independent numpy / pure-Python restatement (tests/golden/gen_golden.py: Philox on Python
integers, dense algebra through LAPACK).  The reference itself holds no numeric fixture for this
path beyond "data/tiny Final Avg RMSE < 3" (data/tiny/run_test.sh:15), which is checked too."""
import os

import numpy as np
import pytest

from tests import util

G = util.GOLDEN


def load(name):
    return np.load(os.path.join(G, name))


def test_rng_streams(oracle):
    z = load("rng.npz")
    for c in (0, 1, 32, 2 ** 32 - 1):
        assert np.array_equal(oracle.randn(c, 64), z["randn_%d" % c])       # same libm: bit-identical
    assert np.array_equal(oracle.words(7, 16), z["words_7"])
    g, _ = oracle.gamma_stream(3, z["gamma_alphas"])
    # the golden stream has no randn() between the gamma draws -> only the first value lines up
    assert g[0] == z["gamma_3"][0]


def test_hyper_draws(oracle):
    z = load("hyper.npz")
    for K, N in ((8, 4), (16, 50), (32, 943)):
        for counter in (0, 5):
            pre = "hyper_K%d_N%d_c%d_" % (K, N, counter)
            mu, LU, LF = oracle.hyper_sample(K, N, z[pre + "cov"], counter)
            scale = np.abs(z[pre + "LF"]).max()
            assert np.allclose(mu, z[pre + "mu"], rtol=1e-10, atol=1e-12)
            assert np.allclose(LU, z[pre + "LU"], rtol=1e-10, atol=1e-12 * scale)
            assert np.allclose(LF, z[pre + "LF"], rtol=1e-10, atol=1e-12 * scale)


def test_tiny_full_run(oracle):
    """data/tiny, K = 8, -i 9 -b 0 (the reference's run_test.sh)."""
    z = load("tiny_k8.npz")
    M, Mt, T, Tt, nu, nm = util.tiny()
    r = oracle.gibbs(8, M, Mt, T, Tt, nsims=9, burnin=0, trace=True)
    assert r["final_rmse_avg"] < 3.0                                        # the reference's own assertion
    assert abs(r["final_rmse_avg"] - float(z["final_rmse_avg"])) < 1e-10
    assert np.allclose(r["rmse"], z["rmse"], rtol=0, atol=1e-10)
    assert np.allclose(r["rmse_avg"], z["rmse_avg"], rtol=0, atol=1e-10)
    assert np.allclose(r["norm_u"], z["norm_u"], rtol=1e-10) and np.allclose(r["norm_m"], z["norm_m"], rtol=1e-10)
    assert np.allclose(r["U"], z["U"][-1], rtol=1e-9, atol=1e-11) and np.allclose(r["V"], z["V"][-1], rtol=1e-9, atol=1e-11)
    assert np.allclose(r["Pavg"], z["Pavg"], rtol=1e-10) and np.allclose(r["Pm2"], z["Pm2"], rtol=1e-8, atol=1e-10)
    K = 8
    for it in range(9):
        assert np.allclose(r["trace"][it, 0, :K], z["mu_m"][it], rtol=1e-9, atol=1e-11)
        assert np.allclose(r["trace"][it, 1, K:].reshape(K, K, order="F"), z["LF_u"][it], rtol=1e-9, atol=1e-10)


def test_ml100k_first_iterations(oracle):
    """MovieLens-100K (the reference's CTest data), K = 32, -i 3 -b 1."""
    z = load("ml100k_k32.npz")
    M, Mt, T, Tt, nu, nm = util.ml100k()
    r = oracle.gibbs(32, M, Mt, T, Tt, nsims=3, burnin=1, trace=True)
    assert np.allclose(r["rmse"], z["rmse"], rtol=0, atol=1e-10)
    assert np.allclose(r["rmse_avg"], z["rmse_avg"], rtol=0, atol=1e-10)
    assert abs(r["final_rmse_avg"] - float(z["final_rmse_avg"])) < 1e-10
    assert abs(r["rmse"][0] - 1.153676) < 1e-3                              # mean predictor
    assert np.allclose(r["U"][z["keep_u"]], z["U"][-1], rtol=1e-8, atol=1e-11)
    assert np.allclose(r["V"][z["keep_m"]], z["V"][-1], rtol=1e-8, atol=1e-11)
    K = 32
    for it in range(3):
        assert np.allclose(r["trace"][it, 0, K:].reshape(K, K, order="F"), z["LF_m"][it], rtol=1e-8, atol=1e-8)
        assert np.allclose(r["trace"][it, 1, :K], z["mu_u"][it], rtol=1e-8, atol=1e-11)
