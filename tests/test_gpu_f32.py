"""Mixed-precision tolerance study of the fp32 large-K path (BASELINE config "MovieLens-1M, K=128,
fp32"): the HIP kernels keep factors, Gram, factorisation and solves in fp32; the oracle is the
fp64 restatement of the reference.  Tolerances are what fp32 costs, stated here:

  * one half-iteration from identical inputs: sampled factors within 2e-3 of max|U| (observed
    ~1e-4: the Cholesky of Lambda* in fp32 loses cond(Lambda*) * 6e-8), sums within 1e-3 relative;
  * a full 7-iteration run on MovieLens-100K: every per-iteration RMSE and the final averaged RMSE
    within 1e-3 of the fp64 chain (the north star's bar for RMSE).
"""
import os

import numpy as np
import pytest

import util
from test_gpu_parity import half_iteration_pair, rel_err

pytestmark = pytest.mark.gpu

K = 128


def test_f32_context_and_bad_combinations(hip_engine_factory):
    import bpmf_amd
    lib = bpmf_amd._lib.load_library()
    assert lib.bpmf_hip_supports(128, 1) == 1 and lib.bpmf_hip_supports(128, 0) == 1 and lib.bpmf_hip_supports(32, 1) == 0
    with pytest.raises(RuntimeError):
        bpmf_amd.HipEngine(32, dtype="f32")
    # fp32 is what the caller asks for, never what num_latent = 128 silently means: the default context is the reference's fp64
    eng = hip_engine_factory(128)
    assert eng.dtype == "f64" and lib.bpmf_hip_ctx_dtype(eng.ctx) == 0 and lib.bpmf_hip_ctx_num_latent(eng.ctx) == 128
    eng32 = hip_engine_factory(128, "f32")
    assert lib.bpmf_hip_ctx_dtype(eng32.ctx) == 1 and lib.bpmf_hip_ctx_ld(eng32.ctx) == 128


def test_f32_half_iterations_against_fp64_oracle(oracle, hip_engine_factory):
    M, Mt, T, Tt, nu, nm = util.ml100k()
    eng = hip_engine_factory(K, "f32")
    rng = np.random.default_rng(7)
    U = 0.3 * rng.standard_normal((nu, K)); V = 0.3 * rng.standard_normal((nm, K))
    A = rng.standard_normal((K, 3 * K)); cov = A @ A.T / (3 * K)
    for (mat, nrows, other, it) in ((M, nu, U, 5), (Mt, nm, V, 7)):
        # the oracle sees the factors the device sees (rounded to fp32 once)
        other32 = other.astype(np.float32).astype(np.float64)
        (items, s, p, n), (items_ref, s_ref, p_ref, n_ref) = half_iteration_pair(oracle, eng, K, mat, nrows, other32, it, cov=cov)
        assert np.all(np.isfinite(items))
        err = rel_err(items, items_ref)
        assert err < 2e-3, err
        assert rel_err(s, s_ref) < 1e-3 and rel_err(p, p_ref) < 1e-3 and abs(n - n_ref) <= 1e-3 * abs(n_ref)


def test_f32_tiny_and_empty_columns(oracle, hip_engine_factory):
    """iteration 0 (zero factors: every column samples from the prior) and the shipped tiny matrix."""
    M, Mt, T, Tt, nu, nm = util.tiny()
    eng = hip_engine_factory(K, "f32")
    (items, s, p, n), (items_ref, s_ref, p_ref, n_ref) = half_iteration_pair(oracle, eng, K, M, nu, np.zeros((nu, K)), 0)
    assert rel_err(items, items_ref) < 2e-3


def test_f32_full_run_rmse_within_1e3_of_fp64(oracle, hip_engine_factory):
    import bpmf_amd
    M, Mt, T, Tt, nu, nm = util.ml100k()
    eng = hip_engine_factory(K, "f32")
    res = bpmf_amd.gibbs(eng, M, Mt, T, nu, nm, nsims=7, burnin=2)
    ref = oracle.gibbs(K, M, Mt, T, Tt, nsims=7, burnin=2, nthreads=max(1, min(os.cpu_count() or 1, 16)))
    assert np.allclose(res["rmse"], ref["rmse"], atol=1e-3), np.abs(np.array(res["rmse"]) - np.array(ref["rmse"])).max()
    assert abs(res["final_rmse_avg"] - ref["final_rmse_avg"]) < 1e-3
    assert 0.9 < res["final_rmse_avg"] < 1.1


def test_k128_statistics_passes_give_the_same_chain(hip_engine_factory, monkeypatch):
    """K = 128 fp32: the column statistics of a half-iteration (c++/sample.cpp:379-384) as a stand-alone pass on the side's
    stream (BPMF_HIP_F32_RIDERS=0: what a sharded or rocprofv3-counter run uses) and as rider workgroups at the head of the
    NEXT k_sample_wg2 launch (the default) sum the same columns in the same order: identical RMSE traces over a pipelined run."""
    import bpmf_amd
    M, Mt, T, Tt, nu, nm = util.ml100k()
    runs = {}
    for name, env in (("alone", {"BPMF_HIP_F32_RIDERS": "0"}), ("head", {"BPMF_HIP_F32_RIDERS": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = hip_engine_factory(K, "f32")
        runs[name] = bpmf_amd.gibbs(eng, M, Mt, T, nu, nm, nsims=6, burnin=2)
    assert runs["head"]["rmse"] == runs["alone"]["rmse"], (runs["head"]["rmse"], runs["alone"]["rmse"])
    assert runs["head"]["final_rmse_avg"] == runs["alone"]["final_rmse_avg"]
