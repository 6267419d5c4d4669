"""Coupled Gibbs CHAINS at BASELINE.json's own single-GPU shapes against the oracle's chain on identical seeds
(main()'s loop, /root/reference c++/bpmf.cpp:180-253; Sys::sample(Sys&) c++/sample.cpp:341-385; Sys::predict :48-96).

tests/test_gpu_fullsize.py compares single half-iterations from random factors; here the STATEFUL pipeline that bench.py
times -- default schedule, fused launches with the gate and the statistics riders, the product-form classes + k_pf_prepare
on the ChEMBL shape, the twin evaluation beside the next samplers, both copies of the factors, the software-pipelined loop
of `bpmf_amd.gibbs(..., pipelined=True)` = the loop of the `bpmf` executable -- runs several iterations from the
reference's start (zero factors, iter = -1) and must stay on the oracle's chain:

  * configs[1]  ML-1M shape, K = 32 fp64, -i 20 -b 5           RMSE traces 1e-6, factors 1e-6 max|U|, norms 1e-7
  * configs[2]  ChEMBL shape, K = 64 fp64, -i 6 -b 2           same
  * ML-1M shape, K = 128 fp64 (what -d 128 means), -i 6 -b 2   same
  * configs[4]  ML-1M shape, K = 128 fp32 opt-in, -i 6 -b 2    RMSE traces 1e-3 (the north star's bar), factors 2e-3 max|U|
  * the form of configs[3]'s ranks: 110 000 x 24 000 x 5 M ratings, K = 32, -i 6 -b 2 (k_sample4, unfused launches)

No sampler-mode override anywhere in this file: what runs is what `bpmf` and bench.py run.
"""
import os

import numpy as np
import pytest

from tests.test_gpu_parity import rel_err

pytestmark = pytest.mark.gpu

NT = max(1, min(os.cpu_count() or 1, 32))
_ref_cache = {}


_data_cache = {}


def _data(shape):
    from bpmf_amd import synth
    if shape not in _data_cache:
        _data_cache.clear()                                      # (one matrix at a time: the ChEMBL shape is 250 MB of factors per copy)
        if shape == "ml1m":
            _data_cache[shape] = synth.ml1m_shaped(seed=42)      # the matrix bench.py times
        elif shape == "large":
            _data_cache[shape] = synth.ratings(110_000, 24_000, 5_000_000, seed=5)
        else:
            _data_cache[shape] = synth.ratings(483500, 5775, 1_023_952, seed=42, real_valued=True)
    return _data_cache[shape]


def _oracle_chain(oracle, shape, K, nsims, burnin):
    key = (shape, K, nsims, burnin)
    if key not in _ref_cache:
        M, Mt, T, Tt, nu, nm = _data(shape)
        _ref_cache[key] = oracle.gibbs(K, M, Mt, T, Tt, nsims=nsims, burnin=burnin, nthreads=NT)
    return _ref_cache[key]


def _chain(oracle, shape, K, nsims, burnin, dtype="f64", rmse_tol=1e-6, item_tol=1e-6, norm_tol=1e-7):
    import bpmf_amd
    M, Mt, T, Tt, nu, nm = _data(shape)
    ref = _oracle_chain(oracle, shape, K, nsims, burnin)
    eng = bpmf_amd.HipEngine(K, dtype=dtype)                    # a context of its own: default environment, default schedule
    try:
        res = bpmf_amd.gibbs(eng, M, Mt, T, nu, nm, nsims=nsims, burnin=burnin, Tt=Tt, pipelined=True)
        names = (eng.kernel_name(res["movies"].side), eng.kernel_name(res["users"].side))
    finally:
        eng.close()
    assert len(res["rmse"]) == nsims and np.all(np.isfinite(res["rmse"]))
    d_rmse = float(np.abs(np.array(res["rmse"]) - ref["rmse"]).max())
    d_avg = float(np.abs(np.array(res["rmse_avg"]) - ref["rmse_avg"]).max())
    d_final = abs(res["final_rmse_avg"] - ref["final_rmse_avg"])
    d_norm = max(float(np.abs(np.array(res["norm_u"]) / ref["norm_u"] - 1).max()), float(np.abs(np.array(res["norm_m"]) / ref["norm_m"] - 1).max()))
    eu, ev = rel_err(res["U"], ref["U"]), rel_err(res["V"], ref["V"])
    print("%s K=%d %s -i %d -b %d: kernels %s | %s; max|dRMSE| %.2e, max|d avg RMSE| %.2e, d Final Avg RMSE %.2e, norms %.2e, U %.2e, V %.2e"
          % (shape, K, dtype, nsims, burnin, names[0], names[1], d_rmse, d_avg, d_final, d_norm, eu, ev))
    assert res["num_predict"] == ref["num_predict"]
    assert d_rmse < rmse_tol and d_avg < rmse_tol and d_final < rmse_tol
    assert d_norm < norm_tol
    assert eu < item_tol and ev < item_tol
    return res, ref


def test_chain_ml1m_k32(oracle):
    """BASELINE configs[1], the headline: the default run `-i 20 -b 5` (c++/bpmf.cpp:30-31 defaults)."""
    res, ref = _chain(oracle, "ml1m", 32, 20, 5)
    assert res["rmse_avg"][-1] < res["rmse"][0]                 # the averaged predictor beats the mean predictor of iteration 0


def test_chain_chembl_k64(oracle):
    """BASELINE configs[2]: the compounds side in the product form (three classes + k_pf_prepare per half-iteration + the slab
    form for the heavier columns), the targets side in k_sample1s, chained through the hyper-parameter draws."""
    _chain(oracle, "chembl", 64, 6, 2)


def test_chain_large_sides_k32(oracle):
    """Sides of >= 20 000 columns (the form BASELINE configs[3] runs on every rank): k_sample4 -- four columns per wave, the
    factorisation on the MFMA in lockstep -- behind the UNFUSED stateful launches (gate kernel on the side's stream, sampler,
    stand-alone statistics pass: k_colstats_wg for the 110 000-column side), 5 M ratings, chained through the hyper-parameter
    draws for six iterations."""
    _chain(oracle, "large", 32, 6, 2)


def test_chain_ml1m_k128_fp64(oracle):
    _chain(oracle, "ml1m", 128, 6, 2)


def test_chain_ml1m_k128_fp32(oracle):
    """BASELINE configs[4] (mixed-precision tolerance study): fp32 factors / Gram / factorisation against the fp64 chain."""
    _chain(oracle, "ml1m", 128, 6, 2, dtype="f32", rmse_tol=1e-3, item_tol=2e-3, norm_tol=1e-3)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_k128_chains_are_bit_reproducible(dtype):
    """The K = 128 workgroups hand operands and tiles over through LDS (staged gathers, the look-ahead factorisation's three
    barriers per step, one of them inside wave 0's diagonal block): a missing barrier shows as bits that differ from run to
    run.  The same `-i 4 -b 1` chain in six fresh engines must give identical factors."""
    import hashlib
    import bpmf_amd
    M, Mt, T, Tt, nu, nm = _data("ml1m")
    seen = set()
    for _ in range(6):
        eng = bpmf_amd.HipEngine(128, dtype=dtype)
        try:
            res = bpmf_amd.gibbs(eng, M, Mt, T, nu, nm, nsims=4, burnin=1, Tt=Tt, pipelined=True)
        finally:
            eng.close()
        seen.add(hashlib.sha1(np.ascontiguousarray(res["U"]).tobytes() + np.ascontiguousarray(res["V"]).tobytes()).hexdigest())
    assert len(seen) == 1, seen
