#!/bin/bash
# K = 128 round 5: parity of the scratch form, then A/B against the previous build (variants/base.so)
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_f32.py tests/test_gpu_latent.py "tests/test_gpu_fullsize.py::test_ml1m_k128_f32_every_column_within_2e3_of_the_fp64_oracle" "tests/test_gpu_fullsize.py::test_ml1m_k128_fp64_every_column_matches_the_oracle" tests/test_gpu_chain.py::test_chain_ml1m_k128_fp32 tests/test_gpu_chain.py::test_chain_ml1m_k128_fp64 -x -q 2>&1 | tail -5
line() { grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1  ms/step %.4f ' % d['ms_per_step'], {k: round(v*1e3,1) for k,v in d['roofline']['launch_ms_per_side'].items()}, 'frac %.3f' % d['roofline']['frac'])"; }
for r in 1 2; do for w in ml1m_k128 ml1m_k128_f64; do
  python bench.py --workload $w --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 50 --warmup 10 2>/dev/null | line "$w new "
  BPMF_HIP_LIBRARY=$PWD/bpmf_amd/csrc/variants/base.so python bench.py --workload $w --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 50 --warmup 10 2>/dev/null | line "$w base"
done; done
