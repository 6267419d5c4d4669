cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_f32.py tests/test_gpu_latent.py tests/test_cli.py -q -x 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -x -k "128" 2>&1 | tail -5
bash tools/sweep.sh ml1m_k128 "X=1" 2>&1 | tee gpurun_out/r4_k128_default.log
