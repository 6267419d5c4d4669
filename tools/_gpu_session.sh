# scratch: the body of the current gpurun call (rewritten per session; see tools/gpu_r4.sh for the round's named stages)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_latent.py tests/test_gpu_scale.py tests/test_cli.py -q -x 2>&1 | tail -3
