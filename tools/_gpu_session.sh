# scratch: the body of the current gpurun call (rewritten per session; see tools/gpu_r4.sh for the round's named stages)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "normal_stream or tiny_first" 2>&1 | tail -3
