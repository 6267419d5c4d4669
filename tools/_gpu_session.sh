cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_f32.py tests/test_cli.py -q -x 2>&1 | tail -3
bash tools/ab_lib.sh ml1m_k128 200 bpmf_amd/csrc/variants/pretwin.so bpmf_amd/libbpmf_hip.so 2>&1 | tee gpurun_out/r4_ab_twin32.log
