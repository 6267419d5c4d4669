cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "normal_stream or tiny_first or low_rank or product_form" 2>&1 | tail -4
bash tools/ab_lib.sh chembl 200 bpmf_amd/csrc/variants/pretrim.so bpmf_amd/libbpmf_hip.so 2>&1 | tee gpurun_out/r4_ab_trim.log
bash tools/ab_lib.sh ml1m 200 bpmf_amd/csrc/variants/pretrim.so bpmf_amd/libbpmf_hip.so 2>&1 | tee -a gpurun_out/r4_ab_trim.log
