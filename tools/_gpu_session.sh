cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "product_form or low_rank" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_latent.py tests/test_gpu_scale.py -q -x -k "chembl or light" 2>&1 | tail -5
BPMF_HIP_PF_MERGE=0 bash tools/ab_lib.sh chembl 200 bpmf_amd/csrc/variants/group.so bpmf_amd/libbpmf_hip.so 2>&1 | tee gpurun_out/r4_ab_pfstream.log
