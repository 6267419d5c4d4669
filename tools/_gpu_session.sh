# scratch: the body of the current gpurun call (rewritten per session; see tools/gpu_r4.sh for the round's named stages)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "low_rank or product_form" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_latent.py -q -x -k "chembl or light" 2>&1 | tail -3
bash tools/ab_lib.sh chembl 200 bpmf_amd/csrc/variants/pretri.so bpmf_amd/libbpmf_hip.so 2>&1 | tee gpurun_out/r4_ab_tri.log
