cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/ab_lib.sh ml1m_k64 200 bpmf_amd/csrc/variants/wps2.so bpmf_amd/libbpmf_hip.so 2>&1 | tee gpurun_out/r4_ab_wps3.log
bash tools/ab_lib.sh chembl 200 bpmf_amd/csrc/variants/wps2.so bpmf_amd/libbpmf_hip.so 2>&1 | tee -a gpurun_out/r4_ab_wps3.log
