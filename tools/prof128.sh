#!/bin/bash
# kernel trace + a few counters of the fp32 K=128 sampler on the ML-1M shape
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/p128; rm -rf $O; mkdir -p $O
CMD="python tools/shape_bench.py 128 6040 3706 1000209 4"
rocprofv3 --kernel-trace -d $O/kt -o r -- $CMD > $O/kt.log 2>&1
python tools/kstats.py $(find $O/kt -name "*.db" | head -1) | head -8
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_BRANCH"; do
  rm -rf $O/pmc; rocprofv3 --pmc $c --kernel-trace -d $O/pmc -o p -- $CMD > /dev/null 2>&1
  python tools/pmc_dump.py "$(find $O/pmc -name '*.db' | head -1)" pmc "%k_sample_wg%"
done
rm -rf $O/kt $O/pmc
