cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
W=${1:-chembl}
PCMD="python bench.py --workload $W --steps 10 --warmup 2 --repeats 1 --prewarm-ms 0 --no-cpu-baseline --no-strong --no-bpmf-exe"
: > gpurun_out/r4_pmc_by_kernel_$W.txt
for c in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"; do
  rm -rf /tmp/prof_pmc; rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_pmc -o p -- $PCMD > /dev/null 2> /tmp/prof_pmc.err
  DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
  python tools/pmc_by_kernel.py "$DB" "%k_sample%" >> gpurun_out/r4_pmc_by_kernel_$W.txt
done
cat gpurun_out/r4_pmc_by_kernel_$W.txt
