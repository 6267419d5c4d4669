#!/bin/bash
# PMC counters of k_sample_slab (separate passes; --pmc only with --kernel-trace): usage pmc_slab.sh <workload> <ablate> <tag>
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
W=${1:-ml1m_k64}; AB=${2:-0}; TAG=${3:-slab}
O=gpurun_out/pmc_$TAG; rm -rf $O; mkdir -p $O
run() { # tag, counters
  tag=$1; pmc=$2
  BPMF_HIP_ABLATE=$AB rocprofv3 --pmc $pmc --kernel-trace -d $O/$tag -o r -- python bench.py --workload $W --steps 6 --warmup 2 --repeats 1 --prewarm-ms 0 --no-cpu-baseline --no-strong > $O/$tag.log 2>&1
  DB=$(find $O/$tag -name "*.db" | head -1)
  python tools/pmc_dump.py "$DB" "$TAG" "%k_sample_slab%" >> $O/summary.txt
  rm -rf $O/$tag
}
run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
run b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
run c "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU"
run d "TCC_HIT_sum TCC_MISS_sum"
run e "FETCH_SIZE"
run f "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"
cat $O/summary.txt
