#!/bin/bash
# PMC counters of the sampler kernel on the bench workload (separate passes, --pmc only with --kernel-trace)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/pmc; rm -rf $O; mkdir -p $O
run() { # tag, counters, cmd...
  tag=$1; pmc=$2; shift 2
  rocprofv3 --pmc $pmc --kernel-trace -d $O/$tag -o r -- "$@" > $O/$tag.log 2>&1
  DB=$(find $O/$tag -name "*.db" | head -1)
  python tools/pmc_dump.py "$DB" "$tag"
  rm -rf $O/$tag
}
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
run bench_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" $B
run bench_b "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" $B
run bench_c "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" $B
