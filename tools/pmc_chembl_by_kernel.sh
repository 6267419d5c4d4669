#!/bin/bash
# per-KERNEL counters of the ChEMBL-shaped workload (compounds side: k_pf_prepare + k_sample_pf<64, 3 | 6 | 16> + k_sample_slab; targets side: k_sample1s)
#   tools/pmc_chembl_by_kernel.sh <round tag>
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R=${1:-r05}; O=gpurun_out/profiles; mkdir -p $O; F=$O/${R}_pmc_by_kernel_chembl.txt
echo "# kernel-source-sha: $(python -c 'import bench; print(bench.kernel_source_sha())')" > $F
echo "# rocprofv3 --pmc <group> --kernel-trace -- python bench.py --workload chembl --steps 10 --warmup 2 ...   (per-kernel, per-launch averages; tools/pmc_by_kernel.py)" >> $F
PCMD="python bench.py --workload chembl --steps 10 --warmup 2 --repeats 1 --prewarm-ms 0 --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity"
for c in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  rm -rf /tmp/prof_pmc; rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_pmc -o p -- $PCMD > /dev/null 2> /tmp/prof_pmc.err
  DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
  echo "## --pmc $c" >> $F
  python tools/pmc_by_kernel.py "$DB" "%k_%" >> $F
done
cat $F
