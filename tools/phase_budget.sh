#!/bin/bash
# Per-phase instruction budget of the headline kernel k_sample1<32>: measured per-launch counters of the
# full kernel and of its ablations (BPMF_HIP_ABLATE: 1 = everything after the Gram skipped, 2 = the Gram skipped, 3 = both),
# then the static per-phase ISA budget of one work item (tools/isa_phases.py).   tools/phase_budget.sh <round tag>
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R=${1:-r05}; O=gpurun_out/profiles; mkdir -p $O; F=$O/${R}_phase_budget_ml1m.txt
echo "# kernel-source-sha: $(python -c 'import bench; print(bench.kernel_source_sha())')" > $F
echo "# k_sample1<32>, ML-1M shape (6040 x 3706 x 900 188 train ratings), per LAUNCH (= half-iteration, average of the two sides): counters of" >> $F
echo "# rocprofv3 --pmc <group> --kernel-trace -- python bench.py --workload ml1m --steps 10 --warmup 2 --repeats 1 --prewarm-ms 0 --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity [--ablate N]" >> $F
for ab in 0 1 2 3; do
  AB=""; [ $ab != 0 ] && AB="--ablate $ab"
  case $ab in 0) echo "## full kernel" >> $F;; 1) echo "## --ablate 1: Gram only (assembly, factorisation, solves skipped; the normal draw still runs)" >> $F;; 2) echo "## --ablate 2: no Gram (zero ratings per item): draw + assembly + factorisation + solves" >> $F;; 3) echo "## --ablate 3: neither (item prologue + draw + store)" >> $F;; esac
  for c in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
    rm -rf /tmp/prof_pmc; rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_pmc -o p -- python bench.py --workload ml1m --steps 10 --warmup 2 --repeats 1 --prewarm-ms 0 --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity $AB > /dev/null 2> /tmp/prof_pmc.err
    DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
    python tools/pmc_dump.py "$DB" pmc "%k_sample1%" >> $F
  done
  python bench.py --workload ml1m --steps 200 --warmup 20 --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity $AB 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('launch (HIP events, 200-step blocks): %.2f us  %s' % (r['launch_ms']*1e3, {k: round(v*1e3,2) for k,v in r['launch_ms_per_side'].items()}))" >> $F
done
echo >> $F
echo "## static ISA budget of one work item (tools/isa_phases.py on the built bpmf_amd/csrc/k32.o; RATINGS=160: a mid-size whole column)" >> $F
RATINGS=160 python tools/isa_phases.py >> $F
cat $F
