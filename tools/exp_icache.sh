#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|INST_CACHE|SQC_" | head -40
for w in ml1m_k128 ml1m; do
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU"; do
  rm -rf /tmp/prof_pmc; env BPMF_HIP_F32_RIDERS=0 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_pmc -o p -- python bench.py --workload $w --steps 10 --warmup 2 --repeats 1 --prewarm-ms 0 --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity > /dev/null 2> /tmp/prof_pmc.err
  DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
  echo "== $w: $c"; python tools/pmc_dump.py "$DB" pmc "%k_sample%" 2>&1 | tail -6
done; done
