#!/bin/bash
# Round profile: rocprofv3 kernel-trace stats of the bench command of one workload + PMC passes (separate --pmc
# passes with --kernel-trace only, as MI355X_MICROARCH.md prescribes).  Output: gpurun_out/profiles/ (copy to profiles/).
#   tools/profile_round.sh <round tag, e.g. r02> <workload: ml1m | ml1m_k64 | chembl | ml1m_k128> [pmc: 1|0]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R=${1:-r02}; W=${2:-ml1m}; PMC=${3:-1}
O=gpurun_out/profiles; mkdir -p $O; rm -f $O/${R}_pmc_$W.txt
# the kernel sources these counters belong to (bench.py reports figures of a profile of OTHER sources as stale)
echo "# kernel-source-sha: $(python -c 'import bench; print(bench.kernel_source_sha())')" > $O/${R}_pmc_$W.txt
echo "# rocprofv3 --pmc <one group per pass> --kernel-trace -- python bench.py --workload $W --steps 10 --warmup 2 --repeats 1 --prewarm-ms 0 --no-cpu-baseline --no-strong   (per-launch averages of the sampler kernels)" >> $O/${R}_pmc_$W.txt
CMD="python bench.py --workload $W --no-cpu-baseline --no-strong --no-bpmf-exe"
PCMD="python bench.py --workload $W --steps 10 --warmup 2 --repeats 1 --prewarm-ms 0 --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity"
PAT="%k_sample%"
PENV=""
if [ "$W" = "ml1m_k128" ]; then
  # counter collection serialises kernels: the fp32 statistics riders (default since round 4) need the partner's sampler and this
  # side's gate kernel in flight together and the bounded gate wait gives up (BPMF_HIP_ENODEV) -- the counter passes take the
  # stand-alone statistics pass instead; the sampler's own workgroups (what the counters are read for) are the same
  PENV="env BPMF_HIP_F32_RIDERS=0"
  echo "# counter passes with BPMF_HIP_F32_RIDERS=0 (kernel serialisation under --pmc; see tools/profile_round.sh)" >> $O/${R}_pmc_$W.txt
fi
if [ "$W" = "strong_10Mx1M" ]; then      # the strong-scaling record of the default workload (k_sample4<32> over the device-generated 10M x 1M matrix)
  CMD="python bench.py --steps 5 --warmup 2 --repeats 1 --prewarm-ms 0 --no-cpu-baseline --no-bpmf-exe --strong-steps 8"
  PCMD="$CMD"; PAT="%k_sample4%"
fi
rm -rf /tmp/prof_kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- $CMD > $O/${R}_bench_under_rocprof_$W.json 2> /tmp/prof_kt.err
find /tmp/prof_kt -name "*kernel_stats.csv" -exec cp {} $O/${R}_kernel_stats_$W.csv \;
python - <<PY
import csv, glob, collections
f = glob.glob('/tmp/prof_kt/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
d = collections.defaultdict(list)
for r in rows: d[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
with open('$O/${R}_kernel_trace_summary_$W.txt', 'w') as out:
    out.write("rocprofv3 --kernel-trace --stats -- $CMD   (per-kernel durations from the trace)\n")
    out.write("%-72s %7s %10s %10s %10s %10s\n" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_ms"))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        out.write("%-72s %7d %10.2f %10.2f %10.2f %10.3f\n" % (k[:72], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, sum(v) / 1e6))
    seen = set()
    for r0 in rows:
        k = r0['Kernel_Name']
        if 'k_sample' in k and k not in seen:
            seen.add(k)
            out.write("\n%s: %s" % (k[:60], {q: r0.get(q) for q in ('VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'LDS_Block_Size', 'Scratch_Size')}))
    out.write("\n")
print(open('$O/${R}_kernel_trace_summary_$W.txt').read())
PY
if [ "$PMC" = "1" ]; then
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"; do
  rm -rf /tmp/prof_pmc; rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_pmc -o p -- $PENV $PCMD > /dev/null 2> /tmp/prof_pmc.err
  DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
  BYG=""; [ "$W" = "strong_10Mx1M" ] && BYG="bygrid"
  python tools/pmc_dump.py "$DB" pmc "$PAT" $BYG >> $O/${R}_pmc_$W.txt
done
cat $O/${R}_pmc_$W.txt
fi
# the 20-step line AFTER the counter passes, with their file in place: `profiled.current` is true, `traffic` is this source's
if [ "$PMC" = "1" ]; then cp $O/${R}_pmc_$W.txt profiles/; fi
if [ "$W" = "strong_10Mx1M" ]; then python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bpmf-exe > $O/${R}_bench20_$W.json 2>/dev/null
else python bench.py --workload $W --steps 20 --warmup 5 --no-strong --no-cpu-baseline --no-bpmf-exe > $O/${R}_bench20_$W.json 2>/dev/null; fi
tail -1 $O/${R}_bench_under_rocprof_$W.json | cut -c1-400
