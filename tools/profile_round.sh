#!/bin/bash
# Round profile: rocprofv3 kernel-trace stats of the default bench command + HBM traffic counters
# (separate --pmc passes, as MI355X_MICROARCH.md prescribes).  Output: gpurun_out/profiles/
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R=${1:-r01}
O=gpurun_out/profiles; mkdir -p $O; rm -f $O/${R}_pmc_sampler.txt
CMD="python bench.py --no-cpu-baseline"
rm -rf /tmp/prof_kt; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kt -o kt -- $CMD > $O/${R}_bench_under_rocprof.json 2> /tmp/prof_kt.err
find /tmp/prof_kt -name "*kernel_stats.csv" -exec cp {} $O/${R}_kernel_stats.csv \;
python - <<PY
import csv, glob, collections
f = glob.glob('/tmp/prof_kt/**/*kernel_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
d = collections.defaultdict(list)
for r in rows: d[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
with open('$O/${R}_kernel_trace_summary.txt', 'w') as out:
    out.write("rocprofv3 --kernel-trace --stats -- $CMD   (per-kernel durations from the trace)\n")
    out.write("%-72s %7s %10s %10s %10s %10s\n" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_ms"))
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        out.write("%-72s %7d %10.2f %10.2f %10.2f %10.3f\n" % (k[:72], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, sum(v) / 1e6))
    r0 = [r for r in rows if 'k_sample' in r['Kernel_Name']][0]
    out.write("\nVGPR/SGPR/LDS of the sampler: " + str({k: r0.get(k) for k in ('VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'LDS_Block_Size', 'Scratch_Size')}) + "\n")
print(open('$O/${R}_kernel_trace_summary.txt').read())
PY
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-24)
  rm -rf /tmp/prof_pmc; rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_pmc -o p -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2> /tmp/prof_pmc.err
  DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
  python tools/pmc_dump.py "$DB" pmc >> $O/${R}_pmc_sampler.txt
done
cat $O/${R}_pmc_sampler.txt
tail -1 $O/${R}_bench_under_rocprof.json | cut -c1-600
