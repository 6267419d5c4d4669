#!/bin/bash
# HBM-side counters of the 10M x 1M x 200 configuration on one GPU (the strong_10Mx1M record of bench.py): separate --pmc passes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/profiles; mkdir -p $O; rm -f $O/r02_pmc_strong_10Mx1M.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/prof_pmc; rocprofv3 --pmc $c --kernel-trace -d /tmp/prof_pmc -o p -- python bench.py --steps 5 --warmup 2 --repeats 1 --prewarm-ms 0 --strong-steps 8 --no-cpu-baseline > /tmp/prof_pmc.out 2> /tmp/prof_pmc.err
  DB=$(find /tmp/prof_pmc -name "*.db" | head -1)
  python tools/pmc_dump.py "$DB" strong "%k_sample4%" >> $O/r02_pmc_strong_10Mx1M.txt
done
cat $O/r02_pmc_strong_10Mx1M.txt
tail -c 1200 /tmp/prof_pmc.out
