cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
PCMD="python bench.py --workload ml1m_k128 --steps 10 --warmup 2 --repeats 1 --prewarm-ms 0 --no-cpu-baseline --no-strong --no-bpmf-exe"
rm -rf /tmp/prof_pmc; timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace -d /tmp/prof_pmc -o p -- $PCMD > /tmp/pmc.out 2> /tmp/pmc.err; echo rc=$?
tail -c 600 /tmp/pmc.out; echo; tail -15 /tmp/pmc.err | cut -c1-300
find /tmp/prof_pmc -name "*.db" | head
