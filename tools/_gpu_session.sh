cd $GRAFT_REPO_ROOT
BPMF_HIP_STAMPS=1 timeout 300 python bench.py --workload chembl --no-cpu-baseline --no-strong --no-bpmf-exe --steps 10 --warmup 3 --repeats 1 --prewarm-ms 0 2>&1 | grep 'bpmf_hip' | cut -c1-400
