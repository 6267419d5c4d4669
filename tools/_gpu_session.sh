cd $GRAFT_REPO_ROOT
for e in "BPMF_HIP_TAIL_NOWT=0" "BPMF_HIP_TAIL_NOWT=1"; do echo "== $e"; env $e BPMF_HIP_TAIL_STATS=1 timeout 300 python bench.py --workload ml1m_k128_f64 --no-cpu-baseline --no-strong --no-bpmf-exe --steps 40 --warmup 5 --repeats 1 --prewarm-ms 0 2>&1 | grep 'GO' | head -4; done
