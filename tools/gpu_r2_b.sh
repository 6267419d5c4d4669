#!/bin/bash
# round 2, second contact: fixed stall test, the new bench.py (prewarm / repeats / strong record / CPU baseline process)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r2b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_waits.py "tests/test_gpu_parity.py::test_sharded_path_over_rccl_single_rank" "tests/test_gpu_parity.py::test_blocking_fallback_paths_give_the_same_chain" -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?"; tail -c 3000 $O/bench_default.json; tail -5 $O/bench_default.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-strong --no-cpu-baseline > $O/bench_20.json 2>&1; python -c "
import json; j=json.loads(open('$O/bench_20.json').read().strip().splitlines()[-1]); print('20-step:', j['value'], j['ms_per_step'], j['ms_per_step_min'], j['ms_per_step_max'], j['ms_per_step_first_block'], j['repeats'], j['prewarm_ms'])"
for w in ml1m_k64 chembl ml1m_k128; do
  timeout 900 python bench.py --workload $w --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err; python -c "
import json; j=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', j['value'], j['ms_per_step'], j['roofline']['launch_ms'], j['roofline']['frac'], j['roofline']['launch_ms_per_side'])"
done
