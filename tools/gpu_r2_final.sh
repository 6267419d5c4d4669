#!/bin/bash
# round-2 wrap-up: the whole GPU suite, smoke, profiles of the four workloads, the default bench line (with CPU baseline + strong record)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/final; mkdir -p $O gpurun_out/profiles
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
for w in ml1m ml1m_k64 chembl ml1m_k128; do bash tools/profile_round.sh r02 $w 1 > $O/profile_$w.log 2>&1; done
timeout 900 python bench.py > gpurun_out/profiles/r02_bench.json 2> $O/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-strong > gpurun_out/profiles/r02_bench_20steps.json 2>> $O/bench.err
python - <<'PY'
import json, glob
for f in ["gpurun_out/profiles/r02_bench.json", "gpurun_out/profiles/r02_bench_20steps.json"] + sorted(glob.glob("gpurun_out/profiles/r02_bench_under_rocprof_*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], j["config"]["name"], round(j["value"] / 1e6, 2), "M/s", round(j["ms_per_step"], 4), "ms", j["roofline"]["launch_ms_per_side"], round(j["roofline"]["frac"], 3),
              (j.get("cpu_baseline") or {}).get("value"), (j.get("strong_10Mx1M") or {}).get("value"))
    except Exception as e:
        print(f, "unreadable", e)
PY
