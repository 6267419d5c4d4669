#!/bin/bash
# Round-3 GPU session script (one gpurun call = one stage list): tools/gpu_r3.sh <stage> [<stage> ...]
# Everything lands under gpurun_out/ (merged back by gpurun); summaries worth keeping are copied to profiles/ by hand.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for stage in "$@"; do
  case "$stage" in
    mr)      timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > gpurun_out/r3_mr.log 2>&1; tail -30 gpurun_out/r3_mr.log ;;
    mrall)   timeout 2400 python -m pytest tests/test_gpu_multirank.py -q -m gpu > gpurun_out/r3_mr.log 2>&1; tail -40 gpurun_out/r3_mr.log ;;
    gpu)     timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r3_gpu.log 2>&1; tail -15 gpurun_out/r3_gpu.log ;;
    smoke)   python __graft_entry__.py smoke > gpurun_out/r3_smoke.log 2>&1; tail -5 gpurun_out/r3_smoke.log ;;
    bench20) python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench20.json 2> gpurun_out/r3_bench20.err; tail -c 3000 gpurun_out/r3_bench20.json; tail -5 gpurun_out/r3_bench20.err ;;
    bench)   python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err; tail -c 3000 gpurun_out/r3_bench.json; tail -5 gpurun_out/r3_bench.err ;;
    benchall) for w in ml1m_k64 chembl ml1m_k128; do python bench.py --workload $w --no-cpu-baseline > gpurun_out/r3_bench_$w.json 2> gpurun_out/r3_bench_$w.err; tail -c 1500 gpurun_out/r3_bench_$w.json; done ;;
    q1)      timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gram1 or four-columns" > gpurun_out/r3_q1.log 2>&1; tail -15 gpurun_out/r3_q1.log
             for m in 1 6; do BPMF_HIP_MODE=$m python bench.py --no-cpu-baseline --no-strong > gpurun_out/r3_bench_mode$m.json 2> gpurun_out/r3_bench_mode$m.err; python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r3_bench_mode$m.json") if l.startswith("{")][-1])
print("mode $m", j["roofline"]["kernel"], "ms/step", j["ms_per_step"], "launch", j["roofline"]["launch_ms_per_side"], "value", j["value"], "rmse", j["rmse"])
PY
             done ;;
    q1ab)    for ab in 1 8 0; do BPMF_HIP_MODE=6 python bench.py --no-cpu-baseline --no-strong --ablate $ab --steps 200 > gpurun_out/r3_q1ab$ab.json 2> gpurun_out/r3_q1ab$ab.err; python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r3_q1ab$ab.json") if l.startswith("{")][-1])
print("mode 6 ablate $ab", j["roofline"]["kernel"], "ms/step", j["ms_per_step"], "launch", j["roofline"]["launch_ms_per_side"])
PY
             done ;;
    q1shapes) for shape in "32 24000 14800 4000000" "32 60400 37060 10000000" "32 300000 100000 20000000" "16 60400 37060 10000000"; do for m in 1 3 6; do
               echo "shape $shape mode $m: $(BPMF_HIP_MODE=$m python tools/shape_bench.py $shape 20 2>&1 | tail -1)"; done; done > gpurun_out/r3_q1shapes.log 2>&1; cat gpurun_out/r3_q1shapes.log ;;
    twin)    timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cli.py -x -q -m gpu -k "twin or rides or cli or g1 or bpmf" > gpurun_out/r3_twin.log 2>&1; tail -5 gpurun_out/r3_twin.log ;;
    inorder) timeout 1500 python -m pytest tests/test_gpu_f32.py tests/test_gpu_fullsize.py tests/test_gpu_waits.py -x -q -m gpu > gpurun_out/r3_inorder.log 2>&1; tail -5 gpurun_out/r3_inorder.log
             for w in ml1m_k128 chembl ml1m_k64; do for io in 1 0; do BPMF_HIP_STATS_INORDER=$io python bench.py --workload $w --no-cpu-baseline --no-strong > gpurun_out/r3_io_${w}_$io.json 2> gpurun_out/r3_io_${w}_$io.err; python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r3_io_${w}_$io.json") if l.startswith("{")][-1])
print("$w inorder=$io", "ms/step", round(j["ms_per_step"],4), "launch", j["roofline"]["launch_ms_per_side"], "value", round(j["value"]), "rmse", j["rmse"])
PY
             done; done
             bash tools/trace_timeline.sh ml1m_k128 > gpurun_out/r3_tl_k128_inorder.txt 2>&1; head -24 gpurun_out/r3_tl_k128_inorder.txt ;;
    twinab)  for w in ml1m ml1m_k64 chembl ml1m_k128; do for fl in "" "--no-users-predict"; do python bench.py --workload $w --no-cpu-baseline --no-strong $fl > gpurun_out/r3_tw.json 2> gpurun_out/r3_tw.err; python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r3_tw.json") if l.startswith("{")][-1])
print("$w '$fl'", "ms/step", round(j["ms_per_step"],4), "launch", {k: round(v,4) for k,v in j["roofline"]["launch_ms_per_side"].items()}, "value", round(j["value"]), "current", j["roofline"]["profiled"]["current"], "traffic", j["roofline"]["traffic"])
PY
             done; done ;;
    riders)  timeout 1500 python -m pytest tests/test_gpu_f32.py tests/test_gpu_fullsize.py tests/test_gpu_waits.py -x -q -m gpu > gpurun_out/r3_riders.log 2>&1; tail -5 gpurun_out/r3_riders.log
             timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank.py -x -q -m gpu -k "128 or fp32 or f32 or parts_single" >> gpurun_out/r3_riders.log 2>&1; tail -3 gpurun_out/r3_riders.log
             for io in 1 0 1 0; do BPMF_HIP_F32_RIDERS=$io python bench.py --workload ml1m_k128 --no-cpu-baseline --no-strong > gpurun_out/r3_rd_$io.json 2> gpurun_out/r3_rd_$io.err; python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r3_rd_$io.json") if l.startswith("{")][-1])
print("ml1m_k128 riders=$io", "ms/step", round(j["ms_per_step"],4), "launch", j["roofline"]["launch_ms_per_side"], "value", round(j["value"]), "rmse", j["rmse"])
PY
             done
             bash tools/trace_timeline.sh ml1m_k128 > gpurun_out/r3_tl_k128_riders.txt 2>&1; head -22 gpurun_out/r3_tl_k128_riders.txt ;;
    m3)      # k_sample4 on the ML-1M shape: chunk sizes and phase switches (is it slots / latency or instructions?)
             pr() { python - "$1" "$2" <<PY
import json, sys
try:
    j=json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print(sys.argv[1], j["roofline"]["kernel"], "ms/step", round(j["ms_per_step"],4), "launch", {k: round(v*1e3,1) for k,v in j["roofline"]["launch_ms_per_side"].items()}, "rmse", round(j["rmse"],4))
except Exception as e: print(sys.argv[1], "failed", e)
PY
             }
             for cfg in "BPMF_HIP_MODE=1" "BPMF_HIP_MODE=1 BPMF_HIP_FUSED=0" "BPMF_HIP_MODE=3" "BPMF_HIP_MODE=3 BPMF_HIP_CHUNK=64" "BPMF_HIP_MODE=3 BPMF_HIP_CHUNK=96" "BPMF_HIP_MODE=3 BPMF_HIP_CHUNK=128" "BPMF_HIP_MODE=3 BPMF_HIP_CHUNK=256" "BPMF_HIP_MODE=3 BPMF_HIP_CHUNK=640"; do
               env $cfg python bench.py --no-cpu-baseline --no-strong --steps 200 > gpurun_out/r3_m3.json 2> gpurun_out/r3_m3.err; pr "$cfg" gpurun_out/r3_m3.json; done
             for ab in 1 2 3; do for cfg in "BPMF_HIP_MODE=1 BPMF_HIP_FUSED=0" "BPMF_HIP_MODE=3" "BPMF_HIP_MODE=3 BPMF_HIP_CHUNK=96"; do
               env $cfg python bench.py --no-cpu-baseline --no-strong --steps 200 --ablate $ab > gpurun_out/r3_m3.json 2> gpurun_out/r3_m3.err; pr "ablate=$ab $cfg" gpurun_out/r3_m3.json; done; done ;;
    x4)      # k_sample1x (mode 7): parity, then A/B against the committed build and mode 1
             timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four-in-a-row or per-item or auto" > gpurun_out/r3_x4.log 2>&1; tail -8 gpurun_out/r3_x4.log
             pr() { python - "$1" "$2" <<PY
import json, sys
try:
    j=json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print(sys.argv[1], j["roofline"]["kernel"], "ms/step", round(j["ms_per_step"],4), "launch", {k: round(v*1e3,1) for k,v in j["roofline"]["launch_ms_per_side"].items()}, "rmse", round(j["rmse"],4), "value", round(j["value"]/1e6,1))
except Exception as e: print(sys.argv[1], "failed", e)
PY
             }
             for rep in 1 2; do for cfg in "BPMF_HIP_LIBRARY=$PWD/bpmf_amd/csrc/variants/base.so" "BPMF_HIP_MODE=1" "BPMF_HIP_MODE=7" "BPMF_HIP_MODE=7 BPMF_HIP_CHUNK=768" "BPMF_HIP_MODE=7 BPMF_HIP_CHUNK=1280" "BPMF_HIP_MODE=7 BPMF_HIP_CHUNK=1536" "BPMF_HIP_MODE=7 BPMF_HIP_CHUNK=2048" "BPMF_HIP_MODE=7 BPMF_HIP_X4_WAVES=1792" "BPMF_HIP_MODE=7 BPMF_HIP_X4_WAVES=2560"; do
               env $cfg python bench.py --no-cpu-baseline --no-strong --steps 200 > gpurun_out/r3_x4.json 2> gpurun_out/r3_x4.err; pr "$cfg" gpurun_out/r3_x4.json; done; done
             for ab in 1 2 3; do for cfg in "BPMF_HIP_MODE=7"; do
               env $cfg python bench.py --no-cpu-baseline --no-strong --steps 200 --ablate $ab > gpurun_out/r3_x4.json 2> gpurun_out/r3_x4.err; pr "ablate=$ab $cfg" gpurun_out/r3_x4.json; done; done ;;
    *) echo "unknown stage $stage" ;;
  esac
done
