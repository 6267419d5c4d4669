#!/bin/bash
# host Normal-Wishart draw (bpmf_hyper_finish) built for different x86 levels, timed on the GPU box's CPU:
#   gpurun -- 'bash tools/hyper_ab.sh'
cd "$GRAFT_REPO_ROOT"; T=/tmp/hab; mkdir -p $T
grep -m1 'model name' /proc/cpuinfo
F="-std=c++17 -O3 -fPIC -pthread -ffp-contract=off -Iinclude"
for v in x86-64-v3 x86-64-v4 znver4 znver5; do
  /opt/rocm/bin/hipcc $F -march=$v -c bpmf_amd/csrc/hyper.cpp -o $T/h_$v.o 2>/dev/null || { echo "$v: not supported"; continue; }
  g++ -O3 tools/probes/hyper_bench.cpp $T/h_$v.o -o $T/hb_$v -lpthread || continue
  for K in 32 64 128; do echo -n "$v  "; $T/hb_$v $K; done
done
/opt/rocm/bin/hipcc $F -march=x86-64-v4 -mprefer-vector-width=512 -c bpmf_amd/csrc/hyper.cpp -o $T/h_v4w.o && g++ -O3 tools/probes/hyper_bench.cpp $T/h_v4w.o -o $T/hb_v4w -lpthread && for K in 32 64 128; do echo -n "v4+512  "; $T/hb_v4w $K; done
