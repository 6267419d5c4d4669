#!/bin/bash
# phase stamps (wave 0, 100 MHz ticks) of two probe items of the last K = 128 launch, full kernel and factorisation only;
# STAMP_LIBS="name ..." adds builds under bpmf_amd/csrc/variants/ beside the tree's
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for wl in ml1m_k128_f64 ml1m_k128; do
 for lib in tree ${STAMP_LIBS:-}; do
  for ab in 0 2; do
    E=""; [ $lib != tree ] && E="BPMF_HIP_LIBRARY=$PWD/bpmf_amd/csrc/variants/$lib.so"
    echo "== $wl $lib ablate=$ab"
    env $E BPMF_HIP_STAMPS=1 BPMF_HIP_F32_RIDERS=0 timeout 300 python bench.py --workload $wl --no-cpu-baseline --no-strong --no-bpmf-exe --no-parity --steps 10 --warmup 2 --repeats 1 --prewarm-ms 0 --ablate $ab 2>&1 | grep "bpmf_hip\] \(stamps\|all\)"
  done
 done
done
