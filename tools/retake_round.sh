#!/bin/bash
# Re-takes the committed measurement set of a round after a kernel-source change (the sha in the headers of profiles/rNN_pmc_* has to
# be the tree's): tools/retake_round.sh <round tag> <stage: a | b | c | d>   -- one stage per gpurun call (each 15 - 25 minutes);
# everything lands in gpurun_out/profiles/ (copy rNN_* to profiles/ afterwards).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R=${1:-r05}; O=gpurun_out/profiles; mkdir -p $O
case "$2" in
  a) for w in ml1m ml1m_k64 chembl; do bash tools/profile_round.sh $R $w 1 > gpurun_out/${R}_prof_$w.log 2>&1; tail -2 gpurun_out/${R}_prof_$w.log | cut -c1-200; done ;;
  b) for w in ml1m_k128 ml1m_k128_f64 strong_10Mx1M; do bash tools/profile_round.sh $R $w 1 > gpurun_out/${R}_prof_$w.log 2>&1; tail -2 gpurun_out/${R}_prof_$w.log | cut -c1-200; done ;;
  c) bash tools/pmc_chembl_by_kernel.sh $R > gpurun_out/${R}_pmc_by_kernel.log 2>&1
     bash tools/phase_budget.sh $R > gpurun_out/${R}_phase_budget.log 2>&1 ;;
  d) # the full bench lines LAST, with the counter files of stages a / b copied to profiles/ (else `profiled.current` is false, `traffic` null)
     python bench.py > $O/${R}_bench.json 2> $O/${R}_bench.err
     python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${R}_bench_20steps.json 2>/dev/null
     for w in ml1m_k64 chembl ml1m_k128 ml1m_k128_f64; do python bench.py --workload $w --steps 20 --warmup 5 --no-strong --no-bpmf-exe > $O/${R}_bench20_full_$w.json 2>/dev/null; done
     for f in $O/${R}_bench.json $O/${R}_bench_20steps.json $O/${R}_bench20_full_*.json; do tail -1 $f | cut -c1-160; done ;;
esac
