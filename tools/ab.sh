#!/bin/bash
# usage: tools/ab.sh "ENV1=.. ENV2=.." "ENV..." ...   -- one short bench per environment setting
cd "$GRAFT_REPO_ROOT"
for e in "$@"; do
  echo "== $e"
  for r in 1 2; do
    env $e timeout 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.4f  sampler %.4f  stats %.4f' % (d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['colstats_ms']))" || true
  done
done
