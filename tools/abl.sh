#!/bin/bash
# per-side sampler kernel time under the profiling ablations / chunk sizes (ML-1M-shaped)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp BPMF_HIP_TIMING_EVERY=1
for e in "$@"; do
  echo "== $e: $(env $e python tools/shape_bench.py 32 6040 3706 1000209 20 2>/dev/null | tail -1)"
done
