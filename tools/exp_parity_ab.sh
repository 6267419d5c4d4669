#!/bin/bash
# usage: tools/exp_parity_ab.sh "PYTEST -k EXPRESSION" "WORKLOADS" ROUNDS lib ...   -- the parity tests of the tree, then tools/exp_variants.sh
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q -k "$1" 2>&1 | tail -4
shift
bash tools/exp_variants.sh "$@"
