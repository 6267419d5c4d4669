#!/bin/bash
# oracle/build_ref.sh -- ONE command that pins the oracle against a real build of the reference.
#
# TEST INFRASTRUCTURE.  Compiles the reference's own sources where they lie (g++ on
# $BPMF_REFERENCE/c++/*.cpp, default /root/reference; nothing is copied, the reference's cmake is
# not run) for the NO_COMM back-end with the reproducible Random123 RNG, once per num_latent, runs
# it single-threaded on the reference's shipped data and leaves binaries + dumps under oracle/_ref/
# (git-ignored, travels with gpurun):
#     oracle/_ref/bpmf_k8, _k10, _k16, _k32, _k64, _k100, _k128, bpmf_k32_nocov      the reference executables (ci/multilatent.sh:5 sizes
#                                                          the GPU tests lean on; _nocov: -DBPMF_NO_COVARIANCE, CMakeLists.txt:26,93-95)
#     oracle/_ref/out/tiny_k8/{U,V}-<i>.ddm, stdout.txt    data/tiny, `-i 9 -b 0 -v` (data/tiny/run_test.sh)
#     oracle/_ref/out/ml100k_k32/{U,V}-<i>.ddm, stdout.txt data/movielens, `-i 3 -b 1 -v`
#     oracle/_ref/out/ml100k_k{10,16,64,100,128}/...       the same at the other sizes (3 iterations: the per-column arithmetic and the
#                                                          RNG stream ids of every num_latent the padded / fp64-128 kernels are tested at)
#     oracle/_ref/out/ml100k_k32_i20/...                   `-i 20 -b 5 -v`: the chain tests/test_gpu_parity.py::test_full_run_ml100k_matches_oracle uses
#     oracle/_ref/out/ml100k_k32_nocov/...                 the BPMF_NO_COVARIANCE build, `-i 3 -b 1 -v`
# tests/test_oracle_vs_ref.py then diffs every dump and every RMSE line against oracle/bpmf_oracle.c
# at 1e-10 (it skips while oracle/_ref/out is absent).
#
# Needs what the reference's CMakeLists.txt:114-120 needs: Eigen3 and Random123 headers.
#     EIGEN3_INCLUDE_DIR=/path/to/eigen3 RANDOM123_INCLUDE_DIR=/path/to/include oracle/build_ref.sh
# Neither is in this image (no network either): the script then stops with exit code 2 and the
# oracle stays "parity unpinned" (DESIGN.md section 2).  No stand-in headers are written, ever.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${BPMF_REFERENCE:-/root/reference}"
OUT="$HERE/_ref"

find_hdr() {  # $1 = header relative path, rest = candidate dirs
    local h="$1"; shift
    for d in "$@"; do [ -n "$d" ] && [ -f "$d/$h" ] && { echo "$d"; return 0; }; done
    return 1
}
EIGEN=$(find_hdr Eigen/Dense "${EIGEN3_INCLUDE_DIR:-}" /usr/include/eigen3 /usr/local/include/eigen3 /opt/conda/include/eigen3 /usr/include) || {
    echo "build_ref: Eigen3 headers not found (set EIGEN3_INCLUDE_DIR): the reference is unbuildable here" >&2; exit 2; }
R123=$(find_hdr Random123/philox.h "${RANDOM123_INCLUDE_DIR:-}" /usr/include /usr/local/include /opt/conda/include) || {
    echo "build_ref: Random123 headers not found (set RANDOM123_INCLUDE_DIR): the reproducible RNG of c++/mvnormal.cpp:18-23 cannot be built here" >&2; exit 2; }
[ -d "$REF/c++" ] || { echo "build_ref: $REF/c++ not found (set BPMF_REFERENCE)" >&2; exit 2; }

mkdir -p "$OUT/out"
build() {  # $1 = num_latent, $2 = suffix of the binary, rest = extra definitions
    # the definitions of CMakeLists.txt:88-118 for -DBPMF_COMM=NO_COMM -DBPMF_NUMLATENT=$1 with Random123 found;
    # -O2 without -march: no FMA contraction, the arithmetic the oracle restates (oracle/Makefile: -ffp-contract=off)
    local k="$1" sfx="${2:-}"; shift; shift || true
    g++ -std=c++17 -O2 -fopenmp -Wno-int-in-bool-context -DBPMF_NO_COMM -DBPMF_NUMLATENT="$k" -DBPMF_RANDOM123 \
        -DEIGEN_DONT_PARALLELIZE -DBPMF_VERSION='"oracle-pin"' "$@" -I"$REF/c++" -I"$EIGEN" -I"$R123" \
        "$REF"/c++/*.cpp -lz -o "$OUT/bpmf_k$k$sfx"
}
run() {  # $1 = binary, $2 = out dir, rest = arguments
    local bin="$1" dir="$2"; shift 2
    rm -rf "$dir"; mkdir -p "$dir"
    # -t 1: one OpenMP thread, so thread_vector's combine (c++/thread_vector.h:62-101) adds in column order
    "$bin" -t 1 -v -o "$dir" "$@" > "$dir/stdout.txt"
}
ML="-n $REF/data/movielens/ml-train.mtx -p $REF/data/movielens/ml-test.mtx"
build 8
run "$OUT/bpmf_k8" "$OUT/out/tiny_k8" -k -i 9 -b 0 -n "$REF/data/tiny/train.mtx" -p "$REF/data/tiny/test.mtx"
for k in 32 10 16 64 100 128; do      # (32 first: the cases of round 2; then the sizes round 4's padded / fp64-128 kernels are tested at)
    build $k
    run "$OUT/bpmf_k$k" "$OUT/out/ml100k_k$k" -i 3 -b 1 $ML
done
run "$OUT/bpmf_k32" "$OUT/out/ml100k_k32_i20" -i 20 -b 5 $ML
build 32 _nocov -DBPMF_NO_COVARIANCE
run "$OUT/bpmf_k32_nocov" "$OUT/out/ml100k_k32_nocov" -i 3 -b 1 $ML
grep -H "Final Avg RMSE" "$OUT"/out/*/stdout.txt
echo "build_ref: done; now run  python -m pytest tests/test_oracle_vs_ref.py -q"
