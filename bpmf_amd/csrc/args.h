// args.h -- argument blocks of the sampler kernels (plain data: shared by the host side, which
// fills them, and kernels.h, which reads them) and the compile-time geometry both sides size buffers with.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bpmf {

// offset of row i in the packed lower-triangular row-major storage of L: row i keeps its
// i+1 entries padded to an even count (so that a pair of columns 2p, 2p+1 is 16-byte aligned)
__host__ __device__ constexpr int tri_off(int i) { return 2 * ((i >> 1) + 1) * ((i >> 1) + (i & 1)); }

template <int K>
struct Geo {
    static constexpr int NT = (K + 15) / 16;             // 16-wide tiles per dimension (K=8 is zero-padded)
    static constexpr int NTRI = NT * (NT + 1) / 2;        // upper-triangular tiles incl. diagonal
    static constexpr int PART = NTRI * 256 + NT * 16;     // doubles in one partial: tiles in accumulator layout + rhs
    // waves per SIMD the sampler is compiled for (bounds the VGPR budget: 512 / WPS)
    static constexpr int WPS = K <= 32 ? 4 : 2;
    // factorisation: one lane per row of Lambda*, C = 64/K columns side by side in one wave
    static constexpr int C = 64 / K;
    static constexpr int NP = K / 2;                      // column pairs = steps of the factorisation
    static constexpr int PLEN = tri_off(K);               // packed L: 544 doubles at K=32
    static constexpr int SLOT = PLEN + 2 * K;             // per column: L | rhs, later y [K] | normals [K]
    static constexpr int LDS_WORDS = C * SLOT + 2 * C;    // + the global column id of each slot
};

// Gram of the one-column-per-wave sampler (K <= 32) on the 4x4x4 shape.  The four blocks of an
// instruction take four DIFFERENT ratings each (16 ratings per instruction) and the same 4x4
// block (g, g') of the Gram; lane (k, b, x) feeds rating slot s = 4 k + b with the latent index
// idx(g, x) = 8 (g / 2) + 2 x + (g & 1)  (so that one 16-byte load per lane brings the operands of
// two groups).  The NB = NG (NG + 1) / 2 upper blocks are NB accumulator registers per lane, each
// holding the contribution of the lane's b; the four b are added once per column (DPP row rotates).
template <int K>
struct Geo44 {
    static constexpr int NG = K / 4;                      // groups of 4 latent indices
    static constexpr int NB = NG * (NG + 1) / 2;          // upper blocks incl. diagonal
    static constexpr int NL = K / 8;                      // 16-byte loads per lane and 16 ratings
    static constexpr int PART = (NB + NG) * 64;           // doubles in one partial of a chunked column
    __host__ __device__ static constexpr int idx(int g, int x) { return 8 * (g >> 1) + 2 * x + (g & 1); }
};

struct SampleArgs {
    // ratings of this rank's columns
    const int32_t *rowidx;
    const double *vals;
    // static schedule of the side: work item = (column, chunk of its ratings)
    const int32_t *wi_col;      // local column
    const int64_t *wi_p0;       // first rating of the chunk
    const int32_t *wi_len;      // ratings in the chunk
    const int32_t *wi_mc;       // heavy column index the chunk belongs to, or -1 (whole column)
    const int32_t *wi_chunk;    // ordinal of the chunk inside its column
    const int32_t *mc_slot0;    // heavy columns: first partial slot, number of chunks
    const int32_t *mc_nchunks;
    unsigned *mc_count;         // arrival counters of the heavy columns (zero between launches)
    double *partials;
    int nwork;
    // factors
    const double *other_items;  // K x nrows
    const double *zero_row;     // K zeros (gather target of the padding slots of a ragged group of ratings)
    double *items;              // K x ncols
    int64_t col_from;           // global id of local column 0
    // per-call
    const double *LambdaF;      // K x K col-major (device)
    const double *Lmu;          // LambdaF * mu (device)
    const double *mu;           // hp.mu (device)
    const float *lf32;          // fp32 path: LambdaF as fp32 in tile layout (k_lf32_tiles), or NULL
    // propagated posterior (-m / -l, c++/sample.cpp:152-174,272-277): one K x K col-major prior
    // precision per LOCAL column replaces LambdaF; rr = Lambda_i * hp.mu keeps the global mu (Q2)
    const double *prop_lambda;
    uint32_t diag_only;         // BPMF_NO_COVARIANCE build of the reference (c++/sample.cpp:300-304): keep only the diagonal of Lambda*
    unsigned long long *fail;   // min global column id whose factorisation failed
    double mean_rating;
    double alpha;
    uint32_t iter_plus_1;
    int ktrue;                  // the caller's num_latent (<= the K the kernel is instantiated for): RNG stream id and number of normals per column
    // in-kernel gate (NULL: the launch itself was ordered behind the staging kernel): the word
    // k_gate_stage sets to `gate_want` once LambdaF | Lmu | fail | mu are in device memory
    const unsigned *gate_flag;
    unsigned gate_want;
    // every in-kernel wait is bounded: after `wait_ticks` of the 100 MHz wall clock a waiter gives up,
    // stores a non-zero code in `tmo` (a sticky word of the side's pinned result blob) and goes on; the
    // host turns that into BPMF_HIP_ENODEV "device wait timed out" and ends the chain
    unsigned long long *tmo;
    unsigned long long wait_ticks;
    unsigned long long *stamps; // profiling only (BPMF_HIP_STAMPS=1): s_memtime at phase boundaries of two probe items, or NULL
    uint32_t ablate;            // profiling only (BPMF_HIP_ABLATE): 1 = skip the factorisation, 2 = skip the Gram, 4 = gather from 64 hot rows only
};

// What else one k_sample1 launch carries besides its work items (see k_sample1 in kernels.h): the gate +
// staging of its own parameters (workgroup 0) and the column statistics of the previous launch's side.
struct FusedArgs {
    const unsigned *gate_host; unsigned gate_want; const double *src_host; double *dst; int n; unsigned *dflag; unsigned dval;
    int nstat; const double *st_items; int64_t st_c0, st_c1; double *st_partials; const unsigned long long *st_fail;
    double *st_out; unsigned *st_ticket; unsigned *st_flag; unsigned st_seq;
    unsigned long long *st_tmo;        // sticky time-out word of the side the statistics belong to
};

// users.predict(movies) fused into movies.predict(users) (k_predict): the same prediction goes into the other side's copy
// of the test entries as well -- entry q of this test matrix is entry perm[q] of the twin (its transpose).  perm = NULL: none.
struct TwinArgs {
    const int32_t *perm;
    double *pavg, *pm2;         // the twin's running mean / M2 (its own entry order)
    double mean;                // the twin side's mean_rating (c++/sample.cpp:78 adds the predicting Sys's own)
    double *partial;            // block partials of the twin's two sums
    double *out;                // the twin's pinned se | se_avg | flag
    unsigned *flag; unsigned seq;
};

// K = 128 fp32: column statistics as rider workgroups of a k_sample_wg2 launch (colstats_f32_rider, kernels_f32.h) -- what
// FusedArgs' st_* fields are to k_sample1: the statistics of the PREVIOUS launch's side as the FIRST workgroups of the grid.
// nblocks = 0: none.
struct StatRiders {
    int nblocks;                       // rider workgroups
    const void *items; int64_t c0, c1; int nsl;      // the side's factors in its own type (float | double)
    double *partials; const unsigned long long *fail_in; double *out;
    unsigned *ticket; unsigned *flag; unsigned seq;
    unsigned long long *tmo; unsigned long long wait_ticks;
};


// columns with a handful of ratings at K = 64 (kernels_lr.h)
struct LrArgs {
    const int32_t *rowidx; const double *vals;
    const int32_t *col; const int64_t *p0; const int32_t *len;   // light work items (len <= NLR)
    int nitems;
    const double *other_items; double *items; int64_t col_from;
    const double *R0;          // K x K, row-major upper factor of LambdaF (zeros below the diagonal)
    const double *S0t;         // K x K: S0t[j * K + i] = (R0^-1)[i][j] -- columns without ratings: x = R0^-1 (y0 + z)
    const double *y0;          // K: R0^-T (LambdaF mu), the forward solve every such column would repeat
    const double *Q;           // nrows x K: R0^-T u_row for every row of the other side (k_pf_prepare), product form only
    const double *Lmu;         // LambdaF * mu
    unsigned long long *fail;
    double mean_rating, alpha, sqrt_alpha;
    uint32_t iter_plus_1;
    int ktrue;                 // (see SampleArgs)
};

// BPMF_REDUCE formulation (kernels_reduce.h): the pass that computes the other side's precomputed Gram parts
struct PrecArgs {
    const int64_t *t_colptr;    // ncols_O + 1: ratings of O's column j that sit in this rank's columns of S
    const int32_t *t_rowidx;    // GLOBAL column ids of S (rows of O's matrix), ascending inside a column
    const double *t_vals;
    const int32_t *order;       // O's columns, longest first
    int64_t ncols;              // of O
    const double *s_items;      // S's factors (only this rank's columns are read)
    const double *zero_row;     // K zeros (gather target of padding slots)
    double *prec;               // O's prec: ncols x PART
    double mean_rating;         // of O (computeMuLambda is O's member: c++/sample.cpp:256)
    double alpha;
};

}  // namespace bpmf
