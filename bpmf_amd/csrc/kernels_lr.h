// k_sample_lr<K>: the column update for columns with only a few ratings, K = 64.
//
// A column with n ratings has  Lambda* = LambdaF + alpha sum_r u_r u_r^T  (c++/sample.cpp:248-258,297-298):
// a rank-n update of a matrix that is the SAME for every column of the half-iteration.  The
// host ships R0 = chol(LambdaF).matrixU() with the parameters; the wave applies n rank-one updates
// to it (Givens form: R^T R + x x^T = R'^T R', R' upper triangular with positive diagonal, i.e.
// THE Cholesky factor the reference computes at :306, up to rounding), then solves as the reference
// does: x = R'^-1 (R'^-T b + z) (:321-323).  O(n K^2) instead of K^3 / 3: on a ChEMBL-shaped side
// (483 500 compounds, ~2 activities each) the full factorisation is >95 % of the work.
//
// One wave per column, lane j owns COLUMN j of R in registers (r[i] = R[i][j], zero below the
// diagonal).  Update step k broadcasts R[k][k] and x[k] (v_readlane), forms the rotation once per
// wave, and every lane rotates its (R[k][j], x[j]) pair.  The forward solve R^T y = b is
// lane-local (lane k needs column k); the backward solve R x = w needs ROWS: the columns pass
// through LDS 16 at a time (a K x 17 tile, conflict-free both ways).
#pragma once
#include "kernels.h"

namespace bpmf {

struct LrArgs {
    const int32_t *rowidx; const double *vals;
    const int32_t *col; const int64_t *p0; const int32_t *len;   // light work items (len <= NLR)
    int nitems;
    const double *other_items; double *items; int64_t col_from;
    const double *R0;          // K x K, row-major upper factor of LambdaF (zeros below the diagonal)
    const double *Lmu;         // LambdaF * mu
    unsigned long long *fail;
    double mean_rating, alpha, sqrt_alpha;
    uint32_t iter_plus_1;
};

__device__ __forceinline__ double readlane_d(double v, int lane)
{
    const long long w = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)w, lane), hi = __builtin_amdgcn_readlane((int)(w >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

template <int K>
__global__ __launch_bounds__(64, 3) void k_sample_lr(LrArgs a)
{
    static_assert(K == 64, "one lane per column of R");
    constexpr int TLD = 17;
    __shared__ double sz[K];
    __shared__ double tile[K * TLD];
    const int lane = threadIdx.x;
    const int w = blockIdx.x;
    const int col = a.col[w];
    const int64_t p0 = a.p0[w];
    const int len = a.len[w];

    // the first ratings' operands are requested before the normal draw
    const int row0 = len > 0 ? a.rowidx[p0] : 0;
    double u_next = len > 0 ? a.other_items[(size_t)row0 * K + lane] : 0.0;
    double wv_next = len > 0 ? (a.vals[p0] - a.mean_rating) * a.alpha : 0.0;   // c++/sample.cpp:256

    // (the normal draw first: its Philox / log / sqrt temporaries are dead before R occupies 128 registers)
    draw_normals<K>(sample_counter<K>(a.col_from + col, a.iter_plus_1), K, sz, lane);
    __builtin_amdgcn_sched_barrier(0);

    double r[K];
#pragma unroll
    for (int i = 0; i < K; ++i) r[i] = a.R0[(size_t)i * K + lane];
    double b = a.Lmu[lane];                                            // rr = LambdaF mu (:285)

    // ---- n rank-one updates of R (R^T R += alpha u u^T) and of the rhs (:251-256)
    for (int t = 0; t < len; ++t) {
        const double u = u_next, wv = wv_next;
        if (t + 1 < len) {                                             // wave-uniform
            const int row = a.rowidx[p0 + t + 1];
            u_next = a.other_items[(size_t)row * K + lane];
            wv_next = (a.vals[p0 + t + 1] - a.mean_rating) * a.alpha;
        }
        b = fma(u, wv, b);
        double x = u * a.sqrt_alpha;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double Rkk = readlane_d(r[k], k), xk = readlane_d(x, k);
            const double inv = rsqrt_nr(fma(xk, xk, Rkk * Rkk));      // 1 / hypot: the rotation (c, s) = (Rkk, xk) / hypot
            const double c = Rkk * inv, s = xk * inv;
            const double rk = r[k];
            r[k] = fma(c, rk, s * x);                                 // lanes j < k: both terms are zero
            x = fma(c, x, -(s * rk));
            // (lane k's x is now ~1e-17 |x_k| instead of 0 and leaks that much into the LOWER triangle of the
            //  later rows; nothing reads it: the solves below touch R[i][j] with i <= j only.  Zeroing it
            //  with a `lane == k` select would keep 64 loop-invariant compare masks -- 128 SGPRs -- alive.)
        }
    }

    // ---- my diagonal entry and its reciprocal
    double dg = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) dg = (lane == i) ? r[i] : dg;
    const double rs = rsqrt_nr(dg);
    const double invd = rs * rs;                                      // 1 / R[lane][lane]; NaN for a non-positive pivot

    // ---- forward solve R^T y = b (:321): lane k accumulates sum_{i<k} R[i][k] y_i from its own column
    // (lane k stops accumulating at step k: (b - acc) * invd is then its y for good -- no per-step capture,
    //  which the compiler turns into 64 live candidates)
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const double yi = readlane_d((b - acc) * invd, i);            // y_i: final in lane i at step i
        acc = (i < lane) ? fma(r[i], yi, acc) : acc;
    }
    const double y = (b - acc) * invd;
    __syncthreads();                                                  // the normals are in LDS
    double wk = y + sz[lane];                                         // :322

    // ---- backward solve R x = w (:323): rows of R through LDS, 16 columns at a time
#pragma unroll
    for (int jb = K / 16 - 1; jb >= 0; --jb) {
        __syncthreads();
        if ((lane >> 4) == jb) {
            const int jj = lane & 15;
#pragma unroll
            for (int i = 0; i < K; ++i)
                if (i < 16 * (jb + 1)) tile[i * TLD + jj] = r[i];     // column `lane`, rows 0 .. 16 jb + 15
        }
        __syncthreads();
#pragma unroll
        for (int jj = 15; jj >= 0; --jj) {
            const int j = 16 * jb + jj;
            const double xj = readlane_d(wk * invd, j);               // x_j: lane j's w no longer changes from here on
            const double Rkj = tile[lane * TLD + jj];                 // R[lane][j] (rows >= 16 (jb + 1): not written, not used)
            wk = (lane < j) ? fma(-Rkj, xj, wk) : wk;
        }
    }
    const double xs = wk * invd;

    // ---- items().col(idx) = rr (:324); a failed factorisation (:308) shows as a non-finite sample
    a.items[(size_t)(a.col_from + col) * K + lane] = xs;
    const bool bad = !(fabs(xs) <= 1.79769313486231570815e+308);
    if (__any(bad)) { if (lane == 0) atomicMin(a.fail, (unsigned long long)(a.col_from + col)); }
}

}  // namespace bpmf
