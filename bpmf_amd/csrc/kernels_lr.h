// k_sample_lr<K, NB>: the column update for columns with only a few ratings, K = 64.
//
// A column with n ratings has  Lambda* = LambdaF + alpha sum_r u_r u_r^T  (c++/sample.cpp:248-258,297-298):
// a rank-n update of a matrix that is the SAME for every column of the half-iteration.  The
// host ships R0 = chol(LambdaF).matrixU() with the parameters; the wave applies the update to it in
// sweeps over NB <= 4 ratings (lr_update_block: one Householder reflector per row; R stays upper
// triangular with a positive diagonal, i.e. THE Cholesky factor the reference computes at :306, up
// to rounding), then solves as the reference does: x = R'^-1 (R'^-T b + z) (:321-323).  O(n K^2)
// instead of K^3 / 3: on a ChEMBL-shaped side (483 500 compounds, ~2 activities each) the full
// factorisation is >95 % of the work.
//
// One wave per column, lane j owns COLUMN j of R in registers (r[i] = R[i][j], zero below the
// diagonal).  Update step k broadcasts R[k][k] and the x_m[k] (v_readlane), forms the reflector once
// per wave, and every lane updates its (R[k][j], x_m[j]).  The forward solve R^T y = b is lane-local
// (lane k needs column k); the backward solve R x = w needs ROWS: the columns pass through LDS 16
// at a time (a K x 17 tile, conflict-free both ways).  Columns without ratings skip all of that:
// x = R0^-1 (y0 + z) with R0^-1 and y0 = R0^-T LambdaF mu from the host.
#pragma once
#include "kernels.h"

namespace bpmf {


__device__ __forceinline__ double readlane_d(double v, int lane)
{
    const long long w = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)w, lane), hi = __builtin_amdgcn_readlane((int)(w >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double rcp_nr(double d)
{
    double y = __builtin_amdgcn_rcp(d);
    double e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    e = fma(-d, y, 1.0);
    return fma(y, e, y);
}

// R^T R += sum_{m < NB} x_m x_m^T for NB ratings at once: step k annihilates x_0[k] .. x_{NB-1}[k] against
// R[k][k] with ONE Householder reflector (v = a + |a| e_1 for a = (R[k][k], x_0[k], ...), row k negated
// afterwards so that its diagonal stays positive: no cancellation, same R as NB Givens sweeps up to
// rounding).  With sigma = |a|, v1 = R[k][k] + sigma, beta = 1 / (sigma v1), t_j = v1 R[k][j] + sum_m x_m[k] x_m[j]:
//     R'[k][j] = t_j / sigma - R[k][j],     x_m'[j] = x_m[j] - (x_m[k] beta) t_j.
// Per step 2 NB + 2 instructions per lane and 4 NB + 17 wave-uniform ones (one 1/sqrt, one reciprocal)
// instead of NB x (4 + 18) for NB separate rotations.
template <int K, int NB>
__device__ __forceinline__ void lr_update_block(double (&r)[K], double &b, const LrArgs &a, int64_t p, int navail, int lane)
{
    double x[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) {                                     // (slots past the column's last rating: a zero vector)
        const bool ok = m < navail;                                   // wave-uniform
        const int row = ok ? a.rowidx[p + m] : 0;
        const double u = ok ? a.other_items[(size_t)row * K + lane] : 0.0;
        const double wv = ok ? (a.vals[p + m] - a.mean_rating) * a.alpha : 0.0;  // c++/sample.cpp:256
        b = fma(u, wv, b);
        x[m] = u * a.sqrt_alpha;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double Rkk = readlane_d(r[k], k);
        double xk[NB];
        double s2 = Rkk * Rkk;
#pragma unroll
        for (int m = 0; m < NB; ++m) { xk[m] = readlane_d(x[m], k); s2 = fma(xk[m], xk[m], s2); }
        const double inv = rsqrt_nr(s2);                              // 1 / sigma
        const double v1 = fma(s2, inv, Rkk);                          // R[k][k] + sigma
        const double beta = inv * rcp_nr(v1);
        const double rk = r[k];
        double t = v1 * rk;
#pragma unroll
        for (int m = 0; m < NB; ++m) t = fma(xk[m], x[m], t);
        r[k] = fma(inv, t, -rk);
#pragma unroll
        for (int m = 0; m < NB; ++m) x[m] = fma(-(xk[m] * beta), t, x[m]);
        // (lane k's x_m is now ~1e-17 |x_m[k]| instead of 0 and leaks that much into the LOWER triangle of the
        //  later rows; nothing reads it: the solves below touch R[i][j] with i <= j only.  Zeroing it
        //  with a `lane == k` select would keep 64 loop-invariant compare masks -- 128 SGPRs -- alive.)
    }
}

// NB = ratings per sweep: one instantiation per class of columns (the host sorts the light columns by
// their number of ratings: 1 | 2 | 3, 5, 6, 9 | 4, 7, 8, 10, 11, 12; a switch between sweep widths inside
// one kernel makes the register allocator spill ~400 registers)
template <int K, int NB>
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

    // (the normal draw first: its Philox / log / sqrt temporaries are dead before R occupies 128 registers)
    draw_normals<K>(sample_counter(a.col_from + col, a.ktrue, a.iter_plus_1), a.ktrue, sz, lane, K);
    __builtin_amdgcn_sched_barrier(0);

    if (len == 0) {
        // no ratings (a third of a ChEMBL-shaped side): Lambda* = LambdaF, so the factor, its inverse and the
        // forward solve are the same for all of them and come from the host: x = R0^-1 (y0 + z), one
        // triangular matrix-vector product (lane i = row i of R0^-1, w_j broadcast)
        __syncthreads();
        const double wv = a.y0[lane] + sz[lane];
        double x0 = 0.0, x1 = 0.0;
#pragma unroll
        for (int j = 0; j < K; j += 2) {
            x0 = fma(a.S0t[(size_t)j * K + lane], readlane_d(wv, j), x0);
            x1 = fma(a.S0t[(size_t)(j + 1) * K + lane], readlane_d(wv, j + 1), x1);
        }
        const double xs0 = x0 + x1;
        a.items[(size_t)(a.col_from + col) * K + lane] = xs0;
        const bool bad0 = !(fabs(xs0) <= 1.79769313486231570815e+308);
        if (__any(bad0)) { if (lane == 0) atomicMin(a.fail, (unsigned long long)(a.col_from + col)); }
        return;
    }

    double r[K];
#pragma unroll
    for (int i = 0; i < K; ++i) r[i] = a.R0[(size_t)i * K + lane];
    double b = a.Lmu[lane];                                            // rr = LambdaF mu (:285)

    // ---- rank-n update of R (R^T R += alpha sum u u^T) and of the rhs (:251-256), NB ratings per sweep
    for (int t = 0; t < len; t += NB) lr_update_block<K, NB>(r, b, a, p0 + t, len - t, lane);

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


// ---------------------------------------------------------------------------
// k_sample_pf<K, NCAP>: the same update in PRODUCT FORM, for columns with at most NCAP <= 16 ratings.
//
// With x_1 = sqrt(alpha) u_1 and p_1 = R0^-T x_1:  Lambda* = R0^T (I + p_1 p_1^T) R0, and the Cholesky
// factor of a rank-one update of the identity is known in closed form: I + p p^T = C^T C with
//     C[k][k] = sqrt(s_{k+1} / s_k),   C[k][j] = p_k p_j / sqrt(s_k s_{k+1})  (j > k),   s_k = 1 + sum_{i<k} p_i^2.
// So R' = C_n ... C_1 R0 (upper triangular, positive diagonal: the reference's factor) never has to be
// formed: with p_m = C_{m-1}^-T ... C_1^-T R0^-T x_m,
//     x = R0^-1 C_1^-1 ... C_n^-1 ( C_n^-T ... C_1^-T (y0 + sum_m kappa_m R0^-T x_m) + z ),
// and a solve with C or C^T collapses to ONE prefix (suffix) sum across the wave:
//     C^T t = c :  t_j = (c_j - p_j B_j / s_j) sqrt(s_j / s_{j+1}),        B_j = sum_{k<j} p_k c_k
//     C  v = w :  v_k = (w_k - p_k sqrt(s_{k+1} / s_k) F_k) sqrt(s_k / s_{k+1}),   F_k = sum_{j>k} g_j w_j,  g = p / sqrt(s s')
// R0^-1 (host side, like R0 itself) sits in LDS once per workgroup, in the order the final GEMM's A operand reads it
// (pf_fill_s0 below); eight waves walk the light columns of the side.  A column costs
// ONE matrix-vector product with R0^-1 (its share of an MFMA GEMM over four columns; the n products R0^-T u_row come
// from k_pf_prepare) and n (n - 1) / 2 + 3 n scans instead of ~25-43 instructions x 64 steps per sweep plus two
// triangular solves.  Round 4: the scans serve two or four columns at a time (pf_group, pf_group_stream below).
// ---------------------------------------------------------------------------
// x + (the value a DPP move fetches): the step of the scans below (no LDS crossbar round trips; a 64-lane __shfl_up
// ladder is six dependent ds_bpermute pairs of ~100+ cycles each, and the scans are the critical chain of a
// product-form column: n (n - 1) / 2 + 3 n of them)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
    const long long w = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)w, CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(w >> 32), CTRL, ROW_MASK, 0xF, true);
    return v + __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// q_row = R0^-T u_row depends on the ROW (a column of the other side), not on the column that reads it: it is
// computed ONCE per half-iteration for every row (Q = U_other R0^-1, nrows x K) instead of once per rating --
// on the ChEMBL-shaped compounds side that turns n + 1 = 2.7 matrix-vector products per column into one (plus n
// gathers of 512 bytes).  Same instruction sequence per entry as the in-line product had: bit-identical q.
template <int K>
__global__ __launch_bounds__(512, 4) void k_pf_prepare(const double *__restrict__ S0t, const double *__restrict__ other_items, int64_t nrows,
                                                       double *__restrict__ Q)
{
    static_assert(K == 64, "one lane per latent index");
    constexpr int LD = K + 1, NW = 8;
    __shared__ double S0[K * LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int q = tid; q < K * K; q += 64 * NW) {                       // S0t[j * K + i] = (R0^-1)[i][j]
        const int j = q / K, i = q % K;
        S0[i * LD + j] = S0t[q];
    }
    __syncthreads();
    for (int64_t row = (int64_t)blockIdx.x * NW + wave; row < nrows; row += (int64_t)gridDim.x * NW) {
        const double u = other_items[(size_t)row * K + lane];
        int ln = lane;                                                // (opaque per row: column `lane` of S0 is not to be hoisted into 128 registers)
        asm volatile("" : "+v"(ln));
        double q = 0.0;                                               // q_j = sum_i S0[i][j] u_i
#pragma unroll
        for (int i = 0; i < K; ++i) q = fma(S0[i * LD + ln], readlane_d(u, i), q);
        Q[(size_t)row * K + lane] = q;
    }
}

// ---------------------------------------------------------------------------
// The product-form solves of SEVERAL columns at once (round 4).  A scan over the 64 latent indices of one column costs
// the wave six DPP steps of three instructions whichever way the data lies, so the layout that pays is the one in
// which a step serves more than one column: NCOL columns share the wave, column c on the LPC = 64 / NCOL lanes
// c LPC .. c LPC + LPC - 1, lane (c, l) holding the E = NCOL consecutive entries l E .. l E + E - 1 of each of its
// column's vectors.  A scan is then E - 1 additions inside the lane, a scan of the lane totals over one DPP row
// (NCOL = 4: row_shr / row_shl 1, 2, 4, 8; NCOL = 2: + one row_bcast:15) and E additions -- ~20 instructions for all
// NCOL columns instead of 18 per column -- while everything element-wise costs what it did (E instructions per lane
// = one per column).  Columns with fewer ratings than their neighbours in the group are padded with zero vectors:
// p = 0 makes C = I, exactly (s = 1, rs = 1, a = 0).
// Per factor a lane keeps p, a = p / s and rs = sqrt(s / s') for its E entries -- with v = w rs - p F in the solve
// with C (sqrt(s'/s) sqrt(s/s') = 1) the fourth value of PfFactor is not needed -- and s'_j = s_{j+1}, so E + 1
// reciprocal square roots per lane serve the 2 E of (s, s').  Registers bound NCOL: 3 factors x 4 entries (<= 3 ratings: 124 registers; 4 factors spill)
// and 6 factors x 2 entries (<= 6) fit the 128 of this kernel's occupancy, 12 factors do not (k_sample_pf<.., 12>
// keeps one column per wave).
// ---------------------------------------------------------------------------
template <int E> struct PfT { double p[E], a[E], rs[E]; };

// LDS layout of the product-form kernels (round 4).  R0^-1 sits in the ORDER the GEMM's A operand reads it: lane (k, b, x) of
// row tile It holds (R0^-1)[16 It + 4 b + x][4 kk + k], kk = 0 .. 15, as 16 consecutive doubles (+ 2 of padding: lane stride
// 18 doubles = 36 banks, 16 lanes of a 16-byte read cover the 64 banks once) -- 32 ds_read_b128 per pass instead of 64
// ds_read_b64 whose row stride of K + 1 put up to four lanes on a bank.  The four columns of a pass are K + 8 doubles apart:
// the B operand's read of entry 4 kk + k of columns 0 .. 3 then hits four different bank groups (at stride K: one).
constexpr int PF_SLD = 18;                                          // doubles per (row tile, lane) of R0^-1 in LDS
template <int K> constexpr int pf_s0_words() { return (K / 16) * 64 * PF_SLD; }
template <int K> constexpr int pf_svld() { return K + 8; }
template <int K>
__device__ __forceinline__ void pf_fill_s0(double *S0, const double *__restrict__ S0t, int tid, int nthreads)
{
    for (int q = tid; q < K * K; q += nthreads) {                      // S0t[j * K + i] = (R0^-1)[i][j]
        const int j = q / K, i = q % K;
        const int It = i >> 4, b = (i >> 2) & 3, x = i & 3, kk = j >> 2, k = j & 3;
        S0[(It * 64 + k * 16 + b * 4 + x) * PF_SLD + kk] = S0t[q];
    }
}

template <int NCOL>
__device__ __forceinline__ double group_incl_prefix(double v)       // inclusive prefix over the LPC lanes of every column
{
    v = dpp_add<0x111, 0xF>(v);                                       // row_shr:1
    v = dpp_add<0x112, 0xF>(v);                                       // row_shr:2
    v = dpp_add<0x114, 0xF>(v);                                       // row_shr:4
    v = dpp_add<0x118, 0xF>(v);                                       // row_shr:8
    if constexpr (NCOL == 2) v = dpp_add<0x142, 0xA>(v);              // row_bcast:15 -> rows 1, 3 (a column = two rows)
    return v;
}
template <int NCOL>
__device__ __forceinline__ double group_incl_suffix(double v, int last_lane_bytes)
{
    if constexpr (NCOL == 4) {
        v = dpp_add<0x101, 0xF>(v);                                   // row_shl:1
        v = dpp_add<0x102, 0xF>(v);                                   // row_shl:2
        v = dpp_add<0x104, 0xF>(v);                                   // row_shl:4
        v = dpp_add<0x108, 0xF>(v);                                   // row_shl:8
        return v;
    } else {
        const double p = group_incl_prefix<NCOL>(v);
        return gbcast(p, last_lane_bytes, 0) - p + v;                 // total of the column (its last lane's prefix) - prefix + own
    }
}
// B_e = base + sum of the entries of the column BEFORE entry e of this lane
template <int NCOL, int E>
__device__ __forceinline__ void group_excl_prefix(const double (&x)[E], double (&B)[E], double base)
{
    double run[E];
    run[0] = x[0];
#pragma unroll
    for (int e = 1; e < E; ++e) run[e] = run[e - 1] + x[e];
    const double O = (group_incl_prefix<NCOL>(run[E - 1]) - run[E - 1]) + base;
    B[0] = O;
#pragma unroll
    for (int e = 1; e < E; ++e) B[e] = O + run[e - 1];
}
// F_e = sum of the entries of the column AFTER entry e of this lane
template <int NCOL, int E>
__device__ __forceinline__ void group_excl_suffix(const double (&x)[E], double (&F)[E], int last_lane_bytes)
{
    double run[E];
    run[E - 1] = x[E - 1];
#pragma unroll
    for (int e = E - 2; e >= 0; --e) run[e] = run[e + 1] + x[e];
    const double O = group_incl_suffix<NCOL>(run[0], last_lane_bytes) - run[0];
    F[E - 1] = O;
#pragma unroll
    for (int e = E - 2; e >= 0; --e) F[e] = O + run[e + 1];
}
template <int NCOL, int E>
__device__ __forceinline__ PfT<E> pft_make(const double (&q)[E])
{
    PfT<E> f;
    double p2[E], S[E], R[E + 1];
#pragma unroll
    for (int e = 0; e < E; ++e) p2[e] = q[e] * q[e];
    group_excl_prefix<NCOL, E>(p2, S, 1.0);                           // s_j = 1 + sum_{i<j} p_i^2
#pragma unroll
    for (int e = 0; e < E; ++e) R[e] = rsqrt_nr(S[e]);
    R[E] = rsqrt_nr(S[E - 1] + p2[E - 1]);                            // (s' of the lane's last entry; the others' s' is the next entry's s)
#pragma unroll
    for (int e = 0; e < E; ++e) {
        f.p[e] = q[e];
        f.a[e] = q[e] * (R[e] * R[e]);                                // p / s
        f.rs[e] = (S[e] * R[e]) * R[e + 1];                           // sqrt(s / s')
    }
    return f;
}
template <int NCOL, int E>
__device__ __forceinline__ void pft_solve_t(const PfT<E> &f, double (&c)[E])       // C^T t = c, in place
{
    double pc[E], B[E];
#pragma unroll
    for (int e = 0; e < E; ++e) pc[e] = f.p[e] * c[e];
    group_excl_prefix<NCOL, E>(pc, B, 0.0);
#pragma unroll
    for (int e = 0; e < E; ++e) c[e] = fma(-f.a[e], B[e], c[e]) * f.rs[e];
}
template <int NCOL, int E>
__device__ __forceinline__ void pft_solve(const PfT<E> &f, double (&w)[E], int last_lane_bytes)   // C v = w, in place
{
    double gw[E], F[E];
#pragma unroll
    for (int e = 0; e < E; ++e) gw[e] = (f.a[e] * f.rs[e]) * w[e];
    group_excl_suffix<NCOL, E>(gw, F, last_lane_bytes);
#pragma unroll
    for (int e = 0; e < E; ++e) w[e] = fma(-f.p[e], F[e], w[e] * f.rs[e]);
}

// the columns w0 + g .. w0 + g + NCOL - 1 of a pass: everything between their normals (in sv) and their v (back into sv)
template <int K, int NCAP, int NCOL>
__device__ __forceinline__ void pf_group(const LrArgs &a, int w0, int g, int wend, double (*sv)[pf_svld<K>()], int lane)
{
    constexpr int E = NCOL, LPC = 64 / NCOL;
    static_assert(K == 64 && LPC * E == K, "a column's K entries over its LPC lanes");
    const int c = lane / LPC, l = lane % LPC;
    const int w = w0 + g + c;
    const bool valid = w < wend;
    const int len = valid ? a.len[w] : 0;
    const int64_t p0 = valid ? a.p0[w] : 0;
    int nmax = __builtin_amdgcn_readlane(len, 0);
#pragma unroll
    for (int cc = 1; cc < NCOL; ++cc) nmax = max(nmax, __builtin_amdgcn_readlane(len, cc * LPC));
    const int last_lane_bytes = 4 * (c * LPC + LPC - 1);
    typedef double dd2 __attribute__((ext_vector_type(2)));
    double cv[E];
    {
        const dd2 *py = reinterpret_cast<const dd2 *>(a.y0 + l * E);
#pragma unroll
        for (int e = 0; e < E; e += 2) { const dd2 t = py[e / 2]; cv[e] = t.x; cv[e + 1] = t.y; }
    }
    PfT<E> f[NCAP];
#pragma unroll
    for (int m = 0; m < NCAP; ++m) {
        if (m < nmax) {                                               // wave-uniform
            const bool has = m < len;
            const int row = has ? a.rowidx[p0 + m] : 0;
            const double wv = has ? (a.vals[p0 + m] - a.mean_rating) * a.alpha : 0.0;       // c++/sample.cpp:256
            const double sa = has ? a.sqrt_alpha : 0.0;               // (a column past its last rating: a zero vector, C = I)
            const dd2 *pq = reinterpret_cast<const dd2 *>(a.Q + (size_t)row * K + l * E);   // q = R0^-T u_row (k_pf_prepare)
            double q[E];
#pragma unroll
            for (int e = 0; e < E; e += 2) { const dd2 t = pq[e / 2]; q[e] = t.x; q[e + 1] = t.y; }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                cv[e] = fma(wv, q[e], cv[e]);                         // R0^-T b = y0 + sum_m wv_m R0^-T u_m
                q[e] *= sa;                                           // R0^-T x_m, x_m = sqrt(alpha) u_m
            }
#pragma unroll
            for (int k = 0; k < m; ++k) pft_solve_t<NCOL, E>(f[k], q);   // p_m = C_{m-1}^-T ... C_1^-T q
            f[m] = pft_make<NCOL, E>(q);
        }
    }
#pragma unroll
    for (int m = 0; m < NCAP; ++m)
        if (m < nmax) pft_solve_t<NCOL, E>(f[m], cv);
    double *slot = &sv[g + c][l * E];
    double v[E];
#pragma unroll
    for (int e = 0; e < E; e += 2) { const dd2 t = *reinterpret_cast<const dd2 *>(slot + e); v[e] = cv[e] + t.x; v[e + 1] = cv[e + 1] + t.y; }   // :322
#pragma unroll
    for (int m = NCAP - 1; m >= 0; --m)
        if (m < nmax) pft_solve<NCOL, E>(f[m], v, last_lane_bytes);
#pragma unroll
    for (int e = 0; e < E; e += 2) {
        dd2 t; t.x = valid ? v[e] : 0.0; t.y = valid ? v[e + 1] : 0.0;
        *reinterpret_cast<dd2 *>(slot + e) = t;
    }
}

// The same for columns with up to 16 ratings: that many factors of three values do not fit the registers, the VECTORS do
// (round 4: a cap of 16 instead of 12 moved 2 957 of the 5 370 heavier columns of the ChEMBL-shaped compounds side out of the
// slab launch, 539 -> 529 us for the side; at 20 the kernel spills 124 registers and the side takes 579 us).
// The rating vectors q_m stay in registers; factor k is made from q_k once every earlier factor has been applied to it,
// applied at once to the later vectors and to the right-hand side (the same solves in the same order per vector as in
// pf_group), dropped -- and made AGAIN from the kept p_k = q_k when the backward pass needs it: n extra pft_make
// (one scan each) for scans that serve two columns instead of one.
template <int K, int NCAP, int NCOL>
__device__ __forceinline__ void pf_group_stream(const LrArgs &a, int w0, int g, int wend, double (*sv)[pf_svld<K>()], int lane)
{
    constexpr int E = NCOL, LPC = 64 / NCOL;
    static_assert(K == 64 && LPC * E == K, "a column's K entries over its LPC lanes");
    const int c = lane / LPC, l = lane % LPC;
    const int w = w0 + g + c;
    const bool valid = w < wend;
    const int len = valid ? a.len[w] : 0;
    const int64_t p0 = valid ? a.p0[w] : 0;
    int nmax = __builtin_amdgcn_readlane(len, 0);
#pragma unroll
    for (int cc = 1; cc < NCOL; ++cc) nmax = max(nmax, __builtin_amdgcn_readlane(len, cc * LPC));
    const int last_lane_bytes = 4 * (c * LPC + LPC - 1);
    typedef double dd2 __attribute__((ext_vector_type(2)));
    double cv[E];
    {
        const dd2 *py = reinterpret_cast<const dd2 *>(a.y0 + l * E);
#pragma unroll
        for (int e = 0; e < E; e += 2) { const dd2 t = py[e / 2]; cv[e] = t.x; cv[e + 1] = t.y; }
    }
    double q[NCAP][E];
#pragma unroll
    for (int m = 0; m < NCAP; ++m) {
        if (m < nmax) {                                               // wave-uniform
            const bool has = m < len;
            const int row = has ? a.rowidx[p0 + m] : 0;
            const double wv = has ? (a.vals[p0 + m] - a.mean_rating) * a.alpha : 0.0;       // c++/sample.cpp:256
            const double sa = has ? a.sqrt_alpha : 0.0;
            const dd2 *pq = reinterpret_cast<const dd2 *>(a.Q + (size_t)row * K + l * E);
#pragma unroll
            for (int e = 0; e < E; e += 2) { const dd2 t = pq[e / 2]; q[m][e] = t.x; q[m][e + 1] = t.y; }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                cv[e] = fma(wv, q[m][e], cv[e]);
                q[m][e] *= sa;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NCAP; ++k) {
        if (k < nmax) {
            const PfT<E> f = pft_make<NCOL, E>(q[k]);                 // q_k has taken C_1^-T .. C_{k-1}^-T: it is p_k
#pragma unroll
            for (int m = k + 1; m < NCAP; ++m)
                if (m < nmax) pft_solve_t<NCOL, E>(f, q[m]);
            pft_solve_t<NCOL, E>(f, cv);
        }
    }
    double *slot = &sv[g + c][l * E];
    double v[E];
#pragma unroll
    for (int e = 0; e < E; e += 2) { const dd2 t = *reinterpret_cast<const dd2 *>(slot + e); v[e] = cv[e] + t.x; v[e + 1] = cv[e + 1] + t.y; }   // :322
#pragma unroll
    for (int k = NCAP - 1; k >= 0; --k) {
        if (k < nmax) {
            const PfT<E> f = pft_make<NCOL, E>(q[k]);
            pft_solve<NCOL, E>(f, v, last_lane_bytes);
        }
    }
#pragma unroll
    for (int e = 0; e < E; e += 2) {
        dd2 t; t.x = valid ? v[e] : 0.0; t.y = valid ? v[e + 1] : 0.0;
        *reinterpret_cast<dd2 *>(slot + e) = t;
    }
}

// One pass of a wave: the items [w0, wend) (at most NB = 4 columns of at most NCAP ratings each) -- normals, the product-form
// solves, x = R0^-1 v of the four as one MFMA GEMM, stores.  S0 = (R0^-1) in LDS (K x (K + 1)), sr / sv this wave's slots.
template <int K, int NCAP>
__device__ __forceinline__ void pf_pass(const LrArgs &a, int w0, int wend, const double *S0, double (*sr)[K], double (*sv)[pf_svld<K>()], double y0, int lane)
{
    constexpr int NB = 4;
    // z ~ N(0, I) of the pass's columns, two columns at a time (the later Philox rounds of a pair are shared), straight
    // into their slots of sv
#pragma unroll 1
    for (int cb = 0; cb < NB; cb += 2) {
        const int w = w0 + cb;
        if (w >= wend) break;                                         // wave-uniform
        const uint32_t cA = sample_counter(a.col_from + a.col[w], a.ktrue, a.iter_plus_1);
        if (w + 1 < wend) {
            const uint32_t cB = sample_counter(a.col_from + a.col[w + 1], a.ktrue, a.iter_plus_1);
            draw_normals_pair<K>(cA, cB, a.ktrue, sv[cb], sv[cb + 1], sr[0], sr[1], lane, K);
        } else {
            draw_normals_deferred<K>(cA, a.ktrue, sv[cb], sr[0], lane, K);
        }
    }
    if constexpr (NCAP <= 6) {
        // several columns per scan (pf_group): four with <= 3 ratings, two with <= 6
        constexpr int NCOL = NCAP <= 3 ? 4 : 2;
#pragma unroll 1
        for (int g = 0; g < NB; g += NCOL) pf_group<K, NCAP, NCOL>(a, w0, g, wend, sv, lane);
    } else {
        // two columns per scan, the factors made twice (pf_group_stream)
#pragma unroll 1
        for (int g = 0; g < NB; g += 2) pf_group_stream<K, NCAP, 2>(a, w0, g, wend, sv, lane);
    }
    // x = R0^-1 v for the NB columns at once: X (K x NB) = S0 (K x K) V (K x NB) on the 4x4x4 shape -- block b of an
    // instruction is row block 4 It + b of S0, the B operand (the four v's, k = lane / 16 picks the latent index
    // 4 kk + k, x = lane % 4 the column) is the same for every b: 64 MFMAs of 16 cycles for four columns against
    // 4 x 64 x (LDS read + two v_readlane + FMA) on the VALU.  D[b][i][j]: lane (i, b, j), register It = x[16 It + 4 b + i] of column j.
    double X[4] = {0.0, 0.0, 0.0, 0.0};
    int ln = lane;                                                    // (opaque: the operand addresses are not to be hoisted out of the column loop)
    asm volatile("" : "+v"(ln));
    const int kq2 = ln >> 4, bq2 = (ln >> 2) & 3, xq2 = ln & 3;        // operand view of v_mfma_f64_4x4x4_4b_f64: lane (k, b, x)
    typedef double dd2 __attribute__((ext_vector_type(2)));
    const dd2 *sop = reinterpret_cast<const dd2 *>(S0 + ln * PF_SLD);   // this lane's operands of row tile 0 (tile It: + 64 PF_SLD doubles)
#pragma unroll
    for (int kk = 0; kk < K / 4; kk += 2) {
        const double vb0 = sv[xq2][4 * kk + kq2], vb1 = sv[xq2][4 * kk + 4 + kq2];   // B[k][j] = v_j[4 kk + k]
#pragma unroll
        for (int It = 0; It < 4; ++It) {
            // R0^-1 is upper triangular: rows 16 It .. of its columns 4 kk .. 4 kk + 7 are exact zeros while kk + 1 < 4 It
            // (adding 0 x v changes nothing for finite v): 40 of the 64 products are issued
            if (kk + 1 < 4 * It) continue;
            const dd2 sa = sop[(It * 64 * PF_SLD + kk) / 2];          // A[b][i][k] = (R0^-1)[16 It + 4 b + i][4 kk + k], kk and kk + 1
            X[It] = mfma44(sa.x, vb0, X[It]);
            X[It] = mfma44(sa.y, vb1, X[It]);
        }
    }
    (void)bq2;
    // back to one lane per latent index (through the same LDS tile), coalesced stores
#pragma unroll
    for (int It = 0; It < 4; ++It) sv[xq2][16 * It + 4 * bq2 + kq2] = X[It];
#pragma unroll 1
    for (int cb = 0; cb < NB; ++cb) {
        const int w = w0 + cb;
        if (w >= wend) break;                                         // wave-uniform
        const int col = a.col[w];
        const double xs = sv[cb][lane];
        a.items[(size_t)(a.col_from + col) * K + lane] = xs;
        const bool bad = !(fabs(xs) <= 1.79769313486231570815e+308);
        if (__any(bad)) { if (lane == 0) atomicMin(a.fail, (unsigned long long)(a.col_from + col)); }
    }
}

template <int K, int NCAP>
__global__ __launch_bounds__(512, 4) void k_sample_pf(LrArgs a)
{
    static_assert(K == 64, "one lane per latent index");
    constexpr int NW = 8, NB = 4;                                     // NB columns per wave and pass: their final products x = R0^-1 v run as ONE MFMA GEMM
    // (round 4, measured: LD = K + 4 / v slots K + 8 -- which a bank model of the GEMM's operand reads says are conflict-free where
    // K + 1 / K put up to 4 lanes on a bank pair -- made the compounds side of the ChEMBL shape SLOWER, 770 against 733 us,
    // interleaved A/B of the two builds: the 25 % of r03_pmc_chembl.txt are not these reads; K + 1 / K stay)
    __shared__ __attribute__((aligned(16))) double S0[pf_s0_words<K>()];   // R0^-1 in operand order (pf_fill_s0)
    __shared__ double sr[NW][2][K];                                   // r2 of the accepted polar attempts of a pair of columns (draw_normals_pair)
    __shared__ __attribute__((aligned(16))) double sv[NW][NB][pf_svld<K>()];   // per column of a pass: its normals z, then v, then x
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    pf_fill_s0<K>(S0, a.S0t, tid, 64 * NW);
    const double y0 = a.y0[lane];
    __syncthreads();
    // passes of four columns, the LAST of the list first: the items are ascending in their number of ratings, so the most
    // expensive passes start the launch and its last round is made of the cheapest
    const int npass = (a.nitems + NB - 1) / NB;
    for (int p = (int)blockIdx.x * NW + wave; p < npass; p += (int)gridDim.x * NW)
        pf_pass<K, NCAP>(a, (npass - 1 - p) * NB, a.nitems, S0, sr[wave], sv[wave], y0, lane);
}

// The three classes of product-form columns (<= 3 | 4..6 | 7..16 ratings: pf_c[0..3], the item list is sorted by the
// number of ratings) in ONE launch.  As three launches each class ended on its own tail -- with 512 resident workgroups
// of eight waves a launch is a whole number of rounds of 16 384 columns: the 20 876 columns with 7..12 ratings (the third class then) of the
// ChEMBL-shaped side took two rounds for 1.27 rounds of work.  Here the passes (four columns of one class) of all three
// form one list, the most expensive first (classes in descending order, inside a class from its end: the item list is
// ascending in the number of ratings), and wave w of the N resident ones takes the passes w, w + N, w + 2 N, ...: every
// round of N passes is of (nearly) one cost, so the waves' totals differ by less than the cost of one pass of the most
// expensive kind and the launch ends on passes of the cheapest.  (A ticket counter handing out the passes one at a time
// was built first and measured 2.4 x SLOWER -- 1 603 against 677 us for the compounds side of the ChEMBL shape: 120 000
// read-modify-writes of one hot word at ~88 per microsecond are 1.4 ms.)  The three bodies are the three instantiations
// of pf_pass; all of them fit the 128 registers the LDS-bound occupancy (2 workgroups per CU) leaves a wave.
template <int K>
__global__ __launch_bounds__(512, 4) void k_sample_pf_all(LrArgs a)
{
    static_assert(K == 64, "one lane per latent index");
    constexpr int NW = 8, NB = 4;
    __shared__ __attribute__((aligned(16))) double S0[pf_s0_words<K>()];
    __shared__ double sr[NW][2][K];
    __shared__ __attribute__((aligned(16))) double sv[NW][NB][pf_svld<K>()];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    pf_fill_s0<K>(S0, a.S0t, tid, 64 * NW);
    const double y0 = a.y0[lane];
    __syncthreads();
    const int np2 = (a.pf_c[3] - a.pf_c[2] + NB - 1) / NB, np1 = (a.pf_c[2] - a.pf_c[1] + NB - 1) / NB, np0 = (a.pf_c[1] - a.pf_c[0] + NB - 1) / NB;
    const int npass = np2 + np1 + np0;
    for (int p = (int)blockIdx.x * NW + wave; p < npass; p += (int)gridDim.x * NW) {
        if (p < np2) {
            const int q = np2 - 1 - p;
            pf_pass<K, 16>(a, a.pf_c[2] + q * NB, a.pf_c[3], S0, sr[wave], sv[wave], y0, lane);
        } else if (p < np2 + np1) {
            const int q = np1 - 1 - (p - np2);
            pf_pass<K, 6>(a, a.pf_c[1] + q * NB, a.pf_c[2], S0, sr[wave], sv[wave], y0, lane);
        } else {
            const int q = np0 - 1 - (p - np2 - np1);
            pf_pass<K, 3>(a, a.pf_c[0] + q * NB, a.pf_c[1], S0, sr[wave], sv[wave], y0, lane);
        }
    }
}

}  // namespace bpmf
