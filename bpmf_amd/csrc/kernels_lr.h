// Product form of the column update for columns with only a few ratings, K = 64 (k_pf_prepare + k_sample_pf).
//
// A column with n ratings has  Lambda* = LambdaF + alpha sum_r u_r u_r^T  (c++/sample.cpp:248-258,297-298):
// a rank-n update of a matrix that is the SAME for every column of the half-iteration.  The host ships
// R0 = chol(LambdaF).matrixU(), R0^-1 and y0 = R0^-T LambdaF mu with the parameters.  O(n K) scans + one K x K
// matrix-vector product instead of K^3 / 3: on a ChEMBL-shaped side (483 500 compounds, ~2 activities each) the full
// factorisation is >95 % of the work.  (Round 2's Householder sweeps over R0 -- k_sample_lr -- lost to this form on
// every column it covered and left the library in round 5: docs/FINDINGS.md.)
#pragma once
#include "kernels.h"

namespace bpmf {


__device__ __forceinline__ double readlane_d(double v, int lane)
{
    const long long w = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)w, lane), hi = __builtin_amdgcn_readlane((int)(w >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// ---------------------------------------------------------------------------
// k_sample_pf<K, NCAP>: the column update in PRODUCT FORM, for columns with at most NCAP <= 16 ratings.
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
// solves, x = R0^-1 v of the four as one MFMA GEMM, stores.  S0 = R0^-1 in LDS in the GEMM's A-operand order (pf_fill_s0: lane
// stride PF_SLD = 18 doubles), sr / sv this wave's slots (the four columns of a pass K + 8 doubles apart).
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
    // (LDS layout: see PF_SLD / pf_svld above.  Round 5, per-kernel counters of the final kernels, profiles/r05_pmc_by_kernel_chembl.txt:
    // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 32 % in k_sample_pf<64, 3>, 26 % / 24 % in <64, 6> / <64, 16> -- with the LDS
    // array 28 % busy in the <64, 3> launch and its VALU 68 %: the conflicts are not what bounds these kernels)
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

}  // namespace bpmf
