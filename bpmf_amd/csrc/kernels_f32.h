// fp32 large-K form of the hot path (K = 128): BASELINE config "MovieLens-1M, K=128, fp32".
//
// The reference computes in fp64 throughout (c++/bpmf.h:55-58); this path keeps the factors,
// the Gram, the factorisation and the solves in fp32 and everything that leaves the column loop
// (hyper-parameters, column statistics, prediction sums, the normal draws) in fp64.  It is the
// "mixed-precision tolerance study" of the north star: tests/test_gpu_f32.py states what the
// fp32 arithmetic costs against the fp64 restatement of the reference.
//
// One workgroup of four waves per column (c++/sample.cpp:263-336 for one idx):
//   * Gram: v_mfma_f32_16x16x4_f32 on the 36 upper 16x16 tiles of the 128x128 Gram, nine tiles
//     per wave; every wave walks all ratings of the column (operands come straight from the
//     gathered registers: lane (kq, li) loads U[row_kq][16 t + li], 64 contiguous bytes per 16
//     lanes), so no cross-wave reduction is needed.
//   * Lambda* = LambdaF + alpha G is formed in the MFMA accumulator tiles and stays there: the
//     blocked right-looking Cholesky (Lambda* = R^T R, 16-wide block rows) updates the register
//     tiles with MFMAs whose operands are the finished block rows of R, parked in LDS (41 KB).
//   * forward solve, + z, backward solve by wave 0 (two rows per lane, no barriers), coalesced store.
// Operand / result layout of v_mfma_f32_16x16x4_f32 (tools/probes/layout16f32_probe.hip):
//   A lane 16 k + i, B lane 16 k + j (one float each);  D[i = 4 (lane / 16) + reg][j = lane % 16].
#pragma once
#include "kernels.h"

namespace bpmf {

typedef float f4 __attribute__((ext_vector_type(4)));

// element type of the factors / tiles: float (K = 128) or double (K = 64: the reference's arithmetic)
template <typename T> struct WgTraits;
template <> struct WgTraits<float> {
    typedef f4 acc_t;
    __device__ static __forceinline__ f4 mfma(float x, float y, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c, 0, 0, 0); }
    __device__ static __forceinline__ int drow(int kq, int reg) { return 4 * kq + reg; }      // D[i = 4 (lane / 16) + reg][j = lane % 16]
    __device__ static __forceinline__ float rsqrt_acc(float d) { return __frsqrt_rn(d); }                // v_rsq_f32 (1 ulp)
    __device__ static __forceinline__ float bcast(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
};
template <> struct WgTraits<double> {
    typedef d4 acc_t;
    __device__ static __forceinline__ d4 mfma(double x, double y, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, c, 0, 0, 0); }
    __device__ static __forceinline__ int drow(int kq, int reg) { return kq + 4 * reg; }      // D[i = lane / 16 + 4 reg][j = lane % 16]
    __device__ static __forceinline__ double rsqrt_acc(double d) { return 1.0 / sqrt(d); }
    __device__ static __forceinline__ double bcast(double v, int l) { return ::bpmf::bcast(v, l); }
};


template <int K>
struct GeoF {
    static constexpr int NT = K / 16;                    // 16-wide tiles per dimension
    static constexpr int NTRI = NT * (NT + 1) / 2;
    // R (upper, Lambda* = R^T R) lives in LDS by block rows: block row s is 16 x (K - 16 s) floats
    // with a row stride of K - 16 s + 4: 40 960 B of LDS per K = 128 fp32 workgroup, i.e. four per CU (+ 8: 43 008 B, three per CU, 5 % slower although its operand reads conflict less)
    __host__ __device__ static constexpr int width(int s) { return K - 16 * s; }
    __host__ __device__ static constexpr int ld(int s) { return width(s) + 4; }
    __host__ __device__ static constexpr int roff(int s) { return 16 * s * (K + 4) - 128 * s * (s - 1); }   // 16 * sum_{t<s} ld(t)
    static constexpr int RWORDS = roff(NT);
    template <typename T> static constexpr size_t lds_bytes() { return (size_t)K * 8 + (size_t)RWORDS * sizeof(T) + 2 * K * sizeof(T); }
    // row-major upper index of tile (I, J), I <= J; with NW waves its owner is wave tri % NW, slot tri / NW
    __host__ __device__ static constexpr int tri(int I, int J) { return I * NT - (I * (I - 1)) / 2 + (J - I); }
    // ... and back: block row / column of tile `t`
    __host__ __device__ static constexpr int tile_i(int t) { int I = 0; while (t >= NT - I) { t -= NT - I; ++I; } return I; }
    __host__ __device__ static constexpr int tile_j(int t) { int I = 0; while (t >= NT - I) { t -= NT - I; ++I; } return I + t; }
};

// ---------------------------------------------------------------------------
// Per-wave part of one column: Gram of the wave's tiles, Lambda* in registers, blocked
// right-looking Cholesky  Lambda* = R^T R  on the register tiles.  Block step s:
//   A  the owners of the tiles (s, J >= s) park them in block row s of the LDS copy of R;
//   B  wave 0 factors the 16x16 diagonal block (lane c < 16 holds column c; pivots and row
//      entries travel through v_readlane) and stores R_ss and 1 / R_kk;
//   C  one thread per remaining column of the block row solves R_ss^T x = a (16 steps);
//   D  every wave applies  A_IJ -= R_sI^T R_sJ  to its own tiles (I, J), s < I <= J, with four
//      v_mfma_f32_16x16x4_f32 per tile whose operands are read from block row s.
// Three workgroup barriers per block step.
// ---------------------------------------------------------------------------
template <int K, typename T, int NW, int W>
__device__ __forceinline__ bool wg_column(const SampleArgsW<T> &a, int col_local, int64_t p0, int len, T *R, T *dinv, T *bv, int tid)
{
    using G = GeoF<K>;
    using X = WgTraits<T>;
    typedef typename X::acc_t acc_t;
    constexpr int NT = G::NT, TPW = (G::NTRI + NW - 1) / NW;
    const int lane = tid & 63;
    const int kq = lane >> 4, li = lane & 15;
    acc_t acc[TPW];
    T r[NT];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NT; ++t) r[t] = 0;
    const int32_t *rowidx = a.rowidx + p0;
    const double *vals = a.vals + p0;
    // 64 ratings per coalesced index block = 4 groups of 4 k-steps (16 ratings); the operands of the
    // next group (32 registers) are in flight while the 36 MFMAs of the current one issue
    int ri_n = (lane < len) ? rowidx[lane] : -1;
    T wv_n = (lane < len) ? (T)((vals[lane] - a.mean_rating) * a.alpha) : (T)0;               // c++/sample.cpp:256
    for (int b0 = 0; b0 < len; b0 += 64) {
        const int ri = ri_n;
        const T wv = wv_n;
        if (b0 + 64 < len) {                                                     // workgroup-uniform
            const int q = b0 + 64 + lane;
            ri_n = (q < len) ? rowidx[q] : -1;
            wv_n = (q < len) ? (T)((vals[q] - a.mean_rating) * a.alpha) : (T)0;
        }
        const int ngroups = (len - b0 >= 64) ? 4 : (len - b0 + 15) >> 4;
        T y[4][NT], yn[4][NT], ww[4], wn[4];
        auto gather = [&](int gg, T (&yy)[4][NT], T (&w1)[4]) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int src = (gg * 4 + st) * 4 + kq;
                const int row = __shfl(ri, src);
                w1[st] = __shfl(wv, src);
                const T *u = a.other_items + (size_t)(row >= 0 ? row : 0) * K + li;
#pragma unroll
                for (int t = 0; t < NT; ++t) yy[st][t] = (row >= 0) ? u[16 * t] : (T)0;
            }
        };
        gather(0, y, ww);
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            if (gg >= ngroups) break;                                            // workgroup-uniform
            const bool more = gg + 1 < ngroups;
            if (gg < 3 && more) gather(gg + 1, yn, wn);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                if (W == 0) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) r[t] = fma(y[st][t], ww[st], r[t]);
                }
#pragma unroll
                for (int I = 0; I < NT; ++I)
#pragma unroll
                    for (int J = I; J < NT; ++J)
                        if ((G::tri(I, J) % NW) == W)
                            acc[G::tri(I, J) / NW] = X::mfma(y[st][I], y[st][J], acc[G::tri(I, J) / NW]);
            }
            if (gg < 3 && more) {
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    ww[st] = wn[st];
#pragma unroll
                    for (int t = 0; t < NT; ++t) y[st][t] = yn[st][t];
                }
            }
        }
    }
    // Lambda* = LambdaF + alpha G in the register tiles (:297-298); b = LambdaF mu + rr (:285,:256)
    const double *LF = a.prop_lambda ? a.prop_lambda + (size_t)col_local * K * K : a.LambdaF;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = I; J < NT; ++J)
            if ((G::tri(I, J) % NW) == W) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int gi = 16 * I + X::drow(kq, reg), gj = 16 * J + li;
                    acc[G::tri(I, J) / NW][reg] = (a.diag_only && gi != gj) ? (T)0 : (T)fma(a.alpha, (double)acc[G::tri(I, J) / NW][reg], LF[gi + (size_t)gj * K]);
                }
            }
    if (W == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            T v = r[t];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (kq == 0) {
                double lm = a.Lmu[16 * t + li];
                if (a.prop_lambda) {                                 // rr = Lambda_i * hp.mu (:285)
                    lm = 0.0;
                    for (int j = 0; j < K; ++j) lm = fma(LF[16 * t + li + (size_t)j * K], a.mu[j], lm);
                }
                bv[16 * t + li] = (T)(lm + (double)v);
            }
        }
    }

    bool bad = false;
#pragma unroll
    for (int s = 0; s < NT; ++s) {
        T *Rs = R + G::roff(s);
        const int LDs = G::ld(s), Ws = G::width(s);
        // A: park the tiles of block row s
#pragma unroll
        for (int J = s; J < NT; ++J)
            if ((G::tri(s, J) % NW) == W) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) Rs[X::drow(kq, reg) * LDs + 16 * (J - s) + li] = acc[G::tri(s, J) / NW][reg];
            }
        __syncthreads();
        // B: diagonal block, upper Cholesky, by the first 16 lanes of wave 0 (column c in registers)
        if (W == 0) {
            const int c = lane & 15;
            T col[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) col[k] = Rs[k * LDs + c];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const T d = X::bcast(col[k], k);
                bad |= !(d > (T)0);
                const T rinv = X::rsqrt_acc(d);
                col[k] *= rinv;                                       // R(k, c), c >= k (entries left of the diagonal are not used)
                if (lane == k) dinv[16 * s + k] = rinv;
#pragma unroll
                for (int m = k + 1; m < 16; ++m) {
                    const T rkm = X::bcast(col[k], m);
                    col[m] = fma(-rkm, col[k], col[m]);               // A(m, c) -= R(k, m) R(k, c)
                }
            }
            if (lane < 16) {
#pragma unroll
                for (int k = 0; k < 16; ++k) Rs[k * LDs + c] = col[k];
            }
        }
        __syncthreads();
        // C: the other columns of the block row: R_ss^T x = a, one thread per column
        for (int cc = tid; cc < Ws - 16; cc += 64 * NW) {
            T *cp = Rs + 16 + cc;
            T x[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = cp[k * LDs];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                x[k] *= dinv[16 * s + k];
#pragma unroll
                for (int m = k + 1; m < 16; ++m) x[m] = fma(-Rs[k * LDs + m], x[k], x[m]);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) cp[k * LDs] = x[k];
        }
        __syncthreads();
        // D: trailing update of this wave's tiles
        if (s + 1 < NT) {
#pragma unroll
            for (int I = s + 1; I < NT; ++I) {
                bool any = false;
#pragma unroll
                for (int J = I; J < NT; ++J) any |= (G::tri(I, J) % NW) == W;
                if (!any) continue;                                  // compile-time
                T opI[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) opI[q] = -Rs[(4 * q + kq) * LDs + 16 * (I - s) + li];
#pragma unroll
                for (int J = I; J < NT; ++J)
                    if ((G::tri(I, J) % NW) == W) {
                        T opJ[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) opJ[q] = Rs[(4 * q + kq) * LDs + 16 * (J - s) + li];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[G::tri(I, J) / NW] = X::mfma(opI[q], opJ[q], acc[G::tri(I, J) / NW]);
                    }
            }
        }
    }
    return bad;
}

template <int K, typename T, int NW>
__global__ __launch_bounds__(64 * NW, 2) void k_sample_wg(SampleArgsW<T> a)
{
    using G = GeoF<K>;
    using X = WgTraits<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[G::template lds_bytes<T>()];
    double *zs = reinterpret_cast<double *>(smem);                   // K normals (fp64 draw, as the reference)
    T *R = reinterpret_cast<T *>(zs + K);                            // R by block rows
    T *bv = R + G::RWORDS;                                           // rhs
    T *dinv = bv + K;                                                // 1 / R(k,k)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int w = blockIdx.x;
    const int col = a.wi_col[w];
    const int64_t p0 = a.wi_p0[w];
    const int len = a.wi_len[w];
    const int64_t idx = a.col_from + col;

    // z ~ N(0, I): stream (idx+1)*K*(iter+1) mod 2^32 (c++/sample.cpp:266); the last wave has the fewest tiles
    if (wave == NW - 1) draw_normals<K>(sample_counter(idx, a.ktrue, a.iter_plus_1), a.ktrue, zs, lane, K);
    bool bad;
    if constexpr (NW == 1) {
        bad = wg_column<K, T, 1, 0>(a, col, p0, len, R, dinv, bv, tid);
    } else if constexpr (NW == 2) {
        if (wave == 0) bad = wg_column<K, T, 2, 0>(a, col, p0, len, R, dinv, bv, tid);
        else bad = wg_column<K, T, 2, 1>(a, col, p0, len, R, dinv, bv, tid);
    } else {
        switch (wave) {
        case 0: bad = wg_column<K, T, 4, 0>(a, col, p0, len, R, dinv, bv, tid); break;
        case 1: bad = wg_column<K, T, 4, 1>(a, col, p0, len, R, dinv, bv, tid); break;
        case 2: bad = wg_column<K, T, 4, 2>(a, col, p0, len, R, dinv, bv, tid); break;
        default: bad = wg_column<K, T, 4, 3>(a, col, p0, len, R, dinv, bv, tid); break;
        }
    }
    __syncthreads();

    // ---- R^T y = b (:321), y += z (:322), R x = y (:323): wave 0, rows (lane, lane + 64).
    // Sixteen steps (one block row of R) at a time: their R entries and 1/R(k,k) are loaded into
    // registers first, so that the dependency chain of a step is readlane -> multiply -> fma only.
    if (wave == 0) {
        T y0 = bv[lane], y1 = (K > 64) ? bv[lane + 64] : (T)0;
        // forward: after y_k is known, b_j -= R(k, j) y_k for j > k (row k of R: contiguous)
        for (int s = 0; s < G::NT; ++s) {
            const T *Rs = R + G::roff(s) - 16 * s;                   // Rs[r * ld + j] = R(16 s + r, j)
            const int LDs = G::ld(s);
            T r0[16], r1[16], di[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = 16 * s + r;
                di[r] = dinv[k];
                r0[r] = (lane > k) ? Rs[r * LDs + lane] : (T)0;
                r1[r] = (K > 64 && lane + 64 > k) ? Rs[r * LDs + lane + 64] : (T)0;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = 16 * s + r;
                const T own = (k < 64) ? y0 : y1;
                const T yk = X::bcast(own, k & 63) * di[r];
                if (lane == (k & 63)) { if (k < 64) y0 = yk; else y1 = yk; }
                y0 = fma(-r0[r], yk, y0);
                if (K > 64) y1 = fma(-r1[r], yk, y1);
            }
        }
        y0 += (T)zs[lane];
        if (K > 64) y1 += (T)zs[lane + 64];
        // backward: x_k = y_k / R(k,k); y_i -= R(i, k) x_k for i < k (column k of R: per-lane row bases)
        int base0, base1;
        {
            const int s0 = lane >> 4, s1 = (lane + 64) >> 4;
            base0 = G::roff(s0) + (lane & 15) * G::ld(s0) - 16 * s0;
            base1 = G::roff(s1) + (lane & 15) * G::ld(s1) - 16 * s1;
        }
        for (int s = G::NT - 1; s >= 0; --s) {
            T c0[16], c1[16], di[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = 16 * s + r;
                di[r] = dinv[k];
                c0[r] = (lane < k) ? R[base0 + k] : (T)0;            // R(lane, k)
                c1[r] = (K > 64 && lane + 64 < k) ? R[base1 + k] : (T)0;
            }
#pragma unroll
            for (int r = 15; r >= 0; --r) {
                const int k = 16 * s + r;
                const T own = (k < 64) ? y0 : y1;
                const T xk = X::bcast(own, k & 63) * di[r];
                if (lane == (k & 63)) { if (k < 64) y0 = xk; else y1 = xk; }
                y0 = fma(-c0[r], xk, y0);
                if (K > 64) y1 = fma(-c1[r], xk, y1);
            }
        }
        T *dst = a.items + (size_t)idx * K;                                     // items().col(idx) = rr (:324)
        dst[lane] = y0;
        if (K > 64) dst[lane + 64] = y1;
        // non-positive pivot or a non-finite sample: "Cholesky failed" (:308)
        const bool nf = !(fabs((double)y0) <= 1.7e308) || !(fabs((double)y1) <= 1.7e308);
        if (__any(nf || bad) && lane == 0) atomicMin(a.fail, (unsigned long long)idx);
    }
}

// ---------------------------------------------------------------------------
// sum x, sum x x^T (fp64 accumulation of the fp32 columns): workgroup w takes a contiguous slice
// of columns, thread t owns the outputs e = t, t + 256, ... of  prod[K*K] | sum[K]; the partials
// are added in workgroup order by k_colstats_f32_final, whose last block publishes the blob.
// ---------------------------------------------------------------------------
// sum x x^T = X X^T on the 16x16x4 f64 MFMA (the fp32 entries widened: products exact, sums fp64).  One single-wave
// workgroup per (slice of columns, upper 16 x 16 tile): nsl x 36 waves spread over the chip, each loading only the two
// row blocks of its tile, four k-steps (16 columns) of loads in flight.  The diagonal tiles also sum x.  Partial of a
// slice: the tiles in accumulator layout | sum[K]; k_colstats_f32_final adds the slices in order and un-tiles.
// (First form: one output per thread and column on the VALU, 128 partials of 132 KB: 55 us alone, 0.2 ms beside the
// next sampler.  With the select wrapped around each load the loads of a k-step were serialised: see kernels_wg2.h.)
// (T = double: the fp64 K = 128 context -- the same pass over fp64 columns)
template <int K, typename T = float>
__global__ __launch_bounds__(64) void k_colstats_f32(const T *__restrict__ items, int64_t c0, int64_t c1, int nsl,
                                                     double *__restrict__ partials)
{
    constexpr int NT = K / 16, NTRI = NT * (NT + 1) / 2, PARTW = NTRI * 256 + K;
    const int tri = blockIdx.x % NTRI, sl = blockIdx.x / NTRI;
    int I = 0, t = tri;
    while (t >= NT - I) { t -= NT - I; ++I; }
    const int J = I + t;
    const int lane = threadIdx.x, kq = lane >> 4, li = lane & 15;
    const int64_t n = c1 - c0;
    const int64_t per = ((n + nsl - 1) / nsl + 3) / 4 * 4;
    const int64_t b = c0 + sl * per, e = (b + per < c1) ? b + per : c1;
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
    double r = 0.0;
    const T *xi = items + 16 * I + li, *xj = items + 16 * J + li;
    // 16 columns per trip; the loads of the next trip are issued before the MFMAs of the current one
    T fa[4], fb[4], na[4], nb[4];
    auto fetch = [&](int64_t c, T (&a4)[4], T (&b4)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t cc = c + 4 * u + kq;
            const size_t at = (size_t)((cc < e) ? cc : b) * K;       // (beyond the slice: any valid column, masked below)
            a4[u] = xi[at];
            b4[u] = xj[at];
        }
    };
    if (b < e) fetch(b, fa, fb);
    for (int64_t c = b; c < e; c += 16) {
        fetch(c + 16 < e ? c + 16 : b, na, nb);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = c + 4 * u + kq < e;
            const double ya = ok ? (double)fa[u] : 0.0, yb = ok ? (double)fb[u] : 0.0;
            acc = mfma16(ya, yb, acc);
            r += ya;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { fa[u] = na[u]; fb[u] = nb[u]; }
    }
    double *p = partials + (size_t)sl * PARTW;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) p[tri * 256 + reg * 64 + lane] = acc[reg];
    if (I == J) {                                                     // (workgroup-uniform)
        r += __shfl_xor(r, 16);
        r += __shfl_xor(r, 32);
        if (kq == 0) p[NTRI * 256 + 16 * I + li] = r;
    }
}

template <int K>
__global__ __launch_bounds__(256) void k_colstats_f32_final(const double *__restrict__ partials, int nsl,
                                                            const unsigned long long *__restrict__ fail_in,
                                                            double *__restrict__ out, unsigned *ticket, unsigned *flag, unsigned seq)
{
    constexpr int NT = K / 16, NTRI = NT * (NT + 1) / 2, PARTW = NTRI * 256 + K, NOUT = K * K + K;
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o < NOUT) {
        int at;
        if (o < K * K) {                                              // prod(gi, gj) at gi + gj K: from tile (min, max) of the block pair
            int gi = o % K, gj = o / K;
            if (gi / 16 > gj / 16) { const int x = gi; gi = gj; gj = x; }
            const int I = gi / 16, J = gj / 16, ii = gi % 16;
            at = (I * NT - (I * (I - 1)) / 2 + (J - I)) * 256 + (ii >> 2) * 64 + (ii & 3) * 16 + (gj % 16);   // D[kq + 4 reg][li] of the f64 16x16x4 shape
        } else {
            at = NTRI * 256 + (o - K * K);
        }
        double s = 0.0;
        for (int q0 = 0; q0 < nsl; q0 += 8) {                         // eight loads in flight, added in slice order
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)((q0 + u < nsl) ? q0 + u : q0) * PARTW + at];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (q0 + u < nsl) ? v[u] : 0.0;
        }
        __hip_atomic_store(&out[o], s, BPMF_RLX_SYSTEM);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long fw = *fail_in;
        __hip_atomic_store(&out[NOUT], (fw == ~0ull) ? 0.0 : (double)(fw + 1ull), BPMF_RLX_SYSTEM);
        __hip_atomic_store(&reinterpret_cast<unsigned long long *>(out)[NOUT + 1], fw, BPMF_RLX_SYSTEM);
    }
    publish_when_last(ticket, gridDim.x, flag, seq);
}

// The same pass as RIDERS at the head of the next sampler launch (k_sample_wg2, workgroups of NW waves): job = (slice, tile)
// per wave as in k_colstats_f32, partials written through, a ticket per workgroup; the last arrivers wait until every
// partial has landed (every workgroup takes its ticket before it waits: the count completes as soon as all riders have
// run, and they are the first workgroups of the grid) and add the slices in order, one output per thread -- the sums of
// k_colstats_f32_final, bit for bit.  Why: as kernels of their own on the side's stream the pass needed a head start over
// the partner's sampler, bought with two cross-queue event hops of ~15 us each per half-iteration (DESIGN.md section 4,
// "what the K = 128 iteration is made of"); as riders the partner's launch follows its predecessor on ONE queue.
template <int K, int NW, typename T = float>
__device__ __forceinline__ void colstats_f32_rider(const StatRiders &r, int rb, int tid)
{
    constexpr int NT = K / 16, NTRI = NT * (NT + 1) / 2, PARTW = NTRI * 256 + K, NOUT = K * K + K, NTH = 64 * NW;
    __shared__ unsigned stk;
    const int wave = tid >> 6, lane = tid & 63, kq = lane >> 4, li = lane & 15;
    if (r.tail) {
        // tail riders: the columns are this launch's own.  Workgroups are dispatched in grid order, so every work item is
        // resident or finished when a rider starts (no deadlock); their samples were stored write-through ahead of their
        // count and are read with device-scope loads below.
        if (tid == 0) {
            const unsigned long long t0 = wall_clock64();
            // (polled: a word of its own that the LAST item sets to this launch's sequence number -- hundreds of riders polling
            //  the counter itself kept its cache line so busy that the items' increments queued behind the polls: launches
            //  90-110 us longer)
            while (__hip_atomic_load(r.done + 16, BPMF_RLX_AGENT) != r.seq) {
                __builtin_amdgcn_s_sleep(32);
                if (r.wait_ticks && wall_clock64() - t0 > r.wait_ticks) { flag_timeout(r.tmo, BPMF_TMO_STATS); break; }
            }
        }
        __syncthreads();
        // (no acquire fence here: a `buffer_inv sc1` per rider wave -- 1 152 cache invalidates behind one another -- made the
        //  launches ~100 us longer; the columns are read with device-scope loads instead, which cannot hit a stale line)
    }
    const int job = rb * NW + wave;
    if (job < r.nsl * NTRI) {                                         // wave-uniform
        const int tri = job % NTRI, sl = job / NTRI;
        int I = 0, t = tri;
        while (t >= NT - I) { t -= NT - I; ++I; }
        const int J = I + t;
        const int64_t n = r.c1 - r.c0;
        const int64_t per = ((n + r.nsl - 1) / r.nsl + 3) / 4 * 4;
        const int64_t b = r.c0 + sl * per, e = (b + per < r.c1) ? b + per : r.c1;
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
        double rs = 0.0;
        const T *xi = reinterpret_cast<const T *>(r.items) + 16 * I + li, *xj = reinterpret_cast<const T *>(r.items) + 16 * J + li;
        T fa[4], fb[4], na[4], nb[4];
        auto fetch = [&](int64_t c, T (&a4)[4], T (&b4)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t cc = c + 4 * u + kq;
                const size_t at = (size_t)((cc < e) ? cc : b) * K;
                if (r.tail) { a4[u] = __hip_atomic_load(&xi[at], BPMF_RLX_AGENT); b4[u] = __hip_atomic_load(&xj[at], BPMF_RLX_AGENT); }
                else { a4[u] = xi[at]; b4[u] = xj[at]; }
            }
        };
        if (b < e) fetch(b, fa, fb);
        for (int64_t c = b; c < e; c += 16) {
            fetch(c + 16 < e ? c + 16 : b, na, nb);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = c + 4 * u + kq < e;
                const double ya = ok ? (double)fa[u] : 0.0, yb = ok ? (double)fb[u] : 0.0;
                acc = mfma16(ya, yb, acc);
                rs += ya;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { fa[u] = na[u]; fb[u] = nb[u]; }
        }
        double *p = r.partials + (size_t)sl * PARTW;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) __hip_atomic_store(&p[tri * 256 + reg * 64 + lane], acc[reg], BPMF_RLX_AGENT);
        if (I == J) {
            rs += __shfl_xor(rs, 16);
            rs += __shfl_xor(rs, 32);
            if (kq == 0) __hip_atomic_store(&p[NTRI * 256 + 16 * I + li], rs, BPMF_RLX_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's partial has landed
    __syncthreads();
    if (tid == 0) stk = __hip_atomic_fetch_add(r.ticket, 1u, BPMF_RLX_AGENT);
    __syncthreads();
    const int tk = (int)stk;
    const int nfin = r.nblocks < (NOUT + NTH - 1) / NTH ? r.nblocks : (NOUT + NTH - 1) / NTH;
    if (tk < r.nblocks - nfin) return;
    const int f = tk - (r.nblocks - nfin);                            // finisher 0 .. nfin-1
    if (tid == 0) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(r.ticket, BPMF_RLX_AGENT) < (unsigned)r.nblocks) {
            __builtin_amdgcn_s_sleep(1);
            if (r.wait_ticks && wall_clock64() - t0 > r.wait_ticks) { flag_timeout(r.tmo, BPMF_TMO_STATS); break; }
        }
    }
    __syncthreads();
    for (int o = f * NTH + tid; o < NOUT; o += nfin * NTH) {
        int at;
        if (o < K * K) {                                              // prod(gi, gj) at gi + gj K: from tile (min, max) of the block pair
            int gi = o % K, gj = o / K;
            if (gi / 16 > gj / 16) { const int x = gi; gi = gj; gj = x; }
            const int I = gi / 16, J = gj / 16, ii = gi % 16;
            at = (I * NT - (I * (I - 1)) / 2 + (J - I)) * 256 + (ii >> 2) * 64 + (ii & 3) * 16 + (gj % 16);
        } else {
            at = NTRI * 256 + (o - K * K);
        }
        double s = 0.0;
        for (int q0 = 0; q0 < r.nsl; q0 += 8) {                       // eight loads in flight, added in slice order (as k_colstats_f32_final)
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __hip_atomic_load(&r.partials[(size_t)((q0 + u < r.nsl) ? q0 + u : q0) * PARTW + at], BPMF_RLX_AGENT);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (q0 + u < r.nsl) ? v[u] : 0.0;
        }
        __hip_atomic_store(&r.out[o], s, BPMF_RLX_SYSTEM);
    }
    if (f == 0 && tid == 0) {
        const unsigned long long fw = *r.fail_in;
        __hip_atomic_store(&r.out[NOUT], (fw == ~0ull) ? 0.0 : (double)(fw + 1ull), BPMF_RLX_SYSTEM);
        __hip_atomic_store(&reinterpret_cast<unsigned long long *>(r.out)[NOUT + 1], fw, BPMF_RLX_SYSTEM);
    }
    publish_when_last(r.ticket + 1, (unsigned)nfin, r.flag, r.seq, r.ticket, r.tail ? r.done : nullptr);
}

// ---------------------------------------------------------------------------
// Sys::predict (c++/sample.cpp:48-96) on fp32 factors: fp64 accumulation of the dot product and
// of everything behind it; same partial / publish scheme as k_predict.
// ---------------------------------------------------------------------------
template <int K>
__global__ __launch_bounds__(256) void k_predict_f32(const int32_t *__restrict__ tcol, const int32_t *__restrict__ trow,
                                                     const double *__restrict__ tval, int64_t nnz,
                                                     const float *__restrict__ items, const float *__restrict__ other,
                                                     int64_t col_from, double mean, int n, double *__restrict__ pavg,
                                                     double *__restrict__ pm2, double *partial, double *__restrict__ out,
                                                     unsigned *ticket, unsigned *flag, unsigned seq)
{
    __shared__ double red[2][4];
    __shared__ double fin[2][256];
    __shared__ unsigned last;
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double se = 0.0, se_avg = 0.0;
    if (q < nnz) {
        const float4 *m = reinterpret_cast<const float4 *>(items + (size_t)(col_from + tcol[q]) * K);
        const float4 *u = reinterpret_cast<const float4 *>(other + (size_t)trow[q] * K);
        double d0 = 0.0, d1 = 0.0;
#pragma unroll 8
        for (int t = 0; t < K / 4; ++t) {
            const float4 x = m[t], y = u[t];
            d0 = fma((double)x.x, (double)y.x, d0);
            d1 = fma((double)x.y, (double)y.y, d1);
            d0 = fma((double)x.z, (double)y.z, d0);
            d1 = fma((double)x.w, (double)y.w, d1);
        }
        const double pred = (d0 + d1) + mean;                       // :78
        const double v = tval[q];
        se = (v - pred) * (v - pred);
        double avg = pavg[q];
        const double delta = pred - avg;
        avg = (n == 0) ? pred : (avg + delta / n);                  // :84 (n, not n+1: reference quirk)
        pavg[q] = avg;
        pm2[q] = (n == 0) ? 0.0 : pm2[q] + delta * (pred - avg);    // :86
        se_avg = (v - avg) * (v - avg);
    }
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
        se += __shfl_xor(se, sh);
        se_avg += __shfl_xor(se_avg, sh);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = se; red[1][wv] = se_avg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&partial[2 * blockIdx.x], (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), BPMF_RLX_AGENT);
        __hip_atomic_store(&partial[2 * blockIdx.x + 1], (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]), BPMF_RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, BPMF_RLX_AGENT);
        last = (t == gridDim.x - 1) ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    const int64_t nblocks = gridDim.x;
    double sa = 0.0, sb = 0.0;
    for (int64_t wgi = threadIdx.x; wgi < nblocks; wgi += 256) {
        sa += __hip_atomic_load(&partial[2 * wgi], BPMF_RLX_AGENT);
        sb += __hip_atomic_load(&partial[2 * wgi + 1], BPMF_RLX_AGENT);
    }
    fin[0][threadIdx.x] = sa; fin[1][threadIdx.x] = sb;
    __syncthreads();
    for (int st = 128; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) { fin[0][threadIdx.x] += fin[0][threadIdx.x + st]; fin[1][threadIdx.x] += fin[1][threadIdx.x + st]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&out[0], fin[0][0], BPMF_RLX_SYSTEM);
        __hip_atomic_store(&out[1], fin[1][0], BPMF_RLX_SYSTEM);
        __hip_atomic_store(ticket, 0u, BPMF_RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(flag, seq, BPMF_RLX_SYSTEM);
    }
}

}  // namespace bpmf
