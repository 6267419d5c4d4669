// kernels_wg2.h -- k_sample_wg2<K, NW, T>: the large-K column update (K = 128), second form.  T = float: the fp32
// mixed-precision path (BASELINE configs[4]); T = double: the reference's own fp64 arithmetic at num_latent 65 .. 128
// (`bpmf-128`, `bpmf-100` ... of ci/multilatent.sh:5) -- same frame, fp64 factors, Gram on v_mfma_f64_16x16x4_f64.
//
// Reference: Sys::sample(long idx, Sys&) + computeMuLambda, c++/sample.cpp:248-336.
//
// k_sample_wg (kernels_f32.h) spent ~57 of the ~60 us of a column in serial chains: 128 sequential pivots of the
// 16x16 diagonal blocks (column c in lane c, v_readlane per entry), one-thread-per-column panel solves, and two
// 128-step triangular solves.  Same frame here -- one workgroup of NW waves per work item, Gram on
// v_mfma_f32_16x16x4_f32 with the tiles dealt round-robin to the waves, R by block rows in LDS -- but:
//   * diagonal block s (16 x 16): widened to fp64 and factored AND inverted by wave 0 with the slab scheme of
//     kernels_slab.h on v_mfma_f64_4x4x4_4b_f64: four steps of a 4x4 pivot block instead of sixteen scalar pivots,
//     the identity riding along as a second slab column, so that R_ss^-T = W_s^T falls out with the factor;
//   * panel  R_sJ = W_s^T A_sJ : four v_mfma_f32_16x16x4_f32 per tile, by the tile's owner (operands from LDS);
//   * forward solve (:321) block by block inside the factorisation: y_s = W_s^T b_s (a 16 x 16 product), then
//     b_J -= R_sJ^T y_s, one thread per remaining entry of b;
//   * backward solve (:323) in 8 block steps: t = y_s - sum_J R_sJ x_J (64 lanes: 16 rows x 4 column groups),
//     x_s = W_s t; the W_s^T stay in LDS (8 x 1 KB);
//   * heavy columns are cut into chunks (partials = the waves' tiles, last workgroup to arrive adds them in chunk
//     order): a 3 000-rating column alone used to be ~0.24 ms of a launch.
// Everything that leaves the column loop stays fp64 (normals, hyper-parameters, statistics), as in kernels_f32.h.
#pragma once
#include <type_traits>
#include "kernels_f32.h"
#include "kernels_slab.h"

namespace bpmf {

// f(integral_constant<int, B>) ... f(integral_constant<int, E - 1>): loops whose index has to be a constant expression
// (tile ownership and register slots of the factorisation are decided per (s, I, J) at compile time)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

template <int K>
struct GeoW2 {
    using F = GeoF<K>;
    static constexpr int NT = F::NT;
    // LDS (floats unless noted): zs [K doubles] | R by block rows | b / y, later x [K] | t [16] | ticket.  W_s^T = R_ss^-T takes
    // the place of the diagonal block R_ss in block row s (nothing reads R_ss once it is inverted), and x_s overwrites
    // y_s in the backward solve: 40 528 B, FOUR workgroups per CU (with W^T and x on their own: 49 232 B, three)
    // (T = double: 80 032 B, two workgroups per CU)
    template <typename T = float> static constexpr size_t lds_bytes() { return (size_t)K * 8 + ((size_t)F::RWORDS + K + 16 + 4) * sizeof(T); }
    static constexpr int PART_FLOATS = F::NTRI * 256 + NT * 16;   // partial of one chunk: all tiles + rhs (elements of T)
};

// Who holds tile (I, J) DURING THE FACTORISATION (round 5, the look-ahead form: see wg2_column).  Wave 0 holds none -- it
// is the chain of diagonal blocks -- and the NW - 1 others hold the tiles of the block rows 1 .. NT - 1 in registers
// (block row 0 is the first pivot row: it stays in LDS) and compute the panels of the tiles they are named for.
// (s, s + 1) and (s + 1, s + 1) are with the same wave: panel -> update -> park of the next diagonal block is one
// wave's straight line.  NW = 4: the table balances the trailing updates of every step over the three waves
// (9 / 9 / 9 tiles at s = 0, 7 / 6 / 7, 5 / 5 / 4, 3 / 3 / 3 ...: tools/own_table.py); NW = 2: wave 1 holds all 28.
template <int NT, int NW>
struct FactOwner {
    static_assert(NT == 8 && (NW == 2 || NW == 4), "K = 128");
    __host__ __device__ static constexpr int of(int I, int J)
    {
        if (NW == 2) return 1;
        constexpr int tab[8][8] = {{0, 1, 3, 3, 2, 1, 2, 1}, {0, 1, 2, 1, 1, 2, 3, 3}, {0, 0, 2, 3, 1, 2, 1, 3}, {0, 0, 0, 3, 1, 2, 2, 3},
                                   {0, 0, 0, 0, 1, 2, 3, 1}, {0, 0, 0, 0, 0, 2, 3, 2}, {0, 0, 0, 0, 0, 0, 3, 1}, {0, 0, 0, 0, 0, 0, 0, 1}};
        return tab[I][J];
    }
    // register slot of tile (I, J), I >= 1, among its wave's tiles (row-major order)
    __host__ __device__ static constexpr int slot(int I, int J)
    {
        int n = 0;
        for (int i = 1; i < NT; ++i)
            for (int j = i; j < NT; ++j) {
                if (i == I && j == J) return n;
                if (of(i, j) == of(I, J)) ++n;
            }
        return n;
    }
    __host__ __device__ static constexpr int count(int W)
    {
        int n = 0;
        for (int i = 1; i < NT; ++i)
            for (int j = i; j < NT; ++j) n += of(i, j) == W;
        return n;
    }
    __host__ __device__ static constexpr int tiles() { int m = 0; for (int w = 1; w < NW; ++w) m = count(w) > m ? count(w) : m; return m; }
};

// 16 x 16 SPD block (upper part used), given as four fp64 slabs A[I]: lane (kq, c) <-> D[4 I + kq][c].
// Out: A[I] = rows of R (D = R^T R; entries left of the diagonal are NOT cleaned), E[I] = rows of R^-T = W^T.
// MIDBAR >= 0: the wave passes ONE workgroup barrier after pivot block MIDBAR (wg2_column's barrier Y: this wave is the
// only one of its workgroup here, the others reach that barrier from their panels).
template <int MIDBAR = -1>
__device__ __forceinline__ void diag16_factor_invert(double (&A)[4], double (&E)[4], int lane)
{
    const int kq = lane >> 4, x = lane & 3, c16 = lane & 15;
    const W44Select wsel(kq, x);
#pragma unroll
    for (int I = 0; I < 4; ++I) E[I] = (4 * I + kq == c16) ? 1.0 : 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double dblk = A[s];
        const double d00 = bcast(dblk, 4 * s + 0), d01 = bcast(dblk, 4 * s + 1), d02 = bcast(dblk, 4 * s + 2), d03 = bcast(dblk, 4 * s + 3),
                     d11 = bcast(dblk, 16 + 4 * s + 1), d12 = bcast(dblk, 16 + 4 * s + 2), d13 = bcast(dblk, 16 + 4 * s + 3),
                     d22 = bcast(dblk, 32 + 4 * s + 2), d23 = bcast(dblk, 32 + 4 * s + 3), d33 = bcast(dblk, 48 + 4 * s + 3);
        double WA;
        factor_block44(d00, d01, d02, d03, d11, d12, d13, d22, d23, d33, wsel, WA);
        A[s] = mfma44(WA, A[s], 0.0);                                 // R_sJ = W^T A_sJ for the four J of the slab
        E[s] = mfma44(WA, E[s], 0.0);
        if (s == MIDBAR) __syncthreads();
        if (s == 3) break;
        // A operand of row block I: quad I of slab s splatted over the quads: lane (k, b, i) <- R[4 s + k][4 I + i]
        double n1 = 0.0, n2 = 0.0, n3 = 0.0;
        if (s < 1) n1 = -quad_splat<1>(A[s]);
        if (s < 2) n2 = -quad_splat<2>(A[s]);
        n3 = -quad_splat<3>(A[s]);
        if (s < 1) { A[1] = mfma44(n1, A[s], A[1]); E[1] = mfma44(n1, E[s], E[1]); }
        if (s < 2) { A[2] = mfma44(n2, A[s], A[2]); E[2] = mfma44(n2, E[s], E[2]); }
        A[3] = mfma44(n3, A[s], A[3]); E[3] = mfma44(n3, E[s], E[3]);
    }
}

// per-wave part of one work item: Gram of the wave's tiles, then (whole column / last chunk) the factorisation
template <int K, int NW, int W, typename T = float>
__device__ __forceinline__ void wg2_column(const SampleArgs &a, int w, unsigned char *smem, int tid)
{
    using G = GeoF<K>;
    using X = WgTraits<T>;
    typedef typename X::acc_t acc_t;
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int NT = G::NT, TPW = (G::NTRI + NW - 1) / NW;
    double *zs = reinterpret_cast<double *>(smem);
    T *R = reinterpret_cast<T *>(zs + K);
    T *bv = R + G::RWORDS, *xs = bv, *ts = bv + K;
    unsigned *sticket = reinterpret_cast<unsigned *>(ts + 16);
    const int lane = tid & 63;
    const int kq = lane >> 4, li = lane & 15;
    const int col = a.wi_col[w];
    const int64_t p0 = a.wi_p0[w];
    const int len = (ablate_bits(a) & 2u) ? 0 : a.wi_len[w];             // (profiling switch: no Gram)
    const int mc = a.wi_mc[w];
    const int64_t idx = a.col_from + col;
    const T *other = reinterpret_cast<const T *>(a.other_items);

    stamp(a, w, 0);

    acc_t acc[TPW];
    T r[NT];
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = acc_t{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NT; ++t) r[t] = 0;
    // Lambda* = LambdaF + alpha G in the register tiles (:297-298).  A whole column (round 5) starts its accumulators at
    // LambdaF / alpha -- the loads of the prior, 2.7 us of L2 round trips per item when they came after the Gram, are in flight
    // beside the first gathers -- and multiplies by alpha at the end; the last chunk of a heavy column, which sums partials,
    // adds the prior as before.  (alpha = 2, the reference's default: LambdaF / alpha and the product are exact; see `whole`.)
    const double *LF = a.prop_lambda ? a.prop_lambda + (size_t)col * K * K : a.LambdaF;
    bool lf_tiles = false;
    if constexpr (F32) lf_tiles = a.lf32 != nullptr;
    const double inv_alpha = 1.0 / a.alpha;
    auto prior = [&](auto firstc) {
        constexpr bool first = decltype(firstc)::value;
    if constexpr (F32) if (lf_tiles) {                                // (workgroup-uniform) the prior as fp32 tiles: one 16-byte load per tile
        const f4 *lt = reinterpret_cast<const f4 *>(a.lf32);
        const T alpha_f = (T)a.alpha, inv_alpha_f = (T)(1.0 / a.alpha);
        f4 lf[TPW];
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int J = I; J < NT; ++J)
                if ((G::tri(I, J) % NW) == W) lf[G::tri(I, J) / NW] = lt[G::tri(I, J) * 64 + lane];
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int J = I; J < NT; ++J)
                if ((G::tri(I, J) % NW) == W) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const T v = first ? lf[G::tri(I, J) / NW][reg] * inv_alpha_f : fmaf(alpha_f, acc[G::tri(I, J) / NW][reg], lf[G::tri(I, J) / NW][reg]);
                        acc[G::tri(I, J) / NW][reg] = (!first && a.diag_only && (16 * I + 4 * kq + reg) != (16 * J + li)) ? (T)0 : v;
                    }
                }
    }
    if (!lf_tiles) {
        // LambdaF(gj, gi): the lower triangle, which is what LLT reads (:306); 16 lanes = one 128-byte line.  The loads of
        // a batch of tiles are issued together and without control flow around them (with the diag_only select wrapped
        // around each load they were 72 serialised L2 round trips: 14.6 of the ~50 us a mid-size column lived)
        constexpr int BATCH = first ? TPW : 6;                       // (before the Gram the registers are free: every load in one batch)
#pragma unroll
        for (int t0 = 0; t0 < TPW; t0 += BATCH) {
            double lf[BATCH][4];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int tri = (t0 + u) * NW + W;
                if (t0 + u < TPW && tri < G::NTRI) {
                    const int I = G::tile_i(tri), J = G::tile_j(tri);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) lf[u][reg] = LF[(16 * J + li) + (size_t)(16 * I + X::drow(kq, reg)) * K];
                }
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int tri = (t0 + u) * NW + W;
                if (t0 + u < TPW && tri < G::NTRI) {
                    const int I = G::tile_i(tri), J = G::tile_j(tri);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const T v = first ? (T)(lf[u][reg] * inv_alpha) : (T)fma(a.alpha, (double)acc[t0 + u][reg], lf[u][reg]);
                        acc[t0 + u][reg] = (!first && a.diag_only && (16 * I + X::drow(kq, reg)) != (16 * J + li)) ? (T)0 : v;
                    }
                }
            }
        }
    }
    };
    // (not where the prior comes as fp32 tiles, one 16-byte load per tile and lane: that is a single round trip, and starting
    // from it measured no gain -- 289 / 352 against 291 / 354 us per launch)
    // Only where it is EXACT: alpha a positive power of two (mantissa bits all zero; the reference's default 2), so that
    // LambdaF / alpha and alpha * (LambdaF / alpha + G) round as LambdaF + alpha G does.  Any other -a F (1.5, 3, 10 ...) would put
    // two more roundings on every entry, and a different result by whether a column was chunked: those take the prior after the
    // Gram, fma(alpha, G, LambdaF), like a chunked column and like every other kernel (tests/test_gpu_alpha.py).
    const bool alpha_pow2 = a.alpha > 0.0 && (__double_as_longlong(a.alpha) & 0x000FFFFFFFFFFFFFll) == 0 &&
                            a.alpha >= 0x1p-500 && a.alpha <= 0x1p500;
    const bool whole = mc < 0 && alpha_pow2 && !lf_tiles;           // (workgroup-uniform)
    {
        const int32_t *rowidx = a.rowidx + p0;
        const double *vals = a.vals + p0;
        // 64 ratings per coalesced index block (lane l holds rating b0 + l) = 4 groups of 4 k-steps (16 ratings).
        // Round 5: the workgroup gathers a group ONCE.  Every wave needs (nearly) every tile row of a factor row as an operand,
        // and with each wave gathering for itself the same K elements per rating crossed L2 -> L1 NW times (the in-flight rows of
        // a CU are 8 x its L1): the Gram phase without its MFMAs took 255 - 280 us per launch in fp64, 85 - 120 in fp32, against
        // 104 - 124 / 60 - 84 with 1 / NW of the loads (tools/exp_halfload.sh).  Wave W gathers the k-steps W, W + NW, ... of the
        // NEXT group while the MFMAs of the current one issue, parks them in LDS (16-byte pieces, lane-contiguous: no bank
        // conflicts; two buffers in the space R takes after the Gram), one barrier per group, and every wave reads its
        // operands of the four k-steps from there.  Same operands, same order of the MFMAs per accumulator: same sums.
        // No control flow around the loads (exact s_waitcnt counts); slots beyond the end of the chunk gather a row of zeros.
        int ri = (lane < len) ? rowidx[lane] : -1;
        T wv = (lane < len) ? (T)((vals[lane] - a.mean_rating) * a.alpha) : (T)0;                 // c++/sample.cpp:256
        int ri_n = (64 + lane < len) ? rowidx[64 + lane] : -1;
        T wv_n = (64 + lane < len) ? (T)((vals[64 + lane] - a.mean_rating) * a.alpha) : (T)0;
        const int rowmask = (ablate_bits(a) & 4u) ? 63 : -1;                 // (profiling switch: gather from 64 hot rows only)
        const bool no_mfma = (ablate_bits(a) & 8u) != 0;                     // (profiling switch: operands are loaded and summed, no MFMA)
        constexpr int SPW = 4 / NW;                                    // k-steps of a group gathered by one wave
        constexpr int EPV = 16 / (int)sizeof(T), VPR = NT / EPV;       // elements per 16-byte piece, pieces per lane and k-step
        typedef T tvec __attribute__((ext_vector_type(EPV)));
        constexpr int SBUF = 16 * K;                                   // elements of one staging buffer: 16 ratings x K
        static_assert(2 * SBUF <= G::RWORDS, "the staging buffers live where R will");
        T pre[SPW][NT];
        auto fetch = [&](int gg) {                                     // gg = 1 .. 3: group of this index block, 4: group 0 of the next
            const bool nx = gg >= 4;
#pragma unroll
            for (int u = 0; u < SPW; ++u) {
                const int st = W + u * NW;
                const int src = ((gg & 3) * 4 + st) * 4 + kq;
                const int row = __shfl(nx ? ri_n : ri, src);
                const T *up = ((row >= 0) ? other + (size_t)(row & rowmask) * K : reinterpret_cast<const T *>(a.zero_row)) + li;
#pragma unroll
                for (int t = 0; t < NT; ++t) pre[u][t] = up[16 * t];
            }
        };
        auto stage = [&](int buf) {                                    // piece v of k-step st: [st][v][lane] x 16 bytes
#pragma unroll
            for (int u = 0; u < SPW; ++u) {
                const int st = W + u * NW;
#pragma unroll
                for (int v = 0; v < VPR; ++v) {
                    tvec x;
#pragma unroll
                    for (int e = 0; e < EPV; ++e) x[e] = pre[u][v * EPV + e];
                    reinterpret_cast<tvec *>(R + buf * SBUF)[(st * VPR + v) * 64 + lane] = x;
                }
            }
        };
        auto step = [&](int buf, int st, T w1) {                       // one k-step (4 ratings): operands from LDS, rhs, MFMAs
            T yy[NT];
#pragma unroll
            for (int v = 0; v < VPR; ++v) {
                const tvec x = reinterpret_cast<const tvec *>(R + buf * SBUF)[(st * VPR + v) * 64 + lane];
#pragma unroll
                for (int e = 0; e < EPV; ++e) yy[v * EPV + e] = x[e];
            }
            if (W == 0 || no_mfma) {
#pragma unroll
                for (int t = 0; t < NT; ++t) r[t] = fma(yy[t], w1, r[t]);
            }
            if (no_mfma) return;
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int J = I; J < NT; ++J)
                    if ((G::tri(I, J) % NW) == W)
                        acc[G::tri(I, J) / NW] = X::mfma(yy[I], yy[J], acc[G::tri(I, J) / NW]);
        };
        const int ngr = (len + 15) >> 4;                               // groups of 16 ratings (the same in every wave)
        if (ngr > 0) fetch(0);
        if (whole) prior(std::true_type{});                            // (its loads fly with the first gathers')
        if (ngr > 0) stage(0);
        __syncthreads();
        for (int g = 0; g < ngr; ++g) {
            const int gl = g & 3;
            const bool more = g + 1 < ngr;
            if (more) fetch(gl + 1);
            T w1[4] = {0, 0, 0, 0};
            if (W == 0 || no_mfma) {
#pragma unroll
                for (int st = 0; st < 4; ++st) w1[st] = __shfl(wv, (gl * 4 + st) * 4 + kq);
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) step(g & 1, st, w1[st]);
            if (more) stage((g + 1) & 1);
            if (gl == 3) {                                             // next index block
                ri = ri_n; wv = wv_n;
                const int q = (g >> 2) * 64 + 128 + lane;
                ri_n = (q < len) ? rowidx[q] : -1;
                wv_n = (q < len) ? (T)((vals[q] - a.mean_rating) * a.alpha) : (T)0;
            }
            __syncthreads();
        }
        if (W == 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                r[t] += __shfl_xor(r[t], 16);
                r[t] += __shfl_xor(r[t], 32);
            }
        }
    }

    // the rest of the item is a latency-bound chain of short VALU / LDS / MFMA steps: let it win the SIMD's issue
    // arbitration over the co-resident workgroups' Gram loops (throughput-bound, they only lose slots they can spare)
    __builtin_amdgcn_s_setprio(3);
    stamp(a, w, 1);
    if (kProfiling && W == 1 && a.stamps && lane == 0 && (w == 0 || w == a.nwork / 2)) a.stamps[(w == 0 ? 0 : 64) + 48] = wall_clock64();   // (wave 1's Gram ends)
    if (mc >= 0) {
        // chunk of a heavy column: every wave parks its tiles (tile `tri` at [tri * 256 + reg * 64 + lane]), wave 0 the rhs;
        // the workgroup that draws the last ticket adds the partials in chunk order
        constexpr int PF = GeoW2<K>::PART_FLOATS;
        const int nch = a.mc_nchunks[mc];
        T *pbase = reinterpret_cast<T *>(a.partials) + (size_t)a.mc_slot0[mc] * PF;
        T *p = pbase + (size_t)a.wi_chunk[w] * PF;
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int J = I; J < NT; ++J)
                if ((G::tri(I, J) % NW) == W) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) __hip_atomic_store(&p[G::tri(I, J) * 256 + reg * 64 + lane], acc[G::tri(I, J) / NW][reg], BPMF_RLX_AGENT);
                }
        if (W == 0 && lane < 16) {
#pragma unroll
            for (int t = 0; t < NT; ++t) __hip_atomic_store(&p[G::NTRI * 256 + t * 16 + lane], r[t], BPMF_RLX_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) *sticket = __hip_atomic_fetch_add(&a.mc_count[mc], 1u, BPMF_RLX_AGENT);
        __syncthreads();
        const unsigned tk = *sticket;
        if ((int)tk != nch - 1) return;                               // (the whole workgroup)
        if (tid == 0) __hip_atomic_store(&a.mc_count[mc], 0u, BPMF_RLX_AGENT);
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = acc_t{0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < NT; ++t) r[t] = 0;
        for (int ch = 0; ch < nch; ++ch) {                            // fixed chunk order: deterministic
            const T *pc = pbase + (size_t)ch * PF;
#pragma unroll
            for (int I = 0; I < NT; ++I)
#pragma unroll
                for (int J = I; J < NT; ++J)
                    if ((G::tri(I, J) % NW) == W) {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) acc[G::tri(I, J) / NW][reg] += __hip_atomic_load(&pc[G::tri(I, J) * 256 + reg * 64 + lane], BPMF_RLX_AGENT);
                    }
            if (W == 0) {
#pragma unroll
                for (int t = 0; t < NT; ++t) r[t] += __hip_atomic_load(&pc[G::NTRI * 256 + t * 16 + li], BPMF_RLX_AGENT);
            }
        }
    }

    if (ablate_bits(a) & 1u) {                                              // (profiling switch: Gram only -- keep it live)
        T v = r[0];
#pragma unroll
        for (int t = 0; t < TPW; ++t) v += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
        if (lane < 16) reinterpret_cast<T *>(a.items)[(size_t)idx * K + 16 * W + lane] = v;
        return;
    }
    // b = LambdaF mu + rr (:285,:256) below; Lambda* = LambdaF + alpha G (:297-298)
    if (whole) {
#pragma unroll
        for (int I = 0; I < NT; ++I)
#pragma unroll
            for (int J = I; J < NT; ++J)
                if ((G::tri(I, J) % NW) == W) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const T v = (T)a.alpha * acc[G::tri(I, J) / NW][reg];
                        acc[G::tri(I, J) / NW][reg] = (a.diag_only && (16 * I + X::drow(kq, reg)) != (16 * J + li)) ? (T)0 : v;
                    }
                }
    } else {
        prior(std::false_type{});
    }
    stamp(a, w, 42);
    if (W == 0 && kq == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            double lm = a.Lmu[16 * t + li];
            if (a.prop_lambda) {                                     // rr = Lambda_i * hp.mu (:285)
                lm = 0.0;
                for (int j = 0; j < K; ++j) lm = fma(LF[16 * t + li + (size_t)j * K], a.mu[j], lm);
            }
            bv[16 * t + li] = (T)(lm + (double)r[t]);
        }
    }

    // ---- factorisation Lambda* = R^T R with the forward solve riding along, LOOK-AHEAD form (round 5) ------------------
    // The form of rounds 2-4 ran every step as  park row s | B: wave 0 factors + inverts the diagonal block, the others
    // wait | C: panels | D: trailing update  with a barrier after each: ~3.3 us per step, 1.5 of them with one wave at
    // work.  Here the diagonal blocks are a chain of their own on wave 0, and block s + 1 is factored WHILE the other
    // waves run the trailing update of step s:
    //   * the Gram's deal of the tiles ends after the assembly: every tile goes to its place in R (LDS holds the whole
    //     upper triangle anyway), and the waves 1 .. NW - 1 take the tiles FactOwner names them for back into registers
    //     (wave 0 none; block row 0 stays in LDS);
    //   * step s, between the barriers X_s (W_s^T and y_s in LDS, block row s parked) and X_{s+1}:
    //       owner of (s, s+1):  panel R_s,s+1 -> LDS, A_s+1,s+1 -= R^T R in its registers, park it      | barrier Z
    //       wave 0:             (b_s+1 -= R_s,s+1^T y_s;) pivot block 0 (NW = 2: 0, 1) of diagonal block s + 1 | barrier Y |
    //                           the other pivot blocks, W_s+1^T -> LDS, y_s+1 = W_s+1^T b_s+1
    //       waves 1 .. NW - 1:  their other panels of block row s | barrier Y | their share of the rhs, the trailing
    //                           update of their tiles, park their tiles of block row s + 1
    //     -- three barriers per step as before, but the 1.5 us of a diagonal block now run beside the trailing update.
    // Every tile sees the same operations on the same operands in the same order as in the old form (panel: four MFMAs
    // over k = 4 kq + q; update: steps 0, 1, ... in turn; rhs: sixteen FMAs per entry and step): bit-identical samples.
    using FO = FactOwner<NT, NW>;
    constexpr int MIDBAR = NW == 2 ? 1 : 0;
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int J = I; J < NT; ++J)
            if ((G::tri(I, J) % NW) == W) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) R[G::roff(I) + X::drow(kq, reg) * G::ld(I) + 16 * (J - I) + li] = acc[G::tri(I, J) / NW][reg];
            }
    __syncthreads();
    stamp(a, w, 2);

    // wave 0: diagonal block s in fp64 on the 4x4x4 shape -- R_ss (upper) and W_s^T = R_ss^-T, which takes the place of
    // R_ss in LDS -- then y_s = W_s^T b_s (c++/sample.cpp:321); s > 0: block s of b first takes the update of step s - 1
    auto diag_step = [&](auto sc, auto mid) {
        constexpr int s = decltype(sc)::value, MID = decltype(mid)::value;
        T *Rs = R + G::roff(s);
        constexpr int LDs = G::ld(s);
        double A16[4], E[4];
#pragma unroll
        for (int I = 0; I < 4; ++I) A16[I] = (double)Rs[(4 * I + kq) * LDs + li];
        diag16_factor_invert<MID>(A16, E, lane);
#pragma unroll
        for (int I = 0; I < 4; ++I) Rs[(4 * I + kq) * LDs + li] = (T)E[I];   // W_s^T [row][li] (lower triangular)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if constexpr (s > 0) {
            const T *Rp = R + G::roff(s - 1);
            constexpr int LDp = G::ld(s - 1);
            if (lane < 16) {                                          // b_j -= sum_k R_s-1[k][j] y_s-1[k], j in block s
                T sacc = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) sacc = fma(Rp[k * LDp + 16 + lane], bv[16 * (s - 1) + k], sacc);
                bv[16 * s + lane] -= sacc;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        T ysum = 0;                                                   // (the wave is in lockstep: no barrier)
        if (lane < 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k) ysum = fma(Rs[lane * LDs + k], bv[16 * s + k], ysum);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < 16) bv[16 * s + lane] = ysum;
    };

    acc_t tl[FO::tiles()];                                            // this wave's tiles in the factorisation (wave 0: none)
    if constexpr (W == 0) {
        diag_step(std::integral_constant<int, 0>{}, std::integral_constant<int, -1>{});
    } else {
        static_for<1, NT>([&](auto Ic) {
            constexpr int I = decltype(Ic)::value;
            static_for<I, NT>([&](auto Jc) {
                constexpr int J = decltype(Jc)::value;
                if constexpr (FO::of(I, J) == W) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) tl[FO::slot(I, J)][reg] = R[G::roff(I) + X::drow(kq, reg) * G::ld(I) + 16 * (J - I) + li];
                }
            });
        });
        // z ~ N(0, I): stream (idx+1)*K*(iter+1) mod 2^32 (c++/sample.cpp:266) -- by the last wave while wave 0 factors the
        // first diagonal block (the backward solve is what reads z)
        if constexpr (W == NW - 1) draw_normals<K>(sample_counter(idx, a.ktrue, a.iter_plus_1), a.ktrue, zs, lane, K);
    }
    __syncthreads();                                                  // X_0
    stamp(a, w, 3);

    static_for<0, NT - 1>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        T *Rs = R + G::roff(s);
        constexpr int LDs = G::ld(s), Ws = G::width(s);
        if constexpr (W == 0) {
            __syncthreads();                                          // Z_s: diagonal block s + 1 is final and parked
            stamp(a, w, 4 + 4 * s);
            diag_step(std::integral_constant<int, s + 1>{}, std::integral_constant<int, MIDBAR>{});   // (passes Y_s)
            stamp(a, w, 5 + 4 * s);
            __syncthreads();                                          // X_{s+1}
            stamp(a, w, 6 + 4 * s);
        } else {
            // Contraction index of MFMA q, lane group kq: k = 4 kq + q (not 4 q + kq).  The B operands -- the bulk of the
            // LDS reads of the factorisation -- then come from rows 4 apart, whose offsets 4 (width + 4) = 16 mod 64 words
            // put the four 16-lane groups on disjoint banks; with consecutive rows (offset width + 4 = 4 mod 64) the groups
            // overlapped in 12 of 16 banks (r03 PMC: 20.6 % of the LDS cycles were bank conflicts).  Any order of k is the
            // same sum up to rounding; the order is fixed, so the result is as reproducible as before.
            T opA[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) opA[q] = Rs[li * LDs + 4 * kq + q];           // A[i = li][k = 4 kq + q] of W_s^T
            auto panel = [&](auto Jc) {                               // R_sJ = W_s^T A_sJ, in place
                constexpr int J = decltype(Jc)::value;
                T opB[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) opB[q] = Rs[(4 * kq + q) * LDs + 16 * (J - s) + li];
                acc_t t4 = acc_t{0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 4; ++q) t4 = X::mfma(opA[q], opB[q], t4);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) Rs[X::drow(kq, reg) * LDs + 16 * (J - s) + li] = t4[reg];
            };
            if constexpr (FO::of(s, s + 1) == W) {
                // the straight line to the next diagonal block: panel (s, s+1), A_s+1,s+1 -= R^T R, park
                panel(std::integral_constant<int, s + 1>{});
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                T opI[4], opJ[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { opJ[q] = Rs[(4 * kq + q) * LDs + 16 + li]; opI[q] = -opJ[q]; }
                constexpr int SL = FO::slot(s + 1, s + 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) tl[SL] = X::mfma(opI[q], opJ[q], tl[SL]);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) R[G::roff(s + 1) + X::drow(kq, reg) * G::ld(s + 1) + li] = tl[SL][reg];
            }
            __syncthreads();                                          // Z_s
            stamp(a, w, 4 + 4 * s);
            static_for<s + 2, NT>([&](auto Jc) {
                constexpr int J = decltype(Jc)::value;
                if constexpr (FO::of(s, J) == W) panel(Jc);
            });
            __syncthreads();                                          // Y_s: block row s of R is complete
            stamp(a, w, 5 + 4 * s);
            // rhs: b_j -= sum_k R_s[k][j] y_s[k] for the entries right of block s + 1 (wave 0 takes block s + 1), one thread each
            for (int cc = 32 + tid - 64; cc < Ws; cc += 64 * (NW - 1)) {
                T sacc = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) sacc = fma(Rs[k * LDs + cc], bv[16 * s + k], sacc);
                bv[16 * s + cc] -= sacc;
            }
            // trailing update of this wave's tiles  A_IJ -= R_sI^T R_sJ
            static_for<s + 1, NT>([&](auto Ic) {
                constexpr int I = decltype(Ic)::value;
                constexpr bool any = [] {
                    bool f = false;
                    for (int J = I; J < NT; ++J) f |= FO::of(I, J) == W && !(I == s + 1 && J == s + 1);
                    return f;
                }();
                if constexpr (any) {
                    T opI[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) opI[q] = -Rs[(4 * kq + q) * LDs + 16 * (I - s) + li];     // (k = 4 kq + q: see the panel)
                    static_for<I, NT>([&](auto Jc) {
                        constexpr int J = decltype(Jc)::value;
                        if constexpr (FO::of(I, J) == W && !(I == s + 1 && J == s + 1)) {
                            T opJ[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) opJ[q] = Rs[(4 * kq + q) * LDs + 16 * (J - s) + li];
                            constexpr int SL = FO::slot(I, J);
#pragma unroll
                            for (int q = 0; q < 4; ++q) tl[SL] = X::mfma(opI[q], opJ[q], tl[SL]);
                        }
                    });
                }
            });
            // block row s + 1 is final: park this wave's tiles of it (the diagonal one went first)
            static_for<s + 2, NT>([&](auto Jc) {
                constexpr int J = decltype(Jc)::value;
                if constexpr (FO::of(s + 1, J) == W) {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) R[G::roff(s + 1) + X::drow(kq, reg) * G::ld(s + 1) + 16 * (J - s - 1) + li] = tl[FO::slot(s + 1, J)][reg];
                }
            });
            __syncthreads();                                          // X_{s+1}
            stamp(a, w, 6 + 4 * s);
        }
    });
    stamp(a, w, 40);

    // ---- y += z (:322); backward solve R x = y (:323), wave 0, COLUMN-ORIENTED (round 6).
    // Rounds 2-5 formed t_s = sum_{J > s} R_sJ x_J when block s was due: per block step a row of up to 112 products split
    // over the four lane groups, two cross-lane reductions, t through LDS, then x_s = W_s t -- six dependent LDS round
    // trips per step on ONE wave with nothing beside it, 9.1 us of a 52 us fp64 item while the workgroup's LDS and its
    // other waves' slots stay taken.  Here every block keeps its own partial sum: as soon as x_s is known, lane group g
    // adds R_Is x_s to the sums of its blocks I < s (I = g and I = 4 + g: one 16-term dot product of a row of tile (I, s)
    // with x_s per lane, 128-bit LDS reads), so that when block s - 1 is due its t is complete in the registers of its
    // group.  Per step: v_s = y_s + z_s - t_s -> LDS | x_s = W_s v_s (every group, redundantly) -> LDS | the updates: two
    // LDS round trips and two 8-deep pairs of FMA chains.  The sums run in another order than before (block by block, J
    // descending): same tolerance class, still one fixed order.
    if (W == 0) {
        typedef T vec_t __attribute__((ext_vector_type(16 / sizeof(T))));      // 128 bits: 4 floats / 2 doubles
        constexpr int VN = 16 / sizeof(T), NV = 16 / VN;
        auto load16 = [&](const T *p, T (&out)[16]) {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const vec_t v = reinterpret_cast<const vec_t *>(p)[q];
#pragma unroll
                for (int e = 0; e < VN; ++e) out[q * VN + e] = v[e];
            }
        };
        // row li of the tiles (I, .) of this lane group's blocks I = kq and I = 4 + kq, as pointers to column block 0
        const T *row0 = R + G::roff(kq) + li * G::ld(kq) - 16 * kq;
        const T *row1 = R + G::roff(4 + kq) + li * G::ld(4 + kq) - 16 * (4 + kq);
        T acc0 = 0, acc1 = 0;
#pragma unroll 1
        for (int s = NT - 1; s >= 0; --s) {                           // (rolled: eight unrolled steps hoisted their loads into 256 registers + spills)
            const T *Rs = R + G::roff(s);
            const int LDs = G::ld(s);
            if (kq == (s & 3)) ts[li] = (T)((double)bv[16 * s + li] + zs[16 * s + li] - (double)((s >> 2) ? acc1 : acc0));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // x_s = W_s v: x_i = sum_k W_s[i][k] v_k = sum_k Wt_s[k][i] v_k
            T v[16];
            load16(ts, v);
            T xa = 0, xb = 0;
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                xa = fma(Rs[k * LDs + li], v[k], xa);
                xb = fma(Rs[(k + 1) * LDs + li], v[k + 1], xb);
            }
            const T xsum = xa + xb;
            if (kq == 0) xs[16 * s + li] = xsum;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (s > 0) {
                T xk[16];
                load16(xs + 16 * s, xk);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    if (4 * m >= s) continue;                         // (wave-uniform: no block of slot m lies above block s)
                    const bool valid = 4 * m + kq < s;
                    const T *row = valid ? (m ? row1 : row0) + 16 * s : xs + 16 * s;      // (lanes without a block read x_s itself: in range)
                    T rr[16];
                    load16(row, rr);
                    T da = 0, db = 0;
#pragma unroll
                    for (int k = 0; k < 16; k += 2) { da = fma(rr[k], xk[k], da); db = fma(rr[k + 1], xk[k + 1], db); }
                    const T dsum = valid ? da + db : (T)0;
                    if (m) acc1 += dsum; else acc0 += dsum;
                }
            }
        }
        T *dst = reinterpret_cast<T *>(a.items) + (size_t)idx * K;                 // items().col(idx) = rr (:324)
        bool nf = false;
        for (int i = lane; i < K; i += 64) {
            const T v = xs[i];
            dst[i] = v;
            nf |= F32 ? !(fabsf((float)v) <= 3.0e38f) : !(fabs((double)v) <= 1.7e308);
        }
        // a non-positive pivot turns into NaN / inf and reaches the sample: "Cholesky failed" (:308)
        if (__any(nf) && lane == 0) atomicMin(a.fail, (unsigned long long)idx);
        stamp(a, w, 41);
    }
}

template <int K, int NW, typename T = float>
__global__ __launch_bounds__(64 * NW, 2) void k_sample_wg2(SampleArgs a, StatRiders r)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[GeoW2<K>::template lds_bytes<T>()];
    const int tid = threadIdx.x, wave = tid >> 6;
    // column statistics as rider workgroups (colstats_f32_rider): the previous launch's side at the head of the grid
    if ((int)blockIdx.x < r.nblocks) { colstats_f32_rider<K, NW, T>(r, (int)blockIdx.x, tid); return; }
    const int w = (int)blockIdx.x - r.nblocks;
    const unsigned long long t_begin = (kProfiling && a.stamps) ? wall_clock64() : 0ull;
    if constexpr (NW == 2) {
        if (wave == 0) wg2_column<K, 2, 0, T>(a, w, smem, tid);
        else wg2_column<K, 2, 1, T>(a, w, smem, tid);
    } else {
        switch (wave) {
        case 0: wg2_column<K, 4, 0, T>(a, w, smem, tid); break;
        case 1: wg2_column<K, 4, 1, T>(a, w, smem, tid); break;
        case 2: wg2_column<K, 4, 2, T>(a, w, smem, tid); break;
        default: wg2_column<K, 4, 3, T>(a, w, smem, tid); break;
        }
    }
    if (kProfiling && a.stamps && tid == 0) {                                        // profiling: sum of the items' lifetimes (wave 0), their number, first start / last end
        atomicAdd(&a.stamps[128 + 0], wall_clock64() - t_begin);
        atomicAdd(&a.stamps[128 + 1], 1ull);
        atomicMin(&a.stamps[128 + 2], t_begin);
        atomicMax(&a.stamps[128 + 3], wall_clock64());
    }
}

}  // namespace bpmf
