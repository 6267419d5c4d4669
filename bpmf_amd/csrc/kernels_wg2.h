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
#include "kernels_f32.h"
#include "kernels_slab.h"

namespace bpmf {

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

// 16 x 16 SPD block (upper part used), given as four fp64 slabs A[I]: lane (kq, c) <-> D[4 I + kq][c].
// Out: A[I] = rows of R (D = R^T R; entries left of the diagonal are NOT cleaned), E[I] = rows of R^-T = W^T.
__device__ __forceinline__ void diag16_factor_invert(double (&A)[4], double (&E)[4], int lane)
{
    const int kq = lane >> 4, x = lane & 3, c16 = lane & 15;
#pragma unroll
    for (int I = 0; I < 4; ++I) E[I] = (4 * I + kq == c16) ? 1.0 : 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const double dblk = A[s];
        const double d00 = bcast(dblk, 4 * s + 0), d01 = bcast(dblk, 4 * s + 1), d02 = bcast(dblk, 4 * s + 2), d03 = bcast(dblk, 4 * s + 3),
                     d11 = bcast(dblk, 16 + 4 * s + 1), d12 = bcast(dblk, 16 + 4 * s + 2), d13 = bcast(dblk, 16 + 4 * s + 3),
                     d22 = bcast(dblk, 32 + 4 * s + 2), d23 = bcast(dblk, 32 + 4 * s + 3), d33 = bcast(dblk, 48 + 4 * s + 3);
        double WA, WB;
        factor_block44(d00, d01, d02, d03, d11, d12, d13, d22, d23, d33, kq, x, WA, WB);
        (void)WB;
        A[s] = mfma44(WA, A[s], 0.0);                                 // R_sJ = W^T A_sJ for the four J of the slab
        E[s] = mfma44(WA, E[s], 0.0);
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
    const int len = (a.ablate & 2u) ? 0 : a.wi_len[w];             // (profiling switch: no Gram)
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
    {
        const int32_t *rowidx = a.rowidx + p0;
        const double *vals = a.vals + p0;
        // 64 ratings per coalesced index block (lane l holds rating b0 + l) = 4 groups of 4 k-steps (16 ratings).  The
        // operands of the next group (32 registers) are in flight while the MFMAs of the current one issue -- across
        // block boundaries too (groups 4, 5 are groups 0, 1 of the NEXT block: at a 512-rating chunk the drained
        // pipeline at every block start was ~25 % of the Gram).  No control flow around the loads (exact s_waitcnt
        // counts); slots beyond the end of the chunk gather a row of zeros.
        int ri = (lane < len) ? rowidx[lane] : -1;
        T wv = (lane < len) ? (T)((vals[lane] - a.mean_rating) * a.alpha) : (T)0;                 // c++/sample.cpp:256
        int ri_n = (64 + lane < len) ? rowidx[64 + lane] : -1;
        T wv_n = (64 + lane < len) ? (T)((vals[64 + lane] - a.mean_rating) * a.alpha) : (T)0;
        T yA[4][NT], yB[4][NT], wA[4], wB[4];
        const int rowmask = (a.ablate & 4u) ? 63 : -1;                 // (profiling switch: gather from 64 hot rows only)
        const bool no_mfma = (a.ablate & 8u) != 0;                     // (profiling switch: operands are loaded and summed, no MFMA)
        auto gather = [&](int gg, T (&yy)[4][NT], T (&w1)[4]) {
            const bool nx = gg >= 4;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int src = ((gg & 3) * 4 + st) * 4 + kq;
                const int row = __shfl(nx ? ri_n : ri, src);
                w1[st] = __shfl(nx ? wv_n : wv, src);
                const T *u = ((row >= 0) ? other + (size_t)(row & rowmask) * K : reinterpret_cast<const T *>(a.zero_row)) + li;
#pragma unroll
                for (int t = 0; t < NT; ++t) yy[st][t] = u[16 * t];
            }
        };
        auto contract = [&](const T (&yy)[4][NT], const T (&w1)[4]) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                if (W == 0 || no_mfma) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) r[t] = fma(yy[st][t], w1[st], r[t]);
                }
                if (no_mfma) continue;
#pragma unroll
                for (int I = 0; I < NT; ++I)
#pragma unroll
                    for (int J = I; J < NT; ++J)
                        if ((G::tri(I, J) % NW) == W)
                            acc[G::tri(I, J) / NW] = X::mfma(yy[st][I], yy[st][J], acc[G::tri(I, J) / NW]);
            }
        };
        if (len > 0) gather(0, yA, wA);
        for (int b0 = 0; b0 < len; b0 += 64) {
            const int ngroups = (len - b0 >= 64) ? 4 : (len - b0 + 15) >> 4;
            for (int gg = 0; gg < ngroups; gg += 2) {
                gather(gg + 1, yB, wB);
                contract(yA, wA);
                gather(gg + 2, yA, wA);
                if (gg + 1 < ngroups) contract(yB, wB);                              // workgroup-uniform
            }
            ri = ri_n; wv = wv_n;
            const int q = b0 + 128 + lane;
            ri_n = (q < len) ? rowidx[q] : -1;
            wv_n = (q < len) ? (T)((vals[q] - a.mean_rating) * a.alpha) : (T)0;
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
    if (W == 1 && a.stamps && lane == 0 && (w == 0 || w == a.nwork / 2)) a.stamps[(w == 0 ? 0 : 64) + 48] = wall_clock64();   // (wave 1's Gram ends)
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

    if (a.ablate & 1u) {                                              // (profiling switch: Gram only -- keep it live)
        T v = r[0];
#pragma unroll
        for (int t = 0; t < TPW; ++t) v += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
        if (lane < 16) reinterpret_cast<T *>(a.items)[(size_t)idx * K + 16 * W + lane] = v;
        return;
    }
    // Lambda* = LambdaF + alpha G in the register tiles (:297-298); b = LambdaF mu + rr (:285,:256)
    const double *LF = a.prop_lambda ? a.prop_lambda + (size_t)col * K * K : a.LambdaF;
    bool lf_tiles = false;
    if constexpr (F32) lf_tiles = a.lf32 != nullptr;
    if constexpr (F32) if (lf_tiles) {                                // (workgroup-uniform) the prior as fp32 tiles: one 16-byte load per tile
        const f4 *lt = reinterpret_cast<const f4 *>(a.lf32);
        const T alpha_f = (T)a.alpha;
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
                        const T v = fmaf(alpha_f, acc[G::tri(I, J) / NW][reg], lf[G::tri(I, J) / NW][reg]);
                        acc[G::tri(I, J) / NW][reg] = (a.diag_only && (16 * I + 4 * kq + reg) != (16 * J + li)) ? (T)0 : v;
                    }
                }
    }
    if (!lf_tiles) {
        // LambdaF(gj, gi): the lower triangle, which is what LLT reads (:306); 16 lanes = one 128-byte line.  The loads of
        // a batch of tiles are issued together and without control flow around them (with the diag_only select wrapped
        // around each load they were 72 serialised L2 round trips: 14.6 of the ~50 us a mid-size column lived)
        constexpr int BATCH = 6;
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
                        const T v = (T)fma(a.alpha, (double)acc[t0 + u][reg], lf[u][reg]);
                        acc[t0 + u][reg] = (a.diag_only && (16 * I + X::drow(kq, reg)) != (16 * J + li)) ? (T)0 : v;
                    }
                }
            }
        }
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
        stamp(a, w, 2 + 4 * s);
        // z ~ N(0, I): stream (idx+1)*K*(iter+1) mod 2^32 (c++/sample.cpp:266) -- by the last wave while wave 0 factors the
        // first diagonal block (it would idle at the next barrier otherwise; the backward solve is what reads z)
        if (s == 0 && W == NW - 1) draw_normals<K>(sample_counter(idx, a.ktrue, a.iter_plus_1), a.ktrue, zs, lane, K);
        // B: diagonal block in fp64 on the 4x4x4 shape: R_ss (upper) and W_s^T = R_ss^-T; then y_s = W_s^T b_s
        if (W == 0) {
            double A16[4], E[4];
#pragma unroll
            for (int I = 0; I < 4; ++I) A16[I] = (double)Rs[(4 * I + kq) * LDs + li];
            diag16_factor_invert(A16, E, lane);
#pragma unroll
            for (int I = 0; I < 4; ++I) {
                const int row = 4 * I + kq;
                Rs[row * LDs + li] = (T)E[I];                             // W_s^T [row][li] (lower triangular) in the place of R_ss
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // forward solve of the block row (:321): y_s = W_s^T b_s, lanes 0..15 (the wave is in lockstep: no barrier)
            T ysum = 0;
            if (lane < 16) {
#pragma unroll
                for (int k = 0; k < 16; ++k) ysum = fma(Rs[lane * LDs + k], bv[16 * s + k], ysum);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < 16) bv[16 * s + lane] = ysum;
        }
        stamp(a, w, 3 + 4 * s);
        __syncthreads();
        stamp(a, w, 4 + 4 * s);
        if (s + 1 < NT) {
            // C: panel  R_sJ = W_s^T A_sJ  by the owner of tile (s, J): operands from LDS, result back to LDS.
            // Contraction index of MFMA q, lane group kq: k = 4 kq + q (not 4 q + kq).  The B operands -- the bulk of the
            // LDS reads of the factorisation -- then come from rows 4 apart, whose offsets 4 (width + 4) = 16 mod 64 words
            // put the four 16-lane groups on disjoint banks; with consecutive rows (offset width + 4 = 4 mod 64) the groups
            // overlapped in 12 of 16 banks (r03 PMC: 20.6 % of the LDS cycles were bank conflicts).  Any order of k is the
            // same sum up to rounding; the order is fixed, so the result is as reproducible as before.
            T opA[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) opA[q] = Rs[li * LDs + 4 * kq + q];           // A[i = li][k = 4 kq + q] of W_s^T
#pragma unroll
            for (int J = s + 1; J < NT; ++J)
                if ((G::tri(s, J) % NW) == W) {
                    T opB[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) opB[q] = Rs[(4 * kq + q) * LDs + 16 * (J - s) + li];
                    acc_t t4 = acc_t{0, 0, 0, 0};
#pragma unroll
                    for (int q = 0; q < 4; ++q) t4 = X::mfma(opA[q], opB[q], t4);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) Rs[X::drow(kq, reg) * LDs + 16 * (J - s) + li] = t4[reg];
                }
            __syncthreads();
            stamp(a, w, 5 + 4 * s);
            // rhs: b_j -= sum_k R_s[k][j] y_s[k] for the entries right of the block, one thread each
            for (int cc = tid; cc < Ws - 16; cc += 64 * NW) {
                T sacc = 0;
#pragma unroll
                for (int k = 0; k < 16; ++k) sacc = fma(Rs[k * LDs + 16 + cc], bv[16 * s + k], sacc);
                bv[16 * (s + 1) + cc] -= sacc;
            }
            // D: trailing update of this wave's tiles  A_IJ -= R_sI^T R_sJ
#pragma unroll
            for (int I = s + 1; I < NT; ++I) {
                bool any = false;
#pragma unroll
                for (int J = I; J < NT; ++J) any |= (G::tri(I, J) % NW) == W;
                if (!any) continue;                                  // compile-time
                T opI[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) opI[q] = -Rs[(4 * kq + q) * LDs + 16 * (I - s) + li];     // (k = 4 kq + q: see the panel)
#pragma unroll
                for (int J = I; J < NT; ++J)
                    if ((G::tri(I, J) % NW) == W) {
                        T opJ[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) opJ[q] = Rs[(4 * kq + q) * LDs + 16 * (J - s) + li];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            acc[G::tri(I, J) / NW] = X::mfma(opI[q], opJ[q], acc[G::tri(I, J) / NW]);
                    }
            }
        }
    }
    __syncthreads();
    stamp(a, w, 40);

    // ---- y += z (:322); backward solve R x = y (:323) block by block, wave 0: lane (kq, li) = row li, column group kq
    if (W == 0) {
#pragma unroll
        for (int s = NT - 1; s >= 0; --s) {
            const T *Rs = R + G::roff(s);
            const int LDs = G::ld(s), Ws = G::width(s);
            // (unrolled with compile-time bounds: the LDS reads of a step are issued together; two partial sums halve the chain)
            T t = 0, t2 = 0;                                          // (fp32 products and sums: half the issue slots of the widened form)
#pragma unroll
            for (int j = 16; j < Ws; j += 8) {
                t = fma(Rs[li * LDs + j + kq], xs[16 * s + j + kq], t);
                if (j + 4 < Ws) t2 = fma(Rs[li * LDs + j + 4 + kq], xs[16 * s + j + 4 + kq], t2);
            }
            t += t2;
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            const double tt = (double)bv[16 * s + li] + zs[16 * s + li] - (double)t;
            if (kq == 0) ts[li] = (T)tt;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // x_s = W_s t: x_i = sum_k W_s[i][k] t_k = sum_k Wt_s[k][i] t_k
            T xsum = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) xsum = fma(Rs[k * LDs + li], ts[k], xsum);
            if (kq == 0) xs[16 * s + li] = xsum;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
    const unsigned long long t_begin = a.stamps ? wall_clock64() : 0ull;
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
    if (a.stamps && tid == 0) {                                        // profiling: sum of the items' lifetimes (wave 0), their number, first start / last end
        atomicAdd(&a.stamps[128 + 0], wall_clock64() - t_begin);
        atomicAdd(&a.stamps[128 + 1], 1ull);
        atomicMin(&a.stamps[128 + 2], t_begin);
        atomicMax(&a.stamps[128 + 3], wall_clock64());
    }
}

}  // namespace bpmf
