// kernels_slab.h -- k_sample_slab<64, double> / k_sample1s<64>: the column update for K = 64 (fp64), one wave per work
// item, Gram and the whole factorisation on v_mfma_f64_4x4x4_4b_f64.  (K = 128 ran here in fp32 until round 5: it is
// k_sample_wg2's now, kernels_wg2.h; factor_block44 and quad_splat below are shared with it.)
//
// Reference: Sys::sample(long idx, Sys&) + computeMuLambda, c++/sample.cpp:248-336.
//
// Why: at K >= 64 a column spent ~60 k cycles (K = 64, finish_single: one lane per row, ~10 % lane
// efficiency) or ~60 us (K = 128: 128 sequential 16-wide pivots, one-thread-per-column panel solves)
// in a factorisation that is only K^3 / 3 flops.  Here Lambda* lives in registers as 4-row SLABS:
//
//     slab (I, q), lane l  <->  Lambda*[4 I + (l >> 4)][16 q + (l & 15)]          I < K/4, q >= I/4
//
// which is at the same time (a) the accumulator layout of v_mfma_f64_16x16x4_f64 -- register `reg` of
// the 16x16 Gram tile (TI, TJ) IS slab (4 TI + reg, TJ), so the Gram hands its registers over as
// they are -- and (b) four 4x4 blocks (I, 4 q + b), b = 0..3, in the result layout of the 4x4x4
// shape.  Blocked right-looking Cholesky Lambda* = R^T R with 4x4 blocks (the scheme of k_sample4,
// kernels_q4.h, with the four blocks of an instruction now four block COLUMNS of one matrix):
//   * the 4x4 diagonal block arrives through v_readlane, is factored and inverted redundantly by
//     every lane (four 1/sqrt);
//   * panel  R_sJ = W^T A_sJ : one MFMA per slab of block row s;
//   * the block row is published to LDS (4 x K doubles); the A operand of the trailing update
//     A_IJ -= R_sI^T R_sJ is read back from there already in operand layout (one 8-byte LDS read per
//     row block I, broadcast over b), the B operand is the slab register as it is;
//   * the rhs rides along as one more block column (forward solve, c++/sample.cpp:321);
//   * backward solve (:323): the blocks of R are needed untransposed (one cross-lane permute per slab),
//     the four blocks of an instruction are four column blocks of the SAME row, their partial sums
//     are added with two DPP row rotates.
// Natural 4-index blocks: R is THE Cholesky factor of the reference's Lambda* in the reference's
// order, so x = R^-1 (R^-T b + z) is the reference's sample for the same z.
// Heavy columns are cut into chunks like everywhere else (partials in tile layout, last arriver adds).
#pragma once
#include <type_traits>
#include "kernels.h"
#include "kernels_f32.h"

namespace bpmf {

template <int K>
struct GeoS {
    static constexpr int NG = K / 4;                      // 4-row blocks
    static constexpr int NQ = K / 16;                     // 16-column slabs per row block
    static constexpr int NT = K / 16;
    static constexpr int NTRI = NT * (NT + 1) / 2;
    __host__ __device__ static constexpr int roff(int I) { int n = 0; for (int t = 0; t < I; ++t) n += NQ - (t >> 2); return n; }
    static constexpr int NREG = roff(NG);                 // slabs kept: q >= I / 4
    __host__ __device__ static constexpr int reg(int I, int q) { return roff(I) + q - (I >> 2); }
    // Gram accumulators (gram_slab): the rotated-block layout of rot44_contract (kernels.h)
    __host__ __device__ static constexpr int nrot(int TI, int TJ) { return Rot44<K>::nrot(TI, TJ); }
    __host__ __device__ static constexpr int aoff(int TI, int TJ) { return Rot44<K>::aoff(TI, TJ); }
    static constexpr int NACC = Rot44<K>::NACC;           // 36 at K = 64 (the slab form itself needs NREG = 40)
    static constexpr int LDR = K + 2;                     // row stride of the published block row (doubles)
    // LDS: z [K] | rhs [K] | block row [4][LDR] | inverted diagonal blocks [NG][16] | (fp32 path) one 16 x 17 tile
    static constexpr int LDS_WORDS = 2 * K + 4 * LDR + 16 * NG + (K == 128 ? 16 * 17 / 2 + 8 : 0);
    // doubles in the partial of one chunk of a heavy column (tile layout)
    static constexpr int PART = K == 128 ? (NTRI * 256 + NT * 16) / 2 : NREG * 64 + NT * 16;
#ifndef BPMF_SLAB_WPS
#define BPMF_SLAB_WPS 2
#endif
    static constexpr int WPS = K == 128 ? 1 : (K == 64 ? BPMF_SLAB_WPS : 4);    // waves per SIMD the kernels are compiled for
    // operand sets of the Gram loop (the gathers run DEPTH - 1 groups of 4 ratings ahead of the MFMAs)
#ifndef BPMF_SLAB_DEPTH
#define BPMF_SLAB_DEPTH 4
#endif
    static constexpr int DEPTH = BPMF_SLAB_DEPTH;
};

// What a lane gets from factor_block44 is ONE entry of the upper triangular W = R_ss^-1: lane (k, b, i) holds W[k][i] (the A operand
// "W^T" of the 4x4x4 shape), zero below the diagonal.  Rounds 2-5 computed all ten entries wave-uniformly (16 operations) and
// selected (34 v_cndmask per pivot block as compiled).  Round 6: the lane solves R w = e_i for ITS column i by back substitution --
// the right-hand side is the lane's unit vector as four factors 1.0 / 0.0 (cx), ten operations -- and takes entry k of it with
// four more (rk): 14 instead of 27 + 34 instructions per pivot block, of which there are 16 per column at K = 64 and 32 per item at
// K = 128, every one on the critical chain.  The eight factors are made once per kernel.  (A failed factorisation -- inf / NaN
// pivots -- gives NaN in every lane: the column is reported as "Cholesky failed" either way.)
struct W44Select {
    double cx[4], rk[4];                                              // (i == j), (k == j), j = 0 .. 3
    __device__ __forceinline__ W44Select(int kq, int x)
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) { cx[j] = (x == j) ? 1.0 : 0.0; rk[j] = (kq == j) ? 1.0 : 0.0; }
    }
};

// W = R_ss^-1 of a 4x4 SPD block given by its 10 upper entries (wave-uniform values): returns the operand register of
// k_sample4's scheme -- WA: lane (k, b, i) holds W[k][i] (A operand "W^T").
__device__ __forceinline__ void factor_block44(double d00, double d01, double d02, double d03, double d11, double d12, double d13,
                                               double d22, double d23, double d33, const W44Select &sel, double &WA)
{
    const double i0 = rsqrt_nr(d00);
    const double R01 = d01 * i0, R02 = d02 * i0, R03 = d03 * i0;
    const double e11 = fma(-R01, R01, d11);
    const double i1 = rsqrt_nr(e11);
    const double R12 = fma(-R01, R02, d12) * i1, R13 = fma(-R01, R03, d13) * i1;
    const double e22 = fma(-R12, R12, fma(-R02, R02, d22));
    const double i2 = rsqrt_nr(e22);
    const double R23 = fma(-R12, R13, fma(-R02, R03, d23)) * i2;
    const double e33 = fma(-R23, R23, fma(-R13, R13, fma(-R03, R03, d33)));
    const double i3 = rsqrt_nr(e33);
    // column i of W: R w = e_i, from the bottom (w_j = 0 for j > i comes out of the zeros of e_i)
    const double w3 = sel.cx[3] * i3;
    const double w2 = fma(-R23, w3, sel.cx[2]) * i2;
    const double w1 = fma(-R13, w3, fma(-R12, w2, sel.cx[1])) * i1;
    const double w0 = fma(-R03, w3, fma(-R02, w2, fma(-R01, w1, sel.cx[0]))) * i0;
    WA = fma(sel.rk[3], w3, fma(sel.rk[2], w2, fma(sel.rk[1], w1, sel.rk[0] * w0)));
}

// value of quad b' of every row of 16 lanes, in all four quads of the row (ds_swizzle, bit mode: no LDS memory)
template <int BQ>
__device__ __forceinline__ double quad_splat(double v)
{
    constexpr int PAT = 0x13 | (BQ << 7);                            // lane' = (lane & 0b10011) | (BQ << 2), per group of 32 lanes
    const long long w = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_ds_swizzle((int)w, PAT), hi = __builtin_amdgcn_ds_swizzle((int)(w >> 32), PAT);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// value of lane G of every row of 16 lanes, in all 16 lanes of the row (ds_swizzle, bit mode: lane' = (lane & 0x10) | G per
// group of 32 lanes; 2.4 cycles per instruction and CU against the 6.3 of a ds_bpermute: profiles/r05_mfma_shapes_probe.txt)
template <int G>
__device__ __forceinline__ int row_lane_i(int v) { return __builtin_amdgcn_ds_swizzle(v, 0x10 | (G << 5)); }
template <int G>
__device__ __forceinline__ double row_lane_d(double v)
{
    const long long w = __builtin_bit_cast(long long, v);
    const int lo = row_lane_i<G>((int)w), hi = row_lane_i<G>((int)(w >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// f(integral_constant<int, B>) ... f(integral_constant<int, E - 1>): a loop whose index has to be a constant expression
template <int B, int E, typename F>
__device__ __forceinline__ void slab_static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        slab_static_for<B + 1, E>(f);
    }
}

// Gram of one chunk on the 4x4x4 shape (fp64, K = 64).  A group is 4 ratings (k = lane >> 4 picks one); the gathered
// register t holds u_k[16 t + c] in lane (k, c), c = 4 b + j.  The four blocks b of v_mfma_f64_4x4x4_4b are independent
// 4 x 4 x 4 products, so with A = register TI as it is (block b = latent rows 16 TI + 4 b ..) and B = register TJ ROTATED by d
// quads (block b = latent columns 16 TJ + 4 ((b + d) % 4) ..) one instruction accumulates the four blocks
// (4 TI + b, 4 TJ + (b + d) % 4) of tile (TI, TJ): d = 0 .. 3 cover an off-diagonal tile; on the diagonal d = 0, 1, 2 do --
// the blocks below the diagonal that d = 1, 2 produce are the transposes of (0,3), (0,2), (1,3).  36 MFMAs per group and
// 16 DPP moves (8 rotated registers x two halves: see contract()); NO cross-lane operation through the LDS pipe.
// Rounds 1-5 made the A operand of row block I by splatting quad I & 3 of register I >> 2 over its row (two ds_swizzle per
// operand, 32 per group, 40 MFMAs per group).  Round 6 measured what that costs: on a SIMD the issue of an LDS-pipe
// instruction ADDS to the MFMA time like a VALU instruction does (tools/probes/cbsz_probe.hip: 28.2 cycles per MFMA with two
// swizzles each against 19.0 without; the Gram loop ran ~1000 cycles per group against 720 of MFMAs) -- fewer MFMAs bought
// with more swizzles / bpermutes gained nothing (-1 %), deeper gathers nothing, the splats one group ahead nothing; what
// counts is the NUMBER of instructions of any kind.  (The MFMA's own A-broadcast controls, cbsz / abid, are accepted by the
// assembler for this shape and ignored by the hardware: tools/probes/cbsz_map_probe.hip.)
// The accumulators are NOT in slab layout: slabs_from_acc() converts once per column.
template <int K>
__device__ __forceinline__ void gram_slab(const int32_t *__restrict__ rowidx, const double *__restrict__ vals, int len,
                                          const double *__restrict__ other, const double *__restrict__ zero_row, double mean, double alpha,
                                          double (&C)[GeoS<K>::NACC], double (&r)[K / 16], int lane)
{
    using G = GeoS<K>;
    constexpr int NT = K / 16;
    const int kq = lane >> 4, li = lane & 15;
    if (len <= 0) return;
    // Index blocks of 64 ratings, the current one and the next one.  Lane (kq, g) = 16 kq + g holds rating 4 g + kq of its block,
    // i.e. the rating group g's lane row kq gathers: the row id and the weight then reach the 16 lanes of the row through a
    // ds_swizzle row broadcast (a compile-time pattern, no address register) instead of a ds_bpermute -- 2.4 against 6.3 cycles
    // per instruction and CU, three per group of four ratings.  The 64 loads of a block still cover 256 contiguous bytes.
    // (The row as a 64-bit byte offset instead of an id -- one add per gather instead of a compare, two selects, a shift and two
    //  adds, at the price of a second permute for the upper half -- measured 1 % slower: 0.2767 against 0.2743 ms, round 6.)
    const int slot = 4 * li + kq;                                     // this lane's rating within an index block
    int ri = (slot < len) ? rowidx[slot] : -1;
    double wv = (slot < len) ? (vals[slot] - mean) * alpha : 0.0;                // c++/sample.cpp:256
    int ri_n = (64 + slot < len) ? rowidx[64 + slot] : -1;
    double wv_n = (64 + slot < len) ? (vals[64 + slot] - mean) * alpha : 0.0;
    // Group ST of the block; ST >= 16 are the first groups of the NEXT block: the gathers run DEPTH - 1 groups ahead of the
    // MFMAs, across block boundaries too.  No control flow around the loads -- the compiler's s_waitcnt counts stay exact --
    // and slots beyond the end of the chunk gather a row of zeros.
    auto gather = [&](auto stc, double (&yy)[NT], double &ww) {
        constexpr int ST = decltype(stc)::value;
        constexpr bool nx = ST >= 16;
        const int row = row_lane_i<(ST & 15)>(nx ? ri_n : ri);
        ww = row_lane_d<(ST & 15)>(nx ? wv_n : wv);
        const double *u = ((row >= 0) ? other + (size_t)row * K : zero_row) + li;
#pragma unroll
        for (int t = 0; t < NT; ++t) yy[t] = u[16 * t];
    };
    auto contract = [&](const double (&yy)[NT], double ww) {
#pragma unroll
        for (int t = 0; t < NT; ++t) r[t] = fma(yy[t], ww, r[t]);
        rot44_contract<K>(yy, C);
    };
    // D operand sets: the gathers run D - 1 groups ahead of the MFMAs (GeoS<K>::DEPTH).  A group whose four ratings all lie
    // beyond the end of the chunk is gathered (a row of zeros: the loads stay unconditional) but NOT contracted: the
    // guard is wave-uniform, one scalar branch per group, and skipping a product with zeros changes no sum.
    constexpr int D = G::DEPTH;
    static_assert(D >= 2 && 16 % D == 0, "the operand sets must rotate evenly through the 16 groups of an index block");
    double y[D][NT], w[D];
    slab_static_for<0, D - 1>([&](auto uc) { gather(uc, y[decltype(uc)::value], w[decltype(uc)::value]); });
    for (int b0 = 0; b0 < len; b0 += 64) {
        const int ngroups = (len - b0 >= 64) ? 16 : (len - b0 + 3) >> 2;         // groups of 4 ratings in this block
        // all 16 groups of the block unrolled: the group number is a compile-time constant, so that which index block a gather
        // reads (this one / the next) and the lane of its row that holds its rating cost no instruction; groups beyond the
        // block's last one are skipped (wave-uniform guards)
        slab_static_for<0, 16>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (g < ngroups) {
                gather(std::integral_constant<int, g + D - 1>{}, y[(g + D - 1) % D], w[(g + D - 1) % D]);
                contract(y[g % D], w[g % D]);
            }
        });
        // next block: its first D - 1 groups are in flight already
        ri = ri_n; wv = wv_n;
        const int q = b0 + 128 + slot;
        ri_n = (q < len) ? rowidx[q] : -1;
        wv_n = (q < len) ? (vals[q] - mean) * alpha : 0.0;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        r[t] += __shfl_xor(r[t], 16);
        r[t] += __shfl_xor(r[t], 32);
    }
}

// The accumulators of gram_slab() -> the slabs the factorisation works on, once per column (a chunked column: after its
// partials were added; the map is a permutation, so it commutes with the sums).  Tile by tile through 2 KB of LDS (the
// block-row buffer and the buffer of the inverted diagonal blocks, both idle until the factorisation, taken in turn): lane
// (i, b, j) of rotation d holds G[16 TI + 4 b + i][16 TJ + 4 ((b + d) % 4) + j] and stores it at [4 b + i][4 ((b + d) % 4) + j]
// of a 16 x 16 image; slab (4 TI + m, TJ) is rows 4 m .. 4 m + 3 of the image, lane (i, c) <- [4 m + i][c].  A diagonal tile
// stores every element at its mirror position too, so that the image is the full symmetric tile (the blocks d = 3 would have
// given are the mirrors of d = 1's; an element written twice is written with the same bits: same products, same order).
// One wave per workgroup: LDS operations of a wave complete in order, the barriers only pin the compiler's order.
template <int K>
__device__ __forceinline__ void slabs_from_acc(const double (&C)[GeoS<K>::NACC], double (&A)[GeoS<K>::NREG], double *buf0, double *buf1, int lane)
{
    // slab (4 TI + m, TJ) is rows 4 m .. 4 m + 3 of the image of tile (TI, TJ): register m of the tile
    rot44_images<K>(C, buf0, buf1, lane, [&](int, int TI, int TJ, int m, double v) { A[GeoS<K>::reg(4 * TI + m, TJ)] = v; });
}

// Lambda* (slabs A) and the rhs in, the sample out.  The rhs is PACKED like a slab column: bv[Q], lane
// (i, b, j = 0) holds element 16 Q + 4 b + i (four row blocks per register, zero in the columns j > 0).
// sz: the K normals; srow: 4 x LDR doubles of LDS (published block row); sw: NG x 16 doubles of LDS
// (the inverted diagonal blocks, for the backward solve).  Single wave.
template <int K>
__device__ __forceinline__ void slab_cholesky_solve(double (&A)[GeoS<K>::NREG], double (&bv)[K / 16], const double *sz, double *srow, double *sw,
                                                    int lane)
{
    using G = GeoS<K>;
    constexpr int NG = G::NG, NQ = G::NQ, LDR = G::LDR;
    const int kq = lane >> 4, b = (lane >> 2) & 3, x = lane & 3, c16 = lane & 15;
    const W44Select wsel(kq, x);
#pragma unroll
    for (int s = 0; s < NG; ++s) {
        const int q0 = s >> 2, b0 = s & 3;
        // the 10 upper entries of the diagonal block: entry (p, q) sits in lane 16 p + 4 b0 + q of slab (s, q0)
        const double dblk = A[G::reg(s, q0)];
        // (through LDS instead -- one store by the sixteen lanes that hold the block, ten broadcast loads -- measured the same:
        //  0.2680 against 0.2687 ms, round 6)
        const double d00 = bcast(dblk, 4 * b0 + 0), d01 = bcast(dblk, 4 * b0 + 1), d02 = bcast(dblk, 4 * b0 + 2), d03 = bcast(dblk, 4 * b0 + 3),
                     d11 = bcast(dblk, 16 + 4 * b0 + 1), d12 = bcast(dblk, 16 + 4 * b0 + 2), d13 = bcast(dblk, 16 + 4 * b0 + 3),
                     d22 = bcast(dblk, 32 + 4 * b0 + 2), d23 = bcast(dblk, 32 + 4 * b0 + 3), d33 = bcast(dblk, 48 + 4 * b0 + 3);
        double WA;
        factor_block44(d00, d01, d02, d03, d11, d12, d13, d22, d23, d33, wsel, WA);
        // W_s as the A operand of the backward solve: entry [kq][x] of the tile has to hold W[x][kq] -- the lane writes
        // ITS W[kq][x] to [x][kq] instead of selecting the transposed entry from the ten candidates a second time
        if (b == 0) sw[16 * s + 4 * x + kq] = WA;
        // forward solve of this block row: y_s = W^T b_s (block b0 of bv[q0]); y_s in every b as a B operand
        const double ys_all = mfma44(WA, bv[q0], 0.0);
        bv[q0] = (b == b0) ? ys_all : bv[q0];
        double ys = (b == b0) ? ys_all : 0.0;
        ys = row_ror_add<0x124>(ys);
        ys = row_ror_add<0x128>(ys);
        // panel: R_sJ = W^T A_sJ (the block J = s becomes R_ss, blocks J < s of the first slab are never read again)
#pragma unroll
        for (int q = q0; q < NQ; ++q) A[G::reg(s, q)] = mfma44(WA, A[G::reg(s, q)], 0.0);
        if (s + 1 < NG) {
            // rhs: b_J -= R_sJ^T y_s for J > s -- the slab of block row s IS the A operand (four J per instruction).
            // -R_sJ is formed ONCE per slab: it is this operand and what gets published (the trailing update reads its A
            // operand back negated already: one sign flip per slab instead of one per row block I > s).
            double nA[NQ];
#pragma unroll
            for (int q = q0; q < NQ; ++q) {
                nA[q] = -A[G::reg(s, q)];
                if (q == q0 && b0 == 3) continue;
                const double nR = (q == q0 && b <= b0) ? 0.0 : nA[q];
                bv[q] = mfma44(nR, ys, bv[q]);
            }
            // publish block row s (negated); the A operand of row block I: lane (k, b, i) <- -R[4 s + k][4 I + i]
#pragma unroll
            for (int q = q0; q < NQ; ++q) srow[kq * LDR + 16 * q + c16] = nA[q];
            __syncthreads();
#pragma unroll
            for (int I = s + 1; I < NG; ++I) {
                const double nI = srow[kq * LDR + 4 * I + x];
#pragma unroll
                for (int q = I >> 2; q < NQ; ++q) A[G::reg(I, q)] = mfma44(nI, A[G::reg(s, q)], A[G::reg(I, q)]);   // A_IJ -= R_sI^T R_sJ
            }
            __syncthreads();
        }
    }

    // ---- y += z (:322); backward solve R x = y (:323).  bv[q] doubles as the B operand "x": lane (k, b, 0) = x[16 q + 4 b + k]
#pragma unroll
    for (int q = 0; q < NQ; ++q) bv[q] += (x == 0) ? sz[16 * q + 4 * b + kq] : 0.0;
    const int tsrc = 16 * x + 4 * b + kq;                             // lane holding the transposed entry of a block
#pragma unroll
    for (int s = NG - 1; s >= 0; --s) {
        const int q0 = s >> 2, b0 = s & 3;
        // sum_{J > s} R_sJ x_J: block b of slab q is J = 4 q + b
        double part = 0.0;
#pragma unroll
        for (int q = q0; q < NQ; ++q) {
            if (q == q0 && b0 == 3) continue;                         // (nothing solved in this slab yet)
            const double RT = __shfl(A[G::reg(s, q)], tsrc);          // R_s,4q+b untransposed as the A operand
            const double xq = (q == q0 && b <= b0) ? 0.0 : bv[q];     // the solved part of x
            part = mfma44(RT, xq, part);
        }
        double tot = row_ror_add<0x124>(part);                        // + b rotated by 1
        tot = row_ror_add<0x128>(tot);                                // + rotated by 2: all four b, in every b
        const double t = bv[q0] - tot;                                // (block b0 is the one that counts)
        const double WB = sw[16 * s + 4 * kq + x];
        const double xs = mfma44(WB, t, 0.0);                         // x_s = W_s t
        bv[q0] = (b == b0) ? xs : bv[q0];
    }
}

// one work item (column or chunk of a heavy column) by one wave; lds: GeoS<K>::LDS_WORDS doubles
template <int K, typename T>
__device__ __forceinline__ void slab_item(const SampleArgs &a, int w, double *lds, int lane)
{
    using G = GeoS<K>;
    constexpr int NG = G::NG, NQ = G::NQ, NT = G::NT, NREG = G::NREG;
    constexpr bool F32 = sizeof(T) == 4;
    double *sz = lds, *sb = lds + K, *srow = lds + 2 * K, *sw = srow + 4 * G::LDR;
    const int kq = lane >> 4, li = lane & 15, x = lane & 3;
    const int col = a.wi_col[w];
    const int64_t p0 = a.wi_p0[w];
    const int len = (ablate_bits(a) & 2u) ? 0 : a.wi_len[w];             // (profiling switch: no Gram)
    const int mc = a.wi_mc[w];
    const int64_t idx = a.col_from + col;

    stamp(a, w, 0);
    // whole column in one item: its normals do not depend on the Gram -- drawn first, in the shadow of the first loads
    if (mc < 0) draw_normals_deferred<K>(sample_counter(idx, a.ktrue, a.iter_plus_1), a.ktrue, sz, srow, lane, K);

    stamp(a, w, 1);
    double A[NREG];
    double rsum[NT];                                                  // rr[16 t + li] (all kq)
    {
        static_assert(!F32 && K == 64, "the slab form is the K = 64 fp64 sampler (K = 128 runs k_sample_wg2)");
        constexpr int NACC = G::NACC;
        double C[NACC];                                               // Gram accumulators, rotated-block layout (gram_slab)
#pragma unroll
        for (int t = 0; t < NACC; ++t) C[t] = 0.0;
#pragma unroll
        for (int t = 0; t < NT; ++t) rsum[t] = 0.0;
        gram_slab<K>(a.rowidx + p0, a.vals + p0, len, a.other_items, a.zero_row, a.mean_rating, a.alpha, C, rsum, lane);
        if (mc >= 0) {
            // chunk of a heavy column: park the accumulators; whichever chunk arrives last adds them up (chunk order)
            constexpr int PART = G::PART;
            static_assert(NACC * 64 + NT * 16 <= PART, "partial slot");
            const int nch = a.mc_nchunks[mc];
            double *pbase = a.partials + (size_t)a.mc_slot0[mc] * PART;
            double *p = pbase + (size_t)a.wi_chunk[w] * PART;
#pragma unroll
            for (int t = 0; t < NACC; ++t) __hip_atomic_store(&p[t * 64 + lane], C[t], BPMF_RLX_AGENT);
            if (lane < 16) {
#pragma unroll
                for (int t = 0; t < NT; ++t) __hip_atomic_store(&p[NACC * 64 + t * 16 + lane], rsum[t], BPMF_RLX_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned tk = 0;
            if (lane == 0) tk = __hip_atomic_fetch_add(&a.mc_count[mc], 1u, BPMF_RLX_AGENT);
            tk = __builtin_amdgcn_readfirstlane(tk);
            if ((int)tk != nch - 1) return;
            if (lane == 0) __hip_atomic_store(&a.mc_count[mc], 0u, BPMF_RLX_AGENT);
#pragma unroll
            for (int t = 0; t < NACC; ++t) C[t] = 0.0;
#pragma unroll
            for (int t = 0; t < NT; ++t) rsum[t] = 0.0;
            for (int ch = 0; ch < nch; ++ch) {                        // fixed chunk order: deterministic
                const double *pc = pbase + (size_t)ch * PART;
                double tmp[NACC];
#pragma unroll
                for (int t = 0; t < NACC; ++t) tmp[t] = __hip_atomic_load(&pc[t * 64 + lane], BPMF_RLX_AGENT);
#pragma unroll
                for (int t = 0; t < NACC; ++t) C[t] += tmp[t];
#pragma unroll
                for (int t = 0; t < NT; ++t) rsum[t] += __hip_atomic_load(&pc[NACC * 64 + t * 16 + li], BPMF_RLX_AGENT);
            }
            draw_normals_deferred<K>(sample_counter(idx, a.ktrue, a.iter_plus_1), a.ktrue, sz, srow, lane, K);
            __syncthreads();                                          // (the draw parks r2 in srow: done before the images go there)
        }
        slabs_from_acc<K>(C, A, srow, sw, lane);
    }

    stamp(a, w, 2);
    if (ablate_bits(a) & 1u) {                                              // (profiling switch: Gram only -- keep it live)
        double v = rsum[0];
#pragma unroll
        for (int t = 0; t < NREG; ++t) v += A[t];
        if (lane < K) reinterpret_cast<T *>(a.items)[(size_t)idx * K + lane] = (T)v;
        return;
    }
    wait_params(a);
    // ---- Lambda* = LambdaF + alpha G (:297-298); b = LambdaF mu + rr (:285,:256)
    const double *LF = a.prop_lambda ? a.prop_lambda + (size_t)col * K * K : a.LambdaF;
#pragma unroll
    for (int I = 0; I < NG; ++I)
#pragma unroll
        for (int q = I >> 2; q < NQ; ++q) {
            const int r_ = 4 * I + kq, c_ = 16 * q + li;
            double v = fma(a.alpha, A[G::reg(I, q)], LF[r_ + (size_t)c_ * K]);
            if (a.diag_only && r_ != c_) v = 0.0;                    // BPMF_NO_COVARIANCE (:300-304)
            A[G::reg(I, q)] = v;
        }
    if (kq == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            double lm = a.Lmu[16 * t + li];
            if (a.prop_lambda) {                                      // rr = Lambda_i * hp.mu (:285)
                lm = 0.0;
                for (int j = 0; j < K; ++j) lm = fma(LF[16 * t + li + (size_t)j * K], a.mu[j], lm);
            }
            sb[16 * t + li] = lm + rsum[t];
        }
    }
    __syncthreads();                                                  // rhs and normals are in LDS
    const int b = (lane >> 2) & 3;
    double bv[NQ];                                                    // packed rhs: lane (i, b, 0) holds element 16 q + 4 b + i
#pragma unroll
    for (int q = 0; q < NQ; ++q) bv[q] = (x == 0) ? sb[16 * q + 4 * b + kq] : 0.0;

    stamp(a, w, 3);
    slab_cholesky_solve<K>(A, bv, sz, srow, sw, lane);
    stamp(a, w, 4);

    // ---- items().col(idx) = rr (:324): through LDS for one coalesced store; a failed factorisation (:308) shows as a non-finite sample
    bool bad = false;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        bad |= !(fabs(bv[q]) <= 1.79769313486231570815e+308);
        if (x == 0) sb[16 * q + 4 * b + kq] = bv[q];
    }
    __syncthreads();
    T *dst = reinterpret_cast<T *>(a.items) + (size_t)idx * K;
    for (int i = lane; i < K; i += 64) dst[i] = (T)sb[i];
    bad = bad && x == 0;
    if (__any(bad) && lane == 0) atomicMin(a.fail, (unsigned long long)idx);
    stamp(a, w, 5);
}

template <int K, typename T>
__global__ __launch_bounds__(64, GeoS<K>::WPS) void k_sample_slab(SampleArgs a)
{
    __shared__ __attribute__((aligned(16))) double lds[GeoS<K>::LDS_WORDS];
    slab_item<K, T>(a, (int)blockIdx.x, lds, (int)threadIdx.x);
}

// The same item body behind k_sample1's launch format (K <= 32, the fused stateful path: workgroup 0 = gate +
// staging of this launch's parameters, the next f.nstat workgroups = column statistics of the previous side).
template <int K>
__global__ __launch_bounds__(64, GeoS<K>::WPS) void k_sample1s(SampleArgs a, FusedArgs f)
{
    __shared__ __attribute__((aligned(16))) double lds[GeoS<K>::LDS_WORDS];
    int bid = blockIdx.x;
    if (f.gate_host) {
        if (bid == 0) { gate_stage_body(0, 1, f.gate_host, f.gate_want, f.src_host, f.dst, f.n, f.dflag, f.dval, a.tmo, a.wait_ticks); return; }
        --bid;
    }
    if (bid < f.nstat) {
        static_assert(!kColstatsRot<K> || GeoS<K>::LDS_WORDS >= 512, "the riders' images");
        colstats_body<K>(bid, f.st_items, f.st_c0, f.st_c1, f.nstat, f.st_partials, f.st_fail, f.st_out, f.st_ticket, f.st_flag, f.st_seq,
                         f.st_tmo, a.wait_ticks, lds);
        return;
    }
    slab_item<K, double>(a, bid - f.nstat, lds, (int)threadIdx.x);
}

}  // namespace bpmf
