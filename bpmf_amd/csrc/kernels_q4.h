// k_sample4<K>: four columns per wave, the whole column update on the 4x4x4 f64 MFMA shape.
//
// v_mfma_f64_4x4x4_4b_f64 computes four INDEPENDENT 4x4x4 products.  k_sample1 gives the four
// blocks four different groups of ratings of ONE column (and has to add the four partial Grams,
// then factorises on the VALU).  Here block b belongs to COLUMN b of a group of four work items:
//   * Gram: an instruction adds 4 ratings of each of the 4 columns to one 4x4 block (g, g') of
//     their Grams -- the same 36 instructions per 16 ratings, no cross-block sum afterwards, and
//     the 36 accumulator registers now hold FOUR matrices;
//   * Lambda* = LambdaF + alpha G stays in those registers.  Blocked right-looking Cholesky
//     Lambda* = R^T R with 4x4 blocks, the four columns in lockstep: the 4x4 diagonal block is
//     factored and inverted redundantly by the 16 lanes of its column (its 10 entries arrive through
//     ds_bpermute), the panel  R_sJ = W^T A_sJ  and the trailing updates  A_IJ -= R_sI^T R_sJ  are
//     MFMAs whose operands are accumulator registers as they are (a D-layout register used as the
//     A operand is the transposed block, which is exactly what both products need);
//   * the forward solve rides in the same loop on a ninth "block column" (b as 4x4 blocks with one
//     live column), the backward solve needs the blocks of R untransposed: one bpermute each;
//   * natural (contiguous) 4-index blocks, so R is THE Cholesky factor of the reference's
//     Lambda* and x = R^-1 (R^-T b + z) is the reference's sample for the same z (c++/sample.cpp:306-323).
// Per column this replaces ~1 900 VALU instructions of assembly + factorisation + solves by ~600
// (of which ~230 are the normal draw) plus ~46 MFMAs.
//
// Lane l = 16 k + 4 b + x:  operand view (k, b, x): A_b[i = x][k], B_b[k][j = x];
//                           result view  (i = l >> 4, b, j = l & 3): D_b[i][j].
#pragma once
#include "kernels.h"

namespace bpmf {

template <int K>
struct Geo4 {
    static constexpr int NG = K / 4;                      // 4-index blocks per dimension
    static constexpr int NB = NG * (NG + 1) / 2;          // upper blocks incl. diagonal
    static constexpr int PART = (NB + NG) * 16;           // doubles one chunk of a heavy column parks (its 16 lanes)
    __host__ __device__ static constexpr int blk(int g, int g2) { return g * NG - (g * (g - 1)) / 2 + (g2 - g); }
    static constexpr int WPS = K == 32 ? 2 : 4;           // K = 32: 72 accumulators + two blocks of gathered operands (128 registers)
};

__device__ __forceinline__ double mfma44n(double a, double b, double c)   // c - a^T-view * b: the A operand negated
{
    return __builtin_amdgcn_mfma_f64_4x4x4f64(-a, b, c, 0, 0, 0);
}

// value held by lane (x = T) of every quad, to all four lanes of the quad (DPP quad_perm)
template <int T>
__device__ __forceinline__ int quad_bcast_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, T * 0x55, 0xF, 0xF, true);
}
template <int T>
__device__ __forceinline__ double quad_bcast_d(double v)
{
    const long long w = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)w, T * 0x55, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(w >> 32), T * 0x55, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

template <int K>
__global__ __launch_bounds__(64, Geo4<K>::WPS) void k_sample4(SampleArgs a)
{
    using G = Geo4<K>;
    constexpr int NG = G::NG, NB = G::NB;
    __shared__ double sz[4][K];                                      // the K normals of each of the four columns
    const int lane = threadIdx.x;
    const int kq = lane >> 4, b = (lane >> 2) & 3, x = lane & 3;      // (kq, x) double as (i, j) of the result view
    const int w = 4 * (int)blockIdx.x + b;
    const bool valid = w < a.nwork;
    const int col = valid ? a.wi_col[w] : -1;
    const int64_t p0 = valid ? a.wi_p0[w] : 0;
    const int len = valid ? a.wi_len[w] : 0;
    const int mc = valid ? a.wi_mc[w] : -1;
    const int glen = (ablate_bits(a) & 2u) ? 0 : len;
    const int32_t *rowidx = a.rowidx + p0;
    const double *vals = a.vals + p0;

    // ---- z ~ N(0, I) of the whole columns (a chunked column draws when its last chunk has arrived)
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) {
        const int c = __builtin_amdgcn_readfirstlane(__shfl(col, 4 * sb));
        const int m = __builtin_amdgcn_readfirstlane(__shfl(mc, 4 * sb));
        if (c >= 0 && m < 0) draw_normals<K>(sample_counter(a.col_from + c, a.ktrue, a.iter_plus_1), a.ktrue, sz[sb], lane, K);
    }

    // ---- Gram: block b of every instruction takes 4 ratings of column b
    double acc[NB], rr[NG];
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[t] = 0.0;
#pragma unroll
    for (int t = 0; t < NG; ++t) rr[t] = 0.0;
    int maxlen = glen;
    maxlen = max(maxlen, __shfl_xor(maxlen, 4));
    maxlen = max(maxlen, __shfl_xor(maxlen, 8));
    maxlen = __builtin_amdgcn_readfirstlane(maxlen);
    {
        // index blocks of 16 ratings per column: lane (k, b, x) holds rating 16 T + 4 x + k of column b,
        // step t' of the block takes its four ratings from the lanes x = t' (quad broadcast)
        auto load_idx = [&](int T, int &ri, double &wv) {
            const int j = 16 * T + 4 * x + kq;
            const bool ok = j < glen;
            ri = ok ? rowidx[j] : -1;
            wv = ok ? (vals[j] - a.mean_rating) * a.alpha : 0.0;                   // c++/sample.cpp:256
        };
        auto gather = [&](int row, double (&R)[NG]) {
            const double *u = ((row >= 0) ? a.other_items + (size_t)row * K : a.zero_row) + x;
#pragma unroll
            for (int g = 0; g < NG; ++g) R[g] = u[4 * g];
        };
        auto contract = [&](const double (&R)[NG], double ww) {
#pragma unroll
            for (int g = 0; g < NG; ++g) rr[g] = fma(R[g], ww, rr[g]);
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int g2 = g; g2 < NG; ++g2) acc[G::blk(g, g2)] = mfma44(R[g], R[g2], acc[G::blk(g, g2)]);
        };
        // a block = 4 steps = 16 ratings per column; the operands of the whole next block (4 x NG
        // registers) are in flight while the 4 x NB MFMAs of the current block issue: a group of four
        // columns is one wave, so there are few waves per SIMD to hide the gather latency behind
        const int nblocks = (maxlen + 15) >> 4;
        auto gather_block = [&](int ri, double (&R)[4][NG]) {
            gather(quad_bcast_i<0>(ri), R[0]);
            gather(quad_bcast_i<1>(ri), R[1]);
            gather(quad_bcast_i<2>(ri), R[2]);
            gather(quad_bcast_i<3>(ri), R[3]);
        };
        auto contract_block = [&](const double (&R)[4][NG], double wv) {
            contract(R[0], quad_bcast_d<0>(wv));
            contract(R[1], quad_bcast_d<1>(wv));
            contract(R[2], quad_bcast_d<2>(wv));
            contract(R[3], quad_bcast_d<3>(wv));
        };
        if (nblocks > 0) {
            int ri0, ri1 = -1;
            double wv0, wv1 = 0.0;
            double RA[4][NG], RB[4][NG];
            load_idx(0, ri0, wv0);
            if (nblocks > 1) load_idx(1, ri1, wv1);
            gather_block(ri0, RA);
            for (int T = 0; T < nblocks; T += 2) {
                int ri2 = -1, ri3 = -1;
                double wv2 = 0.0, wv3 = 0.0;
                if (T + 2 < nblocks) load_idx(T + 2, ri2, wv2);              // wave-uniform
                if (T + 3 < nblocks) load_idx(T + 3, ri3, wv3);
                if (T + 1 < nblocks) gather_block(ri1, RB);
                contract_block(RA, wv0);
                if (T + 1 < nblocks) {
                    if (T + 2 < nblocks) gather_block(ri2, RA);
                    contract_block(RB, wv1);
                }
                ri0 = ri2; wv0 = wv2; ri1 = ri3; wv1 = wv3;
            }
        }
    }
    // rhs sums: over the four k
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        rr[g] += __shfl_xor(rr[g], 16);
        rr[g] += __shfl_xor(rr[g], 32);
    }

    // ---- chunk of a heavy column: park the accumulators; whichever chunk arrives last adds them up
    bool alive = valid;
    if (mc >= 0) {                                                    // (lanes of the chunk slots only)
        const int nch = a.mc_nchunks[mc];
        constexpr int PSTRIDE = Geo44<K>::PART;                       // slot stride of the partial buffer (sized for both layouts)
        double *pbase = a.partials + (size_t)a.mc_slot0[mc] * PSTRIDE;
        double *p = pbase + (size_t)a.wi_chunk[w] * PSTRIDE;
        const int l16 = 4 * kq + x;
#pragma unroll
        for (int t = 0; t < NB; ++t) __hip_atomic_store(&p[t * 16 + l16], acc[t], BPMF_RLX_AGENT);
#pragma unroll
        for (int t = 0; t < NG; ++t) __hip_atomic_store(&p[(NB + t) * 16 + l16], rr[t], BPMF_RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned tk = 0;
        if (l16 == 0) tk = __hip_atomic_fetch_add(&a.mc_count[mc], 1u, BPMF_RLX_AGENT);
        tk = __shfl(tk, 4 * b);                                       // lane (0, b, 0) of this column
        if ((int)tk != nch - 1) {
            alive = false;
        } else {
            if (l16 == 0) __hip_atomic_store(&a.mc_count[mc], 0u, BPMF_RLX_AGENT);     // re-arm
#pragma unroll
            for (int t = 0; t < NB; ++t) acc[t] = 0.0;
#pragma unroll
            for (int t = 0; t < NG; ++t) rr[t] = 0.0;
            for (int ch = 0; ch < nch; ++ch) {
                const double *pc = pbase + (size_t)ch * PSTRIDE;
                double tmp[NB + NG];
#pragma unroll
                for (int t = 0; t < NB + NG; ++t) tmp[t] = __hip_atomic_load(&pc[t * 16 + l16], BPMF_RLX_AGENT);
#pragma unroll
                for (int t = 0; t < NB; ++t) acc[t] += tmp[t];
#pragma unroll
                for (int t = 0; t < NG; ++t) rr[t] += tmp[NB + t];
            }
        }
    }
    // normals of the chunked columns that are complete now
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) {
        const int c = __builtin_amdgcn_readfirstlane(__shfl(col, 4 * sb));
        const int m = __builtin_amdgcn_readfirstlane(__shfl(mc, 4 * sb));
        const int al = __builtin_amdgcn_readfirstlane(__shfl((int)alive, 4 * sb));
        if (c >= 0 && m >= 0 && al) draw_normals<K>(sample_counter(a.col_from + c, a.ktrue, a.iter_plus_1), a.ktrue, sz[sb], lane, K);
    }
    if (ablate_bits(a) & 1u) {
        double v = rr[0];
#pragma unroll
        for (int t = 0; t < NB; ++t) v += acc[t];
        if (alive && kq == 0 && x == 0 && mc < 0) a.items[(size_t)(a.col_from + col) * K] = v;
        return;
    }
    __syncthreads();                                                  // normals are in LDS

    // ---- Lambda* = LambdaF + alpha G (:297-298) in the accumulators; b = LambdaF mu + rr (:285,:256) as block column NG
    const int ii = kq, jj = x;                                        // result view
    const double *LF = (a.prop_lambda && col >= 0) ? a.prop_lambda + (size_t)col * K * K : a.LambdaF;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int g2 = g; g2 < NG; ++g2) {
            const int r_ = 4 * g + ii, c_ = 4 * g2 + jj;
            double v = fma(a.alpha, acc[G::blk(g, g2)], LF[r_ + c_ * K]);
            v = (a.diag_only && r_ != c_) ? 0.0 : v;                 // BPMF_NO_COVARIANCE (:300-304)
            acc[G::blk(g, g2)] = v;
        }
    double bv[NG];                                                    // element 4 g + i of the rhs at lane (i, b, 0), zero elsewhere
    // (per-column priors: a rolled loop of its own -- a branch between the loads of a block row and their uses makes the
    // compiler wait for every load on the spot and keep the results in scratch across the branch: kernels_q1.h)
#pragma unroll
    for (int g = 0; g < NG; ++g) bv[g] = a.Lmu[4 * g + ii];
    if (a.prop_lambda) {                                              // rr = Lambda_i * hp.mu (:285); wave-uniform
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            double lm = 0.0;
#pragma unroll 1
            for (int q = 0; q < K; ++q) lm = fma(LF[4 * g + ii + q * K], a.mu[q], lm);
            bv[g] = lm;
        }
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const double rsel = __shfl(rr[g], (lane & ~3) | ii);          // rr of index 4 g + ii (held by the lanes x = ii of this quad)
        bv[g] = (jj == 0) ? bv[g] + rsel : 0.0;
    }

    // ---- blocked Cholesky Lambda* = R^T R (:306) + forward solve (:321), four columns in lockstep
    double WB[NG];                                                    // operand "X = W_s" of the backward solve, per block step
    const int quadbase = (lane & 0xC);                                // 4 b
#pragma unroll
    for (int s = 0; s < NG; ++s) {
        // the 10 upper entries of the diagonal block of THIS lane's column: entry (p, q) sits in lane (p, b, q)
        const double dblk = acc[G::blk(s, s)];
        const double d00 = __shfl(dblk, 0 + quadbase + 0), d01 = __shfl(dblk, 0 + quadbase + 1), d02 = __shfl(dblk, 0 + quadbase + 2),
                     d03 = __shfl(dblk, 0 + quadbase + 3), d11 = __shfl(dblk, 16 + quadbase + 1), d12 = __shfl(dblk, 16 + quadbase + 2),
                     d13 = __shfl(dblk, 16 + quadbase + 3), d22 = __shfl(dblk, 32 + quadbase + 2), d23 = __shfl(dblk, 32 + quadbase + 3),
                     d33 = __shfl(dblk, 48 + quadbase + 3);
        // 4x4 upper Cholesky (1 / R_pp through v_rsq_f64 + Halley; a non-positive pivot turns into NaN and reaches the sample)
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
        // W = R_ss^-1 (upper): W_pp = 1 / R_pp
        const double W01 = -i0 * R01 * i1, W12 = -i1 * R12 * i2, W23 = -i2 * R23 * i3;
        const double W02 = -i0 * fma(R01, W12, R02 * i2);
        const double W13 = -i1 * fma(R12, W23, R13 * i3);
        const double W03 = -i0 * fma(R01, W13, fma(R02, W23, R03 * i3));
        // operand registers: lane (k, b, i) holds W[k][i] (X = W^T: panel, forward) / W[i][k] (X = W: backward)
        auto pick = [&](int p, int q) -> double {                     // W[p][q], p, q in 0..3 (lane-dependent)
            double v = 0.0;
            v = (p == 0 && q == 0) ? i0 : v; v = (p == 1 && q == 1) ? i1 : v; v = (p == 2 && q == 2) ? i2 : v; v = (p == 3 && q == 3) ? i3 : v;
            v = (p == 0 && q == 1) ? W01 : v; v = (p == 0 && q == 2) ? W02 : v; v = (p == 0 && q == 3) ? W03 : v;
            v = (p == 1 && q == 2) ? W12 : v; v = (p == 1 && q == 3) ? W13 : v; v = (p == 2 && q == 3) ? W23 : v;
            return v;
        };
        const double WA = pick(kq, x);
        WB[s] = pick(x, kq);
        // forward solve of this block row: y_s = W^T b_s
        bv[s] = mfma44(WA, bv[s], 0.0);
        // panel: R_sJ = W^T A_sJ
#pragma unroll
        for (int J = s + 1; J < NG; ++J) acc[G::blk(s, J)] = mfma44(WA, acc[G::blk(s, J)], 0.0);
        // trailing update A_IJ -= R_sI^T R_sJ and rhs b_J -= R_sJ^T y_s
#pragma unroll
        for (int I = s + 1; I < NG; ++I) {
            const double nI = -acc[G::blk(s, I)];
            bv[I] = mfma44(nI, bv[s], bv[I]);
#pragma unroll
            for (int J = I; J < NG; ++J) acc[G::blk(I, J)] = mfma44(nI, acc[G::blk(s, J)], acc[G::blk(I, J)]);
        }
    }

    // ---- y += z (:322); backward solve R x = y (:323)
#pragma unroll
    for (int g = 0; g < NG; ++g) bv[g] += (jj == 0) ? sz[b][4 * g + ii] : 0.0;
    const int tsrc = 16 * x + quadbase + kq;                          // lane holding the transposed entry of a block
#pragma unroll
    for (int s = NG - 1; s >= 0; --s) {
        double t = bv[s];
#pragma unroll
        for (int J = s + 1; J < NG; ++J) {
            const double RT = __shfl(acc[G::blk(s, J)], tsrc);        // R_sJ^T in result layout = "X = R_sJ" as the A operand
            t = mfma44(-RT, bv[J], t);
        }
        bv[s] = mfma44(WB[s], t, 0.0);                                // x_s = W_s t
    }

    // ---- items().col(idx) = rr (:324); a failed factorisation (:308) shows as a non-finite sample
    if (alive && jj == 0) {
        double *dst = a.items + (size_t)(a.col_from + col) * K + ii;
#pragma unroll
        for (int g = 0; g < NG; ++g) dst[4 * g] = bv[g];
    }
    bool bad = false;
#pragma unroll
    for (int g = 0; g < NG; ++g) bad |= !(fabs(bv[g]) <= 1.79769313486231570815e+308);
    bad = bad && alive && jj == 0;
    if (bad) atomicMin(a.fail, (unsigned long long)(a.col_from + col));
}

}  // namespace bpmf
