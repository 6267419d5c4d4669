// kernels_x4.h -- k_sample1x<K> (K <= 32): the Grams of up to three work items by ONE wave, one after the other, then the
// factorisation of their columns in lockstep on the 4x4x4 f64 MFMA.
//
// Reference: Sys::sample(long idx, Sys&) + computeMuLambda, c++/sample.cpp:248-336.
//
// k_sample1 (kernels.h) gives every work item a wave of its own and factorises each column alone on the VALU: ~2 250
// instructions per column after the Gram (b-sum + LDS assembly 550, factorisation 1 100, solves 330, normals 250) -- more
// SIMD time than the Gram itself on an ML-1M-sized side.  k_sample4 (kernels_q4.h) factorises four columns in lockstep
// on the MFMA (~375 VALU instructions + 46 MFMAs per column) but forms their Grams in lockstep too: a wave then walks
// the ratings of four columns at the MFMA rate of one, so items have to be a quarter of the length -- on the ML-1M shape
// most columns become chains of 160-rating chunks whose partials the last arriver adds one memory round trip at a time
// (measured: 36 / 19 us per launch with NEITHER the Gram nor the factorisation switched on).  k_sample1q (kernels_q1.h)
// kept the per-wave Gram and handed the matrices to a factorising wave through memory: the hand-over cost what it saved.
//
// Here the hand-over stays inside the wave.  A wave takes up to THREE work items -- item `w` of each third of the
// cost-sorted list, the middle third backwards, so every wave gets about the same number of ratings -- and for each:
//   * forms the Gram exactly like k_sample1 (64-rating index blocks, normals in their shadow, the four blocks of an
//     instruction = four groups of ratings of the SAME column, chunks of heavy columns parked / summed by the last
//     arriver), but on NATURAL 4-index blocks (8-byte gathers, as k_sample4) so that the accumulators are already the
//     blocks the factorisation wants;
//   * adds the four b of every accumulator (two DPP row rotates: every lane then holds the total); the totals of the
//     first two items go to a 4.6 KB LDS stash each (lanes b = 0 write them, in result layout), the third stays in
//     the accumulator registers;
// then the lanes b = 0, 1 read their stash back into the accumulator registers (block b of a register = column b:
// k_sample4's layout) and the wave runs k_sample4's blocked Cholesky + solves on up to three columns in lockstep
// (finish4_regs; the fourth block of its MFMAs idles).  No K x K LDS matrix assembly, no memory hand-over, no wave waits
// for another, and the registers of k_sample1: three waves per SIMD.  ~1 100 VALU instructions per column instead of
// ~2 250, a third of the workgroups: on the ML-1M shape one generation of waves with two or three items each
// instead of two generations with one.
// (First form of this kernel: four items per wave, the totals kept in 72 stash REGISTERS -- two waves per SIMD.
// Parity-green, 66 / 64 us per launch on the ML-1M shape against k_sample1's 47 / 52: at two waves per SIMD neither the
// gather latency of the Gram nor the start-up latency of four items in a row is hidden.)
// R is the Cholesky factor of the reference's Lambda* in the reference's index order (natural blocks), so
// x = R^-1 (R^-T b + z) is the reference's sample for the same z (c++/sample.cpp:306-323).
#pragma once
#include "kernels.h"
#include "kernels_q4.h"

namespace bpmf {

template <int K>
struct GeoX {
    static constexpr int NG = K / 4;
    static constexpr int NB = NG * (NG + 1) / 2;
    static constexpr int WPS = K == 32 ? 2 : 4;           // (3 -- the registers of k_sample1 -- spills inside the Gram loop and the factorisation: 80 / 80 us per launch against 48 / 54)
    static constexpr int NITEM = 3;                       // work items per wave: two stashed in LDS + one in the accumulators
    static constexpr int SWORDS = NB * 16;                // doubles of one stashed column (result layout of its 16 lanes)
    __host__ __device__ static constexpr int blk(int g, int g2) { return g * NG - (g * (g - 1)) / 2 + (g2 - g); }
};

// gram_chunk44 (kernels.h) on natural blocks: lane (k, b, x) feeds rating slot 4 k + b with u[4 g + x], g = 0 .. NG - 1
// (NG 8-byte loads per lane and 16 ratings instead of NG / 2 16-byte ones; the same bytes).  Everything else -- 64-rating
// index blocks loaded two ahead, row ids through ds_bpermute, gathers one group ahead of the MFMAs in two alternating
// operand sets, padding slots gathering the zero row -- is gram_chunk44's.
template <int K>
__device__ __forceinline__ void gram_chunk44n(const int32_t *__restrict__ rowidx, const double *__restrict__ vals, int len,
                                              const double *__restrict__ other, const double *__restrict__ zero_row,
                                              double mean, double alpha, IdxBlock cur, IdxBlock nxt,
                                              double (&acc)[GeoX<K>::NB], double (&rr)[GeoX<K>::NG], int lane, int rowmask = -1)
{
    using G = GeoX<K>;
    constexpr int NG = G::NG;
    const int slot = lane >> 2, x = lane & 3;
    // (one address register for the four groups of a block: the group's 64 bytes go into the offset field of ds_bpermute)
    const int perm_base = slot * 4;
    auto gather = [&](const IdxBlock &ib, int gg, double (&R)[NG], double &ww) {
        const int src = gg * 16 + slot;
        const int row = __builtin_amdgcn_ds_bpermute(perm_base + gg * 64, ib.ri);
        const double v = __hiloint2double(__builtin_amdgcn_ds_bpermute(perm_base + gg * 64, __double2hiint(ib.v)),
                                          __builtin_amdgcn_ds_bpermute(perm_base + gg * 64, __double2loint(ib.v)));
        ww = (v - mean) * alpha;                                                  // c++/sample.cpp:256 (padding slots: times a row of zeros)
        const double *p = ((ib.base + src < len) ? other + (size_t)(row & rowmask) * K : zero_row) + x;
#pragma unroll
        for (int g = 0; g < NG; ++g) R[g] = p[4 * g];
    };
    auto contract = [&](const double (&R)[NG], double ww) {
#pragma unroll
        for (int g = 0; g < NG; ++g) rr[g] = fma(R[g], ww, rr[g]);
        int blk = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int g2 = g; g2 < NG; ++g2, ++blk) acc[blk] = mfma44(R[g], R[g2], acc[blk]);
    };
    if (len <= 0) return;
    double yA[NG], yB[NG];
    double wA, wB = 0.0;
    gather(cur, 0, yA, wA);
    int b0 = 0;
    for (; b0 + 64 < len; b0 += 64) {
        gather(cur, 1, yB, wB);
        contract(yA, wA);
        gather(cur, 2, yA, wA);
        contract(yB, wB);
        gather(cur, 3, yB, wB);
        contract(yA, wA);
        gather(nxt, 0, yA, wA);                                                  // first group of the next block
        // the index block after the next one: requested here, a whole block (144 MFMAs) ahead of its first use --
        // and not a block earlier, where it would be three more live registers through the loop
        cur = nxt;
        nxt = load_idx_block(rowidx, vals, b0 + 128, lane, len, zero_row);
        contract(yB, wB);
    }
    const int ng = (len - b0 + 15) >> 4;                                         // last block: 1..4 groups
    if (ng > 1) gather(cur, 1, yB, wB);
    contract(yA, wA);
    if (ng > 1) {
        if (ng > 2) gather(cur, 2, yA, wA);
        contract(yB, wB);
        if (ng > 2) {
            if (ng > 3) gather(cur, 3, yB, wB);
            contract(yA, wA);
            if (ng > 3) contract(yB, wB);
        }
    }
}

// The columns of the stash in lockstep: k_sample4's factorisation (kernels_q4.h) on registers.
//   S[blk(g, g2)] at lane (i, b, j): G[4 g + i][4 g2 + j] of the column in slot b;  Sb[g] at lane (i, b, .): rhs sum 4 g + i;
//   fcol: local column of this lane's slot (-1: empty: factorises LambdaF alone and stores nothing);  sz[b][.]: its normals.
template <int K>
__device__ __forceinline__ void finish4_regs(const SampleArgs &a, double (&acc)[GeoX<K>::NB], const double (&Sb)[GeoX<K>::NG], int fcol,
                                             const double *sz, int lane)
{
    using G = GeoX<K>;
    constexpr int NG = G::NG;
    const int kq = lane >> 4, b = (lane >> 2) & 3, x = lane & 3;
    const int ii = kq, jj = x;
    const bool alive = fcol >= 0;
    double bv[NG];
    // Lambda* = LambdaF + alpha G (:297-298); b = LambdaF mu + rr (:285,:256).
    const double *LF = (a.prop_lambda && alive) ? a.prop_lambda + (size_t)fcol * K * K : a.LambdaF;
    // LambdaF mu of this lane's NG rhs elements.  Per-column priors (:285, rr = Lambda_i * hp.mu): a rolled loop ahead of
    // everything else -- no branch may sit between the loads below and their uses (kernels_q1.h).
#pragma unroll
    for (int g = 0; g < NG; ++g) bv[g] = a.Lmu[4 * g + ii];
    if (a.prop_lambda) {                                              // wave-uniform
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            double lm = 0.0;
#pragma unroll 1
            for (int q = 0; q < K; ++q) lm = fma(LF[4 * g + ii + q * K], a.mu[q], lm);
            bv[g] = lm;
        }
    }
    // LambdaF (L2 hits) one block row at a time: all NB of them in flight at once would be NB more live registers
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        double lf[NG];
#pragma unroll
        for (int g2 = g; g2 < NG; ++g2) lf[g2] = LF[4 * g + ii + (4 * g2 + jj) * K];
#pragma unroll
        for (int g2 = g; g2 < NG; ++g2) {
            const int r_ = 4 * g + ii, c_ = 4 * g2 + jj;
            double v = fma(a.alpha, alive ? acc[G::blk(g, g2)] : 0.0, lf[g2]);
            v = (a.diag_only && r_ != c_) ? 0.0 : v;                 // BPMF_NO_COVARIANCE (:300-304)
            acc[G::blk(g, g2)] = v;
        }
        bv[g] = (jj == 0) ? bv[g] + (alive ? Sb[g] : 0.0) : 0.0;
        __builtin_amdgcn_sched_barrier(0);
    }

    // blocked Cholesky Lambda* = R^T R (:306) + forward solve (:321), four columns in lockstep (see k_sample4)
    double WB[NG];
    const int quadbase = (lane & 0xC);
#pragma unroll
    for (int s = 0; s < NG; ++s) {
        const double dblk = acc[G::blk(s, s)];
        const double d00 = __shfl(dblk, 0 + quadbase + 0), d01 = __shfl(dblk, 0 + quadbase + 1), d02 = __shfl(dblk, 0 + quadbase + 2),
                     d03 = __shfl(dblk, 0 + quadbase + 3), d11 = __shfl(dblk, 16 + quadbase + 1), d12 = __shfl(dblk, 16 + quadbase + 2),
                     d13 = __shfl(dblk, 16 + quadbase + 3), d22 = __shfl(dblk, 32 + quadbase + 2), d23 = __shfl(dblk, 32 + quadbase + 3),
                     d33 = __shfl(dblk, 48 + quadbase + 3);
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
        const double W01 = -i0 * R01 * i1, W12 = -i1 * R12 * i2, W23 = -i2 * R23 * i3;
        const double W02 = -i0 * fma(R01, W12, R02 * i2);
        const double W13 = -i1 * fma(R12, W23, R13 * i3);
        const double W03 = -i0 * fma(R01, W13, fma(R02, W23, R03 * i3));
        auto pick = [&](int p, int q) -> double {
            double v = 0.0;
            v = (p == 0 && q == 0) ? i0 : v; v = (p == 1 && q == 1) ? i1 : v; v = (p == 2 && q == 2) ? i2 : v; v = (p == 3 && q == 3) ? i3 : v;
            v = (p == 0 && q == 1) ? W01 : v; v = (p == 0 && q == 2) ? W02 : v; v = (p == 0 && q == 3) ? W03 : v;
            v = (p == 1 && q == 2) ? W12 : v; v = (p == 1 && q == 3) ? W13 : v; v = (p == 2 && q == 3) ? W23 : v;
            return v;
        };
        const double WA = pick(kq, x);
        WB[s] = pick(x, kq);
        bv[s] = mfma44(WA, bv[s], 0.0);                               // forward solve of this block row: y_s = W^T b_s
#pragma unroll
        for (int J = s + 1; J < NG; ++J) acc[G::blk(s, J)] = mfma44(WA, acc[G::blk(s, J)], 0.0);      // panel R_sJ = W^T A_sJ
#pragma unroll
        for (int I = s + 1; I < NG; ++I) {                            // trailing update A_IJ -= R_sI^T R_sJ, rhs b_J -= R_sJ^T y_s
            const double nI = -acc[G::blk(s, I)];
            bv[I] = mfma44(nI, bv[s], bv[I]);
#pragma unroll
            for (int J = I; J < NG; ++J) acc[G::blk(I, J)] = mfma44(nI, acc[G::blk(s, J)], acc[G::blk(I, J)]);
        }
        __builtin_amdgcn_sched_barrier(0);                            // (the next step's pivot algebra is not to be pulled up into this one: registers)
    }
    // y += z (:322); backward solve R x = y (:323)
#pragma unroll
    for (int g = 0; g < NG; ++g) bv[g] += (jj == 0 && alive) ? sz[b * K + 4 * g + ii] : 0.0;
    const int tsrc = 16 * x + quadbase + kq;
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
    // items().col(idx) = rr (:324); a failed factorisation (:308) shows as a non-finite sample
    if (alive && jj == 0) {
        double *dst = a.items + (size_t)(a.col_from + fcol) * K + ii;
#pragma unroll
        for (int g = 0; g < NG; ++g) dst[4 * g] = bv[g];
    }
    bool bad = false;
#pragma unroll
    for (int g = 0; g < NG; ++g) bad |= !(fabs(bv[g]) <= 1.79769313486231570815e+308);
    bad = bad && alive && jj == 0;
    if (bad) atomicMin(a.fail, (unsigned long long)(a.col_from + fcol));
}

// grid: [gate workgroup] + f.nstat statistics riders + nwaves item waves (k_sample1's launch format); wave w takes the
// items  c * nwaves + (c odd ? nwaves - 1 - w : w),  c = 0 .. NITEM - 1, of the cost-sorted list.
template <int K>
__global__ __launch_bounds__(64, GeoX<K>::WPS) void k_sample1x(SampleArgs a, FusedArgs f, int nwaves)
{
    using G = GeoX<K>;
    constexpr int NG = G::NG, NB = G::NB, PART = Geo44<K>::PART, NITEM = G::NITEM;
    __shared__ __attribute__((aligned(16))) double sz[NITEM * K];    // normals of the wave's columns
    __shared__ __attribute__((aligned(16))) double sSb[NITEM * K];   // their rhs sums
    __shared__ __attribute__((aligned(16))) double sS[(NITEM - 1) * G::SWORDS];   // Gram totals of the first NITEM - 1
    const int lane_in = threadIdx.x;
    int bid = blockIdx.x;
    if (f.gate_host) {
        if (bid == 0) { gate_stage_body(0, 1, f.gate_host, f.gate_want, f.src_host, f.dst, f.n, f.dflag, f.dval, a.tmo, a.wait_ticks); return; }
        --bid;
    }
    if (bid < f.nstat) {
        colstats_body<K>(bid, f.st_items, f.st_c0, f.st_c1, f.nstat, f.st_partials, f.st_fail, f.st_out, f.st_ticket, f.st_flag, f.st_seq,
                         f.st_tmo, a.wait_ticks);
        return;
    }
    const int w = bid - f.nstat;
    double acc[NB], rr[NG];
    int cols[NITEM];                                                  // wave-uniform: local column of slot c, or -1
#pragma unroll
    for (int c = 0; c < NITEM; ++c) cols[c] = -1;

    // The schedule entries of the wave's items first (scalar loads), and the first index blocks of item 0: every later
    // item's index blocks are requested as soon as the previous item's Gram loop has ended -- ahead of its chunk
    // hand-over / b-sum / stash -- so that an item does not start with two dependent memory round trips.
    int icol[NITEM], ilen[NITEM], imc[NITEM], ichunk[NITEM];
    int64_t ip0[NITEM];
#pragma unroll
    for (int c = 0; c < NITEM; ++c) {
        const int it = c * nwaves + ((c & 1) ? nwaves - 1 - w : w);
        const bool ok = it < a.nwork;
        const int its = ok ? it : 0;
        icol[c] = a.wi_col[its]; ip0[c] = a.wi_p0[its]; imc[c] = a.wi_mc[its]; ichunk[c] = a.wi_chunk[its];
        ilen[c] = ok ? a.wi_len[its] : -1;                            // (-1: no such item)
    }
    auto first_blocks = [&](int c, IdxBlock &i0, IdxBlock &i1) {
        int64_t p0 = ip0[0]; int len = ilen[0];
#pragma unroll
        for (int q = 1; q < NITEM; ++q) { p0 = (q == c) ? ip0[q] : p0; len = (q == c) ? ilen[q] : len; }
        const int glen = (a.ablate & 2u) ? 0 : (len < 0 ? 0 : len);
        i0 = load_idx_block(a.rowidx + p0, a.vals + p0, 0, lane_in, glen, a.zero_row);
        i1 = load_idx_block(a.rowidx + p0, a.vals + p0, 64, lane_in, glen, a.zero_row);
    };
    IdxBlock ib0, ib1;
    first_blocks(0, ib0, ib1);

    // (rolled: one copy of the Gram loop; unrolled, the three copies spill inside their MFMA loops)
#pragma unroll 1
    for (int c = 0; c < NITEM; ++c) {
        int col = icol[0], len = ilen[0], mc = imc[0], chunk = ichunk[0];
        int64_t p0 = ip0[0];
#pragma unroll
        for (int q = 1; q < NITEM; ++q) {
            col = (q == c) ? icol[q] : col; len = (q == c) ? ilen[q] : len; mc = (q == c) ? imc[q] : mc;
            chunk = (q == c) ? ichunk[q] : chunk; p0 = (q == c) ? ip0[q] : p0;
        }
        if (len < 0) break;                                           // wave-uniform: the list is exhausted (later thirds hold later items)
        // (opaque copies of the lane id: nothing derived from it is to be carried across the Gram loop or out of this loop)
        int lane = lane_in;
        asm volatile("" : "+v"(lane));
        const int glen = (a.ablate & 2u) ? 0 : len;
        // whole column in one item: its normals in the shadow of the index loads
        if (mc < 0) draw_normals<K>(sample_counter(a.col_from + col, a.ktrue, a.iter_plus_1), a.ktrue, sz + c * K, lane, K);
#pragma unroll
        for (int t = 0; t < NB; ++t) acc[t] = 0.0;
#pragma unroll
        for (int t = 0; t < NG; ++t) rr[t] = 0.0;
        gram_chunk44n<K>(a.rowidx + p0, a.vals + p0, glen, a.other_items, a.zero_row, a.mean_rating, a.alpha, ib0, ib1, acc, rr, lane,
                         (a.ablate & 4u) ? 63 : -1);
        if (c + 1 < NITEM) first_blocks(c + 1, ib0, ib1);             // the next item's index blocks: in flight from here
        lane = lane_in;
        asm volatile("" : "+v"(lane));
        if (mc >= 0) {
            // chunk of a heavy column: park the accumulators; whichever chunk arrives last adds them up (chunk order)
            const int nch = a.mc_nchunks[mc];
            double *pbase = a.partials + (size_t)a.mc_slot0[mc] * PART;
            double *p = pbase + (size_t)chunk * PART;
#pragma unroll
            for (int t = 0; t < NB; ++t) __hip_atomic_store(&p[t * 64 + lane], acc[t], BPMF_RLX_AGENT);
#pragma unroll
            for (int t = 0; t < NG; ++t) __hip_atomic_store(&p[(NB + t) * 64 + lane], rr[t], BPMF_RLX_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned tk = 0;
            if (lane == 0) tk = __hip_atomic_fetch_add(&a.mc_count[mc], 1u, BPMF_RLX_AGENT);
            tk = __builtin_amdgcn_readfirstlane(tk);
            if ((int)tk != nch - 1) continue;                         // (this slot of the wave stays empty)
            if (lane == 0) __hip_atomic_store(&a.mc_count[mc], 0u, BPMF_RLX_AGENT);
            draw_normals<K>(sample_counter(a.col_from + col, a.ktrue, a.iter_plus_1), a.ktrue, sz + c * K, lane, K);
#pragma unroll
            for (int t2 = 0; t2 < NB; ++t2) acc[t2] = 0.0;
#pragma unroll
            for (int t2 = 0; t2 < NG; ++t2) rr[t2] = 0.0;
            for (int ch = 0; ch < nch; ++ch) {
                const double *pc = pbase + (size_t)ch * PART;
                double tmp[NB + NG];                                  // all loads of a chunk in flight, then the adds (chunk order)
#pragma unroll
                for (int t2 = 0; t2 < NB + NG; ++t2) tmp[t2] = __hip_atomic_load(&pc[t2 * 64 + lane], BPMF_RLX_AGENT);
#pragma unroll
                for (int t2 = 0; t2 < NB; ++t2) acc[t2] += tmp[t2];
#pragma unroll
                for (int t2 = 0; t2 < NG; ++t2) rr[t2] += tmp[NB + t2];
            }
        }
#pragma unroll
        for (int q = 0; q < NITEM; ++q) cols[q] = (q == c) ? col : cols[q];
        // the four b of every accumulator (fixed order): every lane gets the total; slots 0 .. NITEM - 2 stash it
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            const double v = row_ror_add<0x128>(acc[t]);
            acc[t] = row_ror_add<0x124>(v);
        }
        const int b = (lane >> 2) & 3;
        const int l16 = 4 * (lane >> 4) + (lane & 3);                 // (i, j) of the result view
        if (c < NITEM - 1) {
            if (b == 0) {
#pragma unroll
                for (int t = 0; t < NB; ++t) sS[c * G::SWORDS + t * 16 + l16] = acc[t];
            }
        }
        // rhs sums: over the four b and the four k; element 4 g + x by the lanes (0, 0, x)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            double v = row_ror_add<0x128>(rr[g]);
            v = row_ror_add<0x124>(v);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 4) sSb[c * K + 4 * g + lane] = v;
        }
    }
    bool any = false;
#pragma unroll
    for (int c = 0; c < NITEM; ++c) any |= cols[c] >= 0;
    if (!any) return;                                                 // nothing but unfinished chunks in this wave
    const int lane = lane_in;
    const int b = (lane >> 2) & 3;
    const int l16 = 4 * (lane >> 4) + (lane & 3);
    int fcol = -1;
#pragma unroll
    for (int c = 0; c < NITEM; ++c) fcol = (b == c) ? cols[c] : fcol;
    if (a.ablate & 1u) {                                              // (profiling switch: Gram only -- keep it live)
        double v = rr[0];
#pragma unroll
        for (int t = 0; t < NB; ++t) v += acc[t];
        if (fcol >= 0 && l16 == 0) a.items[(size_t)(a.col_from + fcol) * K] = v;
        return;
    }
    __syncthreads();                                                  // (single wave: stash, rhs sums and normals are in LDS)
    if (b < NITEM - 1) {
#pragma unroll
        for (int t = 0; t < NB; ++t) acc[t] = sS[b * G::SWORDS + t * 16 + l16];
    }
    double Sb[NG];
    {
        const int bs = b < NITEM ? b : NITEM - 1;
#pragma unroll
        for (int g = 0; g < NG; ++g) Sb[g] = sSb[bs * K + 4 * g + (lane >> 4)];
    }
    wait_params(a);
    finish4_regs<K>(a, acc, Sb, fcol, sz, lane);
}

}  // namespace bpmf
