// kernels_q1.h -- k_sample1q<K> (K <= 32): the Gram of a column by ONE wave, the factorisation of FOUR columns by one wave.
//
// Reference: Sys::sample(long idx, Sys&) + computeMuLambda, c++/sample.cpp:248-336.
//
// k_sample1 (kernels.h) gives every work item a wave of its own -- the right grain for the Gram of a side with a
// few thousand columns -- but then factorises each column alone on the VALU: ~1 900 instructions per column against
// ~170 at full lane efficiency, 57 % of the SIMD time of the ML-1M launch.  k_sample4 (kernels_q4.h) factorises four
// columns in lockstep on the 4x4x4 f64 MFMA (~370 VALU instructions + 46 MFMAs per column), but forms their Grams in
// lockstep too: a quarter of the waves, which an ML-1M-sized side cannot afford (1 510 waves on 1 024 SIMDs).
// Here the two are joined through memory instead of through lockstep:
//
//   * every work item (column or chunk of a heavy column) is a single-wave workgroup as in k_sample1: index blocks,
//     normal draw in their shadow, gram_chunk44, chunk partials / last-arriver sum -- unchanged;
//   * the wave that holds a column's complete Gram adds the four b-partials of every accumulator (two DPP row
//     rotates), and writes G, the rhs sums and the K normals into the column's SLOT of its group's scratch area --
//     in the operand layout of k_sample4's factorisation (natural 4-index blocks; the permuted block order of
//     gram_chunk44's 16-byte gathers is undone by the store addresses) -- with write-through stores, drains, and takes
//     a ticket of the group (relaxed device-scope counter: the hand-off idiom of the chunk partials).  A wave that is
//     not the last of its group is done: its slot frees up for the next work item at once;
//   * the wave drawing the last ticket loads the group (one coalesced 512-byte load per block: lane (i, slot, j)
//     gets entry (i, j) of the block of column `slot`), waits for the hyper-parameters if they are still on their
//     way (wait_params: only the factorisation needs them) and runs k_sample4's blocked Cholesky + solves on the
//     four columns in lockstep.
// Groups are four columns that are next to each other in the cost-sorted item list (build_schedule), so they become
// complete at about the same time; a group never waits for anything but its own columns' Grams, and no wave ever
// waits for another wave: no residency requirement.
// R is the Cholesky factor of the reference's Lambda* in the reference's index order (natural blocks), so
// x = R^-1 (R^-T b + z) is the reference's sample for the same z (c++/sample.cpp:306-323).
#pragma once
#include "kernels.h"
#include "kernels_q4.h"

namespace bpmf {

template <int K>
struct GeoQ {
    static constexpr int NG = K / 4;
    static constexpr int NB = NG * (NG + 1) / 2;
    static constexpr int GWORDS = (NB + 2 * NG) * 64;     // doubles of scratch per group: blocks | rhs | normals, 64 lanes each
    static constexpr int WPS = K == 32 ? 3 : 4;           // as k_sample1: the 4x4x4 Gram of K = 32 keeps <= 168 VGPRs
    __host__ __device__ static constexpr int blk(int g, int g2) { return g * NG - (g * (g - 1)) / 2 + (g2 - g); }
};

// The four columns of a complete group in lockstep: k_sample4's factorisation (kernels_q4.h) fed from the group's scratch
// area `sc`.  gcols: the local columns of the four slots (-1: empty); sw: NG x 64 doubles of LDS.
template <int K>
__device__ __forceinline__ void finish_group4(const SampleArgs &a, const double *sc, int4 gcols, double *sw, int lane)
{
    using GQ = GeoQ<K>;
    constexpr int NG = GQ::NG, NB = GQ::NB;
    double acc[NB];
    using G = Geo4<K>;
    const int kq = lane >> 4, b = (lane >> 2) & 3, x = lane & 3;
    const int ii = kq, jj = x;
    const int fcol = b == 0 ? gcols.x : (b == 1 ? gcols.y : (b == 2 ? gcols.z : gcols.w));
    const bool alive = fcol >= 0;
    double bv[NG];
    wait_params(a);
    // Lambda* = LambdaF + alpha G (:297-298); b = LambdaF mu + rr (:285,:256).  One block row at a time: with all
    // 2 x NB loads in flight at once the wave would need twice the registers the factorisation itself does.
    // (An empty slot of the last group factorises LambdaF alone and stores nothing.)
    const double *LF = (a.prop_lambda && alive) ? a.prop_lambda + (size_t)fcol * K * K : a.LambdaF;
    // LambdaF mu of this lane's NG rhs elements.  Per-column priors (:285, rr = Lambda_i * hp.mu): a rolled loop ahead of
    // everything else -- no branch may sit between the loads below and their uses (the compiler would wait for every
    // load on the spot and park it in scratch across the branch).
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
    // the group's blocks and rhs sums: ONE round trip (they were written through to memory: every load is a memory
    // access), straight into the registers the factorisation keeps them in
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[t] = __hip_atomic_load(&sc[t * 64 + lane], BPMF_RLX_AGENT);
    double rs[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) rs[g] = __hip_atomic_load(&sc[(NB + g) * 64 + (lane & ~3)], BPMF_RLX_AGENT);
    // LambdaF (L2 hits) one block row at a time: all NB of them in flight as well would be NB more live registers
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
        bv[g] = (jj == 0) ? bv[g] + (alive ? rs[g] : 0.0) : 0.0;
        __builtin_amdgcn_sched_barrier(0);
    }

    // blocked Cholesky Lambda* = R^T R (:306) + forward solve (:321), four columns in lockstep
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
        sw[s * 64 + lane] = pick(x, kq);
        bv[s] = mfma44(WA, bv[s], 0.0);
#pragma unroll
        for (int J = s + 1; J < NG; ++J) acc[G::blk(s, J)] = mfma44(WA, acc[G::blk(s, J)], 0.0);
#pragma unroll
        for (int I = s + 1; I < NG; ++I) {
            const double nI = -acc[G::blk(s, I)];
            bv[I] = mfma44(nI, bv[s], bv[I]);
#pragma unroll
            for (int J = I; J < NG; ++J) acc[G::blk(I, J)] = mfma44(nI, acc[G::blk(s, J)], acc[G::blk(I, J)]);
        }
        __builtin_amdgcn_sched_barrier(0);                            // (the next step's pivot algebra is not to be pulled up into this one: registers)
    }
    // y += z (:322); backward solve R x = y (:323)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const double zg = __hip_atomic_load(&sc[(NB + NG + g) * 64 + (lane & ~3)], BPMF_RLX_AGENT);
        bv[g] += (jj == 0 && alive) ? zg : 0.0;
    }
    const int tsrc = 16 * x + quadbase + kq;
#pragma unroll
    for (int s = NG - 1; s >= 0; --s) {
        double t = bv[s];
#pragma unroll
        for (int J = s + 1; J < NG; ++J) {
            const double RT = __shfl(acc[G::blk(s, J)], tsrc);
            t = mfma44(-RT, bv[J], t);
        }
        bv[s] = mfma44(sw[s * 64 + lane], t, 0.0);
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

// SPLIT (BPMF_HIP_MODE=8): the kernel ends when the column's slot is written -- plain visibility through the kernel
// boundary, no write-through wait, no ticket -- and k_finish_groups, the next launch on the same stream, factorises every
// group: all groups' lockstep chains run side by side instead of each at the end of its last column's wave.
template <int K, bool SPLIT = false>
__global__ __launch_bounds__(64, GeoQ<K>::WPS) void k_sample1q(SampleArgs a, FusedArgs f)
{
    using GQ = GeoQ<K>;
    using G4 = Geo44<K>;
    constexpr int NG = GQ::NG, NB = GQ::NB, PART = G4::PART;
    __shared__ __attribute__((aligned(16))) double sz[K];
    __shared__ double sw[GeoQ<K>::NG * 64];                           // factorisation: W_s of every block step, lane-private (operand of the backward solve)
    const int lane = threadIdx.x;
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
    const int col = a.wi_col[w];
    const int64_t p0 = a.wi_p0[w];
    const int len = a.wi_len[w];
    const int mc = a.wi_mc[w];

    // ---- Gram of this work item: k_sample1's (index blocks first, the normals in their shadow, 4x4x4 Gram)
    const int glen = (a.ablate & 2u) ? 0 : len;
    const IdxBlock ib0 = load_idx_block(a.rowidx + p0, a.vals + p0, 0, lane, glen, a.zero_row);
    const IdxBlock ib1 = load_idx_block(a.rowidx + p0, a.vals + p0, 64, lane, glen, a.zero_row);
    if (mc < 0) draw_normals<K>(sample_counter(a.col_from + col, a.ktrue, a.iter_plus_1), a.ktrue, sz, lane, K);
    double acc[NB], rr[NG];
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[t] = 0.0;
#pragma unroll
    for (int t = 0; t < NG; ++t) rr[t] = 0.0;
    gram_chunk44<K>(a.rowidx + p0, a.vals + p0, glen, a.other_items, a.zero_row, a.mean_rating, a.alpha, ib0, ib1, acc, rr, lane,
                    (a.ablate & 4u) ? 63 : -1);
    if (mc >= 0) {
        // chunk of a heavy column: park the accumulators; whichever chunk arrives last adds them up (chunk order)
        const int nch = a.mc_nchunks[mc];
        double *pbase = a.partials + (size_t)a.mc_slot0[mc] * PART;
        double *p = pbase + (size_t)a.wi_chunk[w] * PART;
#pragma unroll
        for (int t = 0; t < NB; ++t) __hip_atomic_store(&p[t * 64 + lane], acc[t], BPMF_RLX_AGENT);
#pragma unroll
        for (int t = 0; t < NG; ++t) __hip_atomic_store(&p[(NB + t) * 64 + lane], rr[t], BPMF_RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(&a.mc_count[mc], 1u, BPMF_RLX_AGENT);
        t = __builtin_amdgcn_readfirstlane(t);
        if ((int)t != nch - 1) return;
        if (lane == 0) __hip_atomic_store(&a.mc_count[mc], 0u, BPMF_RLX_AGENT);
        // (the normals now, while no accumulator is live: log / sqrt want ~60 registers of their own)
        draw_normals<K>(sample_counter(a.col_from + col, a.ktrue, a.iter_plus_1), a.ktrue, sz, lane, K);
#pragma unroll
        for (int t2 = 0; t2 < NB; ++t2) acc[t2] = 0.0;
#pragma unroll
        for (int t2 = 0; t2 < NG; ++t2) rr[t2] = 0.0;
        for (int ch = 0; ch < nch; ++ch) {
            const double *pc = pbase + (size_t)ch * PART;
            double tmp[NB + NG];
#pragma unroll
            for (int t2 = 0; t2 < NB + NG; ++t2) tmp[t2] = __hip_atomic_load(&pc[t2 * 64 + lane], BPMF_RLX_AGENT);
#pragma unroll
            for (int t2 = 0; t2 < NB; ++t2) acc[t2] += tmp[t2];
#pragma unroll
            for (int t2 = 0; t2 < NG; ++t2) rr[t2] += tmp[NB + t2];
        }
    }
    if (a.ablate & 1u) {                                              // (profiling switch: Gram only -- keep it live)
        double v = rr[0];
#pragma unroll
        for (int t = 0; t < NB; ++t) v += acc[t];
        if (lane < K) a.items[(size_t)(a.col_from + col) * K + lane] = v;
        return;
    }

    // ---- the complete Gram goes to slot `slot` of group `grp`, in the layout the factorisation loads
    const int gs = a.q_col_slot[col];
    const int grp = gs >> 2, slot = gs & 3;
    double *sc = a.q_scratch + (size_t)grp * GQ::GWORDS;
    // (split form: plain stores -- the kernel boundary makes them visible; fused form: write-through for the ticket)
    auto put = [](double *p, double v) {
        if constexpr (SPLIT) *p = v;
        else __hip_atomic_store(p, v, BPMF_RLX_AGENT);
    };
    {
        const int i = lane >> 4, b = (lane >> 2) & 3, j = lane & 3;
        const int ih = i >> 1, jh = j >> 1, i1 = i & 1, j1 = j & 1;
        // entry (ri, ci) of natural block (R, C), R <= C, of this column: word of the group's scratch
        auto word = [&](int R, int C, int ri, int ci) { return (R * NG - ((R * (R - 1)) >> 1) + (C - R)) * 64 + 16 * ri + 4 * slot + ci; };
        int blk = 0;
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int g2 = g; g2 < NG; ++g2, ++blk) {
                double v = row_ror_add<0x128>(acc[blk]);              // the four b of every accumulator (fixed order)
                v = row_ror_add<0x124>(v);
                // lane (i, b, j) holds G[r][c], r = idx(g, i), c = idx(g2, j) (gram_chunk44's permuted blocks)
                const int R = 2 * (g >> 1) + ih, C = 2 * (g2 >> 1) + jh, ri = 2 * i1 + (g & 1), ci = 2 * j1 + (g2 & 1);
                if ((g >> 1) < (g2 >> 1)) {                          // strictly upper natural block
                    if (b == 0) put(&sc[word(R, C, ri, ci)], v);
                } else if (g == g2) {                                 // both orientations exist among the lanes: each writes the upper one
                    if (b == 0 && R <= C) put(&sc[word(R, C, ri, ci)], v);
                } else {                                              // g = 2 h, g2 = 2 h + 1: one orientation only
                    const bool up = R <= C;
                    if (b == 0) put(&sc[up ? word(R, C, ri, ci) : word(C, R, ci, ri)], v);
                    if (b == 0 && R == C) put(&sc[word(R, R, ci, ri)], v);      // mirror inside the diagonal block
                }
            }
        // rhs sums: over the four b and the four k; element e = idx(g, x) by the lanes x = 0..3 of quad 0
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            double v = row_ror_add<0x128>(rr[g]);
            v = row_ror_add<0x124>(v);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int e = G4::idx(g, lane & 3);
            if (lane < 4) put(&sc[(NB + (e >> 2)) * 64 + 16 * (e & 3) + 4 * slot], v);
        }
        __syncthreads();                                              // (single wave: the normals are in LDS)
        if (lane < K) put(&sc[(NB + NG + (lane >> 2)) * 64 + 16 * (lane & 3) + 4 * slot], sz[lane]);
    }
    if constexpr (SPLIT) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int4 gcols = *reinterpret_cast<const int4 *>(a.q_grp_cols + 4 * grp);
    const int want = (gcols.x >= 0) + (gcols.y >= 0) + (gcols.z >= 0) + (gcols.w >= 0);
    {
        unsigned t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(&a.q_count[grp], 1u, BPMF_RLX_AGENT);
        t = __builtin_amdgcn_readfirstlane(t);
        if ((int)t != want - 1) return;
        if (lane == 0) __hip_atomic_store(&a.q_count[grp], 0u, BPMF_RLX_AGENT);      // re-arm for the next launch
    }
    if (a.ablate & 8u) return;                                        // (profiling switch: Gram + hand-over only)

    finish_group4<K>(a, sc, gcols, sw, lane);
}

// second launch of the split form: one wave per group of four columns
template <int K>
__global__ __launch_bounds__(64, GeoQ<K>::WPS) void k_finish_groups(SampleArgs a, int ngroups)
{
    __shared__ double sw[GeoQ<K>::NG * 64];
    const int grp = blockIdx.x;
    if (grp >= ngroups) return;
    const int4 gcols = *reinterpret_cast<const int4 *>(a.q_grp_cols + 4 * grp);
    finish_group4<K>(a, a.q_scratch + (size_t)grp * GeoQ<K>::GWORDS, gcols, sw, threadIdx.x);
}

}  // namespace bpmf
