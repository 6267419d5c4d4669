// launch_impl.h -- definitions of launch.h's templates: which kernel of kernels*.h runs for which
// form of a side, and with what arguments.  Included by the per-K units only.
#pragma once
#include "launch.h"
#include "kernels.h"
#include "kernels_f32.h"
#include "kernels_q4.h"
#include "kernels_slab.h"
#include "kernels_q1.h"
#include "kernels_x4.h"

namespace bpmf_launch {


// The device work of one half-iteration, in three pieces that the synchronous (stateless) and
// the asynchronous (stateful) paths put on their streams:
//   launch_sampler: the per-column update, reading the parameter blob `d_in`
//   launch_exchange: multi-GPU only, in-place broadcast of every rank's fresh column range
//   launch_stats: sum x / sum x x^T of this rank's columns (+ all-reduce), published to `out_host_dev`
// F32: the fp32 context (K = 128 only).  K = 128 with F32 = false is the reference's fp64 arithmetic at num_latent 65 .. 128.
template <int K, bool F32>
// ev_start / ev_stop (optional): recorded by the dispatch packet of the sampler itself
// (hipExtLaunchKernel) instead of by marker packets before and after it -- every marker is a few
// microseconds on the stream between two samplers.
int sampler_into(bpmf_hip_side *self, double *out_items, const bpmf_hip_side *other, int iter, double alpha, double *d_in, hipStream_t st,
                        hipEvent_t ev_start, hipEvent_t ev_stop)
{
    using namespace bpmf;
    bpmf_hip_ctx *c = self->ctx;
    auto launch = [&](auto kernel, dim3 grid, dim3 block, auto args) {
        BPMF_LAUNCH(kernel, grid, block, st, ev_start, ev_stop, args);
    };
    // one workgroup per column (k_sample_wg): the fp32 large-K path, and K = 64 in fp64
    auto launch_wg = [&](auto zero) {
        typedef decltype(zero) T;
        SampleArgsW<T> f;
        f.rowidx = self->d_rowidx; f.vals = self->d_vals;
        const int fw0 = self->item_n >= 0 ? self->item_off : 0;
        f.wi_col = self->d_wi_col + fw0; f.wi_p0 = self->d_wi_p0 + fw0; f.wi_len = self->d_wi_len + fw0;
        f.other_items = reinterpret_cast<const T *>(other->d_items); f.items = reinterpret_cast<T *>(out_items);
        f.col_from = self->from;
        f.LambdaF = d_in; f.Lmu = d_in + (size_t)K * K;
        f.fail = (unsigned long long *)(d_in + (size_t)K * K + K);
        f.mu = d_in + (size_t)K * K + K + 2; f.prop_lambda = self->d_prop; f.diag_only = c->diag_only;
        f.mean_rating = self->mean_rating; f.alpha = alpha; f.iter_plus_1 = (uint32_t)(iter + 1); f.ktrue = c->Kt;
        // four waves per column at K = 128; one wave owning all tiles at K = 64 (no idle waves in the
        // serial phases of the factorisation: the column-dominated shapes are what K = 64 is run on)
        const int fnw = self->item_n >= 0 ? self->item_n : self->nwork;
        if (fnw > 0) {
            if constexpr (F32) {
                // two waves per column (18 tiles each) by default: three columns in flight per CU instead of two,
                // and one idle wave instead of three through the serial phases (0.66 -> 0.58 ms per launch)
                if (env_int("BPMF_HIP_WG_WAVES", 2) == 4) launch(k_sample_wg<K, T, 4>, dim3(fnw), dim3(256), f);
                else launch(k_sample_wg<K, T, 2>, dim3(fnw), dim3(128), f);
            }
            else if constexpr (K == 64) k64_wg(fnw, st, ev_start, ev_stop, f);
        }
    };
    if constexpr (F32) {
        if (self->mode == 2) { launch_wg(0.0f); return 0; }
    }
    if constexpr (K == 64) {
        if (self->mode == 2) { launch_wg(0.0); return 0; }
    }
    SampleArgs a;
    a.rowidx = self->d_rowidx; a.vals = self->d_vals;
    // (item window: the whole list, or the items of one part of the columns -- bpmf_hip_side_set_overlap)
    const int w0 = self->item_n >= 0 ? self->item_off : 0, nwork = self->item_n >= 0 ? self->item_n : self->nwork;
    a.wi_col = self->d_wi_col + w0; a.wi_p0 = self->d_wi_p0 + w0; a.wi_len = self->d_wi_len + w0; a.wi_mc = self->d_wi_mc + w0; a.wi_chunk = self->d_wi_chunk + w0;
    a.mc_slot0 = self->d_mc_slot0; a.mc_nchunks = self->d_mc_nch; a.mc_count = self->d_mc_count;
    a.partials = self->d_partials; a.nwork = nwork;
    a.other_items = other->d_items; a.items = out_items; a.col_from = self->from;
    a.LambdaF = d_in; a.Lmu = d_in + (size_t)K * K;
    a.fail = (unsigned long long *)(d_in + (size_t)K * K + K);
    a.mu = d_in + (size_t)K * K + K + 2; a.prop_lambda = self->d_prop; a.diag_only = c->diag_only;
    a.mean_rating = self->mean_rating; a.alpha = alpha; a.iter_plus_1 = (uint32_t)(iter + 1); a.ktrue = c->Kt;
    a.ablate = c->ablate; a.stamps = c->d_stamps; a.wt_store = 0u;
    a.gate_flag = self->cur_gate_flag; a.gate_want = self->cur_gate_want;
    a.tmo = self->cur_gate_flag ? tmo_word(self->a_h_out_dev, K) : nullptr; a.wait_ticks = wait_ticks();
    a.zero_row = c->d_zero;
    a.lf32 = (lf32_words(c) && !self->d_prop) ? reinterpret_cast<const float *>(d_in + c->in_words) : nullptr;
    a.q_col_slot = self->d_q_col_slot; a.q_grp_cols = self->d_q_grp_cols; a.q_count = self->d_q_count; a.q_scratch = self->d_q_scratch;
    bpmf::StatRiders rr = self->cur_riders;                          // (K = 128 only)
    if (rr.tail) { rr.items = out_items; rr.nitems = nwork; a.wt_store = 1u; }   // this launch's own columns: the copy it writes
    if constexpr (F32) {                                             // fp32 factors (items / other_items are float arrays)
        if (nwork > 0) {
            if (self->mode == 4) k128_slab(nwork, st, ev_start, ev_stop, a);
            else k128_wg2(nwork, env_int("BPMF_HIP_WG_WAVES", 2) == 4 ? 4 : 2, st, ev_start, ev_stop, a, rr);
        }
        return 0;
    } else if constexpr (K == 128) {                                 // fp64 factors, workgroup of four waves per item (kernels_wg2.h, T = double)
        if (nwork > 0) k128_wg2_f64(nwork, env_int("BPMF_HIP_WG_WAVES_F64", 4) == 2 ? 2 : 4, st, ev_start, ev_stop, a, rr);
        return 0;
    } else {
    if constexpr (K <= 32) {
        if (nwork > 0 && self->mode == 3) {                          // four columns per wave (k_sample4)
            launch(k_sample4<K>, dim3((nwork + 3) / 4), dim3(64), a);
            return 0;
        }
    }
    if constexpr (K == 64) {
        if (self->lr_n > 0 && !self->d_prop && !c->diag_only && !(c->ablate & 3u)) {
            // light columns: rank-n update of the shared factor of LambdaF (k_sample_lr); the others as usual
            if (self->hv_nwork > 0) {
                a.wi_col = self->d_hv_col; a.wi_p0 = self->d_hv_p0; a.wi_len = self->d_hv_len; a.wi_mc = self->d_hv_mc;
                a.wi_chunk = self->d_hv_chunk; a.nwork = self->hv_nwork;
                if (self->mode == 4) {
                    k64_slab(self->hv_nwork, st, ev_start, nullptr, a);
                } else if (self->mode == 1) {
                    const FusedArgs f0{};
                    BPMF_LAUNCH(k_sample1<K>, dim3(self->hv_nwork), dim3(64), st, ev_start, (hipEvent_t) nullptr, a, f0);
                } else {
                    const int grid = std::min(self->hv_nwork, env_int("BPMF_HIP_GRID", c->num_cu * 4 * Geo<K>::WPS));
                    k64_persistent(grid, st, ev_start, nullptr, a);
                }
            }
            LrArgs l;
            l.rowidx = self->d_rowidx; l.vals = self->d_vals; l.col = self->d_lr_col; l.p0 = self->d_lr_p0; l.len = self->d_lr_len;
            l.nitems = self->lr_n; l.other_items = other->d_items; l.items = out_items; l.col_from = self->from;
            l.R0 = d_in + (size_t)K * K + K + 2 + K; l.S0t = l.R0 + (size_t)K * K; l.y0 = l.S0t + (size_t)K * K;
            l.Lmu = a.Lmu; l.fail = a.fail;
            l.mean_rating = self->mean_rating; l.alpha = alpha; l.sqrt_alpha = std::sqrt(alpha); l.iter_plus_1 = (uint32_t)(iter + 1); l.ktrue = c->Kt;
            // product form for the columns with <= 6 ratings: two instantiations (<= 2, <= 6), persistent workgroups
            // of eight waves with R0^-1 in LDS; then one launch per sweep width for the rest (the events ride on
            // the first / last launch of the side)
            l.Q = self->d_pf_q;
            int first = 0, last = 0;
            for (int cls = 1; cls <= 4; ++cls) if (self->lr_class[cls] > self->lr_class[cls - 1]) { if (!first) first = cls; last = cls; }
            bool started = self->hv_nwork > 0;
            if (self->pf_class[3] > self->pf_class[0] && self->d_pf_q) {
                // Q = U_other R0^-1 once per half-iteration (k_pf_prepare), ahead of the product-form launches
                const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((self->nrows + 7) / 8, (int64_t)c->num_cu * 4));
                k64_pf_prepare(grid, st, started ? nullptr : ev_start, l.S0t, other->d_items, self->nrows, self->d_pf_q);
                started = true;
            }
            int last_pf = -1, npf = 0;
            for (int pc = 0; pc < 3; ++pc) if (self->pf_class[pc + 1] > self->pf_class[pc]) { last_pf = pc; ++npf; }
            const int pf_merge = env_int("BPMF_HIP_PF_MERGE", 0);     // (read per launch: the tests flip it)
            if (npf > 1 && pf_merge && !self->d_stat_list) {
                // the three classes in one launch of persistent workgroups, passes dealt round-robin, most expensive first (k_sample_pf_all)
                LrArgs lc = l;
                for (int q = 0; q < 4; ++q) lc.pf_c[q] = self->pf_class[q];
                lc.nitems = self->pf_class[3];
                const bool is_last = last == 0;
                int npass = 0;
                for (int pc = 0; pc < 3; ++pc) npass += (self->pf_class[pc + 1] - self->pf_class[pc] + 3) / 4;
                const int grid = std::max(1, std::min((npass + 7) / 8, c->num_cu * 2));       // two workgroups per CU are resident (58 KB of LDS each)
                k64_pf_all(grid, st, started ? nullptr : ev_start, is_last ? ev_stop : nullptr, lc);
                started = true;
            } else
            for (int pc = 0; pc < 3; ++pc) {
                const int n0 = self->pf_class[pc], n1 = self->pf_class[pc + 1];
                if (n1 <= n0) continue;
                LrArgs lc = l;
                lc.col = l.col + n0; lc.p0 = l.p0 + n0; lc.len = l.len + n0; lc.nitems = n1 - n0;
                const bool is_last = last == 0 && pc == last_pf;
                hipEvent_t e0 = started ? nullptr : ev_start, e1 = is_last ? ev_stop : nullptr;
                if (pc == 0 && !is_last && self->d_stat_list && self->ev_stat_a && self->stat_nA > 0 && self->item_n < 0 && !(c->comm != nullptr && !self->bounds.empty())) {
                    e1 = self->ev_stat_a;                                 // group A of the statistics may start behind this launch
                    self->stat_a_ready = true;
                }
                started = true;
                const int grid = std::max(1, std::min((n1 - n0 + 31) / 32, c->num_cu * 4));     // eight waves x four columns per pass
                k64_pf(pc, grid, st, e0, e1, lc);
            }
            for (int cls = 1; cls <= 4; ++cls) {
                const int n0 = self->lr_class[cls - 1], n1 = self->lr_class[cls];
                if (n1 <= n0) continue;
                LrArgs lc = l;
                lc.col = l.col + n0; lc.p0 = l.p0 + n0; lc.len = l.len + n0; lc.nitems = n1 - n0;
                hipEvent_t e0 = (cls == first && !started) ? ev_start : nullptr, e1 = (cls == last) ? ev_stop : nullptr;
                k64_lr(cls, n1 - n0, st, e0, e1, lc);
            }
            return 0;
        }
    }
    if constexpr (K == 64) {
        if (nwork > 0 && self->mode == 4) {
            const FusedArgs &f = self->cur_fused;                    // (all zero outside the fused stateful path)
            if (f.gate_host || f.nstat) {                            // gate workgroup + statistics riders + items in one launch
                const dim3 grid((unsigned)(nwork + (f.gate_host ? 1 : 0) + f.nstat));
                BPMF_LAUNCH(k_sample1s<K>, grid, dim3(64), st, ev_start, ev_stop, a, f);
            } else {
                k64_slab(nwork, st, ev_start, ev_stop, a);
            }
            return 0;
        }
    }
    if constexpr (K <= 32) {
        if (nwork > 0 && self->mode == 6) {                          // Gram one column per wave, factorisation four columns per wave
            const FusedArgs &f = self->cur_fused;
            const dim3 grid((unsigned)(nwork + (f.gate_host ? 1 : 0) + f.nstat));
            BPMF_LAUNCH(k_sample1q<K>, grid, dim3(64), st, ev_start, ev_stop, a, f);
            return 0;
        }
        if (nwork > 0 && self->mode == 8) {                          // the same in two launches: Grams, then every group's factorisation side by side
            const FusedArgs &f = self->cur_fused;
            const dim3 grid((unsigned)(nwork + (f.gate_host ? 1 : 0) + f.nstat));
            BPMF_LAUNCH((k_sample1q<K, true>), grid, dim3(64), st, ev_start, (hipEvent_t) nullptr, a, f);
            BPMF_LAUNCH(k_finish_groups<K>, dim3((unsigned)self->q_ngroups), dim3(64), st, (hipEvent_t) nullptr, ev_stop, a, self->q_ngroups);
            return 0;
        }
    }
    if constexpr (K <= 32) {
        if (nwork > 0 && self->mode == 7) {                          // up to four items per wave one after the other, factorised in lockstep
            const FusedArgs &f = self->cur_fused;
            // as many waves as the chip holds at this kernel's occupancy (every wave then walks nwork / nwaves items of
            // about the same total length), more only when that would be more than four items per wave
            static const int waves_env = env_int("BPMF_HIP_X4_WAVES", 0);
            const int slots = waves_env > 0 ? waves_env : c->num_cu * 4 * GeoX<K>::WPS;
            const int nwaves = std::max(std::min(nwork, slots), (nwork + GeoX<K>::NITEM - 1) / GeoX<K>::NITEM);
            const dim3 grid((unsigned)(nwaves + (f.gate_host ? 1 : 0) + f.nstat));
            BPMF_LAUNCH(k_sample1x<K>, grid, dim3(64), st, ev_start, ev_stop, a, f, nwaves);
            return 0;
        }
    }
    if (nwork > 0 && self->mode == 1) {
        const FusedArgs &f = self->cur_fused;                        // (all zero outside the fused stateful path)
        const dim3 grid((unsigned)(nwork + (f.gate_host ? 1 : 0) + f.nstat));
        if constexpr (K == 32 || K == 16) {
            // the slab form of the item body (kernels_slab.h) behind the same launch format: BPMF_HIP_SLAB32
            static const int slab = env_int("BPMF_HIP_SLAB32", 0);
            if (slab) {
                BPMF_LAUNCH(k_sample1s<K>, grid, dim3(64), st, ev_start, ev_stop, a, f);
                return 0;
            }
        }
        BPMF_LAUNCH(k_sample1<K>, grid, dim3(64), st, ev_start, ev_stop, a, f);
    } else if (nwork > 0) {
        // persistent waves: as many single-wave workgroups as the chip holds at this kernel's occupancy
        const int resident = c->num_cu * 4 * Geo<K>::WPS;
        const int grid = std::min(nwork, env_int("BPMF_HIP_GRID", resident));
        if constexpr (K == 64) k64_persistent(grid, st, ev_start, ev_stop, a);
        else launch(k_sample<K>, dim3(grid), dim3(64), a);
    }
    return 0;
    }
}

template <int K, bool F32>
int sampler_pair(bpmf_hip_side *A, double *outA, int iterA, double *d_inA, const bpmf::FusedArgs &fa,
                 bpmf_hip_side *B, double *outB, int iterB, double *d_inB, const bpmf::FusedArgs &fb,
                 unsigned gate_wantA, unsigned gate_wantB, double alpha, const bpmf::PairArgs &p, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    using namespace bpmf;
    if constexpr (K > 32 || F32) return fail(BPMF_HIP_EINVAL, "pair launch: K <= 32 in fp64 only");
    else {
    bpmf_hip_ctx *c = A->ctx;
    auto fill = [&](SampleArgs &a, bpmf_hip_side *self, double *out_items, const double *other_items, int iter, double *d_in, unsigned gate_want) {
        a = SampleArgs{};
        a.rowidx = self->d_rowidx; a.vals = self->d_vals;
        a.wi_col = self->d_wi_col; a.wi_p0 = self->d_wi_p0; a.wi_len = self->d_wi_len; a.wi_mc = self->d_wi_mc; a.wi_chunk = self->d_wi_chunk;
        a.mc_slot0 = self->d_mc_slot0; a.mc_nchunks = self->d_mc_nch; a.mc_count = self->d_mc_count;
        a.partials = self->d_partials; a.nwork = self->nwork;
        a.other_items = other_items; a.items = out_items; a.col_from = self->from;
        a.LambdaF = d_in; a.Lmu = d_in + (size_t)K * K;
        a.fail = (unsigned long long *)(d_in + (size_t)K * K + K);
        a.mu = d_in + (size_t)K * K + K + 2; a.prop_lambda = self->d_prop; a.diag_only = c->diag_only;
        a.mean_rating = self->mean_rating; a.alpha = alpha; a.iter_plus_1 = (uint32_t)(iter + 1); a.ktrue = c->Kt;
        a.gate_flag = self->a_dflag; a.gate_want = gate_want;
        a.tmo = tmo_word(self->a_h_out_dev, K); a.wait_ticks = wait_ticks();
        a.zero_row = c->d_zero;
    };
    SampleArgs a, b;
    fill(a, A, outA, B->d_items, iterA, d_inA, gate_wantA);            // A gathers from B's CURRENT copy
    fill(b, B, outB, outA, iterB, d_inB, gate_wantB);                 // B gathers from the copy A writes in this launch
    const unsigned grid = (unsigned)((fa.gate_host ? 1 : 0) + fa.nstat + a.nwork + (fb.gate_host ? 1 : 0) + fb.nstat + b.nwork);
    BPMF_LAUNCH(k_sample1p<K>, dim3(grid), dim3(64), st, ev_start, ev_stop, a, fa, b, fb, p);
    return 0;
    }
}

template <int K, bool F32>
int exchange(bpmf_hip_side *self, hipStream_t st, int sub)
{
    bpmf_hip_ctx *c = self->ctx;
    if (!(c->comm != nullptr && !self->bounds.empty())) return 0;
    COMM_ALIVE_OR_FAIL(c, "exchange");
    // factors are fp64 (8 K bytes per column) or, in the fp32 context, fp32
    const bool f32 = c->dtype == BPMF_HIP_F32;
    const ncclDataType_t ty = f32 ? ncclFloat : ncclDouble;
    const size_t esz = f32 ? sizeof(float) : sizeof(double);
    char *items = reinterpret_cast<char *>(self->d_items);
    Rccl *R = rccl();
    if (!self->conn_send_ptr.empty()) {
        if constexpr (F32) return fail(BPMF_HIP_EINVAL, "the connectivity-aware exchange is fp64 only");
        else {
        // connectivity-aware form (c++/assign.cpp:204-241 conn_map + send_item, c++/sample.cpp:370): a column
        // only travels to the ranks whose ratings / test entries reference it.  Pack the columns of
        // every peer's list into one buffer, one grouped send / receive per peer, scatter what arrived.
        if (sub > 0) return 0;                                      // (not cut into parts: everything goes with part 0)
        const int64_t ns = self->conn_send_ptr.back(), nr = self->conn_recv_ptr.back();
        constexpr int P = K / 2;                                   // 16-byte pieces per column
        if (ns > 0)
            hipLaunchKernelGGL(bpmf::k_pack_cols<K>, dim3((unsigned)((ns * P + 255) / 256)), dim3(256), 0, st,
                               (const double *)self->d_items, (const int32_t *)self->d_conn_send, ns, self->d_conn_sbuf);
        NcclGroup group(R);
        NCCL_TRY(group.start());
        for (int r = 0; r < c->nranks; ++r) {
            const int64_t s0 = self->conn_send_ptr[(size_t)r], s1 = self->conn_send_ptr[(size_t)r + 1];
            const int64_t r0 = self->conn_recv_ptr[(size_t)r], r1 = self->conn_recv_ptr[(size_t)r + 1];
            if (s1 > s0) NCCL_TRY(R->Send(self->d_conn_sbuf + (size_t)s0 * K, (size_t)(s1 - s0) * K, ncclDouble, r, c->comm, st));
            if (r1 > r0) NCCL_TRY(R->Recv(self->d_conn_rbuf + (size_t)r0 * K, (size_t)(r1 - r0) * K, ncclDouble, r, c->comm, st));
        }
        NCCL_TRY(group.end());
        if (nr > 0)
            hipLaunchKernelGGL(bpmf::k_unpack_cols<K>, dim3((unsigned)((nr * P + 255) / 256)), dim3(256), 0, st,
                               (const double *)self->d_conn_rbuf, (const int32_t *)self->d_conn_recv, nr, self->d_items);
        HIP_TRY(hipGetLastError());
        return 0;
        }
    }
    // columns [lo, hi) rank r contributes to this call
    auto range = [&](int r, int64_t &lo, int64_t &hi) {
        if (sub < 0 || self->nsub <= 1) { lo = self->bounds[(size_t)r]; hi = self->bounds[(size_t)r + 1]; }
        else { lo = self->sub_bounds[(size_t)r * (self->nsub + 1) + sub]; hi = self->sub_bounds[(size_t)r * (self->nsub + 1) + sub + 1]; }
    };
    // All-gather-v of disjoint, uneven ranges.  Default: a MESH of point-to-point transfers -- one grouped
    // ncclSend / ncclRecv pair per peer, every pair on its own xGMI link (the links are point-to-point:
    // seven per GPU) -- instead of nranks broadcasts, each of which is a ring / tree collective over all ranks.
    // BPMF_HIP_EXCHANGE=bcast keeps the broadcasts (and is what an RCCL without ncclSend / ncclRecv gets).
    static const bool want_mesh = [] { const char *e = getenv("BPMF_HIP_EXCHANGE"); return !(e && std::string(e) == "bcast"); }();
    int64_t mlo, mhi;
    range(c->rank, mlo, mhi);
    NcclGroup group(R);
    NCCL_TRY(group.start());
    if (want_mesh && R->Send && R->Recv) {
        for (int r = 0; r < c->nranks; ++r) {
            if (r == c->rank) continue;
            int64_t lo, hi;
            range(r, lo, hi);
            if (mhi > mlo) NCCL_TRY(R->Send(items + (size_t)mlo * K * esz, (size_t)(mhi - mlo) * K, ty, r, c->comm, st));
            if (hi > lo) NCCL_TRY(R->Recv(items + (size_t)lo * K * esz, (size_t)(hi - lo) * K, ty, r, c->comm, st));
        }
    } else {
        for (int r = 0; r < c->nranks; ++r) {
            int64_t lo, hi;
            range(r, lo, hi);
            if (hi > lo) {
                char *p = items + (size_t)lo * K * esz;
                NCCL_TRY(R->Broadcast(p, p, (size_t)(hi - lo) * K, ty, r, c->comm, st));
            }
        }
    }
    NCCL_TRY(group.end());
    return 0;
}

template <int K, bool F32>
int stats(bpmf_hip_side *self, hipStream_t st, const double *d_in, double *out_host_dev, unsigned *flag, unsigned seq, unsigned *ticket,
          hipEvent_t ev_done)
{
    using namespace bpmf;
    bpmf_hip_ctx *c = self->ctx;
    const unsigned long long *failp = (const unsigned long long *)(d_in + (size_t)K * K + K);
    if constexpr (K == 128) {                           // one wave per (slice of columns, 16 x 16 tile): fp32 or fp64 factors, fp64 sums
        typedef typename std::conditional<F32, float, double>::type T;
        const bool dist = c->comm != nullptr && !self->bounds.empty();
        const bool own = st != c->stream && c->comm2 && self->a_d_red;
        if (dist) COMM_ALIVE_OR_FAIL(c, "statistics all-reduce");
        double *red = own ? self->a_d_red : c->d_red;
        hipLaunchKernelGGL((k_colstats_f32<K, T>), dim3(self->nstat_waves * (K / 16) * (K / 16 + 1) / 2), dim3(64), 0, st, reinterpret_cast<const T *>(self->d_items),
                           self->from, self->to, self->nstat_waves, self->d_stat_partials);
        // single GPU: the sums go straight to the pinned blob; sharded: into a device blob, all-reduced, then published
        if (ev_done && !dist)
            hipExtLaunchKernelGGL(k_colstats_f32_final<K>, dim3((K * K + K + 255) / 256), dim3(256), 0, st, nullptr, ev_done, 0,
                                  (const double *)self->d_stat_partials, self->nstat_waves, failp, out_host_dev, ticket, flag, seq);
        else
        hipLaunchKernelGGL(k_colstats_f32_final<K>, dim3((K * K + K + 255) / 256), dim3(256), 0, st,
                           (const double *)self->d_stat_partials, self->nstat_waves, failp, dist ? red : out_host_dev, ticket,
                           dist ? ticket + 8 : flag, dist ? 0u : seq);
        if (dist) {
            NCCL_TRY(rccl()->AllReduce(red, red, (size_t)K * K + K + 1, ncclDouble, ncclSum, own ? c->comm2 : c->comm, st));
            publish(red, out_host_dev, K * K + K + 1, flag, seq, K * K + K, st);
        }
        return 0;
    } else {
    if (!(c->comm != nullptr && !self->bounds.empty())) {
        if (self->nstat_wg > 0 && self->stat_a_done) {                // group B + the sum over both groups' partials
            self->stat_a_done = false;
            hipLaunchKernelGGL(k_colstats_wg<K>, dim3(self->stat_wgB), dim3(256), 0, st,
                               (const double *)self->d_items + (size_t)self->from * K, self->stat_nA, self->stat_n, self->stat_wgB, self->d_stat_partials,
                               failp, out_host_dev, ticket, flag, seq, tmo_word(out_host_dev, K), wait_ticks(),
                               (const int32_t *)self->d_stat_list, self->stat_wgA, self->stat_wgA + self->stat_wgB, 1);
        } else if (self->nstat_wg > 0) {
            if (ev_done)
                hipExtLaunchKernelGGL(k_colstats_wg<K>, dim3(self->nstat_wg), dim3(256), 0, st, nullptr, ev_done, 0,
                                      (const double *)self->d_items, self->from, self->to, self->nstat_wg, self->d_stat_partials,
                                      failp, out_host_dev, ticket, flag, seq, tmo_word(out_host_dev, K), wait_ticks(),
                                      (const int32_t *)nullptr, 0, self->nstat_wg, 1);
            else
            hipLaunchKernelGGL(k_colstats_wg<K>, dim3(self->nstat_wg), dim3(256), 0, st,
                               (const double *)self->d_items, self->from, self->to, self->nstat_wg, self->d_stat_partials,
                               failp, out_host_dev, ticket, flag, seq, tmo_word(out_host_dev, K), wait_ticks(),
                               (const int32_t *)nullptr, 0, self->nstat_wg, 1);
        } else {
            if (ev_done)
                hipExtLaunchKernelGGL(k_colstats<K>, dim3(self->nstat_waves), dim3(64), 0, st, nullptr, ev_done, 0,
                                      (const double *)self->d_items, self->from, self->to, self->nstat_waves, self->d_stat_partials,
                                      failp, out_host_dev, ticket, flag, seq, tmo_word(out_host_dev, K), wait_ticks());
            else
        hipLaunchKernelGGL(k_colstats<K>, dim3(self->nstat_waves), dim3(64), 0, st,
                           (const double *)self->d_items, self->from, self->to, self->nstat_waves, self->d_stat_partials,
                           failp, out_host_dev, ticket, flag, seq, tmo_word(out_host_dev, K), wait_ticks());
        }
    } else {
        // local sums into a device blob, all-reduce them (cov is then formed once from the GLOBAL
        // sums: SURVEY Q19) together with the failed-column word, publish to the host
        Rccl *R = rccl();
        COMM_ALIVE_OR_FAIL(c, "statistics all-reduce");
        // on the side's own stream: its own reduction blob and the second communicator
        const bool own = st != c->stream && c->comm2 && self->a_d_red;
        double *red = own ? self->a_d_red : c->d_red;
        if (self->nstat_wg > 0)
            hipLaunchKernelGGL(k_colstats_wg<K>, dim3(self->nstat_wg), dim3(256), 0, st,
                               (const double *)self->d_items, self->from, self->to, self->nstat_wg, self->d_stat_partials,
                               failp, red, ticket, ticket + 8, 0u, tmo_word(out_host_dev, K), wait_ticks(),
                               (const int32_t *)nullptr, 0, self->nstat_wg, 1);
        else
        hipLaunchKernelGGL(k_colstats<K>, dim3(self->nstat_waves), dim3(64), 0, st,
                           (const double *)self->d_items, self->from, self->to, self->nstat_waves, self->d_stat_partials,
                           failp, red, ticket, ticket + 8, 0u, tmo_word(out_host_dev, K), wait_ticks());
        NCCL_TRY(R->AllReduce(red, red, (size_t)K * K + K + 1, ncclDouble, ncclSum, own ? c->comm2 : c->comm, st));   // prod | sum | failed-column word
        publish(red, out_host_dev, K * K + K + 1, flag, seq, K * K + K, st);
    }
    return 0;
    }
}

template <int K, bool F32>
int stats_a(bpmf_hip_side *self, hipStream_t st, const double *d_in, double *out_host_dev, unsigned *ticket)
{
    using namespace bpmf;
    if constexpr (K == 128) return 0;
    else {
    if (!(self->nstat_wg > 0 && self->d_stat_list && self->stat_a_ready)) return 0;
    self->stat_a_ready = false;
    const unsigned long long *failp = (const unsigned long long *)(d_in + (size_t)K * K + K);
    hipLaunchKernelGGL(k_colstats_wg<K>, dim3(self->stat_wgA), dim3(256), 0, st,
                       (const double *)self->d_items + (size_t)self->from * K, (int64_t)0, self->stat_nA, self->stat_wgA, self->d_stat_partials,
                       failp, out_host_dev, ticket, (unsigned *)nullptr, 0u, tmo_word(out_host_dev, K), wait_ticks(),
                       (const int32_t *)self->d_stat_list, 0, self->stat_wgA + self->stat_wgB, 0);
    self->stat_a_done = true;
    return 0;
    }
}

// k_predict on stream `ps` over explicit factor pointers.  in_order: on the main stream behind the
// samplers.  Otherwise (`beside`): behind ev_in (everything that was on the main stream when the
// evaluation was requested), with ev_done recorded after it for launch_sampler's hazard check.
template <int K, bool F32>
void predict(bpmf_hip_test *t, const bpmf_hip_side *self, const void *self_items, const void *other_items, int n,
                    hipStream_t ps, bool beside)
{
    bpmf_hip_ctx *c = self->ctx;
    unsigned *flag = reinterpret_cast<unsigned *>(t->h_res_dev + 2);
    const bool dist = c->comm && !self->bounds.empty();
    t->pstream = ps;
    if (beside) (void)hipStreamWaitEvent(ps, t->in_ev, 0);
    // users.predict(movies) (c++/bpmf.cpp:190): the twin's entries with the roles of the two factor matrices swapped, on
    // the same stream and AHEAD of this evaluation, so that the completion event below covers both
    const bool fused_twin = t->twin && t->d_twin_perm && !dist;
    if (t->twin && !fused_twin && (t->twin->nnz > 0 || dist)) {      // (sharded: its all-reduce is collective, entries or not)
        t->twin->in_ev = t->in_ev;
        predict<K, F32>(t->twin, t->twin->side, other_items, self_items, n, ps, false);
        t->twin->launched = true;
    }
    // se | se_avg of this rank's test ratings: straight to the host, or -> all-reduce -> host
    double *red = c->d_red + c->out_words + (t->owner ? 2 : 0);      // 2 spare words behind the sampler's blob (the twin: the next 2)
    if constexpr (F32) {
        if (fused_twin) {
            // round 4: the twin inside the same kernel here too (k_predict<K, 256, float>) -- as two kernels the second one
            // started when the partner's sampler had filled the chip and ended with it (347 us for 20 us of work), and the
            // host loop, which collects both sums before it enqueues the next iteration, came 45 us late every iteration
            bpmf::TwinArgs tw{};
            bpmf_hip_test *u = t->twin;
            tw.perm = t->d_twin_perm; tw.pavg = u->d_pavg; tw.pm2 = u->d_pm2; tw.mean = u->side->mean_rating;
            tw.partial = u->d_partial; tw.out = u->h_res_dev; tw.flag = reinterpret_cast<unsigned *>(u->h_res_dev + 2); tw.seq = ++u->seq;
            u->pstream = ps; u->launched = true;
            hipLaunchKernelGGL((bpmf::k_predict<K, 256, float>), dim3((unsigned)t->nblocks), dim3(256), 0, ps,
                               (const int32_t *)t->d_tcol, (const int32_t *)t->d_trow, (const double *)t->d_tval, t->nnz,
                               reinterpret_cast<const float *>(self_items), reinterpret_cast<const float *>(other_items), self->from,
                               self->mean_rating, n, t->d_pavg, t->d_pm2, t->d_partial, t->h_res_dev, t->d_ticket, flag, ++t->seq, tw);
        } else
        hipLaunchKernelGGL(bpmf::k_predict_f32<K>, dim3((unsigned)t->nblocks), dim3(256), 0, ps,
                           (const int32_t *)t->d_tcol, (const int32_t *)t->d_trow, (const double *)t->d_tval, t->nnz,
                           reinterpret_cast<const float *>(self_items), reinterpret_cast<const float *>(other_items), self->from,
                           self->mean_rating, n, t->d_pavg, t->d_pm2, t->d_partial, dist ? red : t->h_res_dev, t->d_ticket,
                           dist ? t->d_ticket + 8 : flag, dist ? 0u : ++t->seq);
        if (dist) {
            if (rccl()->AllReduce(red, red, 2, ncclDouble, ncclSum, c->comm, c->stream) != ncclSuccess) return;
            publish(red, t->h_res_dev, 2, flag, ++t->seq, -1, c->stream);
        }
    } else {
    bpmf::TwinArgs tw{};
    if (fused_twin) {                                                 // one kernel, both copies of the test entries
        bpmf_hip_test *u = t->twin;
        tw.perm = t->d_twin_perm; tw.pavg = u->d_pavg; tw.pm2 = u->d_pm2; tw.mean = u->side->mean_rating;
        tw.partial = u->d_partial; tw.out = u->h_res_dev; tw.flag = reinterpret_cast<unsigned *>(u->h_res_dev + 2); tw.seq = ++u->seq;
        u->pstream = ps; u->launched = true;
    }
    const unsigned pseq = dist ? 0u : ++t->seq;
    if (t->wg == 64)                                                  // single-wave workgroups (small test sets: see k_predict)
        hipLaunchKernelGGL((bpmf::k_predict<K, 64>), dim3((unsigned)t->nblocks), dim3(64), 0, ps,
                           (const int32_t *)t->d_tcol, (const int32_t *)t->d_trow, (const double *)t->d_tval, t->nnz,
                           (const double *)self_items, (const double *)other_items, self->from, self->mean_rating, n,
                           t->d_pavg, t->d_pm2, t->d_partial, dist ? red : t->h_res_dev, t->d_ticket,
                           dist ? t->d_ticket + 8 : flag, pseq, tw);
    else
    hipLaunchKernelGGL((bpmf::k_predict<K, 256>), dim3((unsigned)t->nblocks), dim3(256), 0, ps,
                       (const int32_t *)t->d_tcol, (const int32_t *)t->d_trow, (const double *)t->d_tval, t->nnz,
                       (const double *)self_items, (const double *)other_items, self->from, self->mean_rating, n,
                       t->d_pavg, t->d_pm2, t->d_partial, dist ? red : t->h_res_dev, t->d_ticket,
                       dist ? t->d_ticket + 8 : flag, pseq, tw);
    if (dist) {
        if (rccl()->AllReduce(red, red, 2, ncclDouble, ncclSum, c->comm, c->stream) != ncclSuccess) return;
        publish(red, t->h_res_dev, 2, flag, ++t->seq, -1, c->stream);
    }
    }
    if (beside) (void)hipEventRecord(t->ev_done[t->seq & 1u], ps);
}

}  // namespace bpmf_launch

#define BPMF_INSTANTIATE_K(KK, FF)                                                                                               \
    template int bpmf_launch::sampler_into<KK, FF>(bpmf_hip_side *, double *, const bpmf_hip_side *, int, double, double *, hipStream_t, \
                                               hipEvent_t, hipEvent_t);                                                          \
    template int bpmf_launch::exchange<KK, FF>(bpmf_hip_side *, hipStream_t, int);                                                        \
    template int bpmf_launch::stats<KK, FF>(bpmf_hip_side *, hipStream_t, const double *, double *, unsigned *, unsigned, unsigned *, hipEvent_t); \
    template int bpmf_launch::stats_a<KK, FF>(bpmf_hip_side *, hipStream_t, const double *, double *, unsigned *);                       \
    template void bpmf_launch::predict<KK, FF>(bpmf_hip_test *, const bpmf_hip_side *, const void *, const void *, int, hipStream_t, bool);   \
    template int bpmf_launch::sampler_pair<KK, FF>(bpmf_hip_side *, double *, int, double *, const bpmf::FusedArgs &, bpmf_hip_side *, double *, int, double *, \
                                                   const bpmf::FusedArgs &, unsigned, unsigned, double, const bpmf::PairArgs &, hipStream_t, hipEvent_t, hipEvent_t);
