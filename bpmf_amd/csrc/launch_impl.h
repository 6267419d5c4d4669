// launch_impl.h -- definitions of launch.h's templates: which kernel of kernels*.h runs for which
// form of a side, and with what arguments.  Included by the per-K units only.
#pragma once
#include "launch.h"
#include "kernels.h"
#include "kernels_f32.h"
#include "kernels_q4.h"
#include "kernels_slab.h"

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
    a.ablate = c->ablate; a.stamps = c->d_stamps;
    a.gate_flag = self->cur_gate_flag; a.gate_want = self->cur_gate_want;
    a.tmo = self->cur_gate_flag ? tmo_word(self->a_h_out_dev, K) : nullptr; a.wait_ticks = wait_ticks();
    a.zero_row = c->d_zero;
    a.lf32 = (lf32_words(c) && !self->d_prop) ? reinterpret_cast<const float *>(d_in + c->in_words) : nullptr;
    const bpmf::StatRiders &rr = self->cur_riders;                   // (K = 128 fp32 only)
    if constexpr (F32) {                                             // fp32 factors (items / other_items are float arrays): kernels_wg2.h, T = float
        if (nwork > 0) k128_wg2(nwork, st, ev_start, ev_stop, a, rr);
        return 0;
    } else if constexpr (K == 128) {                                 // fp64 factors, workgroup of four waves per item (kernels_wg2.h, T = double)
        if (nwork > 0) k128_wg2_f64(nwork, st, ev_start, ev_stop, a, rr);
        return 0;
    } else {
    if constexpr (K <= 32) {
        if (nwork > 0 && self->mode == 3) {                          // four columns per wave (k_sample4)
            BPMF_LAUNCH(k_sample4<K>, dim3((nwork + 3) / 4), dim3(64), st, ev_start, ev_stop, a);
            return 0;
        }
    }
    if constexpr (K == 64) {
        if (self->lr_n > 0 && !self->d_prop && !c->diag_only && !(c->ablate & 3u)) {
            // light columns (<= 16 ratings): product form over the shared factor of LambdaF (k_sample_pf); the others in the slab form
            if (self->hv_nwork > 0) {
                a.wi_col = self->d_hv_col; a.wi_p0 = self->d_hv_p0; a.wi_len = self->d_hv_len; a.wi_mc = self->d_hv_mc;
                a.wi_chunk = self->d_hv_chunk; a.nwork = self->hv_nwork;
                k64_slab(self->hv_nwork, st, ev_start, nullptr, a);
            }
            LrArgs l;
            l.rowidx = self->d_rowidx; l.vals = self->d_vals; l.col = self->d_lr_col; l.p0 = self->d_lr_p0; l.len = self->d_lr_len;
            l.nitems = self->lr_n; l.other_items = other->d_items; l.items = out_items; l.col_from = self->from;
            l.R0 = d_in + (size_t)K * K + K + 2 + K; l.S0t = l.R0 + (size_t)K * K; l.y0 = l.S0t + (size_t)K * K;
            l.Lmu = a.Lmu; l.fail = a.fail;
            l.mean_rating = self->mean_rating; l.alpha = alpha; l.sqrt_alpha = std::sqrt(alpha); l.iter_plus_1 = (uint32_t)(iter + 1); l.ktrue = c->Kt;
            // three instantiations (<= 3, <= 6, <= 16 ratings), persistent workgroups of eight waves with R0^-1 in LDS
            // (the events ride on the first / last launch of the side)
            l.Q = self->d_pf_q;
            bool started = self->hv_nwork > 0;
            if (self->pf_class[3] > self->pf_class[0] && self->d_pf_q) {
                // Q = U_other R0^-1 once per half-iteration (k_pf_prepare), ahead of the product-form launches
                const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((self->nrows + 7) / 8, (int64_t)c->num_cu * 4));
                k64_pf_prepare(grid, st, started ? nullptr : ev_start, l.S0t, other->d_items, self->nrows, self->d_pf_q);
                started = true;
            }
            int last_pf = -1;
            for (int pc = 0; pc < 3; ++pc) if (self->pf_class[pc + 1] > self->pf_class[pc]) last_pf = pc;
            for (int pc = 0; pc < 3; ++pc) {
                const int n0 = self->pf_class[pc], n1 = self->pf_class[pc + 1];
                if (n1 <= n0) continue;
                LrArgs lc = l;
                lc.col = l.col + n0; lc.p0 = l.p0 + n0; lc.len = l.len + n0; lc.nitems = n1 - n0;
                hipEvent_t e0 = started ? nullptr : ev_start, e1 = pc == last_pf ? ev_stop : nullptr;
                started = true;
                const int grid = std::max(1, std::min((n1 - n0 + 31) / 32, c->num_cu * 4));     // eight waves x four columns per pass
                k64_pf(pc, grid, st, e0, e1, lc);
            }
            return 0;
        }
        if (nwork > 0) {
            const FusedArgs &f = self->cur_fused;                    // (all zero outside the fused stateful path)
            if (f.gate_host || f.nstat) {                            // gate workgroup + statistics riders + items in one launch
                const dim3 grid((unsigned)(nwork + (f.gate_host ? 1 : 0) + f.nstat));
                BPMF_LAUNCH(k_sample1s<K>, grid, dim3(64), st, ev_start, ev_stop, a, f);
            } else {
                k64_slab(nwork, st, ev_start, ev_stop, a);
            }
        }
        return 0;
    } else {
    if (nwork > 0) {                                                 // K <= 32: one work item per single-wave workgroup (k_sample1)
        const FusedArgs &f = self->cur_fused;                        // (all zero outside the fused stateful path)
        const dim3 grid((unsigned)(nwork + (f.gate_host ? 1 : 0) + f.nstat));
        BPMF_LAUNCH(k_sample1<K>, grid, dim3(64), st, ev_start, ev_stop, a, f);
    }
    return 0;
    }
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
        // Not cut into parts: the lists name columns of the whole range, so everything travels behind the LAST part -- the
        // exchange stream has then waited for every part's sampler.  (Rounds 3-5 sent it with part 0, i.e. while the parts
        // 1 .. n - 1 were still being sampled: stale columns to the peers and, at one rank, the scatter racing the samplers.
        // The combination parts + lists had no test; tests/_rccl1_worker.py is it, round 6.)
        if (sub >= 0 && self->nsub > 1 && sub != self->nsub - 1) return 0;
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
        if (self->nstat_wg > 0) {
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
    template void bpmf_launch::predict<KK, FF>(bpmf_hip_test *, const bpmf_hip_side *, const void *, const void *, int, hipStream_t, bool);
