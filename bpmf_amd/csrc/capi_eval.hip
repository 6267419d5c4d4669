// capi_eval.hip -- test sets and Sys::predict (c++/sample.cpp:48-96): launch, twin evaluation, collection
// (one of the translation units of the C ABI of include/bpmf_hip.h: see capi_internal.h for the map)
#include "capi_internal.h"

namespace bpmf_capi {

// ---------------------------------------------------------------------------
extern "C" int bpmf_hip_test_create(bpmf_hip_side *side, const int64_t *tcolptr, const int32_t *trowidx,
                                    const double *tvals, bpmf_hip_test **out)
{
    if (!out) return fail(BPMF_HIP_EINVAL, "test_create: out is NULL");
    *out = nullptr;
    if (!side || !tcolptr) return fail(BPMF_HIP_EINVAL, "test_create: NULL argument");
    const int64_t nloc = side->to - side->from;
    if (tcolptr[0] != 0) return fail(BPMF_HIP_EINVAL, "test_create: tcolptr[0] must be 0");
    const int64_t nnz = tcolptr[nloc];
    if (nnz > 0 && (!trowidx || !tvals)) return fail(BPMF_HIP_EINVAL, "test_create: NULL rowidx/vals");
    HIP_TRY(hipSetDevice(side->ctx->device));
    std::vector<int32_t> tcol((size_t)std::max<int64_t>(nnz, 1));
    for (int64_t c = 0; c < nloc; ++c) {
        if (tcolptr[c + 1] < tcolptr[c]) return fail(BPMF_HIP_EINVAL, "test_create: tcolptr is not monotone");
        for (int64_t p = tcolptr[c]; p < tcolptr[c + 1]; ++p) {
            if (trowidx[p] < 0 || trowidx[p] >= side->nrows) return fail(BPMF_HIP_EINVAL, "test_create: row index out of range");
            tcol[p] = (int32_t)c;
        }
    }
    bpmf_hip_test *t = new (std::nothrow) bpmf_hip_test();
    if (!t) return fail(BPMF_HIP_ENOMEM, "test_create: out of host memory");
    t->side = side; t->nnz = nnz;
    t->h_col.resize((size_t)nnz); t->h_row.assign(trowidx, trowidx + nnz);
    for (int64_t p = 0; p < nnz; ++p) t->h_col[(size_t)p] = (int32_t)(side->from + tcol[(size_t)p]);
    if (hipHostMalloc((void **)&t->h_res, 4 * sizeof(double), hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer((void **)&t->h_res_dev, t->h_res, 0) != hipSuccess) {
        delete t;
        return fail(BPMF_HIP_ENOMEM, "test_create: pinned result allocation failed");
    }
    memset(t->h_res, 0, 4 * sizeof(double));
    if (hipMalloc((void **)&t->d_ticket, 64) != hipSuccess || hipMemset(t->d_ticket, 0, 64) != hipSuccess) {
        (void)hipHostFree(t->h_res);
        delete t;
        return fail(BPMF_HIP_ENOMEM, "test_create: device allocation failed");
    }
    // one lane per test rating, four-wave workgroups; BPMF_HIP_PREDICT_WG=64: single-wave workgroups (up to 4 M ratings, fp64
    // contexts), which find wave slots beside a sampler launch that refills every slot with single-wave workgroups
    // (MEASURED, ML-1M shape: 64-thread workgroups make the evaluation compete with the sampler's items for every slot -- the
    // movies' launch 44.5 -> 53.6 us, the iteration 0.097 -> 0.107 ms; four-wave workgroups wait for the boundary between two
    // launches, where the chip drains anyway.  256 stays the default.)
    t->wg = (nnz <= ((int64_t)4 << 20) && side->ctx->dtype == BPMF_HIP_F64 && env_int("BPMF_HIP_PREDICT_WG", 256) == 64) ? 64 : 256;
    t->nblocks = std::max<int64_t>(1, (nnz + t->wg - 1) / t->wg);
    const int64_t nw = t->nblocks;
    int rc;
    if ((rc = dev_upload(&t->d_tcol, tcol.data(), (size_t)nnz)) || (rc = dev_upload(&t->d_trow, trowidx, (size_t)nnz)) ||
        (rc = dev_upload(&t->d_tval, tvals, (size_t)nnz)) || (rc = dev_upload(&t->d_pavg, tvals, (size_t)nnz)) ||
        (rc = dev_upload(&t->d_pm2, tvals, (size_t)nnz)) || (rc = dev_upload<double>(&t->d_partial, nullptr, (size_t)nw * 2))) {
        bpmf_hip_test_destroy(t);
        return rc;
    }
    // events for an evaluation that runs beside the samplers of the next iteration (launch_predict)
    if (env_int("BPMF_HIP_DBUF", 1) != 0) {
        bool ok = hipEventCreateWithFlags(&t->ev_in, hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess;
        for (auto &e : t->ev_done) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            if (t->ev_in) { (void)hipEventDestroy(t->ev_in); t->ev_in = nullptr; }
        }
    }
    *out = t;
    return BPMF_HIP_OK;
}

// the stream an evaluation was enqueued on, if it still exists (it belongs to a side)
static hipStream_t live_pstream(bpmf_hip_test *t)
{
    bpmf_hip_ctx *c = t->side->ctx;
    if (!t->pstream || t->pstream == c->stream) return c->stream;
    std::lock_guard<std::mutex> lk(c->launch_mutex);
    for (bpmf_hip_side *sd : c->sides) if (sd->saux == t->pstream) return t->pstream;
    return c->stream;
}

extern "C" int bpmf_hip_test_destroy(bpmf_hip_test *t)
{
    if (!t) return BPMF_HIP_OK;
    bpmf_hip_ctx *c = t->side->ctx;
    (void)hipSetDevice(c->device);
    if (t->owner) {                                                   // a twin: its owner's evaluation in flight reads its arrays
        flush_deferred(t->owner);
        (void)bounded_stream_sync(t->owner->side->ctx, live_pstream(t->owner), __func__);
        if (t->owner->d_twin_perm) { (void)hipFree(t->owner->d_twin_perm); t->owner->d_twin_perm = nullptr; }
        t->owner->twin = nullptr; t->owner = nullptr;
    }
    if (t->twin) { flush_deferred(t); t->twin->owner = nullptr; t->twin->launched = false; t->twin = nullptr; }
    if (t->deferred) {                                              // never enqueued: nothing to wait for
        t->deferred = false;
        std::lock_guard<std::mutex> lk(c->launch_mutex);
        for (bpmf_hip_side *sd : c->sides) if (sd->deferred_eval == t) sd->deferred_eval = nullptr;
    }
    (void)bounded_stream_sync(c, c->stream, __func__);
    (void)bounded_stream_sync(t->side->ctx, live_pstream(t), __func__);
    {   // no side may wait for this evaluation any more
        std::lock_guard<std::mutex> lk(c->launch_mutex);
        for (bpmf_hip_side *sd : c->sides)
            for (auto &rd : sd->readers) if (rd.t == t) rd.t = nullptr;
        for (auto &rd : t->side->readers) if (rd.t == t) rd.t = nullptr;
    }
    if (t->ev_in) (void)hipEventDestroy(t->ev_in);
    for (hipEvent_t e : t->ev_done) if (e) (void)hipEventDestroy(e);
    void *ptrs[] = {t->d_tcol, t->d_trow, t->d_tval, t->d_pavg, t->d_pm2, t->d_partial, t->d_ticket, t->d_twin_perm};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (t->h_res) (void)hipHostFree(t->h_res);
    delete t;
    return BPMF_HIP_OK;
}
void dispatch_predict(bpmf_hip_test *t, const bpmf_hip_side *self, const void *self_items, const void *other_items, int n,
                      hipStream_t ps, bool beside)
{
    switch (self->ctx->K) {
    case 8: bpmf_launch::predict<8, false>(t, self, self_items, other_items, n, ps, beside); break;
    case 16: bpmf_launch::predict<16, false>(t, self, self_items, other_items, n, ps, beside); break;
    case 32: bpmf_launch::predict<32, false>(t, self, self_items, other_items, n, ps, beside); break;
    case 64: bpmf_launch::predict<64, false>(t, self, self_items, other_items, n, ps, beside); break;
    case 128:
        if (self->ctx->dtype == BPMF_HIP_F32) bpmf_launch::predict<128, true>(t, self, self_items, other_items, n, ps, beside);
        else bpmf_launch::predict<128, false>(t, self, self_items, other_items, n, ps, beside);
        break;
    default: break;
    }
}

// An evaluation that was requested while both sides keep two copies of their factors is enqueued
// LATER: on the other side's statistics stream, right behind the gate kernel of that side's next
// half-iteration (bpmf_hip_sys_sample), so that it runs beside the samplers that follow instead of
// between them -- they write the other copies.  (Enqueued at once it would sit in front of that
// gate kernel and hold up the sampler behind it; a stream of its own shares a hardware queue with
// one of the other four, and behind a gate kernel that polls the host everything on that queue
// stalls: 0.13 -> 0.34 ms per iteration.)  Whoever needs it earlier flushes it: predict_finish,
// a sampler about to overwrite a copy it reads, test_get, the destructors.
void flush_deferred(bpmf_hip_test *t, bool on_main)
{
    if (t && t->owner) t = t->owner;                                 // a twin is enqueued with the evaluation it belongs to
    if (!t || !t->deferred) return;
    t->deferred = false;
    bpmf_hip_side *o = t->def_other;
    if (o && o->deferred_eval == t) o->deferred_eval = nullptr;
    (void)hipSetDevice(t->side->ctx->device);
    // (on_main: the end of a run -- behind the last sampler on its own stream, no cross-queue hop)
    if (on_main) t->side->ctx->last_sampler_done = nullptr;
    dispatch_predict(t, t->side, t->def_self_items, t->def_other_items, t->def_n, on_main ? t->side->ctx->stream : o->saux, true);
    trace("predict: enqueued", t->side, t->def_n);
}

// users.predict(movies) (c++/bpmf.cpp:190, inside the reference's timed region): `twin` holds the test entries by column of
// the OTHER side (the transpose of `t`'s) and is evaluated with the roles swapped -- pred = users.col(c) . movies.col(r)
// + mean of that side, its own Pavg / Pm2 copies and sums, as the reference's second Sys keeps them -- whenever `t` is.
// Its sums are collected with bpmf_hip_predict_finish(twin, ...).  twin = NULL detaches.
extern "C" int bpmf_hip_test_set_twin(bpmf_hip_test *t, bpmf_hip_test *twin)
{
    if (!t) return fail(BPMF_HIP_EINVAL, "test_set_twin: NULL");
    if (t->launched || (twin && twin->launched)) return fail(BPMF_HIP_EINVAL, "test_set_twin: an evaluation is in flight");
    if (twin && (twin == t || twin->side == t->side || twin->side->ctx != t->side->ctx || twin->side->ncols != t->side->nrows || twin->owner))
        return fail(BPMF_HIP_EINVAL, "test_set_twin: the twin must sit on the other side of the same pair");
    HIP_TRY(hipSetDevice(t->side->ctx->device));
    if (t->twin) t->twin->owner = nullptr;
    if (t->d_twin_perm) { (void)hipFree(t->d_twin_perm); t->d_twin_perm = nullptr; }
    t->twin = twin;
    if (!twin) return BPMF_HIP_OK;
    twin->owner = t;
    // Same entries, transposed?  Then one kernel serves both copies: entry q of `t` is entry perm[q] of the twin.
    // (Otherwise -- shards of different column ranges -- the twin keeps a kernel of its own.)
    const bool whole = t->side->to - t->side->from == t->side->ncols && twin->side->to - twin->side->from == twin->side->ncols;
    if (whole && t->nnz == twin->nnz && t->nnz > 0 && t->nnz < ((int64_t)1 << 31)) {
        const int64_t n = t->nnz, ncm = t->side->ncols;
        std::vector<int32_t> perm((size_t)n);
        // Both copies normally arrive in CSC order with ascending rows inside a column: t's entries of user u then meet, in t's
        // own order, the entries of the twin's column u from first to last -- one cursor per user, O(n) (the two index sorts
        // below took 5.6 ms for the 100 000 test entries of the ML-1M shape, most of the hand-over of that workload).  Any
        // mismatch falls through to the sorts.
        bool ok = true;
        {
            const int64_t nu = twin->side->ncols;
            std::vector<int32_t> cur((size_t)nu + 1, 0);
            for (int64_t p = 0; p < n && ok; ++p) {
                const int64_t u = twin->h_col[(size_t)p];
                ok = u >= 0 && u < nu && (p == 0 || twin->h_col[(size_t)p - 1] <= u);
                if (ok) cur[(size_t)u + 1]++;
            }
            for (int64_t u = 0; u < nu && ok; ++u) cur[(size_t)u + 1] += cur[(size_t)u];     // first entry of column u
            for (int64_t q = 0; q < n && ok; ++q) {
                const int64_t u = t->h_row[(size_t)q];
                ok = u >= 0 && u < nu;
                if (!ok) break;
                const int32_t p = cur[(size_t)u]++;
                ok = p < n && twin->h_col[(size_t)p] == u && twin->h_row[(size_t)p] == t->h_col[(size_t)q] &&
                     (p == 0 || twin->h_col[(size_t)p - 1] != u || twin->h_row[(size_t)p - 1] < twin->h_row[(size_t)p]);     // (no entry twice)
                perm[(size_t)q] = p;
            }
        }
        if (!ok) {
        ok = true;
        std::vector<int64_t> ka((size_t)n), kb((size_t)n);
        std::vector<int32_t> ia((size_t)n), ib((size_t)n);
        for (int64_t q = 0; q < n; ++q) {
            ka[(size_t)q] = (int64_t)t->h_row[(size_t)q] * ncm + t->h_col[(size_t)q];           // (user, movie) of t's entry
            kb[(size_t)q] = (int64_t)twin->h_col[(size_t)q] * ncm + twin->h_row[(size_t)q];     // ... of the twin's
            ia[(size_t)q] = ib[(size_t)q] = (int32_t)q;
        }
        std::sort(ia.begin(), ia.end(), [&](int32_t x, int32_t y) { return ka[(size_t)x] < ka[(size_t)y]; });
        std::sort(ib.begin(), ib.end(), [&](int32_t x, int32_t y) { return kb[(size_t)x] < kb[(size_t)y]; });
        for (int64_t q = 0; q < n && ok; ++q) {
            ok = ka[(size_t)ia[(size_t)q]] == kb[(size_t)ib[(size_t)q]] && (q == 0 || ka[(size_t)ia[(size_t)q]] != ka[(size_t)ia[(size_t)q - 1]]);
            perm[(size_t)ia[(size_t)q]] = ib[(size_t)q];
        }
        }
        if (ok) {
            int rc;
            if ((rc = dev_upload(&t->d_twin_perm, perm.data(), perm.size()))) return rc;
            // the fused kernel writes the twin's block partials with the OWNER's grid
            if (twin->nblocks < t->nblocks) {
                if (twin->d_partial) (void)hipFree(twin->d_partial);
                if ((rc = dev_upload<double>(&twin->d_partial, nullptr, (size_t)t->nblocks * 2))) return rc;
            }
        }
    }
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_predict_launch(bpmf_hip_test *t, const bpmf_hip_side *self, const bpmf_hip_side *other_c, int n)
{
    if (!t || !self || !other_c) return fail(BPMF_HIP_EINVAL, "predict: NULL argument");
    if (t->side != self) return fail(BPMF_HIP_EINVAL, "predict: test matrix belongs to another side");
    if (n < 0) return fail(BPMF_HIP_EINVAL, "predict: n < 0");
    if (t->owner) return fail(BPMF_HIP_EINVAL, "predict_launch: this test matrix is a twin (it is evaluated with its owner)");
    if (t->launched) return fail(BPMF_HIP_EINVAL, "predict_launch: previous launch not finished");
    if (t->twin && t->twin->launched) return fail(BPMF_HIP_EINVAL, "predict_launch: the twin's previous sums were not collected (bpmf_hip_predict_finish)");
    bpmf_hip_ctx *c = self->ctx;
    bpmf_hip_side *other = const_cast<bpmf_hip_side *>(other_c);
    HIP_TRY(hipSetDevice(c->device));
    const bool dist = c->comm && !self->bounds.empty();
    if (dist) COMM_ALIVE_OR_FAIL(c, "predict_launch");               // (its sums are all-reduced)
    if (t->twin && t->twin->nnz == 0 && !dist) t->twin->launched = true;     // (nothing to enqueue for it)
    if (t->nnz == 0 && !dist) {
        t->launched = true;
        if (t->twin && t->twin->nnz > 0) return fail(BPMF_HIP_EINVAL, "predict_launch: empty test matrix with a non-empty twin");
        return BPMF_HIP_OK;
    }
    if (c->K != 8 && c->K != 16 && c->K != 32 && c->K != 64 && c->K != 128) return fail(BPMF_HIP_EINVAL, "predict: unsupported K");
    // (the all-reduce of the sharded form shares the main communicator: that form stays in order)
    const bool beside = t->ev_in && !dist && other->saux && !other->deferred_eval && second_copy_usable(self) && second_copy_usable(other);
    if (beside) {
        // the samplers this evaluation is about: the stop event of the newest one if nothing else
        // went to the main stream since, else a marker (a packet between two samplers)
        if (c->last_sampler_done) t->in_ev = c->last_sampler_done;
        else { HIP_TRY(hipEventRecord(t->ev_in, c->stream)); t->in_ev = t->ev_in; }
        t->deferred = true; t->def_n = n; t->def_other = other;
        t->def_self_items = self->d_items; t->def_other_items = other->d_items;
        other->deferred_eval = t;
        bpmf_hip_side *sm = t->side;
        sm->readers[sm->cur_buf].t = t; sm->readers[sm->cur_buf].seq = t->seq + 1;
        other->readers[other->cur_buf].t = t; other->readers[other->cur_buf].seq = t->seq + 1;
        t->pstream = other->saux;
        t->launched = true;
        if (t->twin) t->twin->launched = true;                        // (enqueued with this one: flush_deferred)
        trace("predict: deferred", self, n);
        return BPMF_HIP_OK;
    }
    c->last_sampler_done = nullptr;
    dispatch_predict(t, self, self->d_items, other->d_items, n, c->stream, false);
    HIP_TRY(hipGetLastError());
    t->launched = true;
    trace("predict: enqueued", self, n);
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_predict_finish(bpmf_hip_test *t, double *se, double *se_avg, int64_t *count)
{
    if (!t || !se || !se_avg || !count) return fail(BPMF_HIP_EINVAL, "predict_finish: NULL argument");
    if (!t->launched) return fail(BPMF_HIP_EINVAL, "predict_finish: nothing launched");
    // Still not enqueued?  Then no sampler launch has come since it was requested: the end of a run of iterations.  The
    // statistics of the newest half-iteration have no launch to ride in either: they start now, beside the evaluation,
    // instead of when somebody finally asks for the side's state (a 20-step block of bench.py ended ~20 us later).
    const bool tail = (t->owner ? t->owner : t)->deferred;
    if (tail && t->side->ctx->pending_stats) (void)flush_pending_stats(t->side->ctx, true);   // (first: its host chain is the longer one)
    flush_deferred(t, tail);
    t->launched = false;
    if (t->owner && t->owner->cancelled) return fail(BPMF_HIP_EINVAL, "predict_finish: the evaluation this twin belongs to was cancelled");
    if (t->cancelled) {
        t->cancelled = false;
        return fail(BPMF_HIP_EINVAL, "predict_finish: the side of this test matrix was destroyed before the evaluation ran");
    }
    bpmf_hip_side *self = t->side;
    bpmf_hip_ctx *c = self->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const bool dist = c->comm && !self->bounds.empty();
    if (t->nnz == 0 && !dist) { *se = 0.0; *se_avg = 0.0; *count = 0; return BPMF_HIP_OK; }
    {   // spin on the sequence number published behind the two sums
        unsigned *flag = reinterpret_cast<unsigned *>(t->h_res + 2);
        const auto t0 = std::chrono::steady_clock::now();
        bool seen = false;
        for (unsigned spins = 0; !seen; ++spins) {
            seen = __atomic_load_n(flag, __ATOMIC_ACQUIRE) == t->seq;
            if (seen || spin_limit_s() <= 0.0) break;
            __builtin_ia32_pause();
            if ((spins & 0xFFFu) == 0xFFFu && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > spin_limit_s()) break;
        }
        if (!seen) {
            { const int rs_ = bounded_stream_sync(t->side->ctx, live_pstream(t), __func__); if (rs_) return rs_; }
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != t->seq) return fail(BPMF_HIP_ENODEV, "device did not publish its results");
        }
    }
    t->done_seq = t->seq;                                           // every block has read its factors
    trace("predict: sums landed", self, 0);
    *se = t->h_res[0];
    *se_avg = t->h_res[1];
    *count = t->nnz;
    if (dist) {
        if (t->global_nnz < 0) {                                   // once: number of test ratings over all ranks
            COMM_ALIVE_OR_FAIL(c, "predict_finish");
            long long v = (long long)t->nnz, *d = reinterpret_cast<long long *>(c->d_red + c->out_words + 4);
            HIP_TRY(hipMemcpyAsync(d, &v, sizeof v, hipMemcpyHostToDevice, c->stream));
            NCCL_TRY(rccl()->AllReduce(d, d, 1, ncclInt64, ncclSum, c->comm, c->stream));
            HIP_TRY(hipMemcpyAsync(&v, d, sizeof v, hipMemcpyDeviceToHost, c->stream));
            { const int rs_ = bounded_stream_sync(c, c->stream, __func__); if (rs_) return rs_; }
            t->global_nnz = v;
        }
        *count = t->global_nnz;
    }
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_predict(bpmf_hip_test *t, const bpmf_hip_side *self, const bpmf_hip_side *other, int n,
                                double *se, double *se_avg, int64_t *count)
{
    if (!se || !se_avg || !count) return fail(BPMF_HIP_EINVAL, "predict: NULL argument");
    const int rc = bpmf_hip_predict_launch(t, self, other, n);
    if (rc) return rc;
    return bpmf_hip_predict_finish(t, se, se_avg, count);
}

extern "C" int bpmf_hip_test_get(bpmf_hip_test *t, double *pavg, double *pm2)
{
    if (!t) return fail(BPMF_HIP_EINVAL, "test_get: NULL");
    HIP_TRY(hipSetDevice(t->side->ctx->device));
    { const int rs_ = bounded_stream_sync(t->side->ctx, t->side->ctx->stream, __func__); if (rs_) return rs_; }
    flush_deferred(t);
    { const int rs_ = bounded_stream_sync(t->side->ctx, live_pstream(t), __func__); if (rs_) return rs_; }
    if (pavg) HIP_TRY(hipMemcpy(pavg, t->d_pavg, (size_t)t->nnz * sizeof(double), hipMemcpyDeviceToHost));
    if (pm2) HIP_TRY(hipMemcpy(pm2, t->d_pm2, (size_t)t->nnz * sizeof(double), hipMemcpyDeviceToHost));
    return BPMF_HIP_OK;
}


}  // namespace bpmf_capi
