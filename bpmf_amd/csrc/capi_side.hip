// capi_side.hip -- static work schedule of a side, side create / destroy, factor storage, priors, launch reports
// (one of the translation units of the C ABI of include/bpmf_hip.h: see capi_internal.h for the map)
#include "capi_internal.h"

namespace bpmf_capi {

int build_schedule(bpmf_hip_side *s, const int64_t *colptr)
{
    const int64_t nloc = s->to - s->from;
    const int K = s->ctx->K;
    // Form of the sampler (s->mode).
    //   K <= 32, up to ~20 000 columns per side: 1 -- every work item gets its own single-wave workgroup and the hardware
    //     dispatcher balances them (k_sample1: Gram on the 4x4x4 MFMA shape, factorisation on the VALU).
    //   K <= 32, more columns: 3 -- four work items per wave with the factorisation on the MFMA as well (k_sample4: a third of
    //     the VALU work per column, but a quarter of the workgroups, which only pays when there are enough of them --
    //     24 000 x 14 800: even; 60 400 x 37 060: 349 / 382 us against 401 / 510 us; 1M x 500K x 45M ratings: 2.35 / 3.13 ms
    //     against 3.5 / 5.3 ms).  BPMF_HIP_MODE=1 | 3 forces one of the two (the tests run small matrices through both).
    //   K = 64: 4 -- the slab form (kernels_slab.h: one wave per item, factorisation on the 4x4x4 f64 MFMA), the product form
    //     for columns with <= 16 ratings when those are at least half of the side (kernels_lr.h).
    //   K = 128: 5 -- a workgroup per item (kernels_wg2.h), fp64 or fp32 factors.
    // The forms that lost their place over rounds 1-4 (persistent waves, workgroup per column, Gram per wave + factorisation
    // per group of four, four items in a row per wave, the pair launch, Householder sweeps for light columns, the K = 128
    // slab form) left the library in round 5; what each measured is in docs/FINDINGS.md.
    const int mode_env = env_int("BPMF_HIP_MODE", -1);
    const bool f32 = s->ctx->dtype == BPMF_HIP_F32;
    const bool big = K == 128;
    if (big) s->mode = 5;
    else if (K == 64) s->mode = 4;
    else s->mode = (mode_env == 1 || mode_env == 3) ? mode_env : (nloc >= 20000 ? 3 : 1);
    // (mode 5 in fp64 -- K = 128 fp64: chunks twice as long as the fp32 form's, ML-1M shape 768 against 384 ratings: 0.609 / 0.722 against
    //  0.630 / 0.746 ms per launch; a chunk's partial is 75 KB there)
    int chunk = env_int("BPMF_HIP_CHUNK", 0);
    if (chunk <= 0) {
        const int64_t simds = (int64_t)s->ctx->num_cu * 4;
        // mode 1: ~one chunk of work per SIMD (ML-1M shape, round 3: 896 ratings 0.1011 ms per iteration, 640: 0.1025, 1 280: 0.1047).
        // Lower bound 16 K: a chunk's partial tiles are ~1.3 K^2 doubles written and read back, against 8 K
        // bytes gathered per rating, so shorter chunks make the partials a first-order traffic term.
        int64_t c = s->mode == 4 ? (s->nnz * 9) / (simds * 16) : (s->mode == 5 ? (s->nnz * (f32 ? 3 : 6)) / (simds * 7) : s->nnz / simds);
        c = (c + 63) / 64 * 64;
        // slab form: ONE wave walks an item, and a rating costs 36 (K = 128) / 10 (K = 64) tile MFMAs per 4
        // ratings: a launch lasts (average load of a wave slot) + (longest item), so items must stay short (ML-1M shape,
        // K = 64: 512-rating chunks 0.208 ms per launch, 256: 0.179)
        const int64_t lo = s->mode == 4 ? 256 : (s->mode == 5 ? 256 : 16 * K);     // (mode 5, ML-1M shape: 384-rating chunks 0.875 ms per iteration, 640: 0.90)
        chunk = (int)std::min<int64_t>(std::max<int64_t>(c, lo), 65536);   // (upper limit: 10M x 1M shards measured best with 64 K-rating chunks)
        // four columns per wave: a wave holds four items (and a chunk's partial is a quarter of the
        // size), so the same work per wave means chunks of a quarter of the length
        if (s->mode == 3) chunk = std::max(chunk / 4, 4 * K);
    }
    chunk = (chunk + 15) / 16 * 16;

    struct Item { int32_t col; int64_t p0; int32_t len; int32_t mc; int32_t chunk; int64_t cost; };
    std::vector<Item> items;
    items.reserve((size_t)nloc + (size_t)(s->nnz / chunk) + 16);
    std::vector<int32_t> mc_slot0, mc_nch;
    // The sort key of the item list: ratings + a SMALL constant for everything after the Gram.  The chunks of the heavy
    // columns -- whose last arriver still has a chunk sum and a factorisation ahead of it -- must start before whole
    // columns of similar length: with the factorisation priced at what it costs (K^2 / 4 + 64 "ratings": rounds 1-2)
    // whole columns overtook them and every launch ended on the heavy columns' last arrivers.  Round 3, ML-1M shape:
    // K = 64 0.3645 -> 0.3045 ms per iteration, K = 128 0.822 -> 0.752, K = 32 0.0993 -> 0.0985; ChEMBL shape
    // 1.065 -> 1.031.  Flat between 1 and ~K^2 / 16; chunks ahead of ALL whole columns measured the same.
    const int64_t fin_cost = 32;
    int32_t slots = 0;
    for (int64_t c = 0; c < nloc; ++c) {
        const int64_t p0 = colptr[c], n = colptr[c + 1] - colptr[c];
        if (n < 0) return fail(BPMF_HIP_EINVAL, "colptr is not monotone");
        if (n <= chunk) {
            items.push_back({(int32_t)c, p0, (int32_t)n, -1, 0, n + fin_cost});
        } else {
            const int nch = (int)((n + chunk - 1) / chunk);
            // equalise the chunks of one column (multiples of 16 ratings)
            const int64_t per = ((n + nch - 1) / nch + 15) / 16 * 16;
            const int32_t mc = (int32_t)mc_slot0.size();
            mc_slot0.push_back(slots); mc_nch.push_back(nch);
            for (int k = 0; k < nch; ++k) {
                const int64_t b = std::min<int64_t>(k * per, n), e = std::min<int64_t>(b + per, n);
                // whichever chunk arrives last also factorises: spread that cost over the chunks
                items.push_back({(int32_t)c, p0 + b, (int32_t)(e - b), mc, k, (e - b) + fin_cost / nch});
            }
            slots += nch;
        }
    }
    // (stable, descending, by a small integer key: a counting sort -- the comparison sort of the 483 500 items of the ChEMBL-shaped
    //  compounds side was most of the 17 ms that side's hand-over took; the order is std::stable_sort's)
    auto sort_desc = [&](auto key) {
        int64_t kmax = 0;
        for (const Item &it : items) { const int64_t k = key(it); if (k < 0) { kmax = -1; break; } kmax = std::max(kmax, k); }
        if (kmax < 0 || kmax > (int64_t)(4 * items.size() + (1 << 20))) {
            std::stable_sort(items.begin(), items.end(), [&](const Item &a, const Item &b) { return key(a) > key(b); });
            return;
        }
        std::vector<size_t> at((size_t)kmax + 2, 0);
        for (const Item &it : items) at[(size_t)(kmax - key(it)) + 1]++;              // bucket 0 = the largest key
        for (size_t k = 1; k < at.size(); ++k) at[k] += at[k - 1];
        std::vector<Item> out(items.size());
        for (const Item &it : items) out[at[(size_t)(kmax - key(it))]++] = it;
        items.swap(out);
    };
    if (s->mode == 3) {
        // four items share a wave and all run as many Gram steps as the longest of them: group by
        // LENGTH (chunks of one column stay together), longest groups first
        sort_desc([](const Item &a) { return (int64_t)a.len; });
    } else {
        sort_desc([](const Item &a) { return a.cost; });
        // (measured again in round 3 with this key: one item from the head of the list, then n from its tail -- Gram-heavy and
        // factorisation-heavy items side by side on a SIMD from the start -- ML-1M shape 0.119 / 0.107 / 0.143 ms per
        // iteration for n = 1 / 2 / 3 against 0.0975: longest first stays)
    }

    // parts (bpmf_hip_side_set_overlap): the items of part c of this rank's columns form a contiguous window of the
    // list (each window keeps the order chosen above), so that part c can be sampled -- and then exchanged -- on its own
    s->sub_item_off.assign(1, 0);
    if (s->nsub > 1 && !s->sub_bounds.empty()) {
        const int64_t *sb = &s->sub_bounds[(size_t)s->ctx->rank * (s->nsub + 1)];
        auto part_of = [&](const Item &it) {
            const int64_t g = s->from + it.col;
            int c = 0;
            while (c + 1 < s->nsub && g >= sb[c + 1]) ++c;
            return c;
        };
        std::stable_sort(items.begin(), items.end(), [&](const Item &a, const Item &b) { return part_of(a) < part_of(b); });
        size_t i = 0;
        for (int c = 0; c < s->nsub; ++c) {
            while (i < items.size() && part_of(items[i]) == c) ++i;
            s->sub_item_off.push_back((int)i);
        }
    }
    const size_t nw = items.size();
    std::vector<int32_t> wcol(nw), wlen(nw), wmc(nw), wchunk(nw);
    std::vector<int64_t> wp0(nw);
    for (size_t i = 0; i < nw; ++i) { wcol[i] = items[i].col; wlen[i] = items[i].len; wmc[i] = items[i].mc; wchunk[i] = items[i].chunk; wp0[i] = items[i].p0; }

    s->nwork = (int)nw; s->nmulti = (int)mc_slot0.size(); s->nslots = slots;
    int rc;
    if (K == 64 && !f32 && s->nsub <= 1) {
        // Product form (k_sample_pf) for the columns with at most BPMF_HIP_PF ratings (default and maximum 16: what
        // k_sample_pf<64, 16> holds; 0: off), sorted by their number so that the waves of a workgroup stay in step:
        // worth launches of their own when they are at least half of the side (ChEMBL-shaped compounds)
        std::vector<int32_t> lc, ll, hc, hl, hm, hk; std::vector<int64_t> lp, hp;
        const int pfmax = std::max(0, std::min(env_int("BPMF_HIP_PF", 16), 16));
        const int nlr = pfmax;
        if (pfmax > 0) {                                              // by number of ratings, list order inside a number: one counting pass
            size_t at[18] = {0};
            for (const Item &it : items)
                if (it.mc < 0 && it.len <= pfmax) at[it.len + 1]++;
            for (int n = 1; n <= pfmax + 1; ++n) at[n] += at[n - 1];
            lc.resize(at[pfmax + 1]); ll.resize(at[pfmax + 1]); lp.resize(at[pfmax + 1]);
            if (pfmax >= 3) s->pf_class[1] = (int)at[4];
            if (pfmax >= 6) s->pf_class[2] = (int)at[7];
            for (const Item &it : items)
                if (it.mc < 0 && it.len <= pfmax) { const size_t q = at[it.len]++; lc[q] = it.col; ll[q] = it.len; lp[q] = it.p0; }
        }
        s->pf_ratings = s->pf_ratings2 = 0;
        for (int32_t l : ll) { s->pf_ratings += l; s->pf_ratings2 += (int64_t)l * l; }
        if (pfmax < 3) s->pf_class[1] = (int)lc.size();
        if (pfmax < 6) s->pf_class[2] = (int)lc.size();
        s->pf_class[3] = (int)lc.size();
        for (const Item &it : items)
            if (!(it.mc < 0 && it.len <= nlr)) { hc.push_back(it.col); hl.push_back(it.len); hm.push_back(it.mc); hk.push_back(it.chunk); hp.push_back(it.p0); }
        if (nlr > 0 && (int64_t)lc.size() * 2 >= nloc && !lc.empty()) {
            s->lr_n = (int)lc.size(); s->hv_nwork = (int)hc.size();
            if (s->pf_class[3] > 0 && (rc = dev_upload<double>(&s->d_pf_q, nullptr, (size_t)s->nrows * K))) return rc;
            if ((rc = dev_upload(&s->d_lr_col, lc.data(), lc.size())) || (rc = dev_upload(&s->d_lr_len, ll.data(), ll.size())) ||
                (rc = dev_upload(&s->d_lr_p0, lp.data(), lp.size())) || (rc = dev_upload(&s->d_hv_col, hc.data(), hc.size())) ||
                (rc = dev_upload(&s->d_hv_len, hl.data(), hl.size())) || (rc = dev_upload(&s->d_hv_mc, hm.data(), hm.size())) ||
                (rc = dev_upload(&s->d_hv_chunk, hk.data(), hk.size())) || (rc = dev_upload(&s->d_hv_p0, hp.data(), hp.size())))
                return rc;
        }
    }
    if ((rc = dev_upload(&s->d_wi_col, wcol.data(), nw))) return rc;
    if ((rc = dev_upload(&s->d_wi_len, wlen.data(), nw))) return rc;
    if ((rc = dev_upload(&s->d_wi_mc, wmc.data(), nw))) return rc;
    if ((rc = dev_upload(&s->d_wi_chunk, wchunk.data(), nw))) return rc;
    if ((rc = dev_upload(&s->d_wi_p0, wp0.data(), nw))) return rc;
    if ((rc = dev_upload(&s->d_mc_slot0, mc_slot0.data(), mc_slot0.size()))) return rc;
    if ((rc = dev_upload(&s->d_mc_nch, mc_nch.data(), mc_nch.size()))) return rc;
    {
        std::vector<unsigned> zeros(std::max<size_t>(mc_slot0.size(), 8 * 32), 0u);
        if ((rc = dev_upload(&s->d_mc_count, zeros.data(), std::max<size_t>(mc_slot0.size(), 1)))) return rc;
    }
    if (big) {
        // column statistics: <= 32 slices of columns x 36 tiles (k_colstats_f32), one partial (tiles | sum) per slice
        // (128 partials of 132 KB were 17 MB written and read back per half-iteration: the two kernels took 55 us alone)
        s->nstat_waves = (int)std::max<int64_t>(1, std::min<int64_t>((nloc + 63) / 64, 32));
        if ((rc = dev_upload<double>(&s->d_stat_partials, nullptr, (size_t)s->nstat_waves * ((size_t)K * K + K)))) return rc;
        if ((rc = dev_upload<double>(&s->d_partials, nullptr, (size_t)slots * part_words_rt(K, f32)))) return rc;     // chunks of heavy columns
        return 0;
    }
    const size_t pw = part_words_rt(K, false);
    if ((rc = dev_upload<double>(&s->d_partials, nullptr, (size_t)slots * pw))) return rc;
    // column statistics: one wave per 64+ columns, at most 2 waves per CU
    // (sides with hundreds of thousands of columns: the pass is a 8 K-byte-per-column stream, four times the waves)
    s->nstat_waves = (int)std::max<int64_t>(1, std::min<int64_t>((nloc + 31) / 32, (int64_t)s->ctx->num_cu * (nloc > 100000 ? 8 : 2)));
    // sides of the one-item-per-wave forms (their statistics ride at the head of the partner's launch): ~160 columns per
    // rider -- every rider is a wave slot the launch's first items do not get (ML-1M shape: 24-40 riders 0.0996 ms per
    // iteration, 189 / 116 riders 0.1011, 16: 0.107)
    if (nloc < 20000) s->nstat_waves = (int)std::max<int64_t>(1, std::min<int64_t>(s->nstat_waves, std::max<int64_t>((nloc + 159) / 160, 24)));
    // big sides: four-wave workgroups with a finisher that reads the partials contiguously (k_colstats_wg)
    s->nstat_wg = (nloc > 100000) ? (int)std::min<int64_t>((nloc + 127) / 128, (int64_t)s->ctx->num_cu * 2) : 0;
    if ((rc = dev_upload<double>(&s->d_stat_partials, nullptr, (size_t)std::max(s->nstat_waves, 2 * s->nstat_wg) * pw))) return rc;
    return 0;
}

// the device arrays build_schedule made (the schedule is rebuilt when the parts of the side change)
void free_schedule(bpmf_hip_side *s)
{
    void **ptrs[] = {(void **)&s->d_wi_col, (void **)&s->d_wi_len, (void **)&s->d_wi_mc, (void **)&s->d_wi_chunk, (void **)&s->d_wi_p0,
                     (void **)&s->d_mc_slot0, (void **)&s->d_mc_nch, (void **)&s->d_mc_count, (void **)&s->d_partials, (void **)&s->d_stat_partials,
                     (void **)&s->d_lr_col, (void **)&s->d_lr_len, (void **)&s->d_lr_p0, (void **)&s->d_hv_col, (void **)&s->d_hv_len,
                     (void **)&s->d_hv_mc, (void **)&s->d_hv_chunk, (void **)&s->d_hv_p0, (void **)&s->d_pf_q};
    for (void **p : ptrs) if (*p) { (void)hipFree(*p); *p = nullptr; }
    s->lr_n = s->hv_nwork = 0;
}


// ---------------------------------------------------------------------------
static int side_create_common(bpmf_hip_ctx *ctx, int64_t ncols, int64_t nrows, int64_t from, int64_t to,
                              const int64_t *colptr, const int32_t *rowidx, const double *vals, bool dev_arrays,
                              double mean_rating, bpmf_hip_side **out)
{
    if (!out) return fail(BPMF_HIP_EINVAL, "side_create: out is NULL");
    *out = nullptr;
    if (!ctx || !colptr || ncols <= 0 || nrows <= 0 || from < 0 || to < from || to > ncols)
        return fail(BPMF_HIP_EINVAL, "side_create: bad argument");
    const int64_t nloc = to - from;
    if (nloc >= (int64_t)1 << 31) return fail(BPMF_HIP_EINVAL, "side_create: more than 2^31-1 local columns");
    if (colptr[0] != 0) return fail(BPMF_HIP_EINVAL, "side_create: colptr[0] must be 0 (pass the local slice)");
    const int64_t nnz = colptr[nloc];
    if (nnz > 0 && (!rowidx || !vals)) return fail(BPMF_HIP_EINVAL, "side_create: NULL rowidx/vals");
    HIP_TRY(hipSetDevice(ctx->device));
    if (!dev_arrays) {
        for (int64_t p = 0; p < nnz; ++p)
            if (rowidx[p] < 0 || rowidx[p] >= nrows) return fail(BPMF_HIP_EINVAL, "side_create: row index out of range");
    }
    bpmf_hip_side *s = new (std::nothrow) bpmf_hip_side();
    if (!s) return fail(BPMF_HIP_ENOMEM, "side_create: out of host memory");
    s->ctx = ctx; s->ncols = ncols; s->nrows = nrows; s->from = from; s->to = to; s->nnz = nnz; s->mean_rating = mean_rating;
    int rc = 0;
    if (dev_arrays) {
        s->d_rowidx = const_cast<int32_t *>(rowidx); s->d_vals = const_cast<double *>(vals); s->own_csc = false;
    } else {
        if ((rc = dev_upload(&s->d_rowidx, rowidx, (size_t)nnz)) || (rc = dev_upload(&s->d_vals, vals, (size_t)nnz))) { bpmf_hip_side_destroy(s); return rc; }
    }
    const size_t words = (size_t)ctx->K * (size_t)ncols;
    const size_t esz = ctx->dtype == BPMF_HIP_F32 ? sizeof(float) : sizeof(double);
    hipError_t e = hipMalloc((void **)&s->d_items, words * esz);
    if (e != hipSuccess) { bpmf_hip_side_destroy(s); return fail(BPMF_HIP_ENOMEM, "side_create: factor matrix allocation failed"); }
    e = hipMemset(s->d_items, 0, words * esz);                      // items().setZero(), c++/sample.cpp:185
    if (e != hipSuccess) { bpmf_hip_side_destroy(s); return fail(BPMF_HIP_ENODEV, "side_create: memset failed"); }
    if (env_int("BPMF_HIP_DBUF", 1) != 0 && hipMalloc((void **)&s->d_items_alt, words * esz) == hipSuccess) {
        if (hipMemset(s->d_items_alt, 0, words * esz) != hipSuccess) { (void)hipFree(s->d_items_alt); s->d_items_alt = nullptr; }
    } else {
        (void)hipGetLastError();                                    // no second copy: samplers write in place
        s->d_items_alt = nullptr;
    }
    s->h_colptr.assign(colptr, colptr + nloc + 1);
    if ((rc = build_schedule(s, colptr))) { bpmf_hip_side_destroy(s); return rc; }
    *out = s;
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_side_create(bpmf_hip_ctx *ctx, int64_t ncols, int64_t nrows, int64_t from, int64_t to,
                                    const int64_t *colptr, const int32_t *rowidx, const double *vals,
                                    double mean_rating, bpmf_hip_side **out)
{
    return side_create_common(ctx, ncols, nrows, from, to, colptr, rowidx, vals, false, mean_rating, out);
}

extern "C" int bpmf_hip_side_create_dev(bpmf_hip_ctx *ctx, int64_t ncols, int64_t nrows, int64_t from, int64_t to,
                                        const int64_t *colptr_host, const int32_t *rowidx_dev, const double *vals_dev,
                                        double mean_rating, bpmf_hip_side **out)
{
    return side_create_common(ctx, ncols, nrows, from, to, colptr_host, rowidx_dev, vals_dev, true, mean_rating, out);
}

extern "C" int bpmf_hip_side_destroy(bpmf_hip_side *s)
{
    if (!s) return BPMF_HIP_OK;
    (void)settle_async(s);
    if (s->worker.joinable()) {
        { std::lock_guard<std::mutex> lk(s->wm); s->wstop = true; }
        s->wcv.notify_all();
        s->worker.join();
    }
    predraw_stop(s);
    (void)hipSetDevice(s->ctx->device);
    if (g_trace_on && s->n_gap > 0)
        fprintf(stderr, "[bpmf_hip] side %04x: previous sampler's end -> this sampler's start: %.2f us (mean of %lld timed launches)\n",
                (unsigned)((uintptr_t)s >> 4) & 0xFFFF, s->tot_gap_ms / (double)s->n_gap * 1e3, (long long)s->n_gap);
    {   // an evaluation over this side's test matrix that was never enqueued dies with the side
        std::lock_guard<std::mutex> lk(s->ctx->launch_mutex);
        for (bpmf_hip_side *sd : s->ctx->sides)
            if (sd->deferred_eval && sd->deferred_eval->side == s) {
                sd->deferred_eval->deferred = false; sd->deferred_eval->cancelled = true;
                sd->deferred_eval = nullptr;
            }
    }
    flush_deferred(s->deferred_eval);                               // (it would go to this side's stream)
    (void)bounded_stream_sync(s->ctx, s->ctx->stream, __func__);
    if (s->saux) {
        (void)bounded_stream_sync(s->ctx, s->saux, __func__); (void)hipStreamDestroy(s->saux);
        std::lock_guard<std::mutex> lk(s->ctx->launch_mutex);
        auto &v = s->ctx->sides;
        v.erase(std::remove(v.begin(), v.end(), s), v.end());
    }
    if (s->sx) { (void)bounded_stream_sync(s->ctx, s->sx, __func__); (void)hipStreamDestroy(s->sx); }
    for (hipEvent_t e : s->sub_ev) if (e) (void)hipEventDestroy(e);
    if (s->sx_done) (void)hipEventDestroy(s->sx_done);
    if (s->ev_stat_go) (void)hipEventDestroy(s->ev_stat_go);
    if (s->own_csc) { if (s->d_rowidx) (void)hipFree(s->d_rowidx); if (s->d_vals) (void)hipFree(s->d_vals); }
    if (s->own_items && s->d_items) (void)hipFree(s->d_items);
    if (s->d_items_alt) (void)hipFree(s->d_items_alt);
    if (s->d_prop) (void)hipFree(s->d_prop);
    if (s->d_aggr_mu) (void)hipFree(s->d_aggr_mu);
    if (s->d_aggr_lambda) (void)hipFree(s->d_aggr_lambda);
    void *ptrs[] = {s->d_wi_col, s->d_wi_len, s->d_wi_mc, s->d_wi_chunk, s->d_wi_p0, s->d_mc_slot0, s->d_mc_nch, s->d_mc_count, s->d_partials,
                    s->d_stat_partials, s->a_d_in,
                    s->d_lr_col, s->d_lr_len, s->d_lr_p0, s->d_hv_col, s->d_hv_len, s->d_hv_mc, s->d_hv_chunk, s->d_hv_p0,
                    s->d_conn_send, s->d_conn_recv, s->d_conn_sbuf, s->d_conn_rbuf, s->d_pf_q,
                    s->d_prec, s->d_t_colptr, s->d_t_rowidx, s->d_t_vals, s->d_t_order};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (s->a_h_in) (void)hipHostFree(s->a_h_in);
    if (s->a_h_out) (void)hipHostFree(s->a_h_out);
    for (auto &set : s->evs) for (hipEvent_t e : set) if (e) (void)hipEventDestroy(e);
    if (s->a_gate) (void)hipHostFree(s->a_gate);
    if (s->a_ticket) (void)hipFree(s->a_ticket);
    if (s->a_dflag) (void)hipFree(s->a_dflag);
    if (s->a_d_red) (void)hipFree(s->a_d_red);
    delete s;
    return BPMF_HIP_OK;
}

// ---- padded num_latent (ctx->Kt < ctx->K) -----------------------------------------------------
// Everything that crosses the C ABI has the caller's size Kt; the device side has the instantiated size K.
// Kt x Kt column-major -> K x K column-major, `diag` on the extra diagonal, zeros elsewhere
void pad_square(int Kt, int K, const double *src, double *dst, double diag)
{
    for (int j = 0; j < K; ++j)
        for (int i = 0; i < K; ++i)
            dst[(size_t)j * K + i] = (i < Kt && j < Kt) ? src[(size_t)j * Kt + i] : ((i == j) ? diag : 0.0);
}
void unpad_square(int Kt, int K, const double *src, double *dst)
{
    for (int j = 0; j < Kt; ++j) memcpy(dst + (size_t)j * Kt, src + (size_t)j * K, sizeof(double) * Kt);
}

// Sys::add_prop_posterior (c++/sample.cpp:157-174): per-column priors from a previous run's
// *-mu.ddm / *-Lambda.ddm.  Like the reference, only Lambda takes part in the update (the loaded
// mu is never used: rr = hp_LambdaF * hp.mu, c++/sample.cpp:285).
extern "C" int bpmf_hip_side_set_prop_posterior(bpmf_hip_side *s, const double *mu, const double *Lambda)
{
    if (!s) return fail(BPMF_HIP_EINVAL, "set_prop_posterior: NULL");
    (void)mu;
    HIP_TRY(hipSetDevice(s->ctx->device));
    { const int rc = settle_async(s); if (rc) return rc; }
    { const int rs_ = bounded_stream_sync(s->ctx, s->ctx->stream, __func__); if (rs_) return rs_; }
    if (s->d_prop) { (void)hipFree(s->d_prop); s->d_prop = nullptr; }
    if (!Lambda) return BPMF_HIP_OK;
    const int K = s->ctx->K, Kt = s->ctx->Kt;
    const size_t nloc = (size_t)(s->to - s->from), words = (size_t)K * K * nloc;
    if (hipMalloc((void **)&s->d_prop, std::max<size_t>(words, 1) * sizeof(double)) != hipSuccess)
        return fail(BPMF_HIP_ENOMEM, "set_prop_posterior: device allocation failed");
    if (Kt == K) {
        HIP_TRY(hipMemcpy(s->d_prop, Lambda, words * sizeof(double), hipMemcpyHostToDevice));
        return BPMF_HIP_OK;
    }
    // padded num_latent: identity in the extra dimensions of every column's prior, a few thousand columns at a time
    const size_t per = std::max<size_t>(1, ((size_t)32 << 20) / ((size_t)K * K * sizeof(double)));
    std::vector<double> buf(std::min(per, std::max<size_t>(nloc, 1)) * (size_t)K * K);
    for (size_t c0 = 0; c0 < nloc; c0 += per) {
        const size_t n = std::min(per, nloc - c0);
        for (size_t c = 0; c < n; ++c) pad_square(Kt, K, Lambda + (c0 + c) * (size_t)Kt * Kt, buf.data() + c * (size_t)K * K, 1.0);
        HIP_TRY(hipMemcpy(s->d_prop + c0 * (size_t)K * K, buf.data(), n * (size_t)K * K * sizeof(double), hipMemcpyHostToDevice));
    }
    return BPMF_HIP_OK;
}

// evaluations that were requested but not enqueued yet and read this side's factors: enqueue them now
// (before the factors are replaced from outside, or a copy they captured goes away)
static void flush_evals_touching(bpmf_hip_side *s)
{
    std::vector<bpmf_hip_test *> pend;
    {
        std::lock_guard<std::mutex> lk(s->ctx->launch_mutex);
        for (bpmf_hip_side *sd : s->ctx->sides)
            if (sd->deferred_eval && (sd->deferred_eval->side == s || sd->deferred_eval->def_other == s)) pend.push_back(sd->deferred_eval);
    }
    for (bpmf_hip_test *t : pend) flush_deferred(t);
}

// the caller is about to use the raw pointer: from here on the samplers write in place
static int drop_second_copy(bpmf_hip_side *s)
{
    s->items_exposed = true;
    if (!s->d_items_alt) return 0;
    flush_evals_touching(s);                                        // (one may have captured the copy about to be freed)
    (void)settle_async(s);
    HIP_TRY(hipSetDevice(s->ctx->device));
    HIP_TRY(hipDeviceSynchronize());
    (void)hipFree(s->d_items_alt);
    s->d_items_alt = nullptr;
    return 0;
}

extern "C" double *bpmf_hip_side_items_dev(bpmf_hip_side *s)
{
    if (!s || s->ctx->dtype != BPMF_HIP_F64) return nullptr;
    if (drop_second_copy(s)) return nullptr;
    return s->d_items;
}

extern "C" int bpmf_hip_side_bind_items(bpmf_hip_side *s, double *items_dev, int ld, size_t bytes)
{
    if (!s || !items_dev) return fail(BPMF_HIP_EINVAL, "bind_items: NULL");
    if (s->ctx->dtype != BPMF_HIP_F64) return fail(BPMF_HIP_EINVAL, "bind_items: fp64 contexts only");
    // the kernels address the storage with the context's leading dimension (bpmf_hip_ctx_ld: 32 for num_latent 20), not with
    // num_latent: a caller that sized its buffer num_latent x ncols would have every sampler launch write past it
    if (ld != s->ctx->K)
        return fail(BPMF_HIP_EINVAL, "bind_items: leading dimension " + std::to_string(ld) + " given, the context's device arrays have " +
                    std::to_string(s->ctx->K) + " (bpmf_hip_ctx_ld; num_latent " + std::to_string(s->ctx->Kt) + ")");
    if (bytes < sizeof(double) * (size_t)s->ctx->K * (size_t)s->ncols)
        return fail(BPMF_HIP_EINVAL, "bind_items: " + std::to_string(bytes) + " bytes given, ld x ncols doubles = " +
                    std::to_string(sizeof(double) * (size_t)s->ctx->K * (size_t)s->ncols) + " needed");
    HIP_TRY(hipSetDevice(s->ctx->device));
    (void)settle_async(s);
    { const int rs_ = bounded_stream_sync(s->ctx, s->ctx->stream, __func__); if (rs_) return rs_; }
    if (s->saux) { const int rs_ = bounded_stream_sync(s->ctx, s->saux, __func__); if (rs_) return rs_; }
    { const int rc = drop_second_copy(s); if (rc) return rc; }
    if (s->own_items && s->d_items) (void)hipFree(s->d_items);
    s->d_items = items_dev; s->own_items = false;
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_side_get_items(bpmf_hip_side *s, double *h)
{
    if (!s || !h) return fail(BPMF_HIP_EINVAL, "get_items: NULL");
    HIP_TRY(hipSetDevice(s->ctx->device));
    { const int rs_ = bounded_stream_sync(s->ctx, s->ctx->stream, __func__); if (rs_) return rs_; }
    const size_t K = (size_t)s->ctx->K, Kt = (size_t)s->ctx->Kt, n = (size_t)s->ncols;     // (device leading dimension K, the caller's rows Kt)
    const size_t words = K * n;
    if (s->ctx->dtype == BPMF_HIP_F32) {                            // fp32 factors: widen on the host
        std::vector<float> tmp(words);
        HIP_TRY(hipMemcpy(tmp.data(), s->d_items, words * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t c = 0; c < n; ++c)
            for (size_t i = 0; i < Kt; ++i) h[c * Kt + i] = (double)tmp[c * K + i];
        return BPMF_HIP_OK;
    }
    if (Kt == K) HIP_TRY(hipMemcpy(h, s->d_items, words * sizeof(double), hipMemcpyDeviceToHost));
    else HIP_TRY(hipMemcpy2D(h, Kt * sizeof(double), s->d_items, K * sizeof(double), Kt * sizeof(double), n, hipMemcpyDeviceToHost));
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_side_set_items(bpmf_hip_side *s, const double *h)
{
    if (!s || !h) return fail(BPMF_HIP_EINVAL, "set_items: NULL");
    HIP_TRY(hipSetDevice(s->ctx->device));
    { const int rc = settle_async(s); if (rc) return rc; }
    flush_evals_touching(s);
    HIP_TRY(hipDeviceSynchronize());                                // (an evaluation beside the samplers may still read the factors)
    const size_t K = (size_t)s->ctx->K, Kt = (size_t)s->ctx->Kt, n = (size_t)s->ncols;     // (rows Kt .. K - 1 of every column stay zero)
    const size_t words = K * n;
    if (s->ctx->dtype == BPMF_HIP_F32) {
        std::vector<float> tmp(words, 0.0f);
        for (size_t c = 0; c < n; ++c)
            for (size_t i = 0; i < Kt; ++i) tmp[c * K + i] = (float)h[c * Kt + i];
        HIP_TRY(hipMemcpy(s->d_items, tmp.data(), words * sizeof(float), hipMemcpyHostToDevice));
        return BPMF_HIP_OK;
    }
    if (Kt == K) HIP_TRY(hipMemcpy(s->d_items, h, words * sizeof(double), hipMemcpyHostToDevice));
    else {
        HIP_TRY(hipMemset(s->d_items, 0, words * sizeof(double)));
        HIP_TRY(hipMemcpy2D(s->d_items, K * sizeof(double), h, Kt * sizeof(double), Kt * sizeof(double), n, hipMemcpyHostToDevice));
    }
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_side_kernel_name(const bpmf_hip_side *s, char *buf, int n)
{
    if (!s || !buf || n <= 0) return fail(BPMF_HIP_EINVAL, "side_kernel_name: bad argument");
    const bpmf_hip_ctx *c = s->ctx;
    const int K = c->K;
    const std::string k = std::to_string(K);
    const bool dist = c->comm != nullptr && !s->bounds.empty();
    const bool fusable = !dist && !s->reduce_on && env_int("BPMF_HIP_FUSED", 1) != 0 && s->nwork > 0;
    std::string name;
    if (s->reduce_on) name = "k_sample_prec<" + k + "> + k_precompute<" + k + ">";
    else if (c->dtype == BPMF_HIP_F32) name = "k_sample_wg2<128,2>";
    else if (K == 128) name = "k_sample_wg2<128,4,double>";
    else if (K == 64) {
        if (s->lr_n > 0 && !s->d_prop && !c->diag_only) {
            static const char *nb[3] = {"3", "6", "16"};
            for (int pc = 0; pc < 3; ++pc)
                if (s->pf_class[pc + 1] > s->pf_class[pc]) name += std::string(name.empty() ? "" : " + ") + "k_sample_pf<64," + nb[pc] + ">";
            if (s->hv_nwork > 0) name += " + k_sample_slab<64>";
        } else name = (fusable && s->lr_n == 0 && s->nsub <= 1) ? "k_sample1s<64>" : "k_sample_slab<64>";
    } else name = (s->mode == 3 ? "k_sample4<" : "k_sample1<") + k + ">";
    snprintf(buf, (size_t)n, "%s", name.c_str());
    return BPMF_HIP_OK;
}

// LDS / register budget and residency of the kernel(s) one sampler launch of the side consists of, asked of the dispatch logic
// itself (launch.h: Probe): per kernel 4 words -- LDS bytes per workgroup, threads per workgroup, workgroups resident per CU
// (hipOccupancyMaxActiveBlocksPerMultiprocessor), VGPRs -- and its name as the launch site spells it, ';'-separated.
// Returns the number of kernels (<= max_kernels), or a negative error code.  Nothing is launched.
extern "C" int bpmf_hip_side_kernel_resources(bpmf_hip_side *s, int64_t *out, int max_kernels, char *names, int names_len)
{
    if (!s || !out || max_kernels <= 0) return fail(BPMF_HIP_EINVAL, "side_kernel_resources: bad argument");
    bpmf_hip_ctx *c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    if (s->reduce_on) return 0;
    bpmf_launch::Probe pr;
    bpmf_launch::probe() = &pr;
    // (every kernel of sampler_into goes through BPMF_LAUNCH, which records instead of launching while the probe is installed)
    const int rc = BPMF_DISPATCH_K(c->K, (bpmf_launch::sampler_into<KK, FF>(s, s->d_items, s, 0, 1.0, c->d_in, c->stream, nullptr, nullptr)));
    bpmf_launch::probe() = nullptr;
    if (rc) return rc;
    std::string all;
    const int n = std::min(pr.n, max_kernels);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < 4; ++j) out[4 * i + j] = pr.v[i][j];
        std::string nm = pr.name[i];
        for (const char *strip : {"bpmf::", "(", ")"}) { size_t p; while ((p = nm.find(strip)) != std::string::npos) nm.erase(p, strlen(strip)); }
        all += (i ? ";" : "") + nm;
    }
    if (names && names_len > 0) snprintf(names, (size_t)names_len, "%s", all.c_str());
    return n;
}

// The static schedule of a side in numbers (build_schedule), for reports: out[0..15] =
//   0 sampler form (mode)   1 work items   2 chunks of heavy columns (partial slots)   3 heavy columns cut into chunks
//   4 light columns in the low-rank / product forms   5 work items of the others   6..8 product-form columns with <= 3 | 4..6 | 7..16 ratings
//   9 (was: columns in k_sample_lr; 0 since round 5)   10 parts (bpmf_hip_side_set_overlap)   11 local columns   12 local ratings
//   13, 14 sum over the product-form columns of their number of ratings n, of n^2   15 reserved (0)
extern "C" int bpmf_hip_side_schedule_info(const bpmf_hip_side *s, int64_t *out, int n)
{
    if (!s || !out || n < 16) return fail(BPMF_HIP_EINVAL, "side_schedule_info: bad argument (16 words)");
    for (int i = 0; i < n; ++i) out[i] = 0;
    out[0] = s->mode; out[1] = s->nwork; out[2] = s->nslots; out[3] = s->nmulti;
    out[4] = s->lr_n; out[5] = s->lr_n > 0 ? s->hv_nwork : s->nwork;
    // (the classes are only in use when the side is split)
    for (int pc = 0; pc < 3; ++pc) out[6 + pc] = s->lr_n > 0 ? s->pf_class[pc + 1] - s->pf_class[pc] : 0;
    out[9] = 0;                                                       // (round 2's reflector sweeps, k_sample_lr: gone)
    out[10] = s->nsub; out[11] = s->to - s->from; out[12] = s->nnz; out[13] = s->lr_n > 0 ? s->pf_ratings : 0; out[14] = s->lr_n > 0 ? s->pf_ratings2 : 0;
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_side_schedule_items(const bpmf_hip_side *s, int32_t *col, int32_t *len, int32_t *heavy, int64_t n, int64_t *nitems)
{
    if (!s || n < 0) return fail(BPMF_HIP_EINVAL, "side_schedule_items: bad argument");
    if (nitems) *nitems = s->nwork;
    const size_t m = (size_t)std::min<int64_t>(n, s->nwork);
    if (m == 0) return BPMF_HIP_OK;
    HIP_TRY(hipSetDevice(s->ctx->device));
    if (col) HIP_TRY(hipMemcpy(col, s->d_wi_col, m * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (len) HIP_TRY(hipMemcpy(len, s->d_wi_len, m * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (heavy) HIP_TRY(hipMemcpy(heavy, s->d_wi_mc, m * sizeof(int32_t), hipMemcpyDeviceToHost));
    return BPMF_HIP_OK;
}

// sum of the sampler / statistics kernel times over all collected launches of the stateful path
extern "C" int bpmf_hip_side_kernel_ms_sum(bpmf_hip_side *s, double *sample_ms, double *reduce_ms, int64_t *launches)
{
    if (!s) return fail(BPMF_HIP_EINVAL, "kernel_ms_sum: NULL");
    { const int rc = settle_async(s); if (rc) return rc; }
    if (sample_ms) *sample_ms = s->tot_sample_ms;
    if (reduce_ms) *reduce_ms = s->tot_reduce_ms;
    if (launches) *launches = s->n_launches;
    return BPMF_HIP_OK;
}


}  // namespace bpmf_capi
