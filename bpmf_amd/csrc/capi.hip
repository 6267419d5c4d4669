// capi.hip -- implementation of include/bpmf_hip.h on top of kernels.h.
//
// Host-side responsibilities: own the device buffers of a `Sys`, build the
// static work schedule of a side once (columns sorted by cost, heavy columns
// cut into nnz chunks), upload hp.mu / hp.LambdaF per half-iteration, launch
// the kernels on one stream and bring the K*K+K+1 reduction words back.
#include <dlfcn.h>

#include "launch.h"

namespace {
thread_local std::string g_err;
}
extern "C" void bpmf_hip_set_error_(const char *msg) { g_err = msg; }

Rccl *rccl()
{
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        // BPMF_HIP_RCCL_LIBRARY: another implementation of the nccl* entry points below -- the tests name their
        // double for ranks that share one GPU (tests/rccl_double), which the real library refuses to serve
        const char *over = getenv("BPMF_HIP_RCCL_LIBRARY");
        if (over && *over) {
            r.handle = dlopen(over, RTLD_NOW | RTLD_LOCAL);
            if (!r.handle) fprintf(stderr, "[bpmf_hip] BPMF_HIP_RCCL_LIBRARY=%s: %s\n", over, dlerror());
        } else
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (r.handle) {
#define BPMF_SYM(f) r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.handle, "nccl" #f))
            BPMF_SYM(GetUniqueId); BPMF_SYM(CommInitRank); BPMF_SYM(CommDestroy); BPMF_SYM(AllReduce);
            BPMF_SYM(Broadcast); BPMF_SYM(GroupStart); BPMF_SYM(GroupEnd); BPMF_SYM(GetErrorString); BPMF_SYM(CommSplit);
            BPMF_SYM(Send); BPMF_SYM(Recv); BPMF_SYM(AllGather); BPMF_SYM(Reduce); BPMF_SYM(CommCount); BPMF_SYM(CommAbort);
#undef BPMF_SYM
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.Broadcast || !r.GroupStart || !r.GroupEnd)
                r.handle = nullptr;
        }
    }
    return r.handle ? &r : nullptr;
}


static int settle_async(struct bpmf_hip_side *s);      // waits until the worker is done with `s`; returns its deferred error
static void flush_deferred(struct bpmf_hip_test *t, bool on_main = false);   // enqueues an evaluation whose launch was put off
namespace { void predraw_stop(struct bpmf_hip_side *s); }   // joins the side's pre-draw helper threads
namespace { int flush_pending_stats(struct bpmf_hip_ctx *c, bool on_main = false); }   // statistics without a launch to ride in: a kernel of their own

struct bpmf_hip_side;
// host-side timeline for BPMF_HIP_TRACE=1: (time, tag, side) records, printed when the context dies
namespace {
struct TraceRec { double us; const char *tag; const void *side; int iter; };
std::vector<TraceRec> g_trace;
std::mutex g_trace_mutex;
const bool g_trace_on = env_int("BPMF_HIP_TRACE", 0) != 0;
const std::chrono::steady_clock::time_point g_trace_t0 = std::chrono::steady_clock::now();
inline void trace(const char *tag, const bpmf_hip_side *s, int iter)
{
    if (!g_trace_on) return;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - g_trace_t0).count();
    std::lock_guard<std::mutex> lk(g_trace_mutex);
    if (g_trace.size() < (1u << 20)) g_trace.push_back({us, tag, s, iter});
}
void trace_dump()
{
    std::lock_guard<std::mutex> lk(g_trace_mutex);
    const size_t from = g_trace.size() > 160 ? g_trace.size() - 160 : 0;
    for (size_t i = from; i < g_trace.size(); ++i)
        fprintf(stderr, "[bpmf_hip] %12.1f us  side %04x  iter %4d  %s\n", g_trace[i].us, (unsigned)((uintptr_t)g_trace[i].side >> 4) & 0xFFFF, g_trace[i].iter, g_trace[i].tag);
    g_trace.clear();
}
struct TraceAtExit { ~TraceAtExit() { if (g_trace_on) trace_dump(); } } g_trace_at_exit;
}  // namespace

namespace {

// ---- bounded host-side waits (multi-GPU) --------------------------------------------------------
double comm_timeout_s()
{
    static const double v = std::max(1, env_int("BPMF_HIP_COMM_TIMEOUT_MS", 60000)) * 1e-3;
    return v;
}

// the peers never completed a collective: abort both communicators (their kernels leave the streams), mark the context
int comm_abort(bpmf_hip_ctx *c, const std::string &what)
{
    std::lock_guard<std::mutex> lk(c->abort_mutex);
    if (!c->comm_dead.exchange(true)) {
        Rccl *R = rccl();
        fprintf(stderr, "[bpmf_hip] rank %d of %d: %s did not complete within %.1f s: a peer rank stalled or died; aborting the communicator(s)\n",
                c->rank, c->nranks, what.c_str(), comm_timeout_s());
        if (R && R->CommAbort) {
            if (c->comm2) (void)R->CommAbort(c->comm2);
            if (c->comm) (void)R->CommAbort(c->comm);
            // (aborted = destroyed.  Both handles stay in place as "this context is sharded / has a second communicator": they
            // are read without a lock by the launch paths, and nothing uses them once comm_dead is set -- COMM_ALIVE_OR_FAIL)
        }
    }
    return fail(BPMF_HIP_ENODEV, "collective timed out (" + what + "): a peer rank stalled or died; the communicator was aborted");
}

// hipStreamSynchronize for a stream that may carry a collective: a poll with a deadline instead of a wait without one
int bounded_stream_sync(bpmf_hip_ctx *c, hipStream_t st, const char *what)
{
    if (!c->comm) { HIP_TRY(hipStreamSynchronize(st)); return 0; }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) return fail(BPMF_HIP_ENODEV, std::string(what) + ": " + hipGetErrorString(q));
        (void)hipGetLastError();
        if (spins < 2000) { __builtin_ia32_pause(); continue; }
        // (once the communicators are dead a stream may hold collective kernels that will never end -- with an RCCL that has no
        // ncclCommAbort for certain: no second full timeout for every later wait, ctx_destroy included)
        const double limit = c->comm_dead.load(std::memory_order_acquire) ? std::min(2.0, comm_timeout_s()) : comm_timeout_s();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) return comm_abort(c, what);
        std::this_thread::sleep_for(std::chrono::microseconds(spins < 20000 ? 20 : 200));
    }
}

int bounded_event_sync(bpmf_hip_ctx *c, hipEvent_t ev, const char *what)
{
    if (!c->comm) { HIP_TRY(hipEventSynchronize(ev)); return 0; }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) return fail(BPMF_HIP_ENODEV, std::string(what) + ": " + hipGetErrorString(q));
        (void)hipGetLastError();
        if (spins < 2000) { __builtin_ia32_pause(); continue; }
        // (once the communicators are dead a stream may hold collective kernels that will never end -- with an RCCL that has no
        // ncclCommAbort for certain: no second full timeout for every later wait, ctx_destroy included)
        const double limit = c->comm_dead.load(std::memory_order_acquire) ? std::min(2.0, comm_timeout_s()) : comm_timeout_s();
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) return comm_abort(c, what);
        std::this_thread::sleep_for(std::chrono::microseconds(spins < 20000 ? 20 : 200));
    }
}

// The kernels write their few result words straight into pinned host memory; the last block
// of the last kernel then publishes a sequence number and the host thread spins on it.  This replaces
// hipMemcpyAsync(D2H) + hipStreamSynchronize (a copy-engine hop and a sleeping wait per
// half-iteration) on a path whose device work is only tens of microseconds.
int wait_host(bpmf_hip_ctx *c)
{
    unsigned *flag = reinterpret_cast<unsigned *>(c->h_out + c->out_words - 1);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == c->seq) return 0;
        if (spin_limit_s() <= 0.0) break;
        __builtin_ia32_pause();
        if ((spins & 0xFFFu) == 0xFFFu) {
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (s > spin_limit_s()) break;       // long kernel (big matrix) or an error: fall back to a blocking wait
        }
    }
    { const int rc = bounded_stream_sync(c, c->stream, "sampler + exchange + all-reduce of a half-iteration"); if (rc) return rc; }
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != c->seq) return fail(BPMF_HIP_ENODEV, "device did not publish its results");
    return 0;
}

// Build the static schedule of a side.  Cost model: one MFMA k-step per 4
// ratings per tile triple, plus a constant for the factorisation.
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
    const int64_t fin_cost = env_int("BPMF_HIP_FINCOST", 0) > 0 ? env_int("BPMF_HIP_FINCOST", 0) : 32;     // (BPMF_HIP_FINCOST: experiments)
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
    if (s->mode == 3) {
        // four items share a wave and all run as many Gram steps as the longest of them: group by
        // LENGTH (chunks of one column stay together), longest groups first
        std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.len > b.len; });
    } else {
        std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.cost > b.cost; });
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
        for (int n = 0; n <= pfmax && pfmax > 0; ++n) {
            for (const Item &it : items)
                if (it.mc < 0 && it.len == n) { lc.push_back(it.col); ll.push_back(it.len); lp.push_back(it.p0); }
            if (n == 3) s->pf_class[1] = (int)lc.size();
            if (n == 6) s->pf_class[2] = (int)lc.size();
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
    if (env_int("BPMF_HIP_NSTAT", 0) > 0) s->nstat_waves = (int)std::min<int64_t>(env_int("BPMF_HIP_NSTAT", 0), std::max<int64_t>(1, (nloc + 31) / 32));   // (experiments)
    // big sides: four-wave workgroups with a finisher that reads the partials contiguously (k_colstats_wg)
    s->nstat_wg = (nloc > 100000 && env_int("BPMF_HIP_STATS_WG", 1) != 0) ? (int)std::min<int64_t>((nloc + 127) / 128, (int64_t)s->ctx->num_cu * 2) : 0;
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

}  // namespace

// ---------------------------------------------------------------------------
extern "C" const char *bpmf_hip_last_error(void) { return g_err.c_str(); }
extern "C" int bpmf_hip_abi_version(void) { return BPMF_HIP_ABI_VERSION; }
extern "C" int bpmf_hip_supports_k(int K) { return K >= 1 && K <= 128; }

extern "C" int bpmf_hip_supports(int K, int dtype)
{
    if (dtype == BPMF_HIP_F64) return bpmf_hip_supports_k(K);
    if (dtype == BPMF_HIP_F32) return K > 64 && K <= 128;
    return 0;
}

// the instantiated num_latent a context of (K, dtype) runs on: 8, 16, 32, 64, 128 (0: unsupported)
extern "C" int bpmf_hip_kernel_k(int K, int dtype)
{
    if (!bpmf_hip_supports(K, dtype)) return 0;
    if (dtype == BPMF_HIP_F32) return 128;
    return K <= 8 ? 8 : K <= 16 ? 16 : K <= 32 ? 32 : K <= 64 ? 64 : 128;
}

extern "C" int bpmf_hip_ctx_ld(const bpmf_hip_ctx *c) { return c ? c->K : 0; }
extern "C" int bpmf_hip_ctx_num_latent(const bpmf_hip_ctx *c) { return c ? c->Kt : 0; }
extern "C" int bpmf_hip_ctx_dtype(const bpmf_hip_ctx *c) { return c ? c->dtype : -1; }

static int ctx_create_impl(int device, int K, int dtype, void *stream, bpmf_hip_ctx **out);

extern "C" int bpmf_hip_ctx_create(int device, int K, void *stream, bpmf_hip_ctx **out)
{
    return ctx_create_impl(device, K, BPMF_HIP_F64, stream, out);
}

extern "C" int bpmf_hip_ctx_create_ex(int device, int K, int dtype, void *stream, bpmf_hip_ctx **out)
{
    return ctx_create_impl(device, K, dtype, stream, out);
}

static int ctx_create_impl(int device, int Ktrue, int dtype, void *stream, bpmf_hip_ctx **out)
{
    if (!out) return fail(BPMF_HIP_EINVAL, "ctx_create: out is NULL");
    *out = nullptr;
    if (!bpmf_hip_supports(Ktrue, dtype))
        return fail(BPMF_HIP_EINVAL, "ctx_create: unsupported num_latent / dtype " + std::to_string(Ktrue) + " / " + std::to_string(dtype) +
                                         " (fp64: 1 .. 128; fp32: 65 .. 128)");
    const int K = bpmf_hip_kernel_k(Ktrue, dtype);                  // what the kernels are instantiated for (>= Ktrue)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(BPMF_HIP_ENODEV, "no HIP device available (the BPMF hot path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(BPMF_HIP_EINVAL, "ctx_create: bad device index");
    HIP_TRY(hipSetDevice(device));
    bpmf_hip_ctx *c = new (std::nothrow) bpmf_hip_ctx();
    if (!c) return fail(BPMF_HIP_ENOMEM, "ctx_create: out of host memory");
    c->device = device; c->K = K; c->Kt = Ktrue; c->dtype = dtype;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
#if defined(BPMF_PROFILING) && BPMF_PROFILING
    c->ablate = (unsigned)env_int("BPMF_HIP_ABLATE", 0);
#else
    // the product build has no profiling hooks in its kernels (kernels.h: kProfiling): asking for them must not pass silently
    if (env_int("BPMF_HIP_ABLATE", 0) || env_int("BPMF_HIP_STAMPS", 0)) {
        delete c;
        return fail(BPMF_HIP_EINVAL, "BPMF_HIP_ABLATE / BPMF_HIP_STAMPS need the profiling build of the library "
                                     "(make -C bpmf_amd/csrc prof; BPMF_HIP_LIBRARY=<repo>/bpmf_amd/libbpmf_hip_prof.so)");
    }
#endif
    if (stream) { c->stream = (hipStream_t)stream; c->own_stream = false; }
    else { HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
    c->in_words = (size_t)K * K + K + 2 + K;                           // LambdaF | Lmu | fail | pad | mu (even: staged as 16-byte words)
    if (K == 64 && dtype == BPMF_HIP_F64) c->in_words += 2 * (size_t)K * K + K;   // | R0 = chol(LambdaF).matrixU() row-major | (R0^-1)^T | R0^-T LambdaF mu (k_sample_pf)
    c->out_words = (size_t)K * K + K + 1 + 1 + 2 + 1;
    HIP_TRY(hipHostMalloc((void **)&c->h_in, c->in_words * sizeof(double), hipHostMallocMapped));
    HIP_TRY(hipHostMalloc((void **)&c->h_out, c->out_words * sizeof(double), hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void **)&c->h_in_dev, c->h_in, 0));
    HIP_TRY(hipHostGetDevicePointer((void **)&c->h_out_dev, c->h_out, 0));
    memset(c->h_out, 0, c->out_words * sizeof(double));
    HIP_TRY(hipMalloc((void **)&c->d_in, (c->in_words + lf32_words(c)) * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&c->d_red, (c->out_words + 8) * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&c->d_ticket, 64));
    HIP_TRY(hipMemset(c->d_ticket, 0, 64));
    HIP_TRY(hipMalloc((void **)&c->d_zero, 1024));
    HIP_TRY(hipMemset(c->d_zero, 0, 1024));
#if defined(BPMF_PROFILING) && BPMF_PROFILING
    if (env_int("BPMF_HIP_STAMPS", 0)) { HIP_TRY(hipMalloc((void **)&c->d_stamps, 4096)); HIP_TRY(hipMemset(c->d_stamps, 0, 4096)); }
#endif
    for (auto &e : c->ev) HIP_TRY(hipEventCreate(&e));
    *out = c;
    return BPMF_HIP_OK;
}

// the BPMF_NO_COVARIANCE build of the reference (c++/sample.cpp:300-304) as a run-time switch
extern "C" int bpmf_hip_ctx_set_no_covariance(bpmf_hip_ctx *c, int on)
{
    if (!c) return fail(BPMF_HIP_EINVAL, "set_no_covariance: NULL");
    c->diag_only = on ? 1u : 0u;
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_ctx_destroy(bpmf_hip_ctx *c)
{
    if (!c) return BPMF_HIP_OK;
    if (g_trace_on) trace_dump();
    (void)hipSetDevice(c->device);
    (void)bounded_stream_sync(c, c->stream, __func__);
    if (c->d_stamps) {                                              // the last launch's stamps of the two probe items
        unsigned long long h[512];
        if (hipMemcpy(h, c->d_stamps, sizeof h, hipMemcpyDeviceToHost) == hipSuccess)
            for (int probe = 0; probe < 2; ++probe) {
                fprintf(stderr, "[bpmf_hip] stamps of probe item %d (100 MHz ticks since its start):", probe);
                for (int i = 1; i < 64; ++i) if (h[probe * 64 + i]) fprintf(stderr, " %d:%lld", i, (long long)(h[probe * 64 + i] - h[probe * 64]));
                fprintf(stderr, "\n");
            }
        if (h[129]) fprintf(stderr, "[bpmf_hip] all launches: %llu items, mean life of wave 0 %.1f us\n", h[129], (double)h[128] / (double)h[129] / 100.0);
        (void)hipFree(c->d_stamps);
    }
    for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->h_in) (void)hipHostFree(c->h_in);
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->d_in) (void)hipFree(c->d_in);
    if (c->d_ticket) (void)hipFree(c->d_ticket);
    if (c->d_zero) (void)hipFree(c->d_zero);
    if (c->d_red) (void)hipFree(c->d_red);
    if (!c->comm_dead.load()) {                                      // (aborted communicators are gone already)
        if (c->comm2 && rccl()) (void)rccl()->CommDestroy(c->comm2);
        if (c->comm && rccl()) (void)rccl()->CommDestroy(c->comm);
    }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_ctx_sync(bpmf_hip_ctx *c)
{
    if (!c) return fail(BPMF_HIP_EINVAL, "ctx_sync: NULL");
    std::vector<bpmf_hip_side *> sides;
    { std::lock_guard<std::mutex> lk(c->launch_mutex); sides = c->sides; }
    int rc = 0;
    trace("ctx_sync: enter", nullptr, 0);
    if (c->pending_stats) { HIP_TRY(hipSetDevice(c->device)); rc = flush_pending_stats(c); }   // (start them before waiting for the other side's collection)
    for (bpmf_hip_side *s : sides) { const int r = settle_async(s); if (r && !rc) rc = r; }
    for (bpmf_hip_side *s : sides) flush_deferred(s->deferred_eval);
    // (a query first: after the collections above the streams are usually idle already, and a blocking
    // synchronize of an idle stream still costs ~10 us each -- 30 us per fence of a 2 ms block of bench.py)
    auto sync_stream = [c](hipStream_t st) -> int {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) return fail(BPMF_HIP_ENODEV, std::string("ctx_sync: ") + hipGetErrorString(q));
        (void)hipGetLastError();                                      // ("not ready" is no error: do not leave it for a later hipGetLastError())
        return bounded_stream_sync(c, st, "ctx_sync");
    };
    { const int r = sync_stream(c->stream); if (r) return r; }
    for (bpmf_hip_side *s : sides) { const int r = sync_stream(s->saux); if (r) return r; }
    trace("ctx_sync: done", nullptr, 0);
    return rc;
}

extern "C" void *bpmf_hip_ctx_stream(bpmf_hip_ctx *c) { return c ? (void *)c->stream : nullptr; }

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
            if (sd->deferred_eval && sd->deferred_eval->side == s) { sd->deferred_eval->deferred = false; sd->deferred_eval->cancelled = true; sd->deferred_eval = nullptr; }
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
    void *ptrs[] = {s->d_wi_col, s->d_wi_len, s->d_wi_mc, s->d_wi_chunk, s->d_wi_p0, s->d_mc_slot0, s->d_mc_nch, s->d_mc_count, s->d_partials, s->d_stat_partials, s->a_d_in,
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
namespace {
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
}  // namespace

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

namespace {

using bpmf_launch::sampler_into;

template <int K, bool F32>
int launch_sampler(bpmf_hip_side *self, const bpmf_hip_side *other, int iter, double alpha, double *d_in, hipStream_t st,
                   hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr)
{
    if (!second_copy_usable(self)) return sampler_into<K, F32>(self, self->d_items, other, iter, alpha, d_in, st, ev_start, ev_stop);
    // the copy about to be overwritten may still be read by an evaluation that has not been collected
    const int tgt = self->cur_buf ^ 1;
    bpmf_hip_side::Reader &rd = self->readers[tgt];
    if (rd.t) {
        if (rd.t->deferred && rd.seq == rd.t->seq + 1) flush_deferred(rd.t);      // (the one that reads this copy, not a later one)
        if (rd.t->done_seq < rd.seq) HIP_TRY(hipStreamWaitEvent(st, rd.t->ev_done[rd.seq & 1u], 0));
    }
    rd.t = nullptr;
    const int rc = sampler_into<K, F32>(self, self->d_items_alt, other, iter, alpha, d_in, st, ev_start, ev_stop);
    if (rc) return rc;
    std::swap(self->d_items, self->d_items_alt);                    // everything enqueued from here on sees the new factors
    self->cur_buf = tgt;
    return 0;
}

// One half-iteration in the BPMF_REDUCE formulation (c++/sample.cpp:289-291,375-377; c++/mpi_reduce.h:24-47):
//   1. multi-GPU: the Gram parts every rank precomputed for this side's columns are summed onto the owner of each
//      range (one ncclReduce per owner, grouped -- MPI_Reduce per owner in the reference)
//   2. the local columns are sampled from prior + precomputed sums (k_sample_prec)
//   3. other.preComputeMuLambda(self): the parts of EVERY column of the other side that come from this rank's
//      fresh columns (k_precompute over the transposed local block)
// The factors themselves are still exchanged afterwards: the sampler no longer needs them, but the evaluation over
// the whole test set and the outputs do (the reference's predict is restricted to local rows in this mode, with a
// warning: c++/sample.cpp:59-61,71-74).
template <int K, bool F32>
int reduce_half_iteration(bpmf_hip_side *self, const bpmf_hip_side *other, int iter, double alpha, double *d_in, hipStream_t st,
                          hipEvent_t ev_start, hipEvent_t ev_stop)
{
    bpmf_hip_ctx *c = self->ctx;
    if (!other->reduce_on || !self->d_prec || !other->d_prec || !self->d_t_colptr)
        return fail(BPMF_HIP_EINVAL, "BPMF_REDUCE formulation: enable it for both sides (bpmf_hip_sys_set_reduce)");
    const size_t part = (size_t)bpmf_launch::reduce_part_words(K);
    const bool dist = c->comm != nullptr && !self->bounds.empty();
    if (dist) {                                                     // (one rank: the reduce is the identity, the path is the same)
        Rccl *R = rccl();
        COMM_ALIVE_OR_FAIL(c, "BPMF_REDUCE half-iteration");
        if (!R->Reduce) return fail(BPMF_HIP_ENODEV, "BPMF_REDUCE formulation: this RCCL has no ncclReduce");
        NcclGroup group(R);
        NCCL_TRY(group.start());
        for (int r = 0; r < c->nranks; ++r) {
            const int64_t lo = self->bounds[(size_t)r], hi = self->bounds[(size_t)r + 1];
            if (hi > lo) {
                double *p = self->d_prec + (size_t)lo * part;
                NCCL_TRY(R->Reduce(p, p, (size_t)(hi - lo) * part, ncclDouble, ncclSum, r, c->comm, st));
            }
        }
        NCCL_TRY(group.end());
    }
    // the factor copy this half-iteration writes (second copy: see launch_sampler)
    double *out_items = self->d_items;
    const bool swap = second_copy_usable(self);
    if (swap) {
        const int tgt = self->cur_buf ^ 1;
        bpmf_hip_side::Reader &rd = self->readers[tgt];
        if (rd.t) {
            if (rd.t->deferred && rd.seq == rd.t->seq + 1) flush_deferred(rd.t);
            if (rd.t->done_seq < rd.seq) HIP_TRY(hipStreamWaitEvent(st, rd.t->ev_done[rd.seq & 1u], 0));
        }
        rd.t = nullptr;
        out_items = self->d_items_alt;                              // (complete after this launch + the exchange: second_copy_usable)
    }
    bpmf::SampleArgs a{};
    a.nwork = (int)(self->to - self->from);
    a.items = out_items; a.col_from = self->from;
    a.LambdaF = d_in; a.Lmu = d_in + (size_t)K * K;
    a.fail = (unsigned long long *)(d_in + (size_t)K * K + K);
    a.mu = d_in + (size_t)K * K + K + 2; a.prop_lambda = self->d_prop; a.diag_only = c->diag_only;
    a.mean_rating = self->mean_rating; a.alpha = alpha; a.iter_plus_1 = (uint32_t)(iter + 1); a.ktrue = c->Kt;
    const int resident = c->num_cu * 4 * bpmf_launch::reduce_waves_per_simd(K);
    const int C = 64 / K;
    const int grid = std::max(1, std::min((a.nwork + C - 1) / C, resident));
    bpmf_launch::reduce_sample(K, grid, st, ev_start, nullptr, a, self->d_prec);
    if (a.nwork <= 0 && ev_start) HIP_TRY(hipEventRecord(ev_start, st));
    if (swap) { std::swap(self->d_items, self->d_items_alt); self->cur_buf ^= 1; }

    bpmf::PrecArgs p{};
    p.t_colptr = self->d_t_colptr; p.t_rowidx = self->d_t_rowidx; p.t_vals = self->d_t_vals; p.order = self->d_t_order;
    p.ncols = other->ncols; p.s_items = self->d_items; p.zero_row = c->d_zero; p.prec = other->d_prec;
    p.mean_rating = other->mean_rating; p.alpha = alpha;
    bpmf_launch::reduce_precompute(K, st, nullptr, ev_stop, p);
    HIP_TRY(hipGetLastError());
    return bpmf_launch::exchange<K, F32>(self, st, -1);
}

// Sampler + exchange of one half-iteration on stream `st`.  Sharded side with parts (bpmf_hip_side_set_overlap):
// part c is sampled on `st`, then exchanged on the side's exchange stream `sx` while part c + 1 is being
// sampled -- what the reference's MPI_ISEND back-end does with its chunks of 100 items sent during compute
// (c++/mpi_isendirecv.h:222-260); `st` continues behind the last exchange.
template <int K, bool F32>
int sample_and_exchange(bpmf_hip_side *self, const bpmf_hip_side *other, int iter, double alpha, double *d_in, hipStream_t st,
                        hipEvent_t ev_start, hipEvent_t ev_stop)
{
    bpmf_hip_ctx *c = self->ctx;
    const bool dist = c->comm != nullptr && !self->bounds.empty();
    if (dist) {
        // test hook: BPMF_HIP_TEST_STALL_RANK="rank:milliseconds[:iteration]" -- that rank goes to sleep before it enqueues this
        // half-iteration (a rank that is descheduled, swapped out or stuck in I/O): its peers' collectives find nobody
        static const char *stall = getenv("BPMF_HIP_TEST_STALL_RANK");
        if (stall && *stall) {
            int r = -1, ms = 0, at = 1;
            if (sscanf(stall, "%d:%d:%d", &r, &ms, &at) >= 2 && r == c->rank && iter == at && ms > 0)
                std::this_thread::sleep_for(std::chrono::milliseconds(ms));
        }
    }
    const bool parts = dist && self->nsub > 1 && self->sx && self->conn_send_ptr.empty() && (int)self->sub_item_off.size() == self->nsub + 1;
    if (self->reduce_on) {
        if constexpr (K == 128) return fail(BPMF_HIP_EINVAL, "the BPMF_REDUCE formulation exists for num_latent <= 64 in fp64");
        else return reduce_half_iteration<K, F32>(self, other, iter, alpha, d_in, st, ev_start, ev_stop);
    }
    // bounded staleness: does part p travel in this half-iteration?  (the side's first half-iteration under a k > 0
    // always exchanges everything, like do_comm of the reference's throttled GASPI back-end, c++/bpmf_gaspi.h:93-99:
    // keyed on the side, not on the iteration number -- a k set in the middle of a chain starts from current replicas too)
    const bool prime = self->stale_k > 0 && !self->stale_primed;
    self->stale_primed = true;
    auto travels = [&](int p) { return self->stale_k <= 0 || prime || ((p + iter) % (self->stale_k + 1)) == 0; };
    if (!parts) {
        int rc = launch_sampler<K, F32>(self, other, iter, alpha, d_in, st, ev_start, ev_stop);
        if (!rc && travels(0)) rc = bpmf_launch::exchange<K, F32>(self, st, -1);
        return rc;
    }
    int rc = 0;
    if (ev_start) HIP_TRY(hipEventRecord(ev_start, st));            // (markers instead of events on the dispatch packets: a part may be empty)
    for (int p = 0; p < self->nsub && !rc; ++p) {
        self->item_off = self->sub_item_off[(size_t)p];
        self->item_n = self->sub_item_off[(size_t)p + 1] - self->item_off;
        if (p == 0) rc = launch_sampler<K, F32>(self, other, iter, alpha, d_in, st, nullptr, nullptr);       // (chooses / swaps the factor copy)
        else rc = sampler_into<K, F32>(self, self->d_items, other, iter, alpha, d_in, st, nullptr, nullptr);
        if (rc) break;
        hipError_t he = hipSuccess;
        if (p == self->nsub - 1 && ev_stop) he = hipEventRecord(ev_stop, st);
        if (he == hipSuccess) he = hipEventRecord(self->sub_ev[p], st);
        if (he == hipSuccess) he = hipStreamWaitEvent(self->sx, self->sub_ev[p], 0);
        if (he != hipSuccess) { rc = fail(BPMF_HIP_ENODEV, std::string("sample_and_exchange: ") + hipGetErrorString(he)); break; }
        if (travels(p)) rc = bpmf_launch::exchange<K, F32>(self, self->sx, p);
    }
    self->item_off = 0; self->item_n = -1;                           // (whatever happened: later launches see the whole item list again)
    if (rc) return rc;
    HIP_TRY(hipEventRecord(self->sx_done, self->sx));
    HIP_TRY(hipStreamWaitEvent(st, self->sx_done, 0));
    return 0;
}

// KK: the instantiated num_latent; FF: the fp32 context (K = 128 only).  Uses the context `c` of the caller.
#define BPMF_DISPATCH_K(K_, ...)                                                     \
    [&]() -> int {                                                                   \
        switch (K_) {                                                                \
        case 8: { constexpr int KK = 8; constexpr bool FF = false; return __VA_ARGS__; }    \
        case 16: { constexpr int KK = 16; constexpr bool FF = false; return __VA_ARGS__; }  \
        case 32: { constexpr int KK = 32; constexpr bool FF = false; return __VA_ARGS__; }  \
        case 64: { constexpr int KK = 64; constexpr bool FF = false; return __VA_ARGS__; }  \
        case 128:                                                                    \
            if (c->dtype == BPMF_HIP_F32) { constexpr int KK = 128; constexpr bool FF = true; return __VA_ARGS__; } \
            else { constexpr int KK = 128; constexpr bool FF = false; return __VA_ARGS__; } \
        default: return fail(BPMF_HIP_EINVAL, "unsupported K");                     \
        }                                                                            \
    }()

// parameter blob of one half-iteration: LambdaF | LambdaF*mu | "no column failed"
// LambdaU (optional): the upper factor the hyper-parameter draw produced, LambdaF = LambdaU^T LambdaU (c++/bpmf.h:101) -- it IS
// chol(LambdaF).matrixU() up to rounding (upper triangular, positive diagonal), so the factorisation below is skipped
void fill_blob(int K, const double *mu, const double *LambdaF, double *h_in, bool with_factor, const double *LambdaU = nullptr)
{
    // rr = hp_LambdaF * hp.mu is the same for every column (c++/sample.cpp:285)
    memcpy(h_in, LambdaF, sizeof(double) * K * K);
    for (int i = 0; i < K; ++i) {
        double s = 0.0;
        for (int j = 0; j < K; ++j) s += LambdaF[(size_t)j * K + i] * mu[j];
        h_in[(size_t)K * K + i] = s;
    }
    const unsigned long long nofail = ~0ull;
    memcpy(&h_in[(size_t)K * K + K], &nofail, sizeof(nofail));
    h_in[(size_t)K * K + K + 1] = 0.0;
    memcpy(&h_in[(size_t)K * K + K + 2], mu, sizeof(double) * K);       // hp.mu itself: the propagated-posterior columns need it
    if (with_factor) {
        // R0 = chol(LambdaF).matrixU(), row-major with zeros below the diagonal: the factor shared by every
        // light column (k_sample_pf).  Not positive definite: NaN, which reaches the samples
        // and is reported as "Cholesky failed" like the reference's own LLT (c++/sample.cpp:306-308).
        double *R = h_in + (size_t)K * K + K + 2 + K;
        bool ok = true;
        if (LambdaU) {
            for (int i = 0; i < K; ++i)
                for (int j = 0; j < K; ++j) R[(size_t)i * K + j] = (j >= i) ? LambdaU[(size_t)j * K + i] : 0.0;     // column-major U(i, j) -> row-major
            for (int i = 0; i < K; ++i) ok = ok && (R[(size_t)i * K + i] > 0.0);
        } else
        for (int i = 0; i < K && ok; ++i) {
            for (int j = 0; j < K; ++j) R[(size_t)i * K + j] = 0.0;
            for (int j = i; j < K; ++j) {
                double v = LambdaF[(size_t)j * K + i];
                for (int k = 0; k < i; ++k) v -= R[(size_t)k * K + i] * R[(size_t)k * K + j];
                if (j == i) { if (!(v > 0.0)) { ok = false; break; } R[(size_t)i * K + i] = std::sqrt(v); }
                else R[(size_t)i * K + j] = v / R[(size_t)i * K + i];
            }
        }
        double *S0t = R + (size_t)K * K, *y0 = S0t + (size_t)K * K;
        if (!ok) {
            for (size_t q = 0; q < 2 * (size_t)K * K + K; ++q) R[q] = std::numeric_limits<double>::quiet_NaN();
        } else {
            // S = R0^-1 (upper), stored transposed (S0t[j*K + i] = S[i][j]); y0 = R0^-T (LambdaF mu): what the
            // columns WITHOUT ratings need (x = S (y0 + z))
            for (size_t q = 0; q < (size_t)K * K; ++q) S0t[q] = 0.0;
            std::vector<double> x(K);
            for (int c = 0; c < K; ++c) {                  // column c of S: R0 x = e_c
                for (int r = 0; r <= c; ++r) x[r] = 0.0;
                x[c] = 1.0;
                for (int j = c; j >= 0; --j) {
                    double v = x[j];
                    for (int m = j + 1; m <= c; ++m) v -= R[(size_t)j * K + m] * x[m];
                    x[j] = v / R[(size_t)j * K + j];
                }
                for (int r = 0; r <= c; ++r) S0t[(size_t)c * K + r] = x[r];
            }
            // Invariant k_sample_pf's final GEMM relies on (kernels_lr.h: the 24 of 64 tile products that lie below the diagonal
            // are not issued): S = R0^-1 has an EXACTLY zero strict lower triangle -- also for a padded num_latent, whose extra
            // dimensions are an identity block.  True by construction (zero fill above, only r <= c written); checked because a
            // later edit of this loop would otherwise fail silently.
            for (int c = 0; c < K && ok; ++c)
                for (int r = c + 1; r < K; ++r)
                    if (S0t[(size_t)c * K + r] != 0.0) { ok = false; break; }
            if (!ok) { for (size_t q = 0; q < 2 * (size_t)K * K + K; ++q) R[q] = std::numeric_limits<double>::quiet_NaN(); return; }
            const double *Lmu = h_in + (size_t)K * K;
            for (int k = 0; k < K; ++k) {                  // R0^T y = Lmu
                double v = Lmu[k];
                for (int i = 0; i < k; ++i) v -= R[(size_t)i * K + k] * y0[i];
                y0[k] = v / R[(size_t)k * K + k];
            }
        }
    }
}

// the same from hyper-parameters of the caller's size Kt: identity precision / zero mean in the extra dimensions
// (their factor rows are zero, their rhs is zero, they draw no normals: x stays exactly 0 there and the leading
// Kt x Kt arithmetic of every column is the unpadded one -- c++/sample.cpp:297-323 with num_latent = Kt)
void fill_blob_ctx(const bpmf_hip_ctx *c, const double *mu, const double *LambdaF, double *h_in, bool with_factor, const double *LambdaU = nullptr)
{
    const int K = c->K, Kt = c->Kt;
    if (Kt == K) { fill_blob(K, mu, LambdaF, h_in, with_factor, LambdaU); return; }
    static thread_local std::vector<double> pm, pf, pu;
    pm.assign((size_t)K, 0.0); pf.resize((size_t)K * K);
    memcpy(pm.data(), mu, sizeof(double) * Kt);
    pad_square(Kt, K, LambdaF, pf.data(), 1.0);
    if (LambdaU) { pu.resize((size_t)K * K); pad_square(Kt, K, LambdaU, pu.data(), 1.0); }
    fill_blob(K, pm.data(), pf.data(), h_in, with_factor, LambdaU ? pu.data() : nullptr);
}

}  // namespace

extern "C" int bpmf_hip_sample_side_launch(bpmf_hip_side *self, const bpmf_hip_side *other, int iter, double alpha,
                                           const double *mu, const double *LambdaF)
{
    if (!self || !other || !mu || !LambdaF) return fail(BPMF_HIP_EINVAL, "sample_side: NULL argument");
    bpmf_hip_ctx *c = self->ctx;
    if (other->ctx != c) return fail(BPMF_HIP_EINVAL, "sample_side: sides belong to different contexts");
    if (other->ncols != self->nrows) return fail(BPMF_HIP_EINVAL, "sample_side: other side has the wrong number of columns");
    if (iter < 0) return fail(BPMF_HIP_EINVAL, "sample_side: iter < 0");
    if (self->pending) return fail(BPMF_HIP_EINVAL, "sample_side_launch: previous launch not finished");
    if (c->comm_dead.load()) return fail(BPMF_HIP_ENODEV, "sample_side: the communicator of this context was aborted (a collective timed out)");
    const int K = c->K;
    HIP_TRY(hipSetDevice(c->device));
    { const int rs = settle_async(self); if (rs) return rs; }
    if (self->saux) { const int rs_ = bounded_stream_sync(self->ctx, self->saux, __func__); if (rs_) return rs_; }
    fill_blob_ctx(c, mu, LambdaF, c->h_in, K == 64 && c->dtype == BPMF_HIP_F64 && self->lr_n > 0);
    bpmf_launch::stage(c->h_in_dev, c->d_in, (int)c->in_words, c->stream);
    if (lf32_words(c)) bpmf_launch::lf32_tiles(c->d_in, reinterpret_cast<float *>(c->d_in + c->in_words), K, c->stream);
    HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    c->last_sampler_done = nullptr;
    int rc = BPMF_DISPATCH_K(K, sample_and_exchange<KK, FF>(self, other, iter, alpha, c->d_in, c->stream, nullptr, nullptr));
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->ev[1], c->stream));
    unsigned *flag = reinterpret_cast<unsigned *>(c->h_out_dev + c->out_words - 1);
    rc = BPMF_DISPATCH_K(K, bpmf_launch::stats<KK, FF>(self, c->stream, c->d_in, c->h_out_dev, flag, ++c->seq, c->d_ticket));
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->ev[2], c->stream));
    // prod | sum | - | fail word land in the pinned result blob; the last wave of k_colstats
    // publishes the sequence number behind them
    HIP_TRY(hipGetLastError());
    self->pending = true;
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_sample_side_finish(bpmf_hip_side *self, double *sum_out, double *prod_out, double *norm_out)
{
    if (!self || !sum_out || !prod_out || !norm_out) return fail(BPMF_HIP_EINVAL, "sample_side_finish: NULL argument");
    if (!self->pending) return fail(BPMF_HIP_EINVAL, "sample_side_finish: nothing launched");
    bpmf_hip_ctx *c = self->ctx;
    const int K = c->K;
    HIP_TRY(hipSetDevice(c->device));
    self->pending = false;
    { const int rcw = wait_host(c); if (rcw) return rcw; }
    { std::string m; if (check_timeout(c->h_out, K, &m)) return fail(BPMF_HIP_ENODEV, m); }
    const int Kt = c->Kt;                                           // (the caller's size; the extra rows / columns of the sums are zero)
    unpad_square(Kt, K, c->h_out, prod_out);
    memcpy(sum_out, c->h_out + (size_t)K * K, sizeof(double) * Kt);
    {   // sum |x|^2 = trace(sum x x^T)
        double nn = 0.0;
        for (int i = 0; i < Kt; ++i) nn += c->h_out[(size_t)i * K + i];
        *norm_out = nn;
    }
    unsigned long long f;
    memcpy(&f, &c->h_out[(size_t)K * K + K + 1], sizeof(f));
    self->timing_valid = false;
    if (f != ~0ull) {
        self->failed_column = (int64_t)f;
        return fail(BPMF_HIP_ECHOL, "Cholesky failed in column " + std::to_string((long long)f));
    }
    self->failed_column = -1;
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_sample_side(bpmf_hip_side *self, const bpmf_hip_side *other, int iter, double alpha,
                                    const double *mu, const double *LambdaF,
                                    double *sum_out, double *prod_out, double *norm_out)
{
    int rc = bpmf_hip_sample_side_launch(self, other, iter, alpha, mu, LambdaF);
    if (rc) return rc;
    return bpmf_hip_sample_side_finish(self, sum_out, prod_out, norm_out);
}

extern "C" int64_t bpmf_hip_failed_column(const bpmf_hip_side *s) { return s ? s->failed_column : -1; }

// aggrMu.col(i) += r; aggrLambda.col(i) += r r^T for this rank's columns (c++/sample.cpp:364-368), on the device
extern "C" int bpmf_hip_side_aggr_add(bpmf_hip_side *s)
{
    if (!s) return fail(BPMF_HIP_EINVAL, "aggr_add: NULL");
    bpmf_hip_ctx *c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    { const int rc = settle_async(s); if (rc) return rc; }
    const size_t K = (size_t)c->Kt, nloc = (size_t)(s->to - s->from);      // (aggrMu / aggrLambda have the caller's size)
    if (!s->d_aggr_mu || !s->d_aggr_lambda) {
        if (s->d_aggr_mu) { (void)hipFree(s->d_aggr_mu); s->d_aggr_mu = nullptr; }
        if (hipMalloc((void **)&s->d_aggr_mu, std::max<size_t>(K * nloc, 1) * sizeof(double)) != hipSuccess ||
            hipMalloc((void **)&s->d_aggr_lambda, std::max<size_t>(K * K * nloc, 1) * sizeof(double)) != hipSuccess) {
            (void)hipGetLastError();
            if (s->d_aggr_mu) { (void)hipFree(s->d_aggr_mu); s->d_aggr_mu = nullptr; }
            s->d_aggr_lambda = nullptr;
            return fail(BPMF_HIP_ENOMEM, "aggr_add: K*K doubles per column do not fit in device memory");
        }
        HIP_TRY(hipMemsetAsync(s->d_aggr_mu, 0, K * nloc * sizeof(double), c->stream));
        HIP_TRY(hipMemsetAsync(s->d_aggr_lambda, 0, K * K * nloc * sizeof(double), c->stream));
    }
    bpmf_launch::aggr_add(s->d_items, c->dtype == BPMF_HIP_F32, c->K, c->Kt, s->from, (int64_t)nloc, s->d_aggr_mu, s->d_aggr_lambda, c->stream);
    HIP_TRY(hipGetLastError());
    c->last_sampler_done = nullptr;
    return BPMF_HIP_OK;
}

// Sys::finalize_mu_lambda (c++/bpmf.cpp:281-295): one K x K inverse per column, batched on the device
extern "C" int bpmf_hip_side_aggr_finalize(bpmf_hip_side *s, int nsamples, double *mu_host, double *lambda_host)
{
    if (!s || !mu_host || !lambda_host) return fail(BPMF_HIP_EINVAL, "aggr_finalize: NULL");
    if (!s->d_aggr_mu || !s->d_aggr_lambda) return fail(BPMF_HIP_EINVAL, "aggr_finalize: nothing was aggregated");
    bpmf_hip_ctx *c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const size_t K = (size_t)c->Kt, nloc = (size_t)(s->to - s->from);
    bpmf_launch::aggr_finalize(c->Kt, nsamples, (int64_t)nloc, s->d_aggr_mu, s->d_aggr_lambda, c->stream);
    { const int rs_ = bounded_stream_sync(c, c->stream, __func__); if (rs_) return rs_; }
    HIP_TRY(hipMemcpy(mu_host, s->d_aggr_mu, K * nloc * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(lambda_host, s->d_aggr_lambda, K * K * nloc * sizeof(double), hipMemcpyDeviceToHost));
    (void)hipFree(s->d_aggr_mu); (void)hipFree(s->d_aggr_lambda);
    s->d_aggr_mu = s->d_aggr_lambda = nullptr;
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_side_last_kernel_ms(bpmf_hip_side *s, float *sample_ms, float *reduce_ms)
{
    if (!s) return fail(BPMF_HIP_EINVAL, "last_kernel_ms: NULL");
    { const int rc = settle_async(s); if (rc) return rc; }      // stateful path: the worker has stored the times
    if (!s->timing_valid) {          // stateless path: the events of the last launch on this context
        bpmf_hip_ctx *c = s->ctx;
        { const int re_ = bounded_event_sync(c, c->ev[2], "last_kernel_ms"); if (re_) return re_; }
        if (hipEventElapsedTime(&s->last_sample_ms, c->ev[0], c->ev[1]) != hipSuccess ||
            hipEventElapsedTime(&s->last_reduce_ms, c->ev[1], c->ev[2]) != hipSuccess) {
            (void)hipGetLastError();                                  // nothing was launched (or timed) yet
            s->last_sample_ms = s->last_reduce_ms = 0.f;
        }
        s->timing_valid = true;
    }
    if (sample_ms) *sample_ms = s->last_sample_ms;
    if (reduce_ms) *reduce_ms = s->last_reduce_ms;
    return BPMF_HIP_OK;
}

// ---------------------------------------------------------------------------
// Stateful form = the virtual the reference's back-ends override: Sys::sample(Sys&)
// (c++/sample.cpp:341-385) including iter++, the host hyper-parameter draw and the cov update.
//
// It is asynchronous inside.  One call enqueues, for half-iteration i of the side,
//     [S0]  k_gate_stage(i)  ->  sampler(i) (+ exchange)        [S1 = the side's own stream]  column statistics(i) -> pinned result blob
// and returns.  The side's host worker thread picks the sums up when they land, forms cov(i), draws
// the hyper-parameters of iteration i+1 (they depend only on cov(i) and on the counter i+1), writes
// them into the side's pinned parameter blob and opens the gate: a word in pinned memory that
// k_gate_stage(i+1) -- usually already queued on S0 behind the other side's sampler -- is polling.
// The gate kernel then copies the blob into device memory and the sampler behind it starts; no
// host thread wake-up, kernel launch or cross-stream event sits between "parameters known" and
// "sampler running".  The caller may run one half-iteration ahead per side (sys_sample(i+1) needs
// collect(i-1) only), so in steady state the GPU never waits for an enqueue and the host work
// (70 us of Normal-Wishart arithmetic per half-iteration) hides behind the other side's sampler.
// Anything that needs host-side state (bpmf_hip_sys_state, destroy, set_items) first drains the
// worker; an error of a half-iteration (Cholesky failed) surfaces at the next such point or at
// the side's next-but-one sys_sample, and the chain is not to be continued after it.
namespace {

int ensure_state(bpmf_hip_side *s)
{
    bpmf_hip_ctx *c = s->ctx;
    const size_t K = (size_t)c->Kt;                                  // (the Sys state -- cov, hp -- has the caller's size)
    if (s->cov.size() == K * K) return 0;
    HIP_TRY(hipHostMalloc((void **)&s->a_h_in, c->in_words * sizeof(double), hipHostMallocMapped));
    HIP_TRY(hipHostMalloc((void **)&s->a_h_out, c->out_words * sizeof(double), hipHostMallocMapped));
    HIP_TRY(hipHostMalloc((void **)&s->a_gate, 64, hipHostMallocMapped));
    HIP_TRY(hipHostGetDevicePointer((void **)&s->a_h_in_dev, s->a_h_in, 0));
    HIP_TRY(hipHostGetDevicePointer((void **)&s->a_h_out_dev, s->a_h_out, 0));
    HIP_TRY(hipHostGetDevicePointer((void **)&s->a_gate_dev, s->a_gate, 0));
    memset(s->a_h_out, 0, c->out_words * sizeof(double));
    memset(s->a_gate, 0, 64);
    HIP_TRY(hipMalloc((void **)&s->a_d_in, (c->in_words + lf32_words(c)) * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&s->a_ticket, 256));                 // [0], [1] statistics tickets (+ spare words)
    HIP_TRY(hipMemset(s->a_ticket, 0, 256));
    HIP_TRY(hipMalloc((void **)&s->a_dflag, 64));
    HIP_TRY(hipMemset(s->a_dflag, 0, 64));
    HIP_TRY(hipMalloc((void **)&s->a_d_red, (c->out_words + 8) * sizeof(double)));
    static const unsigned evflags = env_int("BPMF_HIP_EVENT_FENCE", 0) ? 0u : hipEventDisableSystemFence;
    for (auto &set : s->evs) for (hipEvent_t &e : set) HIP_TRY(hipEventCreateWithFlags(&e, evflags));
    int lo = 0, hi = 0;                                              // numerically lowest = most urgent
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_TRY(hipStreamCreateWithPriority(&s->saux, hipStreamNonBlocking, hi));
    { std::lock_guard<std::mutex> lk(c->launch_mutex); c->sides.push_back(s); }
    s->hp_mu.assign(K, 0.0); s->hp_LambdaU.assign(K * K, 0.0); s->hp_LambdaF.assign(K * K, 0.0);
    s->nx_mu.assign(K, 0.0); s->nx_LambdaU.assign(K * K, 0.0); s->nx_LambdaF.assign(K * K, 0.0);
    s->cov.assign(K * K, 0.0);                                       // cov.setZero(), c++/sample.cpp:188
    return 0;
}

void predraw_main(bpmf_hip_side *s)
{
    auto &P = s->predraw;
    const int K = s->ctx->Kt;
    std::unique_lock<std::mutex> lk(P.m);
    for (;;) {
        P.cv.wait(lk, [&] { return P.stop || P.next <= P.consumed + bpmf_hip_side::Predraw::DEPTH; });
        if (P.stop) return;
        const int it = P.next++;
        auto &sl = P.slot[it % bpmf_hip_side::Predraw::DEPTH];
        lk.unlock();
        sl.au.resize((size_t)K * K); sl.z.resize(K);
        const int rc = bpmf_hyper_draws(K, s->ncols, (uint32_t)it, sl.au.data(), sl.z.data());
        lk.lock();
        sl.iter = rc ? -3 - it : it;                                  // (a failed draw is recomputed inline by the consumer)
        P.cv.notify_all();
    }
}

// the random part of iteration `iter` into rd_au / rd_z (from the ring; iterations are asked for in order)
int predraw_get(bpmf_hip_side *s, int iter)
{
    auto &P = s->predraw;
    const int K = s->ctx->Kt;
    if (P.threads.empty()) {
        const int n = std::max(1, env_int("BPMF_HIP_PREDRAW_THREADS", K >= 128 ? 3 : 1));
        { std::lock_guard<std::mutex> lk(P.m); P.next = iter; P.consumed = iter - 1; }
        for (int i = 0; i < n; ++i) P.threads.emplace_back(predraw_main, s);
    }
    std::unique_lock<std::mutex> lk(P.m);
    auto &sl = P.slot[((iter % bpmf_hip_side::Predraw::DEPTH) + bpmf_hip_side::Predraw::DEPTH) % bpmf_hip_side::Predraw::DEPTH];
    if (iter < P.consumed + 1 || iter >= P.next + bpmf_hip_side::Predraw::DEPTH) {      // out of order (never in a chain): inline
        lk.unlock();
        s->rd_au.resize((size_t)K * K); s->rd_z.resize(K);
        const int rc = bpmf_hyper_draws(K, s->ncols, (uint32_t)iter, s->rd_au.data(), s->rd_z.data());
        if (!rc) s->rd_iter = iter;
        return rc;
    }
    P.cv.wait(lk, [&] { return sl.iter == iter || sl.iter == -3 - iter; });
    const bool ok = sl.iter == iter;
    if (ok) { s->rd_au.swap(sl.au); s->rd_z.swap(sl.z); }
    P.consumed = iter;
    P.cv.notify_all();
    lk.unlock();
    if (!ok) {
        s->rd_au.resize((size_t)K * K); s->rd_z.resize(K);
        const int rc = bpmf_hyper_draws(K, s->ncols, (uint32_t)iter, s->rd_au.data(), s->rd_z.data());
        if (rc) return rc;
    }
    s->rd_iter = iter;
    return 0;
}

void predraw_stop(bpmf_hip_side *s)
{
    auto &P = s->predraw;
    { std::lock_guard<std::mutex> lk(P.m); P.stop = true; }
    P.cv.notify_all();
    for (auto &t : P.threads) if (t.joinable()) t.join();
    P.threads.clear();
}

// hyper-parameters of iteration `iter` from the side's current cov, into (mu, LU, LF); the matching
// parameter blob goes into the side's pinned memory and the gate of that iteration is opened
int draw_and_release(bpmf_hip_side *s, int iter, double *mu, double *LU, double *LF)
{
    bpmf_hip_ctx *c = s->ctx;
    const int K = c->Kt;
    // rng_set_pos(iter); hp.sample(num(), sum = 0, cov)  (c++/sample.cpp:349-350); the random part
    // may have been drawn ahead of time (it does not depend on cov)
    int rc = 0;
    if (s->rd_iter != iter) rc = predraw_get(s, iter);
    if (!rc) rc = bpmf_hyper_finish(K, s->ncols, s->cov.data(), nullptr, s->rd_au.data(), s->rd_z.data(), mu, LU, LF);
    if (!rc) fill_blob_ctx(c, mu, LF, s->a_h_in, c->K == 64 && c->dtype == BPMF_HIP_F64 && s->lr_n > 0, LU);   // (R0, R0^-1: only the low-rank forms read them)
    // the gate is opened even after an error: a sampler may already be queued behind it and must
    // not be left spinning (its results are never looked at: the error is reported first)
    {   // test hook: a host worker that is descheduled for a while (SIGSTOP, debugger, oversubscription)
        static const int stall_ms = env_int("BPMF_HIP_TEST_STALL_WORKER_MS", 0);
        if (stall_ms > 0 && iter > 0) std::this_thread::sleep_for(std::chrono::milliseconds(stall_ms));
    }
    __atomic_store_n(s->a_gate, (unsigned)(iter + 1), __ATOMIC_RELEASE);
    s->gate_iter = iter;
    return rc;
}

// worker side of one half-iteration: wait for the sums, cov, next hyper-parameters, open the gate
void collect(bpmf_hip_side *s, const bpmf_hip_side::Job &job)
{
    bpmf_hip_ctx *c = s->ctx;
    const int K = c->K;
    trace("collect: start", s, job.iter);
    unsigned *flag = reinterpret_cast<unsigned *>(s->a_h_out + c->out_words - 1);
    // while the device is still sampling: the random part of the next draw (gamma / normal stream
    // of WishartUnitChol and MvNormalChol_prec), which needs no result of this half-iteration
    if (s->nx_iter == job.iter) {                                     // the parameters this half-iteration ran with
        s->hp_mu.swap(s->nx_mu); s->hp_LambdaU.swap(s->nx_LambdaU); s->hp_LambdaF.swap(s->nx_LambdaF);
        s->nx_iter = -2;
    }
    trace("collect: draws ready, spinning", s, job.iter);
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (unsigned spins = 0; !seen; ++spins) {
        seen = __atomic_load_n(flag, __ATOMIC_ACQUIRE) == job.seq;
        if (seen || spin_limit_s() <= 0.0) break;
        __builtin_ia32_pause();
        if ((spins & 0xFFFu) == 0xFFFu && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > spin_limit_s()) break;
    }
    trace("collect: sums landed", s, job.iter);
    (void)hipSetDevice(c->device);
    hipEvent_t *ev = s->evs[job.evset];
    int rc = 0;
    std::string msg;
    if (!seen) {                                                      // long kernel or an error: blocking wait
        // (an event, not the stream: our own next gate may be queued on it.)  The statistics may not
        // be enqueued yet: in the fused form they ride in the next sampler launch of the context
        const auto tw = std::chrono::steady_clock::now();
        const double limit = c->comm ? comm_timeout_s() : 60.0;      // (sharded: the pass ends in an all-reduce that needs every peer)
        for (;;) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == job.seq) break;
            hipEvent_t sev = s->stats_ev[job.evset].load(std::memory_order_acquire);
            if (sev) { if (bounded_event_sync(c, sev, "statistics + all-reduce of a half-iteration")) { rc = BPMF_HIP_ENODEV; msg = g_err; } break; }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count() > limit) break;
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        if (!rc && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != job.seq) {
            if (c->comm) { (void)comm_abort(c, "statistics + all-reduce of a half-iteration"); msg = g_err; }
            else msg = "device did not publish its results";
            rc = BPMF_HIP_ENODEV;
        }
    }
    if (!rc) rc = check_timeout(s->a_h_out, K, &msg);     // a bounded in-kernel wait gave up: the sums are not to be used
    if (!rc) {
        const double *prod = s->a_h_out, *sum = s->a_h_out + (size_t)K * K;
        unsigned long long f;
        memcpy(&f, &s->a_h_out[(size_t)K * K + K + 1], sizeof f);
        if (f != ~0ull) {
            s->failed_column = (int64_t)f;
            rc = BPMF_HIP_ECHOL; msg = "Cholesky failed in column " + std::to_string((long long)f);
        } else {
            s->failed_column = -1;
            const int Kt = c->Kt;
            double nn = 0.0;                                          // sum |x|^2 = trace(sum x x^T)  (:381)
            for (int i = 0; i < Kt; ++i) nn += prod[(size_t)i * K + i];
            s->norm = nn;
            { std::lock_guard<std::mutex> lk(s->wm); s->norm_hist[job.iter & 7] = nn; s->collected_iter = job.iter; }
            s->wcv.notify_all();
            if (Kt == K) bpmf_cov_from_sums(K, s->ncols, sum, prod, s->cov.data());   // :383-384
            else {                                                    // padded num_latent: the leading Kt x Kt block of the sums
                static thread_local std::vector<double> pc;
                pc.resize((size_t)Kt * Kt);
                unpad_square(Kt, K, prod, pc.data());
                bpmf_cov_from_sums(Kt, s->ncols, sum, pc.data(), s->cov.data());
            }
        }
    }
    // the next half-iteration of this side: parameters + gate (opened in every case, see above)
    const int rd = draw_and_release(s, job.iter + 1, s->nx_mu.data(), s->nx_LambdaU.data(), s->nx_LambdaF.data());
    trace("collect: gate of the next half-iteration opened", s, job.iter);
    if (!rc && rd) { rc = rd; msg = g_err; }
    if (!rc) s->nx_iter = job.iter + 1;
    {   // kernel times of this launch (its events are complete: the flag is published behind them)
        float a = 0.f, b = 0.f;
        const bool own_stats = s->stats_ev[job.evset].load(std::memory_order_acquire) == ev[2];   // else: inside another launch
        if (job.timed && hipEventSynchronize(own_stats ? ev[2] : ev[1]) == hipSuccess && hipEventElapsedTime(&a, ev[0], ev[1]) == hipSuccess) {
            if (own_stats) (void)hipEventElapsedTime(&b, ev[1], ev[2]);
            s->last_sample_ms = a; s->last_reduce_ms = b; s->timing_valid = true;
            s->tot_sample_ms += a; s->tot_reduce_ms += b; s->n_launches++;
            float g = 0.f;                                         // end of the other side's sampler -> start of this one
            if (job.prev_stop && hipEventElapsedTime(&g, job.prev_stop, ev[0]) == hipSuccess) { s->tot_gap_ms += g; s->n_gap++; }
            else (void)hipGetLastError();
        }
    }
    if (rc && !s->async_rc) { s->async_rc = rc; s->async_msg = msg; }
    trace("collect: done", s, job.iter);
}

void worker_main(bpmf_hip_side *s)
{
    std::unique_lock<std::mutex> lk(s->wm);
    for (;;) {
        s->wcv.wait(lk, [s] { return s->wstop || !s->jobs.empty(); });
        if (s->jobs.empty()) return;                                  // stop requested and nothing left
        const bpmf_hip_side::Job job = s->jobs.front();
        s->jobs.pop_front();
        lk.unlock();
        collect(s, job);
        lk.lock();
        s->in_flight--;
        s->wcv.notify_all();
    }
}

void post_collect(bpmf_hip_side *s, const bpmf_hip_side::Job &job)
{
    std::lock_guard<std::mutex> lk(s->wm);
    if (!s->worker.joinable()) s->worker = std::thread(worker_main, s);
    s->jobs.push_back(job);
    s->in_flight++;
    s->wcv.notify_all();
}

// fused stateful path: the statistics of the newest half-iteration ride in the NEXT sampler launch;
// when somebody needs them and no launch has come, they run as a kernel of their own on the side's stream
// on_main: on the main stream, in order behind P's sampler (end of a run: nothing else is coming on that stream, and a
// kernel on the side's stream would first pay the cross-queue hop -- ~40 us on a queue that has gone idle)
int flush_pending_stats(bpmf_hip_ctx *c, bool on_main)
{
    bpmf_hip_side *P = c->pending_stats;
    if (!P) return 0;
    c->pending_stats = nullptr;
    c->pending_riders = false;
    const int K = c->K;
    HIP_TRY(hipSetDevice(c->device));
    hipEvent_t *ev = P->evs[c->pending_evset];
    hipStream_t sst = on_main ? c->stream : P->saux;
    if (!on_main) HIP_TRY(hipStreamWaitEvent(sst, ev[1], 0));        // (ev[1]: recorded with / behind P's sampler)
    else c->last_sampler_done = nullptr;                              // (the newest thing on S0 is no longer a sampler)
    unsigned *flag = reinterpret_cast<unsigned *>(P->a_h_out_dev + c->out_words - 1);
    const int rc = BPMF_DISPATCH_K(K, bpmf_launch::stats<KK, FF>(P, sst, P->a_d_in, P->a_h_out_dev, flag, c->pending_seq, P->a_ticket));
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ev[2], sst));
    P->stats_ev[c->pending_evset].store(ev[2], std::memory_order_release);
    trace("statistics flushed (no launch to ride in)", P, P->iter);
    return 0;
}

// waits until at most `depth` half-iterations of the side are uncollected; returns a deferred error
int wait_async(bpmf_hip_side *s, int depth)
{
    if (depth == 0 && s->ctx->pending_stats == s) {                   // (main thread: nobody else enqueues)
        const int rc = flush_pending_stats(s->ctx);
        if (rc) return rc;
    }
    {
        std::unique_lock<std::mutex> lk(s->wm);
        s->wcv.wait(lk, [s, depth] { return s->in_flight <= depth; });
        if (!s->async_rc) return 0;
    }
    if (s->ctx->pending_stats == s) (void)flush_pending_stats(s->ctx);   // the chain ends here: no launch will carry them
    {
        std::unique_lock<std::mutex> lk(s->wm);
        s->wcv.wait(lk, [s] { return s->in_flight == 0; });           // an error ends the chain: drain it
    }
    const int rc = s->async_rc;
    g_err = s->async_msg;
    s->async_rc = 0;
    return rc;
}

}  // namespace

static int settle_async(bpmf_hip_side *s) { return wait_async(s, 0); }

namespace {

// the copy of the factors the side's next sampler writes: wait (on `st`) for the evaluation that may still read it
int claim_second_copy(bpmf_hip_side *s, hipStream_t st)
{
    bpmf_hip_side::Reader &rd = s->readers[s->cur_buf ^ 1];
    if (rd.t) {
        if (rd.t->deferred && rd.seq == rd.t->seq + 1) flush_deferred(rd.t);
        if (rd.t->done_seq < rd.seq) HIP_TRY(hipStreamWaitEvent(st, rd.t->ev_done[rd.seq & 1u], 0));
    }
    rd.t = nullptr;
    return 0;
}

}  // namespace

extern "C" int bpmf_hip_sys_sample(bpmf_hip_side *self, bpmf_hip_side *other, double alpha)
{
    if (!self || !other) return fail(BPMF_HIP_EINVAL, "sys_sample: NULL argument");
    bpmf_hip_ctx *c = self->ctx;
    if (other->ctx != c) return fail(BPMF_HIP_EINVAL, "sys_sample: sides belong to different contexts");
    if (other->ncols != self->nrows) return fail(BPMF_HIP_EINVAL, "sys_sample: other side has the wrong number of columns");
    if (self->to - self->from != self->ncols && !(c->comm && !self->bounds.empty()))
        return fail(BPMF_HIP_EINVAL, "sys_sample: the side is a shard: give the context a communicator "
                                     "(bpmf_hip_ctx_comm_init) and the side its ranges (bpmf_hip_side_set_ranges), "
                                     "or use bpmf_hip_sample_side and all-reduce the sums yourself");
    const int K = c->K;
    if (c->comm_dead.load()) return fail(BPMF_HIP_ENODEV, "sys_sample: the communicator of this context was aborted (a collective timed out)");
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_state(self)) || (rc = ensure_state(other))) return rc;
    trace("sys_sample: enter", self, self->iter + 1);
    // one half-iteration of this side may still be uncollected: its worker opens our gate
    if ((rc = wait_async(self, 1))) return rc;
    trace("sys_sample: may enqueue", self, self->iter + 1);
    const int iter = self->iter + 1;                                  // :344
    bool chained;
    { std::lock_guard<std::mutex> lk(self->wm); chained = self->in_flight > 0; }
    if (!chained && self->gate_iter != iter) {
        // first half-iteration (or the chain was broken): draw here, nothing to overlap with
        rc = draw_and_release(self, iter, self->nx_mu.data(), self->nx_LambdaU.data(), self->nx_LambdaF.data());
        if (rc) return rc;
        self->nx_iter = iter;
    }
    self->iter = iter;

    hipStream_t s0 = c->stream, s1 = (c->comm && !c->comm2) ? c->stream : self->saux;   // one stream when there is one communicator only
    const unsigned seq = ++self->a_seq;
    const int evset = (int)(seq & 1u);
    hipEvent_t *ev = self->evs[evset];
    // Fused form (single GPU, K <= 32 in fp64, one item per workgroup): ONE launch on S0 per
    // half-iteration carries the gate + staging of its own parameters (workgroup 0) and the column
    // statistics of the previous launch's side (the next workgroups) -- see FusedArgs in kernels.h.
    // Otherwise: S1 (behind the statistics of the previous half-iteration): gate + staging kernel;
    // S0: sampler, exchange; S1: statistics -> pinned result blob.  (The previous statistics pass
    // has finished reading the columns the sampler overwrites: the gate only opens after its sums
    // were seen.)
    const bool dist = c->comm != nullptr && !self->bounds.empty();
    // (K = 64, slab form without low-rank columns: the same launch format, k_sample1s<64>; only the words the slab
    // form reads are staged -- the R0 / R0^-1 tail of the K = 64 blob belongs to the low-rank forms)
    const size_t stage_words = (K == 64 && self->lr_n == 0) ? (size_t)K * K + K + 2 + K : c->in_words;
    const bool fusable_form = (K <= 32 && self->mode == 1) || (K == 64 && self->lr_n == 0 && self->nsub <= 1);
    const bool fused = s1 != s0 && !dist && stage_words <= 8192 && fusable_form && self->nwork > 0 && !self->reduce_on &&
                       c->dtype == BPMF_HIP_F64 && env_int("BPMF_HIP_FUSED", 1) != 0;
    bpmf::FusedArgs fz{};
    bpmf_hip_side *P = c->pending_stats;
    // statistics waiting for a carrier: they ride here, unless this launch cannot take them, or would
    // overwrite in place the very columns they read (the same side twice in a row without a second copy)
    bool carry = fused && P != nullptr && !c->pending_riders;
    // fp32 path: P's pass as the first workgroups of this side's k_sample_wg2 launch (StatRiders)
    const bool ride_f32 = P != nullptr && c->pending_riders && c->dtype == BPMF_HIP_F32 && self->nwork > 0 && !dist && self->nsub <= 1;
    if (ride_f32) carry = true;
    if (carry && P == self && !second_copy_usable(self)) carry = false;
    if (P && !carry) { if ((rc = flush_pending_stats(c))) return rc; }
    bpmf::StatRiders riders{};
    if (carry && ride_f32) {
        const int nw = 2;                                             // waves per workgroup of k_sample_wg2<128, 2, float>
        const int njobs = P->nstat_waves * (K / 16) * (K / 16 + 1) / 2;
        riders.nblocks = (njobs + nw - 1) / nw;
        riders.items = P->d_items; riders.c0 = P->from; riders.c1 = P->to; riders.nsl = P->nstat_waves;
        riders.partials = P->d_stat_partials;
        riders.fail_in = (const unsigned long long *)(P->a_d_in + (size_t)K * K + K);
        riders.out = P->a_h_out_dev; riders.ticket = P->a_ticket;
        riders.flag = reinterpret_cast<unsigned *>(P->a_h_out_dev + c->out_words - 1); riders.seq = c->pending_seq;
        riders.tmo = tmo_word(P->a_h_out_dev, K); riders.wait_ticks = wait_ticks();
    }
    if (fused) {
        fz.gate_host = self->a_gate_dev; fz.gate_want = (unsigned)(iter + 1); fz.src_host = self->a_h_in_dev;
        fz.dst = self->a_d_in; fz.n = (int)stage_words; fz.dflag = self->a_dflag; fz.dval = seq;
        if (carry && !ride_f32) {
            fz.nstat = P->nstat_waves; fz.st_items = P->d_items; fz.st_c0 = P->from; fz.st_c1 = P->to;
            fz.st_partials = P->d_stat_partials;
            fz.st_fail = (const unsigned long long *)(P->a_d_in + (size_t)K * K + K);
            fz.st_out = P->a_h_out_dev; fz.st_ticket = P->a_ticket;
            fz.st_flag = reinterpret_cast<unsigned *>(P->a_h_out_dev + c->out_words - 1); fz.st_seq = c->pending_seq;
            fz.st_tmo = tmo_word(P->a_h_out_dev, K);
        }
    } else {
        bpmf_launch::gate_stage(c->in_words > 8192 ? 16 : 1, self->a_gate_dev, (unsigned)(iter + 1), self->a_h_in_dev, self->a_d_in, (int)c->in_words,
                                tmo_word(self->a_h_out_dev, K), wait_ticks(), s1);
        if (lf32_words(c)) bpmf_launch::lf32_tiles(self->a_d_in, reinterpret_cast<float *>(self->a_d_in + c->in_words), K, s1);
        if (s1 != s0) {
            HIP_TRY(hipEventRecord(ev[3], s1));
            HIP_TRY(hipStreamWaitEvent(s0, ev[3], 0));
        }
    }
    // an evaluation of the previous iteration that was put off until here: beside the samplers that
    // follow (on S1: in the unfused form behind this gate kernel -- nothing the next sampler needs
    // waits for it -- and ahead of this half-iteration's statistics pass)
    flush_deferred(self->deferred_eval);
    // kernel times come from events around every n-th launch of the side (BPMF_HIP_TIMING_EVERY,
    // default 8; 1 = every launch; 0 = never): the start marker costs a few microseconds on S0
    // A timed launch costs ~8 us (ML-1M shape: every 2nd launch 0.1015 ms per iteration, every 8th 0.0985, every 32nd
    // 0.0975): every n-th launch for the first 64 launches of a side (short records: the 8-step strong-scaling one), every
    // 4 n-th from then on.
    static const int every = env_int("BPMF_HIP_TIMING_EVERY", 8);
    const bool timed = every > 0 && seq % (unsigned)(seq <= 64u ? every : 4 * every) == 0;
    const bool ride = s1 != s0 && self->nwork > 0 && env_int("BPMF_HIP_EXT_EVENTS", 1) != 0;   // events on the sampler's own packet
    if (timed && !ride) HIP_TRY(hipEventRecord(ev[0], s0));
    self->cur_fused = fz;
    self->cur_riders = riders;
    self->cur_gate_flag = fused ? self->a_dflag : nullptr; self->cur_gate_want = seq;
    rc = BPMF_DISPATCH_K(K, sample_and_exchange<KK, FF>(self, other, iter, alpha, self->a_d_in, s0, (ride && timed) ? ev[0] : nullptr,
                                                    ride ? ev[1] : nullptr));
    self->cur_gate_flag = nullptr;
    self->cur_fused = bpmf::FusedArgs{};
    self->cur_riders = bpmf::StatRiders{};
    bpmf_launch::next_flags() = 0;                                    // (a sampler sequence without a kernel leaves it pending)
    if (rc) return rc;
    if (!ride) HIP_TRY(hipEventRecord(ev[1], s0));
    c->last_sampler_done = (env_int("BPMF_HIP_EVAL_MARKER", 0) == 0) ? ev[1] : nullptr;   // an evaluation requested next waits for this: no marker of its own on S0
    if (carry) {                                                      // P's statistics are inside this launch: complete behind ev[1]
        P->stats_ev[c->pending_evset].store(ev[1], std::memory_order_release);
        c->pending_stats = nullptr;
        c->pending_riders = false;
    }
    self->stats_ev[evset].store(nullptr, std::memory_order_release);
    // fp32 path (workgroup-per-item form, single GPU): the pass rides at the head of the next k_sample_wg2 launch of the
    // context -- no stream of its own, no head start to buy with event hops (BPMF_HIP_F32_RIDERS=0: the two kernels on S1).
    // Round 3 measured no gain (the two 30-us gaps go -- rocprofv3 timeline: 9 / 14 us between the samplers -- but the riders,
    // 576 two-wave workgroups that each hold the kernel's 40 KB of LDS, lengthen the launches by ~23 us per iteration, and with
    // the gaps gone the host chain sums -> cov -> 230 us Normal-Wishart finish -> staging became the critical path of one
    // side: 0.806 / 0.864 against 0.810 / 0.833 ms).  Round 4, after the samplers' LDS conflicts were cut: 0.716 / 0.722
    // against 0.730 / 0.731 ms in interleaved runs (0.719 / 0.712 against 0.735 / 0.729 in another session): ON by default.
    // The fp64 form of K = 128 was given the same riders (colstats_f32_rider over doubles) and measured SLOWER, 1.46 / 1.44
    // against 1.386 / 1.380 ms: 288 four-wave workgroups holding 80 KB of LDS each lengthen the two launches by 45 + 70 us,
    // more than the two ~27-us gaps they remove; it keeps its stand-alone pass.
    const int f32_riders = env_int("BPMF_HIP_F32_RIDERS", 1);      // (read per call: the tests flip it)
    const bool riders_next = f32_riders && !fused && !dist && s1 != s0 && c->dtype == BPMF_HIP_F32 && self->nwork > 0 && other->nwork > 0 && self->nsub <= 1;
    if (fused || riders_next) {
        c->pending_stats = self; c->pending_seq = seq; c->pending_evset = evset;     // ride in the next launch
        c->pending_riders = riders_next;
    } else {
        // (fp32 path: the statistics used to take 0.2 ms from the end of the sampler to the sums, on the critical path of
        // the side's host chain: their 256-thread workgroups had to find room beside the NEXT side's sampler, whose
        // 128-thread workgroups refill every slot that frees up.  Now single-wave workgroups without LDS: k_colstats_f32.)
        hipStream_t sst = s1;
        if (sst != s0) HIP_TRY(hipStreamWaitEvent(sst, ev[1], 0));
        // Big side (k_colstats_wg): its 256-thread workgroups only find room beside the partner's sampler if they are
        // dispatched first -- both kernels become ready when this side's sampler ends, and the partner's launch, sitting
        // in the same queue as that sampler, wins by the ~6 us of the cross-queue hop.  S0 therefore waits for a marker
        // S1 passes just ahead of the statistics kernel: the pass (0.1 ms alone) starts a hop ahead of the sampler, keeps
        // its slots, and the side's host chain is done before the partner's sampler is.
        static const int head_start = env_int("BPMF_HIP_STATS_HEADSTART", 1);
        // (fp32 path: 1 152 single-wave tile workgroups, same reasoning: 0.84 -> 0.81 ms.  NOT the fp64 form of K = 128 -- round 4,
        //  interleaved: 1.383 / 1.392 ms without the head start against 1.404 / 1.408 with it: its 22-us pass finds room anyway)
        if (head_start && sst != s0 && !dist && (self->nstat_wg > 0 || (K == 128 && c->dtype == BPMF_HIP_F32))) {
            if (!self->ev_stat_go) HIP_TRY(hipEventCreateWithFlags(&self->ev_stat_go, hipEventDisableTiming));
            HIP_TRY(hipEventRecord(self->ev_stat_go, sst));
            HIP_TRY(hipStreamWaitEvent(s0, self->ev_stat_go, 0));
        }
        unsigned *flag = reinterpret_cast<unsigned *>(self->a_h_out_dev + c->out_words - 1);
        rc = BPMF_DISPATCH_K(K, bpmf_launch::stats<KK, FF>(self, sst, self->a_d_in, self->a_h_out_dev, flag, seq, self->a_ticket));
        if (rc) return rc;
        HIP_TRY(hipEventRecord(ev[2], sst));
        self->stats_ev[evset].store(ev[2], std::memory_order_release);
    }
    HIP_TRY(hipGetLastError());
    self->timing_valid = false;
    self->last_stop = ev[1];
    post_collect(self, {iter, seq, evset, timed, (timed && ride) ? other->last_stop : nullptr});
    trace("sys_sample: enqueued", self, iter);
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_sys_state(const bpmf_hip_side *cs, int *iter, double *norm, double *cov, double *mu,
                                  double *LambdaF, double *LambdaU)
{
    if (!cs) return fail(BPMF_HIP_EINVAL, "sys_state: NULL");
    bpmf_hip_side *s = const_cast<bpmf_hip_side *>(cs);
    { const int rc = settle_async(s); if (rc) return rc; }
    const size_t K = (size_t)s->ctx->Kt;
    if (iter) *iter = s->iter;
    if (norm) *norm = s->norm;
    const bool have = s->cov.size() == K * K;
    if (cov) { if (have) memcpy(cov, s->cov.data(), sizeof(double) * K * K); else memset(cov, 0, sizeof(double) * K * K); }
    if (mu) { if (have) memcpy(mu, s->hp_mu.data(), sizeof(double) * K); else memset(mu, 0, sizeof(double) * K); }
    if (LambdaF) { if (have) memcpy(LambdaF, s->hp_LambdaF.data(), sizeof(double) * K * K); else memset(LambdaF, 0, sizeof(double) * K * K); }
    if (LambdaU) { if (have) memcpy(LambdaU, s->hp_LambdaU.data(), sizeof(double) * K * K); else memset(LambdaU, 0, sizeof(double) * K * K); }
    return BPMF_HIP_OK;
}

// norm (c++/sample.cpp:381) of half-iteration `iter` of the side (one of its last 8), waiting only until THAT half-iteration has
// been collected -- later ones may be in flight: the pipelined loop of the `bpmf` executable prints the line of iteration i - 1
// after it has enqueued iteration i, and must not drain the side for it (bpmf_hip_sys_state does).
extern "C" int bpmf_hip_sys_norm(bpmf_hip_side *s, int iter, double *norm)
{
    if (!s || !norm || iter < 0) return fail(BPMF_HIP_EINVAL, "sys_norm: bad argument");
    if (iter > s->iter) return fail(BPMF_HIP_EINVAL, "sys_norm: that half-iteration has not been enqueued");
    if (s->ctx->pending_stats == s && s->iter == iter) {             // its statistics still wait for a launch to ride in: start them
        HIP_TRY(hipSetDevice(s->ctx->device));
        const int rc = flush_pending_stats(s->ctx);
        if (rc) return rc;
    }
    std::unique_lock<std::mutex> lk(s->wm);
    s->wcv.wait(lk, [s, iter] { return s->collected_iter >= iter || s->async_rc != 0 || s->in_flight == 0; });
    if (s->async_rc) { const int rc = s->async_rc; g_err = s->async_msg; return rc; }      // (left in place: the next sys_sample / sys_state reports it too)
    if (s->collected_iter < iter || iter <= s->collected_iter - 8) return fail(BPMF_HIP_EINVAL, "sys_norm: that half-iteration is not among the last 8 collected");
    *norm = s->norm_hist[iter & 7];
    return BPMF_HIP_OK;
}

// Which kernel(s) the sampler launch of this side is, as the dispatch in launch_impl.h (sampler_into) decides it:
// what a profile of the run shows, for the labels of bench.py's roofline object.
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
    for (int pc = 0; pc < 3; ++pc) out[6 + pc] = s->lr_n > 0 ? s->pf_class[pc + 1] - s->pf_class[pc] : 0;    // (the classes are only in use when the side is split)
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

// ---------------------------------------------------------------------------
// Multi-GPU: one process per GPU, RCCL over xGMI (stands in for the reference's MPI/GASPI
// back-ends: send_item + reduce_sum_cov_norm, c++/mpi_common.h:44-50, c++/mpi_bcast.h:21-30).
extern "C" int bpmf_hip_comm_unique_id(void *id128)
{
    if (!id128) return fail(BPMF_HIP_EINVAL, "comm_unique_id: NULL");
    Rccl *R = rccl();
    if (!R) return fail(BPMF_HIP_ENODEV, "RCCL (librccl.so.1) could not be loaded");
    ncclUniqueId id;
    NCCL_TRY(R->GetUniqueId(&id));
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof id);
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_ctx_comm_init(bpmf_hip_ctx *c, int nranks, int rank, const void *id128)
{
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return fail(BPMF_HIP_EINVAL, "ctx_comm_init: bad argument");
    if (c->comm) return fail(BPMF_HIP_EINVAL, "ctx_comm_init: the context already has a communicator");
    Rccl *R = rccl();
    if (!R) return fail(BPMF_HIP_ENODEV, "RCCL (librccl.so.1) could not be loaded");
    HIP_TRY(hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    NCCL_TRY(R->CommInitRank(&c->comm, nranks, id, rank));
    c->nranks = nranks; c->rank = rank;
    if (R->CommSplit && env_int("BPMF_HIP_COMM_STREAMS", 2) >= 2) {
        // every rank, same colour: a duplicate of the communicator.  Without it (old RCCL, or
        // BPMF_HIP_COMM_STREAMS=1) the statistics pass stays on the main stream.
        if (R->CommSplit(c->comm, 0, rank, &c->comm2, nullptr) != ncclSuccess) c->comm2 = nullptr;
    }
    return BPMF_HIP_OK;
}

// number of ranks of the context's communicator as the communication library itself counts them (ncclCommCount);
// 1 without a communicator
extern "C" int bpmf_hip_ctx_comm_nranks(const bpmf_hip_ctx *c)
{
    if (!c) return 0;
    if (!c->comm) return 1;
    Rccl *R = rccl();
    int n = 0;
    if (R && R->CommCount && R->CommCount(c->comm, &n) == ncclSuccess) return n;
    return c->nranks;
}

extern "C" int bpmf_hip_side_set_ranges(bpmf_hip_side *s, const int64_t *bounds)
{
    if (!s || !bounds) return fail(BPMF_HIP_EINVAL, "side_set_ranges: NULL");
    bpmf_hip_ctx *c = s->ctx;
    if (!c->comm) return fail(BPMF_HIP_EINVAL, "side_set_ranges: the context has no communicator");
    if (bounds[0] != 0 || bounds[c->nranks] != s->ncols || bounds[c->rank] != s->from || bounds[c->rank + 1] != s->to)
        return fail(BPMF_HIP_EINVAL, "side_set_ranges: ranges do not tile the columns or disagree with this rank's slice");
    for (int r = 0; r < c->nranks; ++r)
        if (bounds[r + 1] < bounds[r]) return fail(BPMF_HIP_EINVAL, "side_set_ranges: ranges are not monotone");
    s->bounds.assign(bounds, bounds + c->nranks + 1);
    // parts by default when the exchange is worth hiding: BPMF_HIP_OVERLAP = number of parts (0 / 1: off; unset: 4 parts
    // once a half-iteration moves >= 64 MB of fresh columns into this rank)
    // The decision must be the same on every rank (set_overlap is collective, and the per-part messages of two
    // ranks must pair up): it is taken from `bounds`, which every rank holds -- what the rank with the NARROWEST
    // range receives -- not from this rank's own width.  K = 64 in fp64 keeps the uncut form unless asked: the
    // low-rank / product-form split of a side (build_schedule) exists for the uncut item list only.
    const int want = env_int("BPMF_HIP_OVERLAP", -1);
    const size_t esz = c->dtype == BPMF_HIP_F32 ? 4 : 8;
    int64_t narrowest = s->ncols;
    for (int r = 0; r < c->nranks; ++r) narrowest = std::min(narrowest, bounds[r + 1] - bounds[r]);
    const size_t incoming = (size_t)(s->ncols - narrowest) * (size_t)c->K * esz;
    const size_t threshold = (size_t)std::max(1, env_int("BPMF_HIP_OVERLAP_MIN_KB", 64 << 10)) << 10;    // (the tests lower it)
    const bool auto_ok = !(c->K == 64 && c->dtype == BPMF_HIP_F64);
    const int nsub = want >= 0 ? want : (c->nranks > 1 && auto_ok && incoming >= threshold ? 4 : 1);
    // BPMF_HIP_STALE (`bpmf`, the tests): only when set, and never over a k given through bpmf_hip_side_set_staleness
    if (getenv("BPMF_HIP_STALE") && !s->stale_explicit) {
        const int k = std::max(0, std::min(env_int("BPMF_HIP_STALE", 0), 64));
        if (k > 0 && !s->conn_send_ptr.empty()) return fail(BPMF_HIP_EINVAL, "side_set_ranges: BPMF_HIP_STALE cannot be combined with the connectivity-aware exchange");
        if (k != s->stale_k) s->stale_primed = false;
        s->stale_k = k;
    }
    if (nsub > 1) return bpmf_hip_side_set_overlap(s, nsub);
    return BPMF_HIP_OK;
}

// The BPMF_REDUCE formulation of the reference for a pair of sides (see reduce_half_iteration): storage for the
// precomputed parts (zero, like Sys::init: c++/sample.cpp:192-195) and, per side, the transpose of this rank's block
// of ratings.  on = 0 returns to the gather formulation (the storage is kept).
static int reduce_prepare(bpmf_hip_side *s, const bpmf_hip_side *o)
{
    bpmf_hip_ctx *c = s->ctx;
    const size_t part = (size_t)bpmf_launch::reduce_part_words(c->K);
    if (!s->d_prec) {
        HIP_TRY(hipMalloc((void **)&s->d_prec, std::max<size_t>(1, (size_t)s->ncols * part) * sizeof(double)));
    }
    HIP_TRY(hipMemset(s->d_prec, 0, std::max<size_t>(1, (size_t)s->ncols * part) * sizeof(double)));
    if (s->d_t_colptr) return 0;
    // transpose of the local block: for every column j of the other side, the local columns of this side (global
    // ids, ascending) with a rating in row j
    const int64_t nloc = s->to - s->from, nnz = s->nnz, nr = s->nrows;
    if ((int64_t)s->h_colptr.size() != nloc + 1) return fail(BPMF_HIP_EINVAL, "set_reduce: the side has no host column pointers");
    std::vector<int32_t> ri((size_t)std::max<int64_t>(nnz, 1)); std::vector<double> rv((size_t)std::max<int64_t>(nnz, 1));
    if (nnz > 0) {
        HIP_TRY(hipMemcpy(ri.data(), s->d_rowidx, (size_t)nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(rv.data(), s->d_vals, (size_t)nnz * sizeof(double), hipMemcpyDeviceToHost));
    }
    std::vector<int64_t> tp((size_t)nr + 1, 0);
    for (int64_t q = 0; q < nnz; ++q) tp[(size_t)ri[(size_t)q] + 1]++;
    for (int64_t j = 0; j < nr; ++j) tp[(size_t)j + 1] += tp[(size_t)j];
    std::vector<int64_t> fill(tp.begin(), tp.end() - 1);
    std::vector<int32_t> tr((size_t)std::max<int64_t>(nnz, 1)); std::vector<double> tv((size_t)std::max<int64_t>(nnz, 1));
    for (int64_t cl = 0; cl < nloc; ++cl)
        for (int64_t q = s->h_colptr[(size_t)cl]; q < s->h_colptr[(size_t)cl + 1]; ++q) {
            const int64_t d = fill[(size_t)ri[(size_t)q]]++;
            tr[(size_t)d] = (int32_t)(s->from + cl); tv[(size_t)d] = rv[(size_t)q];
        }
    std::vector<int32_t> order((size_t)nr);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return tp[(size_t)x + 1] - tp[(size_t)x] > tp[(size_t)y + 1] - tp[(size_t)y]; });
    int rc;
    if ((rc = dev_upload(&s->d_t_colptr, tp.data(), tp.size())) || (rc = dev_upload(&s->d_t_rowidx, tr.data(), (size_t)nnz)) ||
        (rc = dev_upload(&s->d_t_vals, tv.data(), (size_t)nnz)) || (rc = dev_upload(&s->d_t_order, order.data(), order.size()))) return rc;
    (void)o;
    return 0;
}

extern "C" int bpmf_hip_sys_set_reduce(bpmf_hip_side *a, bpmf_hip_side *b, int on)
{
    if (!a || !b) return fail(BPMF_HIP_EINVAL, "sys_set_reduce: NULL");
    bpmf_hip_ctx *c = a->ctx;
    if (b->ctx != c || a->ncols != b->nrows || b->ncols != a->nrows) return fail(BPMF_HIP_EINVAL, "sys_set_reduce: the two sides do not belong together");
    if (c->dtype != BPMF_HIP_F64 || bpmf_launch::reduce_part_words(c->K) == 0)
        return fail(BPMF_HIP_EINVAL, "sys_set_reduce: the BPMF_REDUCE formulation exists for fp64, K = 8 .. 64");
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = settle_async(a)) || (rc = settle_async(b))) return rc;
    { const int rs_ = bounded_stream_sync(c, c->stream, __func__); if (rs_) return rs_; }
    if (!on) { a->reduce_on = b->reduce_on = false; return BPMF_HIP_OK; }
    if (!a->conn_send_ptr.empty() || !b->conn_send_ptr.empty())
        return fail(BPMF_HIP_EINVAL, "sys_set_reduce: not together with the connectivity-aware exchange");
    if ((rc = reduce_prepare(a, b)) || (rc = reduce_prepare(b, a))) return rc;
    a->reduce_on = b->reduce_on = true;
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_side_set_overlap(bpmf_hip_side *s, int nparts)
{
    if (!s || nparts < 1 || nparts > 8) return fail(BPMF_HIP_EINVAL, "side_set_overlap: 1..8 parts");
    bpmf_hip_ctx *c = s->ctx;
    if (!c->comm || s->bounds.empty()) return fail(BPMF_HIP_EINVAL, "side_set_overlap: set the communicator and the ranges first");
    COMM_ALIVE_OR_FAIL(c, "side_set_overlap");
    int rc;
    if ((rc = settle_async(s))) return rc;
    HIP_TRY(hipSetDevice(c->device));
    { const int rs_ = bounded_stream_sync(c, c->stream, __func__); if (rs_) return rs_; }
    if (s->saux) { const int rs_ = bounded_stream_sync(s->ctx, s->saux, __func__); if (rs_) return rs_; }
    Rccl *R = rccl();
    if (nparts > 1 && (!R->AllGather || !R->Send || !R->Recv)) nparts = 1;          // (old RCCL: no parts)
    const int64_t nloc = s->to - s->from;
    // this rank's parts: equal work, a column counted as (K^2 / 4 + 64) ratings like in the schedule's cost model
    std::vector<int64_t> mine((size_t)nparts + 1, s->from);
    {
        const double c0 = (double)c->K * c->K / 4.0 + 64.0;
        const double total = (double)s->h_colptr[(size_t)nloc] + c0 * (double)nloc;
        int64_t col = 0;
        for (int p = 1; p < nparts; ++p) {
            const double goal = total * p / nparts;
            while (col < nloc && (double)s->h_colptr[(size_t)col + 1] + c0 * (double)(col + 1) <= goal) ++col;
            mine[(size_t)p] = s->from + col;
        }
        mine[(size_t)nparts] = s->to;
    }
    // ... of every rank: one small all-gather (device buffers; once per side)
    std::vector<int64_t> all((size_t)c->nranks * (nparts + 1));
    if (nparts > 1) {
        int64_t *d = nullptr;
        HIP_TRY(hipMalloc((void **)&d, all.size() * sizeof(int64_t)));
        HIP_TRY(hipMemcpy(d + (size_t)c->rank * (nparts + 1), mine.data(), mine.size() * sizeof(int64_t), hipMemcpyHostToDevice));
        ncclResult_t nr = R->AllGather(d + (size_t)c->rank * (nparts + 1), d, (size_t)nparts + 1, ncclInt64, c->comm, c->stream);
        hipError_t he = hipSuccess;
        if (nr == ncclSuccess && bounded_stream_sync(c, c->stream, "side_set_overlap: all-gather of the parts") != 0) { (void)hipFree(d); return BPMF_HIP_ENODEV; }
        if (nr == ncclSuccess && he == hipSuccess) he = hipMemcpy(all.data(), d, all.size() * sizeof(int64_t), hipMemcpyDeviceToHost);
        (void)hipFree(d);
        if (nr != ncclSuccess) return fail(BPMF_HIP_ENODEV, "side_set_overlap: ncclAllGather failed");
        if (he != hipSuccess) return fail(BPMF_HIP_ENODEV, "side_set_overlap: HIP error");
        for (int r = 0; r < c->nranks; ++r)
            if (all[(size_t)r * (nparts + 1)] != s->bounds[(size_t)r] || all[(size_t)r * (nparts + 1) + nparts] != s->bounds[(size_t)r + 1])
                return fail(BPMF_HIP_EINVAL, "side_set_overlap: the ranks disagree about the ranges or the number of parts");
    }
    if (nparts > 1 && !s->sx) {
        HIP_TRY(hipStreamCreateWithFlags(&s->sx, hipStreamNonBlocking));
        for (hipEvent_t &e : s->sub_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s->sx_done, hipEventDisableTiming));
    }
    s->nsub = nparts;
    s->sub_bounds = nparts > 1 ? all : std::vector<int64_t>();
    free_schedule(s);
    return build_schedule(s, s->h_colptr.data());
}

// Bounded-staleness exchange: the third variant of SURVEY 8 f4.  The reference's GASPI back-end can skip sends at random
// (`send_prob`, c++/bpmf_gaspi.h:91-104: a column then stays as the peer last saw it) and its all-reduce back-end keeps
// blocks up to `slack` iterations old (c++/mpi_allreduce.h:134-175): relaxations for fabrics where waiting is the cost.
// Here, deterministic and rank-invariant: part p of a side (bpmf_hip_side_set_overlap; the whole range when the side is
// uncut) is exchanged only in the half-iterations with (p + iter) % (k + 1) == 0, and always in iteration 0 -- a
// remote copy is at most k half-iterations of that side old, the traffic drops to 1 / (k + 1).  Columns a rank owns
// are always current on that rank; the statistics (sum, cov, norm) are all-reduced exactly as ever.  k = 0: the exact
// chain.  Collective in effect: every rank sets the same k.  bpmf_hip_side_exchange brings every replica up to date
// (before outputs).  The chain is NOT the reference's NO_COMM chain any more: a property-tested relaxation
// (tests/test_gpu_multirank.py), never a default.
extern "C" int bpmf_hip_side_set_staleness(bpmf_hip_side *s, int k)
{
    if (!s || k < 0 || k > 64) return fail(BPMF_HIP_EINVAL, "side_set_staleness: k = 0 .. 64");
    if (k > 0 && !s->conn_send_ptr.empty()) return fail(BPMF_HIP_EINVAL, "side_set_staleness: not together with the connectivity-aware exchange");
    { const int rc = settle_async(s); if (rc) return rc; }
    if (k != s->stale_k) s->stale_primed = false;                      // the next half-iteration exchanges every part
    s->stale_k = k; s->stale_explicit = true;
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_side_set_conn(bpmf_hip_side *s, const int64_t *send_ptr, const int32_t *send_cols,
                                      const int64_t *recv_ptr, const int32_t *recv_cols)
{
    if (!s) return fail(BPMF_HIP_EINVAL, "side_set_conn: NULL");
    bpmf_hip_ctx *c = s->ctx;
    if (!c->comm || s->bounds.empty()) return fail(BPMF_HIP_EINVAL, "side_set_conn: set the communicator and the ranges first");
    COMM_ALIVE_OR_FAIL(c, "side_set_conn");
    if (c->dtype != BPMF_HIP_F64) return fail(BPMF_HIP_EINVAL, "side_set_conn: the packed exchange is fp64 only (the fp32 context uses the all-gather form)");
    if (s->stale_k > 0 && (send_ptr || recv_ptr)) return fail(BPMF_HIP_EINVAL, "side_set_conn: not together with the bounded-staleness exchange");
    int rc;
    if ((rc = settle_async(s))) return rc;
    HIP_TRY(hipSetDevice(c->device));
    { const int rs_ = bounded_stream_sync(c, c->stream, __func__); if (rs_) return rs_; }
    for (void *p : {(void *)s->d_conn_send, (void *)s->d_conn_recv, (void *)s->d_conn_sbuf, (void *)s->d_conn_rbuf})
        if (p) HIP_TRY(hipFree(p));
    s->d_conn_send = s->d_conn_recv = nullptr; s->d_conn_sbuf = s->d_conn_rbuf = nullptr;
    s->conn_send_ptr.clear(); s->conn_recv_ptr.clear();
    if (!send_ptr && !recv_ptr) return BPMF_HIP_OK;                     // back to the all-gather form
    if (!send_ptr || !recv_ptr) return fail(BPMF_HIP_EINVAL, "side_set_conn: both lists or none");
    Rccl *R = rccl();
    if (!R || !R->Send || !R->Recv) return fail(BPMF_HIP_ENODEV, "side_set_conn: this RCCL has no ncclSend / ncclRecv");
    const int n = c->nranks;
    if (send_ptr[0] != 0 || recv_ptr[0] != 0) return fail(BPMF_HIP_EINVAL, "side_set_conn: list offsets must start at 0");
    for (int r = 0; r < n; ++r)
        if (send_ptr[r + 1] < send_ptr[r] || recv_ptr[r + 1] < recv_ptr[r]) return fail(BPMF_HIP_EINVAL, "side_set_conn: list offsets are not monotone");
    const int64_t ns = send_ptr[n], nr = recv_ptr[n];
    if ((ns > 0 && !send_cols) || (nr > 0 && !recv_cols)) return fail(BPMF_HIP_EINVAL, "side_set_conn: NULL column list");
    // what leaves must be this rank's to give, what arrives must land in the sender's range
    for (int64_t i = 0; i < ns; ++i)
        if (send_cols[i] < s->from || send_cols[i] >= s->to) return fail(BPMF_HIP_EINVAL, "side_set_conn: send list names a column outside this rank's range");
    for (int r = 0; r < n; ++r)
        for (int64_t i = recv_ptr[r]; i < recv_ptr[r + 1]; ++i)
            if (recv_cols[i] < s->bounds[(size_t)r] || recv_cols[i] >= s->bounds[(size_t)r + 1])
                return fail(BPMF_HIP_EINVAL, "side_set_conn: receive list names a column outside the sender's range");
    if ((rc = dev_upload<int32_t>(&s->d_conn_send, send_cols, (size_t)std::max<int64_t>(ns, 1)))) return rc;
    if ((rc = dev_upload<int32_t>(&s->d_conn_recv, recv_cols, (size_t)std::max<int64_t>(nr, 1)))) return rc;
    if ((rc = dev_upload<double>(&s->d_conn_sbuf, nullptr, (size_t)std::max<int64_t>(ns, 1) * c->K))) return rc;
    if ((rc = dev_upload<double>(&s->d_conn_rbuf, nullptr, (size_t)std::max<int64_t>(nr, 1) * c->K))) return rc;
    s->conn_send_ptr.assign(send_ptr, send_ptr + n + 1);
    s->conn_recv_ptr.assign(recv_ptr, recv_ptr + n + 1);
    return BPMF_HIP_OK;
}

extern "C" int bpmf_hip_side_exchange(bpmf_hip_side *s)
{
    if (!s) return fail(BPMF_HIP_EINVAL, "side_exchange: NULL");
    bpmf_hip_ctx *c = s->ctx;
    if (!c->comm || s->bounds.empty()) return fail(BPMF_HIP_EINVAL, "side_exchange: set the communicator and the ranges first");
    COMM_ALIVE_OR_FAIL(c, "side_exchange");
    int rc;
    if ((rc = settle_async(s))) return rc;
    HIP_TRY(hipSetDevice(c->device));
    c->last_sampler_done = nullptr;
    rc = BPMF_DISPATCH_K(c->K, (bpmf_launch::exchange<KK, FF>(s, c->stream, -1)));
    if (rc) return rc;
    { const int rs_ = bounded_stream_sync(c, c->stream, __func__); if (rs_) return rs_; }
    return BPMF_HIP_OK;
}

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
namespace {
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
}  // namespace

// An evaluation that was requested while both sides keep two copies of their factors is enqueued
// LATER: on the other side's statistics stream, right behind the gate kernel of that side's next
// half-iteration (bpmf_hip_sys_sample), so that it runs beside the samplers that follow instead of
// between them -- they write the other copies.  (Enqueued at once it would sit in front of that
// gate kernel and hold up the sampler behind it; a stream of its own shares a hardware queue with
// one of the other four, and behind a gate kernel that polls the host everything on that queue
// stalls: 0.13 -> 0.34 ms per iteration.)  Whoever needs it earlier flushes it: predict_finish,
// a sampler about to overwrite a copy it reads, test_get, the destructors.
static void flush_deferred(bpmf_hip_test *t, bool on_main)
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
        std::vector<int64_t> ka((size_t)n), kb((size_t)n);
        std::vector<int32_t> ia((size_t)n), ib((size_t)n);
        for (int64_t q = 0; q < n; ++q) {
            ka[(size_t)q] = (int64_t)t->h_row[(size_t)q] * ncm + t->h_col[(size_t)q];           // (user, movie) of t's entry
            kb[(size_t)q] = (int64_t)twin->h_col[(size_t)q] * ncm + twin->h_row[(size_t)q];     // ... of the twin's
            ia[(size_t)q] = ib[(size_t)q] = (int32_t)q;
        }
        std::sort(ia.begin(), ia.end(), [&](int32_t x, int32_t y) { return ka[(size_t)x] < ka[(size_t)y]; });
        std::sort(ib.begin(), ib.end(), [&](int32_t x, int32_t y) { return kb[(size_t)x] < kb[(size_t)y]; });
        std::vector<int32_t> perm((size_t)n);
        bool ok = true;
        for (int64_t q = 0; q < n && ok; ++q) {
            ok = ka[(size_t)ia[(size_t)q]] == kb[(size_t)ib[(size_t)q]] && (q == 0 || ka[(size_t)ia[(size_t)q]] != ka[(size_t)ia[(size_t)q - 1]]);
            perm[(size_t)ia[(size_t)q]] = ib[(size_t)q];
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
    if (t->cancelled) { t->cancelled = false; return fail(BPMF_HIP_EINVAL, "predict_finish: the side of this test matrix was destroyed before the evaluation ran"); }
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

extern "C" int bpmf_hip_randn_stream(bpmf_hip_ctx *c, uint32_t counter, int n, double *out)
{
    if (!c || !out || n < 0 || n > 128) return fail(BPMF_HIP_EINVAL, "randn_stream: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    double *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, 128 * sizeof(double)));
    bpmf_launch::randn_probe(counter, n, d, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(out, d, n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(BPMF_HIP_ENODEV, std::string("randn_stream: ") + hipGetErrorString(e));
    return BPMF_HIP_OK;
}

