// capi_context.hip -- errors, RCCL entry points, host trace, bounded waits, context create / destroy / sync, the host normal stream
// (one of the translation units of the C ABI of include/bpmf_hip.h: see capi_internal.h for the map)
#include "capi_internal.h"

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

namespace bpmf_capi {

thread_local std::string g_err;
extern "C" void bpmf_hip_set_error_(const char *msg) { g_err = msg; }

// host-side timeline for BPMF_HIP_TRACE=1: (time, tag, side) records, printed when the context dies
struct TraceRec { double us; const char *tag; const void *side; int iter; };
static std::vector<TraceRec> g_trace;
static std::mutex g_trace_mutex;
const bool g_trace_on = env_int("BPMF_HIP_TRACE", 0) != 0;
static const std::chrono::steady_clock::time_point g_trace_t0 = std::chrono::steady_clock::now();
void trace(const char *tag, const bpmf_hip_side *s, int iter)
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
        fprintf(stderr, "[bpmf_hip] %12.1f us  side %04x  iter %4d  %s\n", g_trace[i].us, (unsigned)((uintptr_t)g_trace[i].side >> 4) & 0xFFFF,
                g_trace[i].iter, g_trace[i].tag);
    g_trace.clear();
}
static struct TraceAtExit { ~TraceAtExit() { if (g_trace_on) trace_dump(); } } g_trace_at_exit;

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
    // | R0 = chol(LambdaF).matrixU() row-major | (R0^-1)^T | R0^-T LambdaF mu (k_sample_pf)
    if (K == 64 && dtype == BPMF_HIP_F64) c->in_words += 2 * (size_t)K * K + K;
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



}  // namespace bpmf_capi
