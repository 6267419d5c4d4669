// state.h -- host-side state behind the opaque handles of include/bpmf_hip.h, shared by capi_*.hip
// (the C ABI) and the per-K launch units (k8.hip ... k128.hip, kcommon.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is dlopen'ed on first use

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <deque>
#include <limits>
#include <new>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/bpmf_hip.h"
#include "args.h"

extern "C" void bpmf_hip_set_error_(const char *msg);

inline int fail(int code, const std::string &msg)
{
    bpmf_hip_set_error_(msg.c_str());
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(e_ == hipErrorOutOfMemory ? BPMF_HIP_ENOMEM : BPMF_HIP_ENODEV,                 \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                            \
    } while (0)

template <typename T>
inline int dev_upload(T **dst, const T *src, size_t n)
{
    *dst = nullptr;
    if (n == 0) n = 1;
    HIP_TRY(hipMalloc((void **)dst, n * sizeof(T)));
    if (src) HIP_TRY(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

inline int env_int(const char *name, int dflt);
inline unsigned long long wait_ticks();
// how long a host thread spins on a result word before it falls back to a blocking wait on the
// event behind the kernels (BPMF_HIP_SPIN_MS, default 50; 0 = always block: used by the tests)
inline double spin_limit_s()
{
    static const double v = env_int("BPMF_HIP_SPIN_MS", 50) * 1e-3;
    return v;
}

inline int env_int(const char *name, int dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

// bound of every in-kernel wait (gate of the hyper-parameters, staged-parameter word, arrival count of
// the statistics waves), in ticks of the 100 MHz wall clock the kernels read: BPMF_HIP_WAIT_TIMEOUT_MS,
// default 20 s.  A wait that runs into it sets the sticky word `tmo_word` of the result blob and the
// host reports BPMF_HIP_ENODEV "device wait timed out" instead of using the results.
inline unsigned long long wait_ticks()
{
    static const unsigned long long v = (unsigned long long)std::max(1, env_int("BPMF_HIP_WAIT_TIMEOUT_MS", 20000)) * 100000ull;
    return v;
}
static const char *const kTimeoutWhat[5] = {"", "the gate of the hyper-parameters never opened (host worker stalled?)",
                                     "the staged parameters never arrived (gate workgroup not scheduled?)",
                                     "the column statistics never completed (waves not scheduled?)",
                                     "(unused)"};

// RCCL entry points, resolved at run time: single-GPU users never load the library, and inside a
// torch process the already-loaded librccl.so.1 is reused (one communicator runtime per process).
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;     // optional (bpmf_hip_side_set_overlap: the parts of every rank)
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommSplit) CommSplit = nullptr;       // optional (second communicator for the statistics streams)
    decltype(&ncclSend) Send = nullptr;                 // optional (connectivity-aware exchange)
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;             // optional (BPMF_REDUCE formulation: the parts of a side's Gram onto the owners)
    decltype(&ncclCommCount) CommCount = nullptr;       // optional (bpmf_hip_ctx_comm_nranks)
    decltype(&ncclCommAbort) CommAbort = nullptr;       // optional (a collective that never completes: comm_abort in capi_context.hip)
};

Rccl *rccl();      // capi_context.hip

// Every path that would enqueue a collective asks first: after a timed-out collective the communicators were aborted
// (capi_context.hip comm_abort) and their handles must not be used again -- the call fails with BPMF_HIP_ENODEV instead.
#define COMM_ALIVE_OR_FAIL(ctx_, who_)                                                                         \
    do {                                                                                                       \
        if ((ctx_)->comm_dead.load(std::memory_order_acquire))                                                 \
            return fail(BPMF_HIP_ENODEV, std::string(who_) + ": the communicator of this context was aborted (a collective timed out)"); \
    } while (0)

#define NCCL_TRY(expr)                                                                                 \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess)                                                                         \
            return fail(BPMF_HIP_ENODEV, std::string(#expr) + ": " +                                   \
                        (rccl() && rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "RCCL error")); \
    } while (0)


// ncclGroupStart / ncclGroupEnd as a scope: an operation that fails inside a group must not leave the group open
// (every later collective of the thread would be queued instead of issued, and hang)
struct NcclGroup {
    Rccl *R;
    bool open = false;
    explicit NcclGroup(Rccl *r) : R(r) {}
    ncclResult_t start() { const ncclResult_t r = R->GroupStart(); open = r == ncclSuccess; return r; }
    ncclResult_t end() { open = false; return R->GroupEnd(); }
    ~NcclGroup() { if (open) (void)R->GroupEnd(); }
};

struct bpmf_hip_ctx {
    int device = 0;
    // K: the num_latent the kernels are instantiated for -- the leading dimension of EVERY device array and blob (factors,
    // LambdaF, sums).  Kt <= K: the caller's num_latent (c++/bpmf.h:22-24 BPMF_NUMLATENT), the size of everything that crosses
    // the C ABI.  Kt < K: the extra dimensions carry zero factor rows and an identity block of the prior precision, draw no
    // normals and stay exactly zero; RNG stream ids use Kt (c++/sample.cpp:266).
    int K = 0, Kt = 0;
    int dtype = BPMF_HIP_F64;            // arithmetic of the column loop and storage of the factors (BPMF_HIP_F32: K = 128)
    hipStream_t stream = nullptr;        // S0: samplers, exchange, predict
    hipEvent_t last_sampler_done = nullptr;   // stop event of the newest thing on S0 when that is a stateful sampler (+ exchange), else NULL
    // fused stateful path: the side whose newest half-iteration still has its statistics to run (they
    // ride in the next k_sample1 launch; flush_pending_stats launches them alone if none comes)
    struct bpmf_hip_side *pending_stats = nullptr; unsigned pending_seq = 0; int pending_evset = 0;
    // ... or, unfused sides with a stand-alone statistics pass (big sides, the fp32 path): the pass goes onto S0 itself just
    // ahead of the NEXT sampler launch, which is then launched "any order" (no barrier bit): see bpmf_hip_sys_sample
    // ... or, fp32 path: as rider workgroups at the head of the next k_sample_wg2 launch (StatRiders, args.h)
    bool pending_riders = false;
    std::vector<bpmf_hip_side *> sides;  // stateful sides with a statistics stream of their own (for ctx_sync)
    bool own_stream = false;
    int num_cu = 256;
    unsigned ablate = 0;
    unsigned diag_only = 0;              // BPMF_NO_COVARIANCE variant (bpmf_hip_ctx_set_no_covariance)
    // per-call parameter blob: LambdaF[K*K] | Lmu[K] | fail (u64); pinned host copy + device copy
    double *h_in = nullptr, *h_in_dev = nullptr, *d_in = nullptr;
    // result blob in pinned host memory the kernels write directly (zero-copy):
    // prod[K*K] | sum[K] | - | fail (u64) | se | se_avg | flag (u32)
    double *h_out = nullptr, *h_out_dev = nullptr;
    size_t in_words = 0, out_words = 0;
    std::mutex launch_mutex;             // kernel launches come from the caller's thread and from the sides' workers
    // multi-GPU: RCCL communicator (one rank per process / GPU) and a device staging blob for the
    // all-reduced sums: prod[K*K] | sum[K] | - | fail (u64) | se | se_avg | count
    ncclComm_t comm = nullptr;
    // second communicator over the same ranks (ncclCommSplit): the all-reduce of a side's column
    // statistics runs on the side's own stream, beside the other side's sampler and exchange, which
    // two collectives on ONE communicator could not do.  NULL: everything on the main stream.
    ncclComm_t comm2 = nullptr;
    // A collective whose peer never shows up would hold this rank for ever: every host-side wait on a stream that may carry
    // one is bounded (BPMF_HIP_COMM_TIMEOUT_MS, default 60 s; bounded_stream_sync / bounded_event_sync in capi_context.hip); when it
    // runs out both communicators are aborted (ncclCommAbort), the context is dead for collectives and every later call that
    // needs them fails with BPMF_HIP_ENODEV -- the reference's MPI_ERRORS_ARE_FATAL / SUCCESS_OR_DIE (c++/mpi_common.h:16,
    // c++/bpmf_gaspi.h:26-64) as an error code instead of a hang.
    std::atomic<bool> comm_dead{false};
    std::mutex abort_mutex;
    int nranks = 1, rank = 0;
    double *d_red = nullptr;
    unsigned seq = 0;                    // value the next publishing kernel writes behind its results
    unsigned *d_ticket = nullptr;        // arrival counters of k_colstats' waves (stateless path)
    unsigned long long *d_stamps = nullptr;   // BPMF_HIP_STAMPS=1: phase time stamps of two probe work items (printed when the context dies)
    double *d_zero = nullptr;            // K zeros: the row padding slots of a ragged rating group gather from
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
};

struct bpmf_hip_side {
    bpmf_hip_ctx *ctx = nullptr;
    int64_t ncols = 0, nrows = 0, from = 0, to = 0, nnz = 0;
    double mean_rating = 0.0;
    int32_t *d_rowidx = nullptr; double *d_vals = nullptr; bool own_csc = true;
    double *d_items = nullptr; bool own_items = true;
    // Second copy of the factor matrix: a sampler writes the copy that is NOT current and the two swap
    // roles behind it, so that an evaluation of the previous iteration (k_predict on its own stream)
    // can still read the factors the sampler replaces.  Only while the library owns the storage, the
    // raw pointer was never handed out, and this rank's launches rewrite or receive every column.
    double *d_items_alt = nullptr; bool items_exposed = false; int cur_buf = 0;
    struct Reader { struct bpmf_hip_test *t = nullptr; unsigned seq = 0; } readers[2];   // last evaluation that read buffer 0 / 1
    struct bpmf_hip_test *deferred_eval = nullptr;      // evaluation waiting for this side's next gate kernel (flush_deferred)
    double *d_aggr_mu = nullptr, *d_aggr_lambda = nullptr;   // -o: aggrMu (K x nloc) / aggrLambda (K*K x nloc) of the local columns
    double *d_prop = nullptr;            // propagated posterior (-m / -l): K x K prior precision per local column, or NULL
    // schedule
    int nwork = 0, nmulti = 0, nslots = 0, mode = 0;
    // K = 64: columns with <= 16 ratings take the product form (k_sample_pf), the rest the slab form --
    // lr_n light items + hv_nwork others (the full list above stays for per-column priors etc.)
    int lr_n = 0, hv_nwork = 0;
    int64_t pf_ratings2 = 0;
    int64_t pf_ratings = 0;                // ratings of the product-form columns (bpmf_hip_side_schedule_info)
    int pf_class[4] = {0, 0, 0, 0};        // the light items by class: <= 3 ratings, 4..6, 7..16 -- class c is [pf_class[c], pf_class[c+1])
    double *d_pf_q = nullptr;              // product form: R0^-T u_row for every row of the other side (nrows x K), per half-iteration
    int32_t *d_lr_col = nullptr, *d_lr_len = nullptr; int64_t *d_lr_p0 = nullptr;
    int32_t *d_hv_col = nullptr, *d_hv_len = nullptr, *d_hv_mc = nullptr, *d_hv_chunk = nullptr; int64_t *d_hv_p0 = nullptr;
    int32_t *d_wi_col = nullptr, *d_wi_len = nullptr, *d_wi_mc = nullptr, *d_wi_chunk = nullptr;
    int64_t *d_wi_p0 = nullptr;
    int32_t *d_mc_slot0 = nullptr, *d_mc_nch = nullptr;
    unsigned *d_mc_count = nullptr;
    double *d_partials = nullptr;
    int nstat_waves = 0;
    int nstat_wg = 0;                    // > 0: the side's stand-alone statistics pass runs as this many four-wave workgroups (k_colstats_wg: big sides)
    hipEvent_t ev_stat_go = nullptr;     // big side: S0 continues only once S1 has reached the statistics kernel (head start for its workgroups)
    double *d_stat_partials = nullptr;
    std::vector<int64_t> bounds;         // multi-GPU: column range of every rank (nranks + 1 entries)
    // overlap of exchange and sampling (bpmf_hip_side_set_overlap): every rank's range is cut into nsub parts of
    // equal work; part c of every rank is exchanged on the stream `sx` while part c + 1 is being sampled
    int nsub = 1;
    // bounded staleness (bpmf_hip_side_set_staleness): part p of this side travels only in the half-iterations with
    // (p + iter) % (stale_k + 1) == 0 (and in iteration 0); in between the peers sample from the copy they have
    int stale_k = 0;
    bool stale_explicit = false;         // k came through bpmf_hip_side_set_staleness (the environment then does not override it)
    bool stale_primed = false;           // every part has travelled once since k was set (the first half-iteration always exchanges everything)
    std::vector<int64_t> sub_bounds;     // nranks x (nsub + 1): global column bounds of the parts of every rank
    std::vector<int> sub_item_off;       // nsub + 1: the work items of part c are [off[c], off[c+1]) of the item arrays
    int item_off = 0, item_n = -1;       // item window of the launch being enqueued (-1: the whole list)
    hipStream_t sx = nullptr;            // exchange stream
    hipEvent_t sub_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, sx_done = nullptr;
    // BPMF_REDUCE formulation (bpmf_hip_sys_set_reduce, kernels_reduce.h): this side's precomputed Gram parts
    // (ncols x part words, every column -- not only the local ones) and the transpose of this rank's block of
    // ratings (for every column of the OTHER side: the local columns of this side that rate it), longest first
    bool reduce_on = false;
    double *d_prec = nullptr;
    int64_t *d_t_colptr = nullptr; int32_t *d_t_rowidx = nullptr; double *d_t_vals = nullptr; int32_t *d_t_order = nullptr;
    std::vector<int64_t> h_colptr;       // host copy of the local column pointers (schedules are rebuilt when the parts change)
    // connectivity-aware exchange (bpmf_hip_side_set_conn): per peer, the columns of this rank's range the
    // peer reads (send) and the columns of the peer's range this rank reads (recv), as global column ids
    std::vector<int64_t> conn_send_ptr, conn_recv_ptr;
    int32_t *d_conn_send = nullptr, *d_conn_recv = nullptr;
    double *d_conn_sbuf = nullptr, *d_conn_rbuf = nullptr;
    int64_t failed_column = -1;
    bool pending = false;
    float last_sample_ms = 0.f, last_reduce_ms = 0.f;
    bool timing_valid = true;
    // asynchronous (stateful) path: own parameter / result blobs, gate word, events
    double *a_h_in = nullptr, *a_h_in_dev = nullptr, *a_d_in = nullptr;
    double *a_h_out = nullptr, *a_h_out_dev = nullptr;
    bpmf::FusedArgs cur_fused{};         // gate + statistics riders of the k_sample1 launch being enqueued (fused stateful path)
    bpmf::StatRiders cur_riders{};       // fp32 path: statistics riders of the k_sample_wg2 launch being enqueued
    // the event behind which this side's statistics of the job with event set 0 / 1 are complete, once they
    // have been enqueued (inside the next sampler launch, or as a kernel of their own): the collector's blocking wait
    std::atomic<hipEvent_t> stats_ev[2] = {{nullptr}, {nullptr}};
    unsigned *a_dflag = nullptr;         // device word k_gate_stage sets when the parameters are staged (in-kernel gate of the sampler)
    const unsigned *cur_gate_flag = nullptr; unsigned cur_gate_want = 0;   // what the launch being enqueued polls (NULL: ordered by the queue)
    unsigned *a_gate = nullptr, *a_gate_dev = nullptr;   // pinned word the host sets to iter + 1 when a_h_in holds that iteration's parameters
    unsigned *a_ticket = nullptr;                        // arrival counters of this side's k_colstats waves
    double *a_d_red = nullptr;                           // multi-GPU: this side's device blob for the all-reduced sums
    unsigned a_seq = 0;
    hipEvent_t evs[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};   // (start, sampled, stats, staged) of the two half-iterations that may be in flight
    hipStream_t saux = nullptr;          // this side's statistics stream (high priority: its few blocks must not queue behind the other side's sampler)
    // host worker of this side: collects its sums when they land, forms cov, draws its next
    // hyper-parameters and releases the gate of its next sampler, all while the GPU samples the other side
    struct Job { int iter; unsigned seq; int evset; bool timed; hipEvent_t prev_stop; };
    hipEvent_t last_stop = nullptr;      // stop event of this side's newest sampler (diagnostic: boundary to the next launch)
    double tot_gap_ms = 0.0; int64_t n_gap = 0;
    std::thread worker;
    std::mutex wm;
    std::condition_variable wcv;
    std::deque<Job> jobs;
    int in_flight = 0;                   // half-iterations enqueued and not collected yet (at most 2)
    bool wstop = false;
    int async_rc = 0;                    // deferred error of a half-iteration (e.g. Cholesky failed)
    std::string async_msg;
    int gate_iter = -2;                  // iteration whose parameters the gate has been opened for
    double tot_sample_ms = 0.0, tot_reduce_ms = 0.0;
    long long n_launches = 0;
    // state of the reference's Sys (c++/bpmf.h:139,221-226) for bpmf_hip_sys_sample
    int iter = -1;
    double norm = 0.0;
    // norm of the last few collected half-iterations (bpmf_hip_sys_norm: the line of iteration i - 1 is printed after iteration i
    // has been enqueued; asking bpmf_hip_sys_state for it would drain the side's pipeline first).  Guarded by `wm`.
    int collected_iter = -1;
    double norm_hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<double> cov, hp_mu, hp_LambdaU, hp_LambdaF;      // current
    std::vector<double> nx_mu, nx_LambdaU, nx_LambdaF;           // pre-drawn for iteration nx_iter
    int nx_iter = -2;
    std::vector<double> rd_au, rd_z;                             // cov-independent random part, drawn for iteration rd_iter
    int rd_iter = -2;
    // The random part of a hyper-parameter draw (Wishart gammas / normals, K (K - 1) + 2 K numbers from the Philox
    // stream `iter`) depends on nothing but the iteration number and costs more than a whole sampler launch at
    // K >= 64 (K = 128: ~1 ms on the host): helper threads draw a few iterations AHEAD into a ring, the collector
    // thread only picks the finished draw up.
    struct Predraw {
        static constexpr int DEPTH = 8;
        struct Slot { std::vector<double> au, z; int iter = -1; };
        Slot slot[DEPTH];
        std::vector<std::thread> threads;
        std::mutex m;
        std::condition_variable cv;
        int next = 0;                    // first iteration no helper has claimed
        int consumed = -1;               // last iteration handed out
        bool stop = false;
    } predraw;
};

struct bpmf_hip_test {
    bpmf_hip_side *side = nullptr;
    int64_t nnz = 0;
    int32_t *d_tcol = nullptr, *d_trow = nullptr;
    double *d_tval = nullptr, *d_pavg = nullptr, *d_pm2 = nullptr, *d_partial = nullptr;
    int64_t nblocks = 0;
    int wg = 256;                        // threads per workgroup of k_predict (64: single-wave workgroups, small test sets)
    int64_t global_nnz = -1;             // multi-GPU: test ratings over all ranks (all-reduced once)
    double *h_res = nullptr, *h_res_dev = nullptr;       // pinned: se | se_avg | flag
    unsigned *d_ticket = nullptr;                        // arrival counter of k_predict's blocks
    unsigned seq = 0, done_seq = 0;
    bool launched = false;
    hipEvent_t ev_in = nullptr, ev_done[2] = {nullptr, nullptr}, in_ev = nullptr;
    hipStream_t pstream = nullptr;       // where the launch in flight was enqueued (the main stream, or the other side's)
    // requested, not yet enqueued (flush_deferred): the factor copies it reads, captured at the request
    bool deferred = false, cancelled = false; int def_n = 0; struct bpmf_hip_side *def_other = nullptr;
    const void *def_self_items = nullptr, *def_other_items = nullptr;
    // users.predict(movies) of c++/bpmf.cpp:190: the test matrix of the OTHER side (transposed entries), evaluated with
    // the roles swapped whenever this one is (bpmf_hip_test_set_twin); owner: the test matrix this one is the twin of
    struct bpmf_hip_test *twin = nullptr, *owner = nullptr;
    // single GPU, fp64: the twin's entries are this matrix's, transposed -- entry q here is entry twin_perm[q] there, and
    // ONE kernel writes both copies (k_predict's TwinArgs).  NULL: the twin runs as a kernel of its own (sharded, fp32).
    int32_t *d_twin_perm = nullptr;
    std::vector<int32_t> h_col, h_row;   // global column / row of every entry (kept for the matching)
};

// sticky "a device-side wait timed out" word of a result blob (prod | sum | failD | fail | TMO | - | flag)
inline unsigned long long *tmo_word(double *blob, int K) { return reinterpret_cast<unsigned long long *>(blob + (size_t)K * K + K + 2); }
inline int check_timeout(double *h_blob, int K, std::string *msg)
{
    unsigned long long *w = tmo_word(h_blob, K);
    const unsigned long long v = __atomic_load_n(w, __ATOMIC_ACQUIRE);
    if (!v) return 0;
    __atomic_store_n(w, 0ull, __ATOMIC_RELEASE);
    *msg = std::string("device wait timed out: ") + kTimeoutWhat[v < 5 ? v : 0];
    return BPMF_HIP_ENODEV;
}

// doubles in the partial of one chunk of a heavy column: the larger of the two accumulator layouts
// (16x16x4 tiles of k_sample, 4x4x4 blocks of k_sample1 for K <= 32)
template <int K>
inline size_t part_words()
{
    size_t w = (size_t)bpmf::Geo<K>::PART;
    if constexpr (K <= 32) w = std::max(w, (size_t)bpmf::Geo44<K>::PART);
    return w;
}

inline size_t part_words_rt(int K, bool f32)
{
    switch (K) {
    case 8: return part_words<8>();
    case 16: return part_words<16>();
    case 32: return part_words<32>();
    case 64: return part_words<64>();
    case 128: return (size_t)(36 * 256 + 8 * 16) / (f32 ? 2 : 1);       // 36 tiles + rhs in the element type of the factors (GeoW2<128>::PART_FLOATS, GeoS<128>::PART)
    }
    return 0;
}


// doubles behind the parameter blob of a half-iteration that hold LambdaF as fp32 tiles (fp32 path only)
inline size_t lf32_words(const bpmf_hip_ctx *c) { return c->dtype == BPMF_HIP_F32 ? (size_t)(c->K / 16) * (c->K / 16 + 1) / 2 * 256 / 2 : 0; }

// may this side's samplers write the second copy of the factors?  Every column of the new copy must
// be produced by this launch or arrive through the exchange that follows it.
inline bool second_copy_usable(const bpmf_hip_side *s)
{
    if (!s->d_items_alt || !s->own_items || s->items_exposed || s->nwork <= 0) return false;
    // bounded staleness: a part that does not travel must keep the value the peer LAST RECEIVED, and that sits in the
    // current copy only -- with two copies a skipped part would fall back to what the other copy held two
    // half-iterations earlier: the samplers of such a side write in place
    if (s->stale_k > 0) return false;
    const bool dist = s->ctx->comm != nullptr && !s->bounds.empty();
    return dist || (s->from == 0 && s->to == s->ncols);
}
