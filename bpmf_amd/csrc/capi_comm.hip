// capi_comm.hip -- communicator, column ranges, parts, staleness, packed connectivity exchange, BPMF_REDUCE between ranks
// (one of the translation units of the C ABI of include/bpmf_hip.h: see capi_internal.h for the map)
#include "capi_internal.h"

namespace bpmf_capi {

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

extern "C" int bpmf_hip_ctx_comm_streams(const bpmf_hip_ctx *c)
{
    if (!c || !c->comm) return 0;
    return c->comm2 ? 2 : 1;
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
        if (k > 0 && !s->conn_send_ptr.empty())
            return fail(BPMF_HIP_EINVAL, "side_set_ranges: BPMF_HIP_STALE cannot be combined with the connectivity-aware exchange");
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
        if (nr == ncclSuccess && bounded_stream_sync(c, c->stream, "side_set_overlap: all-gather of the parts") != 0) {
            (void)hipFree(d);
            return BPMF_HIP_ENODEV;
        }
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


}  // namespace bpmf_capi
