// capi_sample.hip -- sampler launches, the stateless half-iteration, posterior aggregation, the stateful pipeline (bpmf_hip_sys_sample)
// (one of the translation units of the C ABI of include/bpmf_hip.h: see capi_internal.h for the map)
#include "capi_internal.h"

namespace bpmf_capi {


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
    static const unsigned evflags = hipEventDisableSystemFence;
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


int settle_async(bpmf_hip_side *s) { return wait_async(s, 0); }


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
    const bool ride = s1 != s0 && self->nwork > 0;   // events on the sampler's own packet
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
    // an evaluation requested next waits for this: no marker of its own on S0
    c->last_sampler_done = ev[1];
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
        constexpr int head_start = 1;
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
    if (s->collected_iter < iter || iter <= s->collected_iter - 8)
        return fail(BPMF_HIP_EINVAL, "sys_norm: that half-iteration is not among the last 8 collected");
    *norm = s->norm_hist[iter & 7];
    return BPMF_HIP_OK;
}

// Which kernel(s) the sampler launch of this side is, as the dispatch in launch_impl.h (sampler_into) decides it:
// what a profile of the run shows, for the labels of bench.py's roofline object.

}  // namespace bpmf_capi
