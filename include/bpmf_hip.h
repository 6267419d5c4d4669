/*
 * bpmf_hip.h -- C ABI of the MI355X-native BPMF Gibbs hot path (libbpmf_hip.so).
 *
 * This is the drop-in boundary for the per-column sampler of ExaScience/bpmf.
 * The reference has no FFI: its boundary is the C++ class `struct Sys`
 * (c++/bpmf.h:112-239) selected at compile time through `#define SYS <Backend>_Sys`
 * (c++/nocomm.h:6, c++/bpmf.cpp:19-39,131-132).  A new back-end header
 * (INTEGRATION.md shows `hip_sys.h`) subclasses Sys and forwards the virtuals
 * below to these entry points; each prototype cites the reference member it
 * replaces.
 *
 * Conventions
 *   - every function returns 0 on success or a negative BPMF_HIP_E* code and
 *     leaves a message for bpmf_hip_last_error() (thread-local);
 *   - matrices are column-major doubles exactly like Eigen's MatrixNNd / the
 *     `items()` map (c++/bpmf.h:56,193-194): a factor matrix is K x N with one
 *     contiguous K-vector per user / item, so `.ddm` dumps stay byte-compatible;
 *   - sparse matrices are CSC with ascending row indices per column (what
 *     Eigen::SparseMatrix<double> holds after setFromTriplets, c++/io.cpp:521),
 *     int64 column pointers (2e9 nnz configs overflow Eigen's int), int32 rows;
 *   - host pointers unless a name ends in `_dev`; the caller keeps ownership of
 *     everything it passes in; handles own their device memory;
 *   - one context per process per GPU, used from one host thread at a time
 *     (the reference calls Sys::sample from the main thread, c++/bpmf.cpp:184-185).
 *   - there is NO CPU fallback: without a usable HIP device every entry point
 *     that touches the device fails with BPMF_HIP_ENODEV.
 */
#ifndef BPMF_HIP_H
#define BPMF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BPMF_HIP_ABI_VERSION 1

#if defined(__GNUC__)
#define BPMF_API __attribute__((visibility("default")))
#else
#define BPMF_API
#endif

enum {
    BPMF_HIP_OK = 0,
    BPMF_HIP_EINVAL = -1,   /* bad argument (unsupported K, NULL, range)            */
    BPMF_HIP_ENODEV = -2,   /* no HIP device / HIP runtime error                     */
    BPMF_HIP_ENOMEM = -3,   /* device or host allocation failed                      */
    BPMF_HIP_ECHOL = -4,    /* "Cholesky failed" (c++/sample.cpp:308); see _failed_column */
    BPMF_HIP_ENUM = -5      /* host-side numerical failure in the hyper-parameter draw */
};

typedef struct bpmf_hip_ctx bpmf_hip_ctx;     /* one GPU + stream + scratch            */
typedef struct bpmf_hip_side bpmf_hip_side;   /* one `Sys`: ratings CSC + factor matrix */
typedef struct bpmf_hip_test bpmf_hip_test;   /* test matrix T with Pavg / Pm2           */

/* message of the last failing call on this thread */
BPMF_API const char *bpmf_hip_last_error(void);
BPMF_API int bpmf_hip_abi_version(void);
/* 1 if a context of num_latent K (BPMF_NUMLATENT, c++/bpmf.h:22-24,53: the reference ships bpmf-8 ... bpmf-128 incl. 10, 20 ... 100,
 * ci/multilatent.sh:5) can be created in fp64: 1 <= K <= 128.  The kernels are instantiated for 8, 16, 32, 64, 128; any other K
 * runs on the next instantiated size (bpmf_hip_kernel_k) with zero factor rows and an identity block of the prior precision in
 * the extra dimensions, which draw no normals and stay exactly zero: the per-column RNG stream id (idx+1)*K*(iter+1) and the
 * number of normals per column are taken from the caller's K (c++/sample.cpp:266,322), the hyper-parameter draw runs at K, and
 * everything that crosses this interface (items, sums, cov, priors, outputs) has the caller's K x ... sizes. */
BPMF_API int bpmf_hip_supports_k(int K);
/* arithmetic of the column loop / storage type of the factors on the device.  The reference is
 * fp64 throughout (c++/bpmf.h:55-58); BPMF_HIP_F32 is the large-K mixed-precision path (K = 128:
 * fp32 factors, Gram, factorisation and solves; fp64 hyper-parameters, statistics, RMSE sums and
 * normal draws).  The host-side interface (items, sums) stays double in both cases. */
#define BPMF_HIP_F64 0
#define BPMF_HIP_F32 1
BPMF_API int bpmf_hip_supports(int K, int dtype);    /* fp64: 1 .. 128; fp32: 65 .. 128 (opt-in, never chosen silently) */
/* the instantiated size a context of (K, dtype) runs on (8, 16, 32, 64 or 128; 0: unsupported) */
BPMF_API int bpmf_hip_kernel_k(int K, int dtype);

/* ---- context --------------------------------------------------------------
 * Replaces Sys::Init / Sys::Finalize (c++/nocomm.h:19-27).  `stream` is a
 * hipStream_t to launch on (e.g. torch's current stream) or NULL to let the
 * context create its own non-blocking stream. */
BPMF_API int bpmf_hip_ctx_create(int device, int K, void *stream, bpmf_hip_ctx **out);      /* fp64 */
BPMF_API int bpmf_hip_ctx_create_ex(int device, int K, int dtype, void *stream, bpmf_hip_ctx **out);
/* run-time form of the reference's BPMF_NO_COVARIANCE build (c++/sample.cpp:300-304): only the
 * diagonal of Lambda* = LambdaF + alpha G enters the factorisation of every column */
BPMF_API int bpmf_hip_ctx_set_no_covariance(bpmf_hip_ctx *ctx, int on);
/* what the context was created with, and the leading dimension of its DEVICE arrays (= bpmf_hip_kernel_k): a factor matrix
 * handed out / bound as a raw device pointer (bpmf_hip_side_items_dev, _bind_items) is ld x N column-major with rows
 * num_latent .. ld-1 zero.  ld == num_latent for 8, 16, 32, 64, 128: byte-compatible with the reference's items(). */
BPMF_API int bpmf_hip_ctx_num_latent(const bpmf_hip_ctx *ctx);
BPMF_API int bpmf_hip_ctx_dtype(const bpmf_hip_ctx *ctx);
BPMF_API int bpmf_hip_ctx_ld(const bpmf_hip_ctx *ctx);
BPMF_API int bpmf_hip_ctx_destroy(bpmf_hip_ctx *ctx);
BPMF_API int bpmf_hip_ctx_sync(bpmf_hip_ctx *ctx);
BPMF_API void *bpmf_hip_ctx_stream(bpmf_hip_ctx *ctx);

/* ---- multi-GPU -----------------------------------------------------------------
 * One process per GPU.  Replaces the reference's MPI/GASPI/ArgoDSM back-ends (send_item of every
 * fresh column + reduce_sum_cov_norm, c++/mpi_common.h:44-50, c++/mpi_bcast.h:21-30) by RCCL:
 * rank 0 calls _comm_unique_id and ships the 128 bytes to the other ranks (torch.distributed,
 * MPI, a file ...); every rank calls _ctx_comm_init.  A side that holds a shard is then given the
 * column range of EVERY rank (_side_set_ranges, nranks+1 bounds, contiguous, tiling [0, ncols));
 * from then on bpmf_hip_sample_side / bpmf_hip_sys_sample additionally broadcast each rank's fresh
 * range in place (all-gather-v over xGMI) and all-reduce sum | prod | norm on the device, so that
 * the values returned (and the cov formed from them) are the GLOBAL ones on every rank, and
 * bpmf_hip_predict returns the all-reduced se / se_avg / count.  The all-gather-v is a mesh of grouped ncclSend /
 * ncclRecv pairs, one pair per peer and xGMI link (BPMF_HIP_EXCHANGE=bcast: one ncclBroadcast per owner instead).  RCCL is loaded on first use
 * (dlopen of librccl.so.1), single-GPU use never touches it. */
BPMF_API int bpmf_hip_comm_unique_id(void *id128);
BPMF_API int bpmf_hip_ctx_comm_init(bpmf_hip_ctx *ctx, int nranks, int rank, const void *id128);
/* ranks of the context's communicator as the communication library itself counts them (ncclCommCount); 1 without one.
 * What `nprocs` is to the reference (Sys::nprocs, c++/mpi_common.h:14-22): reports print it next to the numbers. */
BPMF_API int bpmf_hip_ctx_comm_nranks(const bpmf_hip_ctx *ctx);
/* communicators the context holds: 0 = none (single GPU), 1 = one (exchange and statistics on the main stream), 2 = a second one
 * was split off with ncclCommSplit for the statistics / evaluation streams (the default where the library offers it;
 * BPMF_HIP_COMM_STREAMS=1 keeps one).  The reference's back-ends have one MPI_COMM_WORLD (c++/mpi_common.h:44-50); what
 * the second communicator stands for is their second thread of progress (c++/mpi_isendirecv.h:222-260).  A run reports it
 * so that a silent fall-back to one communicator (ncclCommSplit refused) shows. */
BPMF_API int bpmf_hip_ctx_comm_streams(const bpmf_hip_ctx *ctx);
BPMF_API int bpmf_hip_side_set_ranges(bpmf_hip_side *side, const int64_t *bounds);
/* Overlap of exchange and sampling, the job of the reference's MPI_ISEND back-end (chunks of 100 fresh items are sent
 * while the next ones are sampled, c++/mpi_isendirecv.h:13-14,222-260): every rank's column range is cut into `nparts`
 * (1..8) parts of equal work; part c of all ranks is exchanged on a stream of its own while part c + 1 is being sampled.
 * Collective: every rank calls it with the same nparts after _side_set_ranges (which already picks 4 parts when a
 * half-iteration brings >= 64 MB of fresh columns to the rank with the narrowest range -- a quantity every rank computes
 * from the same bounds; not for K = 64 fp64, whose low-rank column forms need the uncut item list; BPMF_HIP_OVERLAP=n
 * overrides, 1 = off).  Same
 * samples as without parts. */
BPMF_API int bpmf_hip_side_set_overlap(bpmf_hip_side *side, int nparts);

/* Bounded-staleness exchange (third variant; the reference's relaxations: random send throttling of the
 * GASPI back-end, `send_prob`, c++/bpmf_gaspi.h:91-104, and the blocks up to `slack` iterations old of
 * c++/mpi_allreduce.h:134-175).  Part p of the side (the whole range if uncut) travels only in the half-iterations with
 * (p + iter) % (k + 1) == 0 and in iteration 0; in between the peers sample from the copy they have: at most k
 * half-iterations old, 1 / (k + 1) of the traffic.  Own columns and the all-reduced statistics stay exact.  k = 0
 * (default): the exact chain.  Every rank sets the same k (or BPMF_HIP_STALE=k in the environment of all ranks).
 * bpmf_hip_side_exchange brings the replicas up to date.  A relaxation: results differ from the reference's. */
BPMF_API int bpmf_hip_side_set_staleness(bpmf_hip_side *side, int k);

/* The BPMF_REDUCE formulation of the reference (`Sys::preComputeMuLambda`, c++/sample.cpp:234-246; the
 * sampler reading precMu / precLambda, :289-291; `other.preComputeMuLambda(*this)` after a side's columns, :375-377; the
 * per-owner MPI_Reduce of c++/mpi_reduce.h:24-47).  For a pair of sides: after side S has been sampled, the Gram and rhs
 * parts of EVERY column of the other side that come from this rank's columns of S are computed and kept
 * (ncols x ~(K^2/2 + K) doubles per side and rank: the K^2 N memory the reference pays); before a side is sampled the
 * parts of all ranks are summed onto the owners (ncclReduce per owner range) and the column update adds the prior to the
 * sums instead of gathering the other side's factors.  Both sides start from zero parts, as Sys::init leaves them
 * (c++/sample.cpp:192-195) -- the chain is the reference's BPMF_REDUCE chain, which differs from the default one only in
 * the order of the floating-point sums.  fp64, K = 8 .. 64; not together with _side_set_conn.  The fresh columns are
 * still exchanged afterwards for _predict over the whole test set and for the outputs (the reference's predict covers
 * local rows only in this mode and says so: c++/sample.cpp:59-61).  on = 0: back to the gather formulation.
 * `bpmf`: BPMF_REDUCE=1 in the environment. */
BPMF_API int bpmf_hip_sys_set_reduce(bpmf_hip_side *a, bpmf_hip_side *b, int on);

/* Connectivity-aware exchange.  Replaces Sys::update_conn's conn_map and the
 * per-item sends it steers (c++/assign.cpp:204-241; send_item at c++/sample.cpp:370,
 * c++/mpi_isendirecv.h:222-250, c++/bpmf_gaspi.h:140-172: `if (!conn(i, k)) continue`): a fresh column travels only to the ranks whose stored
 * ratings or test entries reference it, instead of to everyone.  After _side_set_ranges, give the
 * side, per peer rank r (offsets *_ptr[nranks+1], CSR style; global column ids, int32):
 *   send_cols[send_ptr[r] .. send_ptr[r+1])  columns of THIS rank's range that rank r reads,
 *   recv_cols[recv_ptr[r] .. recv_ptr[r+1])  columns of rank r's range that THIS rank reads
 * (the two must mirror each other across ranks: r's receive list from q == q's send list to r, same
 * order).  The exchange then packs, does one grouped ncclSend / ncclRecv per peer, and scatters;
 * columns nobody asked for stay stale in the replica (they are never read on this rank).  Both
 * pointers NULL: back to the all-gather form.  bpmf_hip_side_exchange runs the side's exchange on
 * its own (blocking), e.g. to replicate factors written with _side_set_items. */
BPMF_API int bpmf_hip_side_set_conn(bpmf_hip_side *side, const int64_t *send_ptr, const int32_t *send_cols,
                                    const int64_t *recv_ptr, const int32_t *recv_cols);
BPMF_API int bpmf_hip_side_exchange(bpmf_hip_side *side);

/* ---- one side (= one Sys) ---------------------------------------------------
 * Replaces Sys::Sys + alloc_and_init + Sys::init (c++/sample.cpp:112-137,179-226,
 * c++/nocomm.h:29-33).  The factor matrix has `ncols` columns (all of them,
 * replicated on every GPU) and is zero-initialised (items().setZero(), :185).
 * This rank samples columns [col_from, col_to) -- Sys::from()/to(),
 * c++/bpmf.h:170-172 -- and passes the CSC slice of exactly those columns:
 * colptr has col_to-col_from+1 entries starting at 0, rowidx are row ids of the
 * ratings = column ids of the OTHER side (must be < nrows).  mean_rating is
 * M.sum()/M.nonZeros() over the whole matrix (c++/sample.cpp:183).
 * The arrays are copied to the device; the `_dev` variant adopts device arrays
 * the caller keeps alive (e.g. a synthetic matrix generated on the GPU). */
BPMF_API int bpmf_hip_side_create(bpmf_hip_ctx *ctx, int64_t ncols, int64_t nrows, int64_t col_from, int64_t col_to,
                         const int64_t *colptr, const int32_t *rowidx, const double *vals,
                         double mean_rating, bpmf_hip_side **out);
BPMF_API int bpmf_hip_side_create_dev(bpmf_hip_ctx *ctx, int64_t ncols, int64_t nrows, int64_t col_from, int64_t col_to,
                             const int64_t *colptr_host, const int32_t *rowidx_dev, const double *vals_dev,
                             double mean_rating, bpmf_hip_side **out);
/* Propagated-posterior priors (-m / -l of the reference: Sys::add_prop_posterior,
 * c++/sample.cpp:157-174, used at :272-277).  Lambda: K*K doubles per column of the side's slice
 * [from, to) (each a column-major K x K precision, the layout of U-Lambda.ddm / V-Lambda.ddm); it
 * replaces hp.LambdaF in those columns' updates.  mu (K per column) is accepted and, like in the
 * reference, not used (rr = hp_LambdaF * hp.mu, c++/sample.cpp:285).  Lambda = NULL removes them. */
BPMF_API int bpmf_hip_side_set_prop_posterior(bpmf_hip_side *side, const double *mu, const double *Lambda);
BPMF_API int bpmf_hip_side_destroy(bpmf_hip_side *side);

/* items(): device address of the K x ncols factor matrix (c++/bpmf.h:193-194);
 * bind_items makes the side use caller-owned device storage instead (so a
 * torch tensor / RCCL buffer can be exchanged in place; replaces the
 * backend's malloc in alloc_and_init, c++/nocomm.h:31).  Either call pins the
 * factors to ONE address: the side gives up its second copy (see
 * bpmf_hip_predict_launch) and its samplers write in place from then on.
 * The caller states what it allocated: `ld` (rows per column of its storage) must equal
 * bpmf_hip_ctx_ld -- NOT num_latent when that is not one of 8 / 16 / 32 / 64 / 128 -- and `bytes`
 * must cover ld x ncols doubles; otherwise BPMF_HIP_EINVAL and nothing is bound. */
BPMF_API double *bpmf_hip_side_items_dev(bpmf_hip_side *side);
BPMF_API int bpmf_hip_side_bind_items(bpmf_hip_side *side, double *items_dev, int ld, size_t bytes);
/* host <-> device copies of the whole K x ncols matrix (the -v / -o dumps,
 * c++/bpmf.cpp:206-207,234-239) */
BPMF_API int bpmf_hip_side_get_items(bpmf_hip_side *side, double *items_host);
BPMF_API int bpmf_hip_side_set_items(bpmf_hip_side *side, const double *items_host);

/* ---- the hot path ----------------------------------------------------------
 * Replaces the column loop of Sys::sample(Sys&) (c++/sample.cpp:352-384) with
 * Sys::sample(long,Sys&) + computeMuLambda (:248-336) and the Philox/polar
 * draw (c++/mvnormal.cpp:18-47) inside: for every column idx in
 * [col_from,col_to) of `self`, from the current factor of `other`,
 *     Lambda* = LambdaF + alpha * sum_j u_j u_j^T,  b = LambdaF*mu + alpha * sum_j (r_ij - mean) u_j,
 *     x = L^-T (L^-1 b + z),  z ~ N(0,I) from stream (idx+1)*K*(iter+1) mod 2^32,
 * and writes x into self.items[:, idx].  `iter` is the value of Sys::iter after
 * the `iter++` at :344 (0 for the first call).  mu / LambdaF are the output of
 * bpmf_hyper_sample (hp.mu / hp.LambdaF).  On return the three reductions
 * over this rank's columns are on the host (thread_vector combine, :379-381):
 *     sum_out[K] = sum x,  prod_out[K*K] = sum x x^T (col-major),  *norm_out = sum |x|^2.
 * The call blocks until they have arrived.  BPMF_HIP_ECHOL if a pivot is not
 * positive (THROWERROR("Cholesky failed"), :308). */
BPMF_API int bpmf_hip_sample_side(bpmf_hip_side *self, const bpmf_hip_side *other, int iter, double alpha,
                         const double *mu, const double *LambdaF,
                         double *sum_out, double *prod_out, double *norm_out);
/* the same split in two so that an exchange of the fresh columns can overlap
 * the reductions: _launch enqueues the kernels, _finish waits for the partials */
BPMF_API int bpmf_hip_sample_side_launch(bpmf_hip_side *self, const bpmf_hip_side *other, int iter, double alpha,
                                const double *mu, const double *LambdaF);
BPMF_API int bpmf_hip_sample_side_finish(bpmf_hip_side *self, double *sum_out, double *prod_out, double *norm_out);
/* Stateful form, the virtual every reference back-end overrides: Sys::sample(Sys&)
 * (c++/sample.cpp:341-385; e.g. c++/mpi_bcast.h:21-30 wraps it).  Does iter++ (:344),
 * rng_set_pos(iter) + hp.sample(num(), sum = 0, cov) on the host (:349-350), the column loop
 * on the device, and cov = (prod - sum sum^T/N)/(N-1), norm (:379-384).  iter starts at -1
 * (:113), cov at 0 (:188).  Only for a side that owns all its columns (NO_COMM); a shard uses
 * bpmf_hip_sample_side and all-reduces the sums (or gives the context a communicator, see above).
 * The call is asynchronous inside: it enqueues the sampler and the column statistics and returns;
 * a host worker thread of the context collects the sums when they land, forms cov, draws this
 * side's NEXT hyper-parameters (they depend only on that cov and on iter+1) and stages them on the
 * device, while the caller samples the other side.  The only observable differences from a blocking
 * call: bpmf_hip_sys_state / the side's next bpmf_hip_sys_sample wait for that collection, and an
 * error of the half-iteration (BPMF_HIP_ECHOL) is reported by whichever of them comes first. */
BPMF_API int bpmf_hip_sys_sample(bpmf_hip_side *self, bpmf_hip_side *other, double alpha);
/* Sys::iter, Sys::norm, Sys::cov, hp.mu, hp.LambdaF, hp.LambdaU (c++/bpmf.h:86-89,139,222-223);
 * any output pointer may be NULL */
BPMF_API int bpmf_hip_sys_state(const bpmf_hip_side *side, int *iter, double *norm, double *cov, double *mu,
                                double *LambdaF, double *LambdaU);
/* norm (c++/sample.cpp:381: sum of the squared samples) of half-iteration `iter` of the side -- one of its last 8 -- waiting only
 * until that half-iteration's sums have landed; later half-iterations may be in flight (bpmf_hip_sys_state would wait for them).
 * What Sys::print of iteration i - 1 needs (c++/sample.cpp:101-107) when iteration i is already enqueued. */
BPMF_API int bpmf_hip_sys_norm(bpmf_hip_side *side, int iter, double *norm);
/* Posterior aggregation for the -o outputs.  _aggr_add replaces `aggrMu.col(i) += r; aggrLambda.col(i) += r r^T` of
 * Sys::sample(Sys&) (c++/sample.cpp:364-368): call it after a post-burn-in bpmf_hip_sys_sample; the K + K*K doubles
 * per LOCAL column live on the device.  _aggr_finalize replaces Sys::finalize_mu_lambda (c++/bpmf.cpp:281-295):
 * cov = (prod - sum sum^T / n) / (n - 1), Lambda = cov^-1 (batched, one workgroup per column), mu = sum / n; it
 * copies this rank's K x nloc means and K*K x nloc precisions (column-major per column, the layout of U-mu.ddm /
 * U-Lambda.ddm) to the host and frees the device buffers.  A singular covariance (n <= K) gives NaN. */
BPMF_API int bpmf_hip_side_aggr_add(bpmf_hip_side *side);
BPMF_API int bpmf_hip_side_aggr_finalize(bpmf_hip_side *side, int nsamples, double *mu_host, double *lambda_host);
/* global id of the first column whose factorisation failed, or -1 */
BPMF_API int64_t bpmf_hip_failed_column(const bpmf_hip_side *side);

/* ---- prediction / RMSE -------------------------------------------------------
 * Replaces Sys::predict (c++/sample.cpp:48-96).  The test matrix slice covers
 * the same columns [col_from,col_to) as `side`; Pavg = Pm2 = T initially
 * (c++/sample.cpp:123).  n = iter < burnin ? 0 : iter - burnin (:50).  Returns
 * the partial sums of this rank: se = sum (r-pred)^2, se_avg = sum (r-avg)^2,
 * count = number of predictions; rmse = sqrt(se/count) (:93-95). */
BPMF_API int bpmf_hip_test_create(bpmf_hip_side *side, const int64_t *tcolptr, const int32_t *trowidx,
                         const double *tvals, bpmf_hip_test **out);
BPMF_API int bpmf_hip_test_destroy(bpmf_hip_test *test);
BPMF_API int bpmf_hip_predict(bpmf_hip_test *test, const bpmf_hip_side *self, const bpmf_hip_side *other, int n,
                     double *se, double *se_avg, int64_t *count);
/* the same in two halves: _launch requests the evaluation of the factors as they are after the
 * samplers enqueued so far, _finish waits for the two sums.  A caller may enqueue the next
 * iteration in between (one evaluation per test matrix may be outstanding).  While both sides
 * keep two copies of their factors (library-owned storage, raw pointer never requested) those
 * samplers do not wait for the evaluation: they write the other copies, and the kernel is enqueued
 * beside them.  Otherwise it runs in order on the main stream.  Same results either way. */
BPMF_API int bpmf_hip_predict_launch(bpmf_hip_test *test, const bpmf_hip_side *self, const bpmf_hip_side *other, int n);
BPMF_API int bpmf_hip_predict_finish(bpmf_hip_test *test, double *se, double *se_avg, int64_t *count);
/* `users.predict(movies)` of the reference's loop (c++/bpmf.cpp:190; inside its timed region, its results never
 * printed): `twin` is a test matrix created on the OTHER side from the transposed test entries; it is evaluated with
 * the roles of the two factor matrices swapped whenever `test` is (same launch sequence, its own Pavg / Pm2 and sums).
 * Collect its sums with bpmf_hip_predict_finish(twin, ...) after those of `test`.  twin = NULL detaches. */
BPMF_API int bpmf_hip_test_set_twin(bpmf_hip_test *test, bpmf_hip_test *twin);
/* Pavg / Pm2 in the nnz order of the slice passed to _test_create (Pavg.sdm /
 * Pm2.sdm outputs, c++/bpmf.cpp:229-230) */
BPMF_API int bpmf_hip_test_get(bpmf_hip_test *test, double *pavg_host, double *pm2_host);

/* ---- hyper-parameters (host) --------------------------------------------------
 * Replaces rng_set_pos(iter) + HyperParams::sample (c++/sample.cpp:349-350,
 * c++/bpmf.h:98-103) = CondNormalWishart/NormalWishart/WishartChol/
 * WishartUnitChol/MvNormalChol_prec (c++/mvnormal.cpp:56-135) with the fixed
 * prior mu0=0, b0=2, WI=I, df=K.  Runs on the host with libstdc++'s
 * normal/gamma distributions on the Philox stream `counter` (= iter).
 * cov is K x K; Um is sum/N or NULL for the reference's behaviour (its member
 * `sum` is never updated, so it always passes 0).  Outputs: mu[K], LambdaU
 * (upper Cholesky factor, K x K), LambdaF = LambdaU^T LambdaU. */
BPMF_API int bpmf_hyper_sample(int K, int64_t N, const double *cov, const double *Um, uint32_t counter,
                      double *mu, double *LambdaU, double *LambdaF);
/* The same draw in two steps, so that the expensive, cov-independent part can be produced ahead of
 * time: _draws consumes the whole Philox stream `counter` (unit-Wishart factor au[K*K], upper, and
 * the K normals z of MvNormalChol_prec), _finish does the algebra once cov is known.
 * bpmf_hyper_sample(...) == _draws followed by _finish. */
BPMF_API int bpmf_hyper_draws(int K, int64_t N, uint32_t counter, double *au, double *z);
BPMF_API int bpmf_hyper_finish(int K, int64_t N, const double *cov, const double *Um, const double *au, const double *z,
                               double *mu, double *LambdaU, double *LambdaF);
/* cov = (prod - sum sum^T / N) / (N - 1)  (c++/sample.cpp:383-384) */
BPMF_API void bpmf_cov_from_sums(int K, int64_t N, const double *sum, const double *prod, double *cov);
/* the per-column normal stream, for tests: out[i] = i-th randn() after
 * rng_set_pos(counter) (c++/mvnormal.cpp:34-43) */
BPMF_API void bpmf_randn_stream(uint32_t counter, int n, double *out);
/* the same n draws produced by the device sampler (n <= 128) */
BPMF_API int bpmf_hip_randn_stream(bpmf_hip_ctx *ctx, uint32_t counter, int n, double *out);

/* Reporting (the reference's counters.cpp / measure_perf hooks have no numeric equivalent; these serve bench.py):
 * the kernel(s) a sampler launch of this side consists of, by name, as a profile shows them; and the side's static
 * schedule in numbers (16 words, see capi_side.hip: form, work items, chunks, columns per product-form class ...). */
BPMF_API int bpmf_hip_side_kernel_name(const bpmf_hip_side *side, char *buf, int n);
/* LDS / register budget and residency of the kernel(s) of one sampler launch of the side (what `LDS occupancy on the Cholesky`
 * of the north star is computed from): per kernel 4 words in `out` -- static LDS bytes per workgroup, threads per workgroup,
 * workgroups resident per CU as the runtime's occupancy query reports it, VGPRs -- in launch order; `names`: the kernels as the
 * launch sites spell them, ';'-separated.  Returns the number of kernels (<= max_kernels) or a negative error.  Launches nothing. */
BPMF_API int bpmf_hip_side_kernel_resources(bpmf_hip_side *side, int64_t *out, int max_kernels, char *names, int names_len);
BPMF_API int bpmf_hip_side_schedule_info(const bpmf_hip_side *side, int64_t *out16, int n);
/* the side's work items in launch order (what replaces the `#pragma omp parallel for schedule(guided)` over the columns,
 * c++/sample.cpp:353-356): local column, number of ratings, and the ordinal of the item's heavy column (-1: the item is
 * a whole column).  Copies min(n, work items) entries; returns the number of work items through *nitems. */
BPMF_API int bpmf_hip_side_schedule_items(const bpmf_hip_side *side, int32_t *col, int32_t *len, int32_t *heavy, int64_t n, int64_t *nitems);
/* sums of the sampler / statistics kernel times (ms, HIP events on their streams) over all
 * half-iterations of the stateful path collected so far, and their number */
BPMF_API int bpmf_hip_side_kernel_ms_sum(bpmf_hip_side *side, double *sample_ms, double *reduce_ms, int64_t *launches);
/* kernel timing of the last _sample_side on this side, in milliseconds, from
 * HIP events recorded on the context's stream (for bench.py's roofline line) */
BPMF_API int bpmf_hip_side_last_kernel_ms(bpmf_hip_side *side, float *sample_ms, float *reduce_ms);

#ifdef __cplusplus
}
#endif
#endif /* BPMF_HIP_H */
