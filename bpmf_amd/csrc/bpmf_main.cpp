// bpmf_main.cpp -- the `bpmf` executable of this repo: the reference's command line, stdout lines
// and output files (c++/bpmf.cpp:41-260) on top of libbpmf_hip.so.
//
// Host C++ only: it reads the matrices (io.cpp), mirrors them to the device through the C ABI of
// include/bpmf_hip.h and runs main()'s Gibbs loop; every column update happens in the HIP kernels.
// Flags: -n TRAIN -p TEST [-o DIR] [-i N] [-b N] [-a F] [-d K] [-t N] [-f N] [-k] [-r] [-v] [-g N] [--fp32]
// -g N (or BPMF_NGPU=N): N GPUs of this node, the job of `mpirun -np N bpmf` (c++/bpmf.cpp:111-117, c++/mpi_common.h:14-50):
// one host thread + one context per GPU in this process, RCCL id shared in memory, columns of both sides sharded,
// `nprocs: N`, every rank writes bpmf_<rank>.out like the reference's ranks do.
// -m / -l "MU_FILE,LAMBDA_FILE": propagated posteriors of a previous run (c++/bpmf.cpp:134-135).
// K (the reference's compile-time BPMF_NUMLATENT, c++/bpmf.h:22-24; ci/multilatent.sh:5 builds bpmf-8 ... bpmf-128 incl. 10, 20 ...
// 100) is chosen at run time: -d K, else the environment variable BPMF_NUMLATENT, else 32.  Any 1 <= K <= 128, always in the
// reference's fp64 arithmetic.  --fp32 (or BPMF_HIP_F32=1; K > 64 only) opts into the library's mixed-precision large-K path
// and says so on stdout: never chosen silently.
#include <getopt.h>
#include <fcntl.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bpmf_hip.h"
#include "io.h"
#include "../../include/bpmf_io.h"

namespace {

using bpmf::io::Csc;
using bpmf::io::Dense;

double tick()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void usage()
{
    std::cout << "Usage: bpmf -n <MTX> -p <MTX> [-o DIR/] [-i N] [-b N] [-f N] [-a F] [-d K] [-krv] [-t N] [-m MTX,MTX] [-l MTX,MTX] [-g N] [--fp32]\n"
              << "\n"
              << "Parameters:\n"
              << "  -n MTX: training matrix (rows = users, columns = items)\n"
              << "  -p MTX: test matrix\n"
              << "  [-o DIR]: directory for the model and the predictions\n"
              << "  [-i N]: total number of Gibbs iterations (20)\n"
              << "  [-b N]: number of burn-in iterations (5)\n"
              << "  [-f N]: update frequency (accepted, unused)\n"
              << "  [-a F]: noise precision alpha (2.0)\n"
              << "  [-d K]: number of latent dimensions, 1 .. 128, fp64 (32, or $BPMF_NUMLATENT)\n"
              << "  [--fp32]: mixed-precision column update (fp32 factors / Gram / factorisation; K > 64 only; or BPMF_HIP_F32=1)\n"
              << "\n"
              << "  [-l MTX,MTX]: propagated posterior mu and Lambda matrices for U\n"
              << "  [-m MTX,MTX]: propagated posterior mu and Lambda matrices for V\n"
              << "\n"
              << "  [-g N]: shard users and items over N GPUs of this node (RCCL over xGMI; default $BPMF_NGPU or 1)\n"
              << "  [-k]: do not balance the item-to-GPU assignment on work (equal column counts per GPU instead)\n"
              << "  [-r]: redirect stdout to bpmf_<rank>.out (always with more than one GPU)\n"
              << "  [-v]: write every sample (U-<i>.ddm, V-<i>.ddm)\n"
              << "  [-t N]: host threads (accepted; the column loop runs on the GPU)\n"
              << "\n"
              << "Matrix formats (by extension, optionally .gz):\n"
              << "  *.mtx / *.mm: MatrixMarket, sparse (coordinate) or dense (array)\n"
              << "  *.sdm / *.sbm: sparse binary double / pattern\n"
              << "  *.ddm / *.csv: dense binary double / text\n"
              << std::endl;
}

// With -g N > 1 the ranks are threads of this process: a rank that fails (Cholesky failed, device wait timed out, a
// collective that gave up) must not run static destructors and the HIP / RCCL teardown under the other ranks' live
// threads -- they may be blocked inside a collective.  Sys::Abort of the reference's MPI back-ends is MPI_Abort
// (c++/mpi_common.h:28-31): the message, then the process ends at once.
std::atomic<bool> g_rank_threads{false};
std::mutex g_die_mutex;
std::vector<std::string> g_rank_files;                               // bpmf_<rank>.out of every rank (-g N / -r): the failure message is appended to each

[[noreturn]] void die(const std::string &msg)
{
    if (g_rank_threads.load()) {
        {   // _Exit runs no destructors.  Every line a rank writes ends in std::endl, so its log file is current up to the line it
            // is formatting right now; the failure message goes into every rank's log through a descriptor of its own (O_APPEND +
            // write): the other ranks' std::ofstream objects are NOT touched from this thread -- they may be inside operator<<
            // (inserting into / flushing them from here was a data race on iostream state that could lose the very
            // diagnostic it was meant to save).
            std::lock_guard<std::mutex> lk(g_die_mutex);
            std::cerr << "bpmf: " << msg << std::endl;
            const std::string line = "bpmf: " + msg + "\n";
            for (const std::string &f : g_rank_files) {
                const int fd = ::open(f.c_str(), O_WRONLY | O_APPEND);
                if (fd >= 0) { ssize_t w = ::write(fd, line.data(), line.size()); (void)w; ::close(fd); }
            }
        }
        std::_Exit(1);
    }
    std::cerr << "bpmf: " << msg << std::endl;
    exit(1);
}

void check(int rc)
{
    if (rc) die(bpmf_hip_last_error());
}

// Sys::init prints (c++/sample.cpp:203-223)
void print_init(std::ostream &os, const char *name, const Csc &M, int64_t test_nnz, double mean_rating)
{
    const int breakpoint1 = 24, breakpoint2 = 10500;        // c++/bpmf.h:255-256
    int64_t bp1 = 0, bp2 = 0;
    for (int64_t k = 0; k < M.ncols; ++k) {
        const int64_t c = M.colptr[(size_t)k + 1] - M.colptr[(size_t)k];
        if (c > breakpoint1) bp1++;
        if (c > breakpoint2) bp2++;
    }
    os << "mean rating: " << mean_rating << std::endl;
    os << "total number of ratings in train: " << M.nnz() << std::endl;
    os << "total number of ratings in test: " << test_nnz << std::endl;
    os << "average ratings per row: " << (double)M.nnz() / (double)M.ncols << std::endl;
    os << "rows > break_point1: " << 100. * (double)bp1 / (double)M.ncols << std::endl;
    os << "rows > break_point2: " << 100. * (double)bp2 / (double)M.ncols << std::endl;
    os << "num " << name << ": " << M.ncols << std::endl;
}

// Contiguous column ranges of equal work, work = c0 + nnz per column -- the reference's assign() balances the same
// quantity with c0 = 10 (c++/assign.cpp:109-120) and then permutes the columns; contiguous cuts of the original
// order keep the column ids, hence the per-column RNG streams and the samples, independent of the GPU count.
// c0: a column costs about as much as 64 ratings here (10M x 1M shard, K = 32: 463 555 columns cost 1.14 ms more
// than 190 with the same 250 M ratings, a rating 0.038 ns: profiles/r01_shard_10Mx1M.txt); BPMF_ASSIGN_COST overrides.
// balance = false (-k): equal column counts, the reference's `!permute` branch (c++/assign.cpp:60-65).
std::vector<int64_t> column_ranges(const Csc &M, int nparts, bool balance)
{
    std::vector<int64_t> b((size_t)nparts + 1, 0);
    const int64_t n = M.ncols;
    if (!balance) {
        const int64_t per = n / nparts;
        for (int p = 0; p < nparts; ++p) b[(size_t)p] = (int64_t)p * per;
        b[(size_t)nparts] = n;
        return b;
    }
    const double c0 = getenv("BPMF_ASSIGN_COST") ? atof(getenv("BPMF_ASSIGN_COST")) : 64.0;
    const double total = (double)M.nnz() + c0 * (double)n;
    int64_t col = 0;
    for (int p = 1; p < nparts; ++p) {
        const double goal = total * p / nparts;
        while (col < n && (double)M.colptr[(size_t)col + 1] + c0 * (double)(col + 1) <= goal) ++col;
        b[(size_t)p] = col;
    }
    b[(size_t)nparts] = n;
    return b;
}

// B = A with its columns renumbered (column j of B = column col_new2old[j] of A) and its row ids mapped through
// row_old2new, rows ascending inside every column again: Sys::permuteCols (c++/assign.cpp:17-36) for both sides at once
Csc permute(const Csc &A, const std::vector<int64_t> &col_new2old, const std::vector<int64_t> &row_old2new)
{
    Csc B;
    B.nrows = A.nrows; B.ncols = A.ncols;
    B.colptr.assign((size_t)A.ncols + 1, 0);
    B.rowidx.resize(A.rowidx.size()); B.vals.resize(A.vals.size());
    std::vector<std::pair<int32_t, double>> tmp;
    int64_t q = 0;
    for (int64_t j = 0; j < A.ncols; ++j) {
        const int64_t o = col_new2old[(size_t)j];
        tmp.clear();
        for (int64_t p = A.colptr[(size_t)o]; p < A.colptr[(size_t)o + 1]; ++p)
            tmp.emplace_back((int32_t)row_old2new[(size_t)A.rowidx[(size_t)p]], A.vals[(size_t)p]);
        std::sort(tmp.begin(), tmp.end(), [](const std::pair<int32_t, double> &a, const std::pair<int32_t, double> &b) { return a.first < b.first; });
        for (auto &e : tmp) { B.rowidx[(size_t)q] = e.first; B.vals[(size_t)q] = e.second; ++q; }
        B.colptr[(size_t)j + 1] = q;
    }
    return B;
}

std::vector<int64_t> inverse(const std::vector<int64_t> &p)
{
    std::vector<int64_t> r(p.size());
    for (size_t i = 0; i < p.size(); ++i) r[(size_t)p[i]] = (int64_t)i;
    return r;
}

// dense K' x N matrix (column-major, one block of `rows` doubles per column): out column j = in column map[j]
void permute_columns(std::vector<double> &d, int64_t rows, const std::vector<int64_t> &map)
{
    if (d.empty()) return;
    std::vector<double> o(d.size());
    for (size_t j = 0; j < map.size(); ++j) memcpy(&o[j * (size_t)rows], &d[(size_t)map[j] * (size_t)rows], sizeof(double) * (size_t)rows);
    d.swap(o);
}

struct Job {
    // inputs (read-only for the ranks)
    Csc M, Mt, T, Tt;                                                // Tt: the test entries by user (users.predict(movies), c++/bpmf.cpp:190)
    int K = 32, dtype = BPMF_HIP_F64, nsims = 20, burnin = 5, update_freq = 1, nthrds = -1, nranks = 1;
    double alpha = 2.0, mean_m = 0.0, mean_u = 0.0;
    bool verbose = false, redirect = false, sharded = false;
    std::string odirname;
    Dense prop_m_mu, prop_m_lambda, prop_u_mu, prop_u_lambda;       // -m / -l (empty: none)
    std::vector<int64_t> bm, bu;                                     // column ranges of the ranks
    std::vector<int64_t> perm_m, perm_u;                             // greedy assignment: new column id -> original (empty: none)
    char rccl_id[128];
    std::vector<int> devices;                                        // BPMF_HIP_DEVICES: device of rank r (default: device r)
    // results
    std::vector<double> pavg, pm2;                                   // test-set order of T; every rank fills its slice
    std::vector<double> u_mu, u_lambda, m_mu, m_lambda;              // -o: K x N means, K*K x N precisions; every rank fills its columns
    double elapsed = 0.0, rmse_avg = NAN;
    int64_t num_predict = 0;
    long double average_items_sec = 0, average_ratings_sec = 0;
    std::vector<std::string> errors;                                 // per rank
};

// one rank = one GPU: Sys::Sys + init (its shard), the Gibbs loop of c++/bpmf.cpp:180-253
void rank_main(Job &J, int rank, std::ostream &os)
{
    const int K = J.K;
    const int64_t nmovies = J.M.ncols, nusers = J.M.nrows;
    const bool aggregate = !J.odirname.empty();
    bpmf_hip_ctx *ctx = nullptr;
    check(bpmf_hip_ctx_create_ex(J.devices.empty() ? rank : J.devices[(size_t)rank % J.devices.size()], K, J.dtype, nullptr, &ctx));
    if (J.sharded) check(bpmf_hip_ctx_comm_init(ctx, J.nranks, rank, J.rccl_id));
    const int64_t m0 = J.bm[(size_t)rank], m1 = J.bm[(size_t)rank + 1], u0 = J.bu[(size_t)rank], u1 = J.bu[(size_t)rank + 1];
    auto slice_ptr = [](const Csc &A, int64_t c0, int64_t c1) {       // colptr of the columns [c0, c1), rebased to 0
        std::vector<int64_t> cp((size_t)(c1 - c0) + 1);
        for (int64_t c = c0; c <= c1; ++c) cp[(size_t)(c - c0)] = A.colptr[(size_t)c] - A.colptr[(size_t)c0];
        return cp;
    };
    bpmf_hip_side *movies = nullptr, *users = nullptr;
    bpmf_hip_test *test = nullptr, *test_u = nullptr;
    {
        const std::vector<int64_t> cp = slice_ptr(J.M, m0, m1);
        const size_t off = (size_t)J.M.colptr[(size_t)m0];
        check(bpmf_hip_side_create(ctx, nmovies, nusers, m0, m1, cp.data(), J.M.rowidx.data() + off, J.M.vals.data() + off, J.mean_m, &movies));
    }
    print_init(os, "movs", J.M, J.T.nnz(), J.mean_m);
    {
        const std::vector<int64_t> cp = slice_ptr(J.Mt, u0, u1);
        const size_t off = (size_t)J.Mt.colptr[(size_t)u0];
        check(bpmf_hip_side_create(ctx, nusers, nmovies, u0, u1, cp.data(), J.Mt.rowidx.data() + off, J.Mt.vals.data() + off, J.mean_u, &users));
    }
    if (J.sharded) {
        check(bpmf_hip_side_set_ranges(movies, J.bm.data()));
        check(bpmf_hip_side_set_ranges(users, J.bu.data()));
    }
    // Sys::add_prop_posterior (c++/sample.cpp:157-174): the K*K x N precisions of this rank's columns
    if (!J.prop_m_lambda.data.empty()) {
        check(bpmf_hip_side_set_prop_posterior(movies, J.prop_m_mu.data.data() + (size_t)K * m0, J.prop_m_lambda.data.data() + (size_t)K * K * m0));
        os << "with propagated posterior" << std::endl;               // c++/sample.cpp:221-222
    }
    print_init(os, "users", J.Mt, J.T.nnz(), J.mean_u);
    if (!J.prop_u_lambda.data.empty()) {
        check(bpmf_hip_side_set_prop_posterior(users, J.prop_u_mu.data.data() + (size_t)K * u0, J.prop_u_lambda.data.data() + (size_t)K * K * u0));
        os << "with propagated posterior" << std::endl;
    }
    // the reference's BPMF_REDUCE build (a compile-time variant there, c++/bpmf.h:30-42; a run-time switch here)
    if (const char *e = getenv("BPMF_REDUCE")) {
        if (atoi(e) != 0) check(bpmf_hip_sys_set_reduce(movies, users, 1));
    }
    const size_t toff = (size_t)J.T.colptr[(size_t)m0];
    {
        const std::vector<int64_t> cp = slice_ptr(J.T, m0, m1);
        check(bpmf_hip_test_create(movies, cp.data(), J.T.rowidx.data() + toff, J.T.vals.data() + toff, &test));
    }
    {   // users.predict(movies) of the reference's loop (c++/bpmf.cpp:190): the users' copy of the test entries (T = Pavg^T of
        // the second Sys, c++/sample.cpp:132-137) is evaluated with every movies.predict(users); nothing prints its results
        const std::vector<int64_t> cp = slice_ptr(J.Tt, u0, u1);
        const size_t off = (size_t)J.Tt.colptr[(size_t)u0];
        check(bpmf_hip_test_create(users, cp.data(), J.Tt.rowidx.data() + off, J.Tt.vals.data() + off, &test_u));
        check(bpmf_hip_test_set_twin(test, test_u));
    }
    double se_u, se_avg_u;
    int64_t num_u = 0;
    auto finish_both = [&](double *se_p, double *se_avg_p, int64_t *num_p) {
        check(bpmf_hip_predict_finish(test, se_p, se_avg_p, num_p));
        check(bpmf_hip_predict_finish(test_u, &se_u, &se_avg_u, &num_u));
    };

    char host[1024];
    gethostname(host, sizeof host);
    os << "hostname: " << host << std::endl;
    os << "pid: " << getpid() << std::endl;
    if (getenv("PBS_JOBID")) os << "jobid: " << getenv("PBS_JOBID") << std::endl;
    os << "num_latent: " << K << std::endl;
    if (J.dtype == BPMF_HIP_F32) os << "arithmetic: fp32 factors / Gram / factorisation (--fp32), fp64 hyper-parameters and sums" << std::endl;
    if (bpmf_hip_kernel_k(K, J.dtype) != K) os << "kernels: num_latent " << bpmf_hip_kernel_k(K, J.dtype) << ", " << bpmf_hip_kernel_k(K, J.dtype) - K << " padded dimensions" << std::endl;
    os << "nprocs: " << J.nranks << std::endl;
    os << "nthrds: " << (J.nthrds > 0 ? J.nthrds : 1) << std::endl;
    os << "nsims: " << J.nsims << std::endl;
    os << "burnin: " << J.burnin << std::endl;
    os << "alpha: " << J.alpha << std::endl;
    os << "update_freq: " << J.update_freq << std::endl;
    if (!J.perm_m.empty()) os << "assignment: greedy (c++/assign.cpp), columns renumbered" << std::endl;
    if (J.sharded) os << "movs domain: [" << m0 << ", " << m1 << ")  users domain: [" << u0 << ", " << u1 << ")" << std::endl;

    const int nsims = J.nsims, burnin = J.burnin;
    const double alpha = J.alpha;
    long double average_items_sec = 0, average_ratings_sec = 0;
    double rmse = NAN, rmse_avg = NAN, se, se_avg;
    int64_t num_predict = 0;
    int iter = -1;
    const double begin = tick();
    // Sys::print, c++/sample.cpp:101-107
    auto print_line = [&](int it, double rm, double rma, double nu, double nm, double secs) {
        const double items_per_sec = (double)(nusers + nmovies) / secs;
        const double ratings_per_sec = (double)J.M.nnz() / secs;
        char buf[1024];
        snprintf(buf, sizeof buf, "%d: %s iteration %d:\t RMSE: %3.4f\tavg RMSE: %3.4f\tFU(%6.2f)\tFM(%6.2f)\titems/sec: %6.2f\tratings/sec: %6.2fM\n",
                 rank, (it < burnin) ? "Burnin" : "Sampling", it, rm, rma, std::sqrt(nu), std::sqrt(nm), items_per_sec, ratings_per_sec / 1e6);
        os << buf << std::flush;
        average_items_sec += items_per_sec;
        average_ratings_sec += ratings_per_sec;
    };
    if (J.odirname.empty() && !J.verbose) {
        // Plain sampling run: the loop of c++/bpmf.cpp:180-198 software-pipelined by one half-iteration.
        // The library only enqueues in bpmf_hip_sys_sample; the line of iteration i-1 (its RMSE sums
        // and norms) is collected after iteration i has been queued, so the device
        // never waits for the host's printing.  The per-iteration rate is the time between two lines.
        double mark = tick(), norm_m = 0.0, norm_u = 0.0;
        for (int i = 0; i < nsims; ++i) {
            check(bpmf_hip_sys_sample(movies, users, alpha));   // movies.sample(users)
            check(bpmf_hip_sys_sample(users, movies, alpha));   // users.sample(movies)
            if (i > 0) {
                // norms of iteration i-1 (bpmf_hip_sys_norm waits for THAT half-iteration's sums only: asking bpmf_hip_sys_state
                // here drained each side's pipeline once per iteration -- 83 M against the 100 M samples/s of the same loop
                // without it, bench.py's bpmf_exe record of round 4)
                check(bpmf_hip_sys_norm(movies, i - 1, &norm_m));
                check(bpmf_hip_sys_norm(users, i - 1, &norm_u));
                // the evaluation of iteration i-1 ran beside the two samplers just queued
                finish_both(&se, &se_avg, &num_predict);
                rmse = std::sqrt(se / (double)num_predict);
                rmse_avg = std::sqrt(se_avg / (double)num_predict);
                const double now = tick();
                print_line(i - 1, rmse, rmse_avg, norm_u, norm_m, now - mark);
                mark = now;
            }
            iter = i;
            check(bpmf_hip_predict_launch(test, movies, users, (iter < burnin) ? 0 : (iter - burnin)));
        }
        if (nsims > 0) {
            finish_both(&se, &se_avg, &num_predict);
            rmse = std::sqrt(se / (double)num_predict);
            rmse_avg = std::sqrt(se_avg / (double)num_predict);
            check(bpmf_hip_sys_state(movies, nullptr, &norm_m, nullptr, nullptr, nullptr, nullptr));
            check(bpmf_hip_sys_state(users, nullptr, &norm_u, nullptr, nullptr, nullptr, nullptr));
            print_line(nsims - 1, rmse, rmse_avg, norm_u, norm_m, tick() - mark);
        }
    } else
    for (int i = 0; i < nsims; ++i) {
        const double start = tick();
        check(bpmf_hip_sys_sample(movies, users, alpha));       // movies.sample(users)
        check(bpmf_hip_sys_sample(users, movies, alpha));       // users.sample(movies)
        iter = i;
        const int n = (iter < burnin) ? 0 : (iter - burnin);
        check(bpmf_hip_predict_launch(test, movies, users, n));
        finish_both(&se, &se_avg, &num_predict);
        rmse = std::sqrt(se / (double)num_predict);
        rmse_avg = std::sqrt(se_avg / (double)num_predict);
        const double stop = tick();
        double norm_u, norm_m;
        check(bpmf_hip_sys_state(users, nullptr, &norm_u, nullptr, nullptr, nullptr, nullptr));
        check(bpmf_hip_sys_state(movies, nullptr, &norm_m, nullptr, nullptr, nullptr, nullptr));
        print_line(iter, rmse, rmse_avg, norm_u, norm_m, stop - start);

        // aggrMu / aggrLambda of this rank's columns, on the device (c++/sample.cpp:364-368)
        if (aggregate && iter >= burnin) { check(bpmf_hip_side_aggr_add(users)); check(bpmf_hip_side_aggr_add(movies)); }
        // (-v: the replicas of both factor matrices are complete on every rank -- the all-gather form of the exchange;
        // users.bcast() / movies.bcast() of c++/bpmf.cpp:202-203 have nothing left to do)
        if (J.verbose && rank == 0) {
            Dense d;
            d.nrows = K;
            d.ncols = nusers; d.data.resize((size_t)K * nusers);
            check(bpmf_hip_side_get_items(users, d.data.data()));
            if (!J.perm_u.empty()) permute_columns(d.data, K, inverse(J.perm_u));
            bpmf::io::write_dense(J.odirname + "/U-" + std::to_string(i) + ".ddm", d);
            d.ncols = nmovies; d.data.resize((size_t)K * nmovies);
            check(bpmf_hip_side_get_items(movies, d.data.data()));
            if (!J.perm_m.empty()) permute_columns(d.data, K, inverse(J.perm_m));
            bpmf::io::write_dense(J.odirname + "/V-" + std::to_string(i) + ".ddm", d);
        }
    }
    const double elapsed = tick() - begin;

    // movies.predict(users, true) once more with the same iter (c++/bpmf.cpp:225,242: SURVEY Q6)
    if (nsims > 0) {
        const int n = (iter < burnin) ? 0 : (iter - burnin);
        check(bpmf_hip_predict_launch(test, movies, users, n));
        finish_both(&se, &se_avg, &num_predict);
        rmse_avg = std::sqrt(se_avg / (double)num_predict);
    }
    if (!J.odirname.empty()) {                                        // this rank's slice of Pavg / Pm2 and of the posterior
        const size_t tn = (size_t)(J.T.colptr[(size_t)m1] - J.T.colptr[(size_t)m0]);
        if (tn) check(bpmf_hip_test_get(test, J.pavg.data() + toff, J.pm2.data() + toff));
        const int nsamples = nsims - burnin;
        if (nsamples > 0) {                                           // Sys::finalize_mu_lambda (c++/bpmf.cpp:281-295), batched on the device
            check(bpmf_hip_side_aggr_finalize(users, nsamples, J.u_mu.data() + (size_t)K * u0, J.u_lambda.data() + (size_t)K * K * u0));
            check(bpmf_hip_side_aggr_finalize(movies, nsamples, J.m_mu.data() + (size_t)K * m0, J.m_lambda.data() + (size_t)K * K * m0));
        }
    }
    if (rank == 0) {
        J.elapsed = elapsed; J.rmse_avg = rmse_avg; J.num_predict = num_predict;
        J.average_items_sec = average_items_sec; J.average_ratings_sec = average_ratings_sec;
    }
    bpmf_hip_test_destroy(test);
    bpmf_hip_test_destroy(test_u);
    bpmf_hip_side_destroy(movies);
    bpmf_hip_side_destroy(users);
    bpmf_hip_ctx_destroy(ctx);
}

}  // namespace

int main(int argc, char *argv[])
{
    Job J;
    std::string fname, probename, mname, lname;
    bool balance = true;
    int K = 32, ngpu = getenv("BPMF_NGPU") ? atoi(getenv("BPMF_NGPU")) : 0;
    if (const char *e = getenv("BPMF_NUMLATENT")) K = atoi(e);

    bool fp32 = getenv("BPMF_HIP_F32") && atoi(getenv("BPMF_HIP_F32")) != 0;
    static const struct option long_opts[] = {{"fp32", no_argument, nullptr, 1000}, {nullptr, 0, nullptr, 0}};
    int ch;
    while ((ch = getopt_long(argc, argv, "krvn:t:p:i:b:f:o:m:l:a:d:g:h", long_opts, nullptr)) != -1) {
        switch (ch) {
        case 1000: fp32 = true; break;
        case 'i': J.nsims = atoi(optarg); break;
        case 'b': J.burnin = atoi(optarg); break;
        case 'f': J.update_freq = atoi(optarg); break;
        case 't': J.nthrds = atoi(optarg); break;
        case 'a': J.alpha = atof(optarg); break;
        case 'd': K = atoi(optarg); break;
        case 'g': ngpu = atoi(optarg); break;
        case 'n': fname = optarg; break;
        case 'p': probename = optarg; break;
        case 'o': J.odirname = optarg; break;
        case 'm': mname = optarg; break;
        case 'l': lname = optarg; break;
        case 'r': J.redirect = true; break;
        case 'k': balance = false; break;
        case 'v': J.verbose = true; break;
        default: usage(); return 1;
        }
    }
    if (fname.empty() || probename.empty()) { usage(); return 1; }
    // fp64 like the reference (c++/bpmf.h:55-58) for every num_latent; the fp32 large-K path only when asked for
    J.K = K;
    J.dtype = fp32 ? BPMF_HIP_F32 : BPMF_HIP_F64;
    if (!bpmf_hip_supports(K, J.dtype))
        die("unsupported number of latent dimensions " + std::to_string(K) + (fp32 ? " with --fp32 (65 .. 128)" : " (1 .. 128)"));
    if (J.verbose && J.odirname.empty()) die("-v needs -o DIR");   // the reference would write to "/U-0.ddm" 
    // -g N: the sharded path (N = 1 too: one rank with a communicator -- what the tests can run on one GPU); no -g: NO_COMM
    J.sharded = ngpu >= 1;
    J.nranks = std::max(ngpu, 1);

    // Sys::Sys (c++/sample.cpp:112-137): read, grow both to the common shape, transpose for the users
    try {
        J.M = bpmf::io::read_sparse(fname);
        J.T = bpmf::io::read_sparse(probename);
    } catch (const std::exception &e) { die(e.what()); }
    const int64_t rows = std::max(J.M.nrows, J.T.nrows), cols = std::max(J.M.ncols, J.T.ncols);
    bpmf::io::resize(J.M, rows, cols);
    bpmf::io::resize(J.T, rows, cols);
    if (J.M.nnz() == 0) die("the training matrix is empty");
    J.Mt = bpmf::io::transpose(J.M);
    const int64_t nmovies = cols, nusers = rows;
    double msum = 0.0, usum = 0.0;                              // mean_rating = M.sum()/M.nonZeros() per Sys (:183)
    for (double v : J.M.vals) msum += v;
    for (double v : J.Mt.vals) usum += v;
    J.mean_m = msum / (double)J.M.nnz(); J.mean_u = usum / (double)J.Mt.nnz();

    // Sys::add_prop_posterior (c++/sample.cpp:157-174): "mu_file,lambda_file"; K x N and K*K x N dense matrices
    auto read_prop = [&](const std::string &fnames, int64_t n, const char *what, Dense &mu, Dense &lambda) {
        if (fnames.empty()) return;
        const size_t pos = fnames.find_first_of(",");
        if (pos == std::string::npos) die(std::string("-") + what + " expects MU_FILE,LAMBDA_FILE");
        try {
            mu = bpmf::io::read_dense(fnames.substr(0, pos));
            lambda = bpmf::io::read_dense(fnames.substr(pos + 1));
        } catch (const std::exception &e) { die(e.what()); }
        if (mu.ncols != n || lambda.ncols != n || mu.nrows != K || lambda.nrows != (int64_t)K * K)
            die(std::string("propagated posterior (-") + what + "): expected " + std::to_string(K) + " x " + std::to_string(n) + " and " +
                std::to_string(K * K) + " x " + std::to_string(n) + " matrices");
    };
    read_prop(mname, nmovies, "m", J.prop_m_mu, J.prop_m_lambda);
    read_prop(lname, nusers, "l", J.prop_u_mu, J.prop_u_lambda);

    // Assignment of the columns to the GPUs.  -k: equal column counts (c++/assign.cpp:60-65).  Otherwise the reference's
    // greedy, permuting assign() (BPMF_ASSIGN=greedy, the default with more than one GPU: `bpmf -g N` then behaves like
    // `mpirun -np N bpmf`, including column ids -- and RNG streams -- that depend on N), or BPMF_ASSIGN=contiguous:
    // cuts of the original order at equal work, which keep the samples independent of N.
    const char *assign_env = getenv("BPMF_ASSIGN");
    // (test hook: BPMF_TEST_ASSIGN_PARTS=P renumbers as for P ranks while running on the ranks there are -- what one
    // GPU can check of the permute / unpermute plumbing)
    const int assign_parts = getenv("BPMF_TEST_ASSIGN_PARTS") ? atoi(getenv("BPMF_TEST_ASSIGN_PARTS")) : J.nranks;
    const bool greedy = balance && assign_parts > 1 && !(assign_env && std::string(assign_env) == "contiguous");
    if (greedy) {
        J.perm_m.resize((size_t)nmovies); J.perm_u.resize((size_t)nusers);
        for (int64_t i = 0; i < nmovies; ++i) J.perm_m[(size_t)i] = i;
        for (int64_t i = 0; i < nusers; ++i) J.perm_u[(size_t)i] = i;
        std::vector<int64_t> dm((size_t)assign_parts + 1, 0), du((size_t)assign_parts + 1, 0);
        for (int pass = 0; pass < 2; ++pass) {                       // movies.assign(users); users.assign(movies); twice (c++/bpmf.cpp:140-143)
            for (int side = 0; side < 2; ++side) {
                std::vector<int64_t> &perm = side == 0 ? J.perm_m : J.perm_u;
                const Csc &A = side == 0 ? J.M : J.Mt;
                std::vector<int64_t> cp(perm.size() + 1, 0), order(perm.size());
                for (size_t j = 0; j < perm.size(); ++j) cp[j + 1] = cp[j] + (A.colptr[(size_t)perm[j] + 1] - A.colptr[(size_t)perm[j]]);
                if (bpmf_assign_greedy((int64_t)perm.size(), cp.data(), assign_parts, order.data(), side == 0 ? dm.data() : du.data()))
                    die("assignment failed");
                std::vector<int64_t> np(perm.size());
                for (size_t j = 0; j < perm.size(); ++j) np[j] = perm[(size_t)order[j]];
                perm.swap(np);
            }
        }
        if (assign_parts == J.nranks) { J.bm = dm; J.bu = du; }
        else { J.bm = {0, nmovies}; J.bu = {0, nusers}; if (J.nranks != 1) die("BPMF_TEST_ASSIGN_PARTS needs one rank"); }
        const std::vector<int64_t> inv_u = inverse(J.perm_u);
        J.M = permute(J.M, J.perm_m, inv_u);
        J.T = permute(J.T, J.perm_m, inv_u);
        J.Mt = bpmf::io::transpose(J.M);
        permute_columns(J.prop_m_mu.data, K, J.perm_m); permute_columns(J.prop_m_lambda.data, (int64_t)K * K, J.perm_m);
        permute_columns(J.prop_u_mu.data, K, J.perm_u); permute_columns(J.prop_u_lambda.data, (int64_t)K * K, J.perm_u);
    } else {
        J.bm = column_ranges(J.M, J.nranks, balance);
        J.bu = column_ranges(J.Mt, J.nranks, balance);
    }
    // BPMF_HIP_DEVICES=d0,d1,...: the HIP device of every rank (default: rank r on device r).  Two ranks on one device
    // need a communication library that serves them (BPMF_HIP_RCCL_LIBRARY: RCCL itself refuses) -- the tests' set-up.
    if (const char *e = getenv("BPMF_HIP_DEVICES")) {
        std::stringstream ss(e);
        std::string tok;
        while (std::getline(ss, tok, ',')) if (!tok.empty()) J.devices.push_back(atoi(tok.c_str()));
    }
    J.Tt = bpmf::io::transpose(J.T);
    if (J.sharded) check(bpmf_hip_comm_unique_id(J.rccl_id));      // (also loads RCCL before the rank threads start)
    if (!J.odirname.empty()) {
        J.pavg.assign(J.T.vals.size(), 0.0); J.pm2.assign(J.T.vals.size(), 0.0);
        const double nan = std::nan("");                              // (no post-burn-in sample: the reference divides 0 by 0)
        J.u_mu.assign((size_t)K * nusers, nan); J.u_lambda.assign((size_t)K * K * nusers, nan);
        J.m_mu.assign((size_t)K * nmovies, nan); J.m_lambda.assign((size_t)K * K * nmovies, nan);
    }

    // stdout of the ranks: bpmf_<rank>.out when there are several or with -r (c++/bpmf.cpp:111-117)
    const bool to_files = J.nranks > 1 || J.redirect;
    std::vector<std::unique_ptr<std::ofstream>> files;
    auto rank_os = [&](int r) -> std::ostream & {
        if (!to_files) return std::cout;
        return *files[(size_t)r];
    };
    if (to_files) for (int r = 0; r < J.nranks; ++r) {
        // emptied, then opened in APPEND mode: die() adds its line through a descriptor of its own (O_APPEND), and a rank thread
        // that writes its next line afterwards must append behind it, not overwrite it at its own file offset
        const std::string name = "bpmf_" + std::to_string(r) + ".out";
        { std::ofstream empty(name, std::ios::out | std::ios::trunc); }
        files.emplace_back(new std::ofstream(name, std::ios::out | std::ios::app));
        g_rank_files.push_back(name);
    }

    if (J.nranks == 1) {
        rank_main(J, 0, rank_os(0));
    } else {
        std::vector<std::thread> th;
        g_rank_threads.store(true);
        for (int r = 0; r < J.nranks; ++r) th.emplace_back([&J, r, &rank_os] { rank_main(J, r, rank_os(r)); });
        for (auto &t : th) t.join();
        g_rank_threads.store(false);
    }
    std::ostream &os = rank_os(0);

    if (!J.odirname.empty()) {
        try {
            // everything that is written is in the ORIGINAL numbering (users.unpermuteCols / movies.unpermuteCols,
            // c++/bpmf.cpp:223-224; the reference leaves items() / aggrMu in the permuted order: not reproduced)
            Csc P = J.T;
            P.vals = J.pavg;
            if (!J.perm_m.empty()) P = permute(P, inverse(J.perm_m), J.perm_u);
            bpmf::io::write_sparse(J.odirname + "/Pavg.sdm", P);
            P = J.T; P.vals = J.pm2;
            if (!J.perm_m.empty()) P = permute(P, inverse(J.perm_m), J.perm_u);
            bpmf::io::write_sparse(J.odirname + "/Pm2.sdm", P);
            if (!J.perm_m.empty()) {
                const std::vector<int64_t> im = inverse(J.perm_m), iu = inverse(J.perm_u);
                permute_columns(J.u_mu, K, iu); permute_columns(J.u_lambda, (int64_t)K * K, iu);
                permute_columns(J.m_mu, K, im); permute_columns(J.m_lambda, (int64_t)K * K, im);
            }
            Dense d;
            d.nrows = K; d.ncols = nusers; d.data.swap(J.u_mu);
            bpmf::io::write_dense(J.odirname + "/U-mu.ddm", d);
            d.nrows = (int64_t)K * K; d.data.swap(J.u_lambda);
            bpmf::io::write_dense(J.odirname + "/U-Lambda.ddm", d);
            d.nrows = K; d.ncols = nmovies; d.data.swap(J.m_mu);
            bpmf::io::write_dense(J.odirname + "/V-mu.ddm", d);
            d.nrows = (int64_t)K * K; d.data.swap(J.m_lambda);
            bpmf::io::write_dense(J.odirname + "/V-Lambda.ddm", d);
        } catch (const std::exception &e) { die(e.what()); }
    }

    os << "Total time: " << J.elapsed << std::endl;
    os << "Final Avg RMSE: " << J.rmse_avg << std::endl;
    os << "  computed on " << J.num_predict << " items (" << (J.T.nnz() ? int(100. * (double)J.num_predict / (double)J.T.nnz()) : 0)
       << "% of total items in test set)" << std::endl;
    // the reference divides by movies.iter = nsims-1; this build reports the true mean
    os << "Average items/sec: " << (double)(J.average_items_sec / std::max(J.nsims, 1)) << std::endl;
    os << "Average ratings/sec: " << (double)(J.average_ratings_sec / std::max(J.nsims, 1)) << std::endl;
    return 0;
}
