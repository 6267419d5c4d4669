// bpmf_main.cpp -- the `bpmf` executable of this repo: the reference's command line, stdout lines
// and output files (c++/bpmf.cpp:41-260) on top of libbpmf_hip.so.
//
// Host C++ only: it reads the matrices (io.cpp), mirrors them to the device through the C ABI of
// include/bpmf_hip.h and runs main()'s Gibbs loop; every column update happens in the HIP kernels.
// Flags: -n TRAIN -p TEST [-o DIR] [-i N] [-b N] [-a F] [-d K] [-t N] [-f N] [-k] [-r] [-v]
// -m / -l "MU_FILE,LAMBDA_FILE": propagated posteriors of a previous run (c++/bpmf.cpp:134-135).
// K (the reference's compile-time BPMF_NUMLATENT) is chosen at run time: -d K, else the
// environment variable BPMF_NUMLATENT, else 32.
#include <getopt.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/bpmf_hip.h"
#include "io.h"

namespace {

using bpmf::io::Csc;
using bpmf::io::Dense;

double tick()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void usage()
{
    std::cout << "Usage: bpmf -n <MTX> -p <MTX> [-o DIR/] [-i N] [-b N] [-f N] [-a F] [-d K] [-krv] [-t N]\n"
              << "\n"
              << "Parameters:\n"
              << "  -n MTX: training matrix (rows = users, columns = items)\n"
              << "  -p MTX: test matrix\n"
              << "  [-o DIR]: directory for the model and the predictions\n"
              << "  [-i N]: total number of Gibbs iterations (20)\n"
              << "  [-b N]: number of burn-in iterations (5)\n"
              << "  [-f N]: update frequency (accepted, unused)\n"
              << "  [-a F]: noise precision alpha (2.0)\n"
              << "  [-d K]: number of latent dimensions: 8, 16, 32 or 64 (32, or $BPMF_NUMLATENT)\n"
              << "\n"
              << "  [-k]: do not optimise the item-to-node assignment (single process: no effect)\n"
              << "  [-r]: redirect stdout to bpmf_0.out\n"
              << "  [-v]: write every sample (U-<i>.ddm, V-<i>.ddm)\n"
              << "  [-t N]: host threads (accepted; the column loop runs on the GPU)\n"
              << "\n"
              << "Matrix formats (by extension, optionally .gz):\n"
              << "  *.mtx / *.mm: MatrixMarket, sparse (coordinate) or dense (array)\n"
              << "  *.sdm / *.sbm: sparse binary double / pattern\n"
              << "  *.ddm / *.csv: dense binary double / text\n"
              << std::endl;
}

[[noreturn]] void die(const std::string &msg)
{
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

// K x K inverse by Gauss-Jordan with partial pivoting (finalize_mu_lambda's cov.inverse(), c++/bpmf.cpp:291)
bool invert_inplace(int K, std::vector<double> &a)
{
    std::vector<double> inv((size_t)K * K, 0.0);
    for (int i = 0; i < K; ++i) inv[(size_t)i * K + i] = 1.0;
    auto A = [&](int r, int c) -> double & { return a[(size_t)c * K + r]; };
    auto B = [&](int r, int c) -> double & { return inv[(size_t)c * K + r]; };
    for (int c = 0; c < K; ++c) {
        int p = c;
        for (int r = c + 1; r < K; ++r) if (std::fabs(A(r, c)) > std::fabs(A(p, c))) p = r;
        if (A(p, c) == 0.0) return false;
        if (p != c) for (int j = 0; j < K; ++j) { std::swap(A(c, j), A(p, j)); std::swap(B(c, j), B(p, j)); }
        const double d = A(c, c);
        for (int j = 0; j < K; ++j) { A(c, j) /= d; B(c, j) /= d; }
        for (int r = 0; r < K; ++r) {
            if (r == c) continue;
            const double f = A(r, c);
            if (f == 0.0) continue;
            for (int j = 0; j < K; ++j) { A(r, j) -= f * A(c, j); B(r, j) -= f * B(c, j); }
        }
    }
    a.swap(inv);
    return true;
}

// posterior aggregation of one side (aggrMu / aggrLambda, c++/sample.cpp:364-368; c++/bpmf.cpp:281-295)
struct Aggregate {
    int K = 0; int64_t N = 0;
    std::vector<double> mu, lambda, items;
    void init(int K_, int64_t N_) { K = K_; N = N_; mu.assign((size_t)K * N, 0.0); lambda.assign((size_t)K * K * N, 0.0); items.resize((size_t)K * N); }
    void add(bpmf_hip_side *side)
    {
        check(bpmf_hip_side_get_items(side, items.data()));
        for (int64_t c = 0; c < N; ++c) {
            const double *r = &items[(size_t)c * K];
            double *m = &mu[(size_t)c * K], *l = &lambda[(size_t)c * K * K];
            for (int j = 0; j < K; ++j) {
                m[j] += r[j];
                for (int i = 0; i < K; ++i) l[(size_t)j * K + i] += r[i] * r[j];
            }
        }
    }
    void finalize(int nsamples)
    {
        std::vector<double> cov((size_t)K * K);
        const double nan = std::nan("");
        for (int64_t c = 0; c < N; ++c) {
            double *m = &mu[(size_t)c * K], *l = &lambda[(size_t)c * K * K];
            for (int j = 0; j < K; ++j)
                for (int i = 0; i < K; ++i) cov[(size_t)j * K + i] = (l[(size_t)j * K + i] - (m[i] * m[j] / nsamples)) / (nsamples - 1);
            if (!invert_inplace(K, cov)) std::fill(cov.begin(), cov.end(), nan);     // singular when nsamples <= K
            memcpy(l, cov.data(), sizeof(double) * K * K);
            for (int j = 0; j < K; ++j) m[j] /= nsamples;
        }
    }
};

}  // namespace

int main(int argc, char *argv[])
{
    std::string fname, probename, mname, lname, odirname;
    int nsims = 20, burnin = 5, update_freq = 1, nthrds = -1, K = 32;
    double alpha = 2.0;
    bool redirect = false, verbose = false, k_given = false;
    if (const char *e = getenv("BPMF_NUMLATENT")) K = atoi(e);

    int ch;
    while ((ch = getopt(argc, argv, "krvn:t:p:i:b:f:o:m:l:a:d:h")) != -1) {
        switch (ch) {
        case 'i': nsims = atoi(optarg); break;
        case 'b': burnin = atoi(optarg); break;
        case 'f': update_freq = atoi(optarg); break;
        case 't': nthrds = atoi(optarg); break;
        case 'a': alpha = atof(optarg); break;
        case 'd': K = atoi(optarg); break;
        case 'n': fname = optarg; break;
        case 'p': probename = optarg; break;
        case 'o': odirname = optarg; break;
        case 'm': mname = optarg; break;
        case 'l': lname = optarg; break;
        case 'r': redirect = true; break;
        case 'k': k_given = true; break;
        case 'v': verbose = true; break;
        default: usage(); return 1;
        }
    }
    (void)k_given;
    if (fname.empty() || probename.empty()) { usage(); return 1; }
    // fp64 like the reference for 8..64 latent dimensions; 128 selects the fp32 large-K path of the library
    const int dtype = (K == 128) ? BPMF_HIP_F32 : BPMF_HIP_F64;
    if (!bpmf_hip_supports(K, dtype)) die("unsupported number of latent dimensions " + std::to_string(K) + " (8, 16, 32, 64; 128 in fp32)");

    std::ofstream redirected;
    if (redirect) redirected.open("bpmf_0.out");
    std::ostream &os = redirect ? static_cast<std::ostream &>(redirected) : std::cout;

    // Sys::Sys (c++/sample.cpp:112-137): read, grow both to the common shape, transpose for the users
    Csc M, T;
    try {
        M = bpmf::io::read_sparse(fname);
        T = bpmf::io::read_sparse(probename);
    } catch (const std::exception &e) { die(e.what()); }
    const int64_t rows = std::max(M.nrows, T.nrows), cols = std::max(M.ncols, T.ncols);
    bpmf::io::resize(M, rows, cols);
    bpmf::io::resize(T, rows, cols);
    if (M.nnz() == 0) die("the training matrix is empty");
    const Csc Mt = bpmf::io::transpose(M);
    const int64_t nmovies = cols, nusers = rows;

    double msum = 0.0, usum = 0.0;                              // mean_rating = M.sum()/M.nonZeros() per Sys (:183)
    for (double v : M.vals) msum += v;
    for (double v : Mt.vals) usum += v;
    const double mean_m = msum / (double)M.nnz(), mean_u = usum / (double)Mt.nnz();

    bpmf_hip_ctx *ctx = nullptr;
    check(bpmf_hip_ctx_create_ex(0, K, dtype, nullptr, &ctx));
    bpmf_hip_side *movies = nullptr, *users = nullptr;
    bpmf_hip_test *test = nullptr;
    check(bpmf_hip_side_create(ctx, nmovies, nusers, 0, nmovies, M.colptr.data(), M.rowidx.data(), M.vals.data(), mean_m, &movies));
    print_init(os, "movs", M, T.nnz(), mean_m);
    check(bpmf_hip_side_create(ctx, nusers, nmovies, 0, nusers, Mt.colptr.data(), Mt.rowidx.data(), Mt.vals.data(), mean_u, &users));
    // Sys::add_prop_posterior (c++/sample.cpp:157-174): "mu_file,lambda_file"; K x N and K*K x N dense matrices
    auto add_prop_posterior = [&](bpmf_hip_side *side, const std::string &fnames, int64_t n, const char *what) {
        if (fnames.empty()) return;
        const size_t pos = fnames.find_first_of(",");
        if (pos == std::string::npos) die(std::string("-") + what + " expects MU_FILE,LAMBDA_FILE");
        const Dense mu = bpmf::io::read_dense(fnames.substr(0, pos));
        const Dense lambda = bpmf::io::read_dense(fnames.substr(pos + 1));
        if (mu.ncols != n || lambda.ncols != n || mu.nrows != K || lambda.nrows != (int64_t)K * K)
            die(std::string("propagated posterior (-") + what + "): expected " + std::to_string(K) + " x " + std::to_string(n) + " and " +
                std::to_string(K * K) + " x " + std::to_string(n) + " matrices");
        check(bpmf_hip_side_set_prop_posterior(side, mu.data.data(), lambda.data.data()));
    };
    add_prop_posterior(movies, mname, nmovies, "m");
    add_prop_posterior(users, lname, nusers, "l");
    print_init(os, "users", Mt, T.nnz(), mean_u);
    check(bpmf_hip_test_create(movies, T.colptr.data(), T.rowidx.data(), T.vals.data(), &test));

    char host[1024];
    gethostname(host, sizeof host);
    os << "hostname: " << host << std::endl;
    os << "pid: " << getpid() << std::endl;
    if (getenv("PBS_JOBID")) os << "jobid: " << getenv("PBS_JOBID") << std::endl;
    os << "num_latent: " << K << std::endl;
    os << "nprocs: " << 1 << std::endl;
    os << "nthrds: " << (nthrds > 0 ? nthrds : 1) << std::endl;
    os << "nsims: " << nsims << std::endl;
    os << "burnin: " << burnin << std::endl;
    os << "alpha: " << alpha << std::endl;
    os << "update_freq: " << update_freq << std::endl;

    Aggregate agg_u, agg_m;
    const bool aggregate = !odirname.empty();
    if (aggregate) { agg_u.init(K, nusers); agg_m.init(K, nmovies); }

    long double average_items_sec = 0, average_ratings_sec = 0;
    double rmse = NAN, rmse_avg = NAN, se, se_avg;
    int64_t num_predict = 0;
    int iter = -1;
    const double begin = tick();
    // Sys::print, c++/sample.cpp:101-107
    auto print_line = [&](int it, double rm, double rma, double nu, double nm, double secs) {
        const double items_per_sec = (double)(nusers + nmovies) / secs;
        const double ratings_per_sec = (double)M.nnz() / secs;
        char buf[1024];
        snprintf(buf, sizeof buf, "%d: %s iteration %d:\t RMSE: %3.4f\tavg RMSE: %3.4f\tFU(%6.2f)\tFM(%6.2f)\titems/sec: %6.2f\tratings/sec: %6.2fM\n",
                 0, (it < burnin) ? "Burnin" : "Sampling", it, rm, rma, std::sqrt(nu), std::sqrt(nm), items_per_sec, ratings_per_sec / 1e6);
        os << buf << std::flush;
        average_items_sec += items_per_sec;
        average_ratings_sec += ratings_per_sec;
    };
    if (!aggregate && !verbose) {
        // Plain sampling run: the loop of c++/bpmf.cpp:180-198 software-pipelined by one half-iteration.
        // The library only enqueues in bpmf_hip_sys_sample; the line of iteration i-1 (its RMSE sums
        // and norms) is collected after iteration i has been queued, so the device
        // never waits for the host's printing.  The per-iteration rate is the time between two lines.
        double mark = tick(), norm_m = 0.0, norm_u = 0.0;
        for (int i = 0; i < nsims; ++i) {
            if (i > 0) check(bpmf_hip_sys_state(movies, nullptr, &norm_m, nullptr, nullptr, nullptr, nullptr));   // of iteration i-1
            check(bpmf_hip_sys_sample(movies, users, alpha));   // movies.sample(users)
            if (i > 0) check(bpmf_hip_sys_state(users, nullptr, &norm_u, nullptr, nullptr, nullptr, nullptr));    // of iteration i-1
            check(bpmf_hip_sys_sample(users, movies, alpha));   // users.sample(movies)
            if (i > 0) {
                // the evaluation of iteration i-1 ran beside the two samplers just queued
                check(bpmf_hip_predict_finish(test, &se, &se_avg, &num_predict));
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
            check(bpmf_hip_predict_finish(test, &se, &se_avg, &num_predict));
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
        check(bpmf_hip_predict(test, movies, users, n, &se, &se_avg, &num_predict));
        rmse = std::sqrt(se / (double)num_predict);
        rmse_avg = std::sqrt(se_avg / (double)num_predict);
        const double stop = tick();
        double norm_u, norm_m;
        check(bpmf_hip_sys_state(users, nullptr, &norm_u, nullptr, nullptr, nullptr, nullptr));
        check(bpmf_hip_sys_state(movies, nullptr, &norm_m, nullptr, nullptr, nullptr, nullptr));
        print_line(iter, rmse, rmse_avg, norm_u, norm_m, stop - start);

        if (aggregate && iter >= burnin) { agg_u.add(users); agg_m.add(movies); }
        if (verbose) {
            if (odirname.empty()) die("-v needs -o DIR");       // the reference would write to "/U-0.ddm" 
            Dense d;
            d.nrows = K;
            d.ncols = nusers; d.data.resize((size_t)K * nusers);
            check(bpmf_hip_side_get_items(users, d.data.data()));
            bpmf::io::write_dense(odirname + "/U-" + std::to_string(i) + ".ddm", d);
            d.ncols = nmovies; d.data.resize((size_t)K * nmovies);
            check(bpmf_hip_side_get_items(movies, d.data.data()));
            bpmf::io::write_dense(odirname + "/V-" + std::to_string(i) + ".ddm", d);
        }
    }
    const double elapsed = tick() - begin;

    // movies.predict(users, true) once more with the same iter (c++/bpmf.cpp:225,242: SURVEY Q6)
    if (nsims > 0) {
        const int n = (iter < burnin) ? 0 : (iter - burnin);
        check(bpmf_hip_predict(test, movies, users, n, &se, &se_avg, &num_predict));
        rmse_avg = std::sqrt(se_avg / (double)num_predict);
    }
    if (aggregate) {
        try {
            Csc P = T;
            std::vector<double> pm2(T.vals.size());
            check(bpmf_hip_test_get(test, P.vals.data(), pm2.data()));
            bpmf::io::write_sparse(odirname + "/Pavg.sdm", P);
            P.vals = pm2;
            bpmf::io::write_sparse(odirname + "/Pm2.sdm", P);
            const int nsamples = nsims - burnin;
            Dense d;
            agg_u.finalize(nsamples);
            d.nrows = K; d.ncols = nusers; d.data = agg_u.mu;
            bpmf::io::write_dense(odirname + "/U-mu.ddm", d);
            d.nrows = (int64_t)K * K; d.data = agg_u.lambda;
            bpmf::io::write_dense(odirname + "/U-Lambda.ddm", d);
            agg_m.finalize(nsamples);
            d.nrows = K; d.ncols = nmovies; d.data = agg_m.mu;
            bpmf::io::write_dense(odirname + "/V-mu.ddm", d);
            d.nrows = (int64_t)K * K; d.data = agg_m.lambda;
            bpmf::io::write_dense(odirname + "/V-Lambda.ddm", d);
        } catch (const std::exception &e) { die(e.what()); }
    }

    os << "Total time: " << elapsed << std::endl;
    os << "Final Avg RMSE: " << rmse_avg << std::endl;
    os << "  computed on " << num_predict << " items (" << (T.nnz() ? int(100. * (double)num_predict / (double)T.nnz()) : 0)
       << "% of total items in test set)" << std::endl;
    // the reference divides by movies.iter = nsims-1; this build reports the true mean
    os << "Average items/sec: " << (double)(average_items_sec / std::max(nsims, 1)) << std::endl;
    os << "Average ratings/sec: " << (double)(average_ratings_sec / std::max(nsims, 1)) << std::endl;

    bpmf_hip_test_destroy(test);
    bpmf_hip_side_destroy(movies);
    bpmf_hip_side_destroy(users);
    bpmf_hip_ctx_destroy(ctx);
    return 0;
}
