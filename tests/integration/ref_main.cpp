// TEST INFRASTRUCTURE -- the NO_COMM loop of the reference's main() (/root/reference c++/bpmf.cpp:83-109 flags, :131-136 the two
// Sys, :180-198 the iteration, :206-207 the -v dumps, :221-253 the outputs and the closing lines) over ref_shim.h's stand-in `Sys`
// and the back-end header `hip_sys.h` that tests/test_integration_stub.py extracts from INTEGRATION.md: what a maintainer's
// `bpmf` built with -DBPMF_HIP_COMM does, minus Eigen.  The calls marked HIP are the places INTEGRATION.md tells the
// maintainer to touch in bpmf.cpp.
#include <chrono>
#include <cstring>

#include "ref_shim.h"

bool Sys::verbose = false;
int Sys::nprocs = 1, Sys::procid = 0;
int Sys::burnin = 5, Sys::nsims = 20;
double Sys::alpha = 2.0;
std::string Sys::odirname;

#include "hip_sys.h"                                                    // defines SYS, HIP_Sys and Sys::Init / Finalize / sync / Abort

static double tick() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    std::string fname, probename;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> std::string { if (i + 1 >= argc) { std::cerr << "missing value for " << a << std::endl; exit(2); } return argv[++i]; };
        if (a == "-n") fname = next();
        else if (a == "-p") probename = next();
        else if (a == "-i") Sys::nsims = atoi(next().c_str());
        else if (a == "-b") Sys::burnin = atoi(next().c_str());
        else if (a == "-a") Sys::alpha = atof(next().c_str());
        else if (a == "-o") Sys::odirname = next();
        else if (a == "-v") Sys::verbose = true;
        else { std::cerr << "unknown flag " << a << std::endl; return 2; }
    }
    Sys::Init();
    int rc = 0;
    try {
        SYS movies("movs", fname, probename);
        SYS users("users", movies.M, movies.Pavg);
        movies.alloc_and_init();
        users.alloc_and_init();
        Sys::cout() << "num_latent: " << num_latent << std::endl;
        long double average_items_sec = .0;
        const double begin = tick();
        for (int i = 0; i < Sys::nsims; ++i) {
            const double start = tick();
            movies.sample(users);
            users.sample(movies);
            movies.hip_predict(users, false);                           // HIP: bpmf.cpp:189
            users.hip_predict(movies, false);                           // HIP: bpmf.cpp:190
            const double stop = tick();
            const double items_per_sec = (users.num() + movies.num()) / (stop - start);
            movies.print(items_per_sec, users.nnz() / (stop - start), sqrt(users.norm), sqrt(movies.norm));
            average_items_sec += items_per_sec;
            if (Sys::verbose) {                                          // items() of both sides are current on the host: the stub fetched them
                DenseD u(num_latent, users.num()), v(num_latent, movies.num());
                memcpy(u.data(), users.items_ptr, sizeof(double) * (size_t)u.size());
                memcpy(v.data(), movies.items_ptr, sizeof(double) * (size_t)v.size());
                write_matrix(Sys::odirname + "/U-" + std::to_string(i) + ".ddm", u);
                write_matrix(Sys::odirname + "/V-" + std::to_string(i) + ".ddm", v);
            }
        }
        const double elapsed = tick() - begin;
        movies.hip_predict(users, true);                                // HIP: bpmf.cpp:225 / :242
        if (Sys::odirname.size()) {
            movies.fetch_predictions();                                  // HIP: before bpmf.cpp:229-230
            write_matrix(Sys::odirname + "/Pavg.sdm", movies.Pavg);
            write_matrix(Sys::odirname + "/Pm2.sdm", movies.Pm2);
            users.hip_finalize_mu_lambda();                              // HIP: instead of users.finalize_mu_lambda(), bpmf.cpp:232
            write_matrix(Sys::odirname + "/U-mu.ddm", users.aggrMu);
            write_matrix(Sys::odirname + "/U-Lambda.ddm", users.aggrLambda);
            movies.hip_finalize_mu_lambda();                             // HIP: instead of movies.finalize_mu_lambda(), bpmf.cpp:236
            write_matrix(Sys::odirname + "/V-mu.ddm", movies.aggrMu);
            write_matrix(Sys::odirname + "/V-Lambda.ddm", movies.aggrLambda);
        }
        Sys::cout() << "Total time: " << elapsed << std::endl;
        Sys::cout() << "Final Avg RMSE: " << movies.rmse_avg << std::endl;
        Sys::cout() << "  computed on " << movies.num_predict << " items (" << int(100. * movies.num_predict / movies.T.nonZeros())
                    << "% of total items in test set)" << std::endl;
        Sys::cout() << "Average items/sec: " << (double)(average_items_sec / movies.iter) << std::endl;
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << std::endl;
        rc = 1;
    }
    Sys::Finalize();
    return rc;
}
