// host micro-benchmark of bpmf_hyper_finish (the cov-dependent part of the Normal-Wishart draw): g++ -O3 tools/probes/hyper_bench.cpp bpmf_amd/csrc/hyper.o -o /tmp/hb -lpthread; BPMF_HIP_FINISH_THREADS=T /tmp/hb K
#include <chrono>
#include <cstdio>
#include <vector>
#include <cmath>
extern "C" int bpmf_hyper_draws(int K, long N, unsigned counter, double *au, double *z);
extern "C" int bpmf_hyper_finish(int K, long N, const double *cov, const double *Um, const double *au, const double *z, double *mu, double *LU, double *LF);
extern "C" void bpmf_hip_set_error_(const char *m) { fprintf(stderr, "err %s\n", m); }
int main(int argc, char **argv) {
    int K = argc > 1 ? atoi(argv[1]) : 128; long N = 6040;
    std::vector<double> cov(K*K), au(K*K), z(K), mu(K), LU(K*K), LF(K*K);
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) cov[i*K+j] = (i==j ? 1.0 : 0.0) + 0.01 * std::cos(i*0.3+j*0.7) * std::cos(j*0.3+i*0.7);
    for (int i = 0; i < K; ++i) for (int j = 0; j < i; ++j) cov[i*K+j] = cov[j*K+i];
    bpmf_hyper_draws(K, N, 12345u, au.data(), z.data());
    double best = 1e9, cs = 0;
    for (int r = 0; r < 50; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        bpmf_hyper_finish(K, N, cov.data(), nullptr, au.data(), z.data(), mu.data(), LU.data(), LF.data());
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us < best) best = us;
        cs += LF[5*K+7];
    }
    double sum = 0; for (double v : LF) sum += v; for (double v : mu) sum += v;
    printf("K=%d finish best %.1f us  checksum %.17g\n", K, best, sum);
}
