/*
 * pin_libstdcxx.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Drives the REAL libstdc++ std::normal_distribution<> / std::gamma_distribution<>
 * (the third-party code the reference calls at c++/mvnormal.cpp:42,68) with the
 * Philox word stream, exactly the way the reference does:
 *   randn():  std::normal_distribution<>()(rng)        -- temporary per call
 *   gamma:    std::gamma_distribution<> gam(a); gam(rng)
 * so that tests can pin oracle/bpmf_oracle.c's C restatement of those
 * transforms bit-for-bit.  The URNG below is the restated MicroURNG
 * (word order r[3],r[2],r[1],r[0]; counter {c,0,0,n}; key {42,0}); Philox
 * itself comes from the oracle and is pinned by the Random123 KAT vectors.
 */
#include <cstdint>
#include <random>

extern "C" void bpmf_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

namespace {
struct MicroPhilox {
    typedef uint32_t result_type;
    uint32_t c0, n; int last; uint32_t r[4];
    explicit MicroPhilox(uint32_t c) : c0(c), n(0), last(0) {}
    static constexpr result_type min() { return 0; }
    static constexpr result_type max() { return 0xFFFFFFFFu; }
    result_type operator()() {
        if (last == 0) {
            const uint32_t ctr[4] = {c0, 0u, 0u, n};
            const uint32_t key[2] = {42u, 0u};
            bpmf_oracle_philox4x32_10(ctr, key, r);
            ++n; last = 4;
        }
        return r[--last];
    }
};
}

extern "C" __attribute__((visibility("default")))
void pin_randn_stream(uint32_t counter, int n, double *out)
{
    MicroPhilox rng(counter);
    for (int i = 0; i < n; ++i) out[i] = std::normal_distribution<>()(rng);
}

extern "C" __attribute__((visibility("default")))
void pin_gamma_stream(uint32_t counter, int n, const double *alphas, double *out_gamma, double *out_randn_after)
{
    MicroPhilox rng(counter);
    for (int i = 0; i < n; ++i) {
        std::gamma_distribution<> gam(alphas[i]);
        out_gamma[i] = gam(rng);
        out_randn_after[i] = std::normal_distribution<>()(rng);
    }
}
