// philox.h -- counter-based RNG shared by the host hyper-sampler and the HIP kernels.
//
// Replaces the reference's `r123::MicroURNG<r123::Philox4x32> rng({{0}},{{42}})`
// (c++/mvnormal.cpp:18-23) and `rng_set_pos` (c++/mvnormal.cpp:34-39):
//   block n of stream c = Philox4x32-10(counter = {c,0,0,n}, key = {42,0});
//   the URNG hands a block's words out last-to-first (w3,w2,w1,w0).
// Each libstdc++ polar-method attempt consumes exactly one block
// (2 x generate_canonical<double,53> = 4 words), which is what makes the
// per-column stream block-addressable on the device.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BPMF_HD __host__ __device__ __forceinline__
#else
#define BPMF_HD inline
#endif

namespace bpmf {

struct Philox4 { uint32_t w[4]; };

BPMF_HD uint32_t mulhi32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

// Philox4x32 with 10 rounds (Random123 philox.h; constants from the paper
// "Parallel random numbers: as easy as 1, 2, 3", SC'11).
BPMF_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 product per multiplier (v_mad_u64_u32 on the device: both halves from ONE quarter-rate
        // instruction instead of v_mul_hi_u32 + v_mul_lo_u32)
        const uint64_t p0 = (uint64_t)M0 * (uint64_t)c0, p1 = (uint64_t)M1 * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    Philox4 o; o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
    return o;
}

// block n of the reference's stream `rng_set_pos(c)`
BPMF_HD Philox4 stream_block(uint32_t c, uint32_t n) { return philox4x32_10(c, 0u, 0u, n, 42u, 0u); }

// libstdc++ generate_canonical<double,53> fed two 32-bit words (first word is
// the low half): (w_first + w_second * 2^32) / 2^64, clamped below 1.
BPMF_HD double canonical53(uint32_t w_first, uint32_t w_second)
{
    // the product is exact, so one rounding happens in the add, as in libstdc++
    const double s = (double)w_first + (double)w_second * 4294967296.0;
    const double r = s * 5.421010862427522170037264004349708557128906250e-20;   // 2^-64, exact
    return r >= 1.0 ? 0.99999999999999988897769753748434595763683319091796875 : r;
}

// Host-side URNG with the MicroURNG interface expected by <random>.
struct MicroPhilox {
    typedef uint32_t result_type;
    uint32_t c0 = 0, n = 0;
    int last = 0;
    uint32_t r[4] = {0, 0, 0, 0};
    MicroPhilox() {}
    explicit MicroPhilox(uint32_t c) : c0(c) {}
    void reset(uint32_t c) { c0 = c; n = 0; last = 0; }
    static constexpr result_type min() { return 0u; }
    static constexpr result_type max() { return 0xFFFFFFFFu; }
    result_type operator()()
    {
        if (last == 0) {
            const Philox4 b = stream_block(c0, n);
            r[0] = b.w[0]; r[1] = b.w[1]; r[2] = b.w[2]; r[3] = b.w[3];
            ++n; last = 4;
        }
        return r[--last];
    }
};

}  // namespace bpmf
