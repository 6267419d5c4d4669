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
        // one 32 x 32 -> 64 product per multiplier (v_mad_u64_u32 on the device: both halves from ONE instruction instead of
        // v_mul_hi_u32 + v_mul_lo_u32; all three issue at the rate of a v_fma_f64 on gfx950: tools/probes/valu_rate_probe.hip)
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
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmin(r, 0.99999999999999988897769753748434595763683319091796875);   // (r <= 1, never NaN: one v_min_f64 for a compare + two selects)
#else
    return r >= 1.0 ? 0.99999999999999988897769753748434595763683319091796875 : r;
#endif
}

// The factor of the polar method, sqrt(-2 log(r2) / r2) for 0 < r2 <= 1 (libstdc++ normal_distribution::operator(),
// bits/random.tcc: `__mult = std::sqrt(-2 * std::log(__r2) / __r2)`).
//
// The device's math library evaluates the three library calls of that expression in ~125 VALU instructions (a
// double-double logarithm, a correctly rounded division, a correctly rounded square root).  A column's normals do not
// need any of the three to be correctly rounded -- they need to be the reference's normals to a few ulp (the device
// already differs from an x86 run by the rounding of its own log) -- and 125 instructions are 5 % of a K = 32 column
// and a sixth of a product-form column.  Here, in ~45:
//   * log after fdlibm's e_log.c (r2 = 2^k (1 + f), sqrt(2)/2 < 1 + f < sqrt(2); s = f / (2 + f);
//     log(1 + f) = f - hfsq + s (hfsq + R(s^2)), hfsq = f^2 / 2, R the degree-14 minimax polynomial: error < 1 ulp);
//     the quotient s only enters the small correction term, so a reciprocal with two Newton steps is enough;
//   * sqrt(L / r2) = L / sqrt(L r2) with one reciprocal square root (hardware seed + one third-order step), L = -2 log r2.
// Measured against glibc's long-double evaluation: <= 2 ulp (tools/probes/polar_mult_check.cpp, tests/test_polar_mult.py);
// the device stream test keeps its bound of 8 ulp on the normals.  r2 = 1 gives 0 like the reference (log 1 = 0).
#if !defined(__HIP_DEVICE_COMPILE__)
inline double polar_chop_(double v)                                    // keeps 23 bits of the mantissa: what a hardware seed is good for
{
    uint64_t u;
    __builtin_memcpy(&u, &v, 8);
    u &= ~((1ull << 29) - 1ull);
    __builtin_memcpy(&v, &u, 8);
    return v;
}
#endif
BPMF_HD double polar_rcp_seed(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(d);
#else
    return polar_chop_(1.0 / d);                                       // host stand-in of the hardware seed (tests only)
#endif
}
BPMF_HD double polar_rsq_seed(double d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsq(d);
#else
    return polar_chop_(1.0 / __builtin_sqrt(d));
#endif
}
BPMF_HD double polar_mult(double r2)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
#if defined(__HIP_DEVICE_COMPILE__)
    double m = __builtin_amdgcn_frexp_mant(r2);                         // r2 = m 2^k, m in [0.5, 1)
    int k = __builtin_amdgcn_frexp_exp(r2);
#else
    int k;
    double m = __builtin_frexp(r2, &k);
#endif
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;                                               // m in [sqrt(2)/2, sqrt(2))
    k = low ? k - 1 : k;
    const double f = m - 1.0, d = m + 1.0;
    double r = polar_rcp_seed(d);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.0), r);
    const double s = f * r;
    const double z = s * s, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, Lg6, Lg4), Lg2);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
    const double hfsq = 0.5 * f * f;
    const double dk = (double)k;
    const double corr = __builtin_fma(s, hfsq + (t2 + t1), dk * ln2_lo);
    const double L = -2.0 * __builtin_fma(dk, ln2_hi, -((hfsq - corr) - f));   // -2 log(r2) >= 0
    double p = L * r2;
    p = p > 1e-300 ? p : 1e-300;                                       // r2 = 1: L = 0, the product below stays 0
    const double y0 = polar_rsq_seed(p);
    const double e = __builtin_fma(-p, y0 * y0, 1.0);
    const double y = __builtin_fma(y0, e * __builtin_fma(0.375, e, 0.5), y0);   // 1 / sqrt(L r2)
    return L * y;
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
