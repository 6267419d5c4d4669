// Does v_mfma_f64_4x4x4_4b_f64 honour the A-matrix broadcast controls (cbsz / abid) on gfx950?
// With cbsz = 2 block `abid` of the A operand should feed all four blocks -- i.e. lane (k, b, i) reads lane (k, abid, i):
// exactly what quad_splat<abid>() of kernels_slab.h builds with two ds_swizzle per operand.  The probe compares, for random
// operands, mfma(a, b, c, cbsz = 2, abid = m) with mfma(splat_m(a), b, c) bit for bit, and times both forms.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/cbsz_probe.hip -o tools/probes/cbsz_probe && tools/probes/cbsz_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int BQ>
__device__ __forceinline__ double quad_splat(double v)
{
    constexpr int PAT = 0x13 | (BQ << 7);
    const long long w = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_ds_swizzle((int)w, PAT), hi = __builtin_amdgcn_ds_swizzle((int)(w >> 32), PAT);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

__global__ void k_check(const double *a, const double *b, const double *c, double *out)
{
    const int l = threadIdx.x;
    const double av = a[l], bv = b[l], cv = c[l];
    out[0 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(quad_splat<0>(av), bv, cv, 0, 0, 0);
    out[1 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(quad_splat<1>(av), bv, cv, 0, 0, 0);
    out[2 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(quad_splat<2>(av), bv, cv, 0, 0, 0);
    out[3 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(quad_splat<3>(av), bv, cv, 0, 0, 0);
    out[4 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, cv, 2, 0, 0);
    out[5 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, cv, 2, 1, 0);
    out[6 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, cv, 2, 2, 0);
    out[7 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, cv, 2, 3, 0);
    out[8 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, cv, 0, 0, 0);        // no broadcast (reference for "ignored")
    // blgp on the f64 shapes is the NEG field (negate A | B | C): -A as an operand modifier would save the sign flips of the trailing update
    out[9 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, cv, 0, 0, 1);
    out[10 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(-av, bv, cv, 0, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(64, 2) void k_time(const double *a, double *out, int iters)
{
    const int l = threadIdx.x;
    double y0 = a[l], y1 = a[64 + l];
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(quad_splat<0>(y0), y1, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(quad_splat<1>(y0), y1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(quad_splat<2>(y0), y1, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(quad_splat<3>(y0), y1, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f64_4x4x4f64(quad_splat<0>(y1), y0, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f64_4x4x4f64(quad_splat<1>(y1), y0, acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f64_4x4x4f64(quad_splat<2>(y1), y0, acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f64_4x4x4f64(quad_splat<3>(y1), y0, acc[7], 0, 0, 0);
        } else {
            acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(y0, y1, acc[0], 2, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(y0, y1, acc[1], 2, 1, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(y0, y1, acc[2], 2, 2, 0);
            acc[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(y0, y1, acc[3], 2, 3, 0);
            acc[4] = __builtin_amdgcn_mfma_f64_4x4x4f64(y1, y0, acc[4], 2, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f64_4x4x4f64(y1, y0, acc[5], 2, 1, 0);
            acc[6] = __builtin_amdgcn_mfma_f64_4x4x4f64(y1, y0, acc[6], 2, 2, 0);
            acc[7] = __builtin_amdgcn_mfma_f64_4x4x4f64(y1, y0, acc[7], 2, 3, 0);
        }
        y0 += 1e-300; y1 += 1e-300;
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 64 + l] = s;
}

int main()
{
    std::vector<double> h(3 * 64);
    srand(5);
    for (auto &v : h) v = (rand() % 2001 - 1000) / 64.0;
    double *d, *o;
    hipMalloc(&d, 3 * 64 * 8); hipMalloc(&o, 4096 * 64 * 8);
    hipMemcpy(d, h.data(), 3 * 64 * 8, hipMemcpyHostToDevice);
    k_check<<<1, 64>>>(d, d + 64, d + 128, o);
    std::vector<double> r(11 * 64);
    hipMemcpy(r.data(), o, 11 * 64 * 8, hipMemcpyDeviceToHost);
    for (int m = 0; m < 4; ++m) {
        int same = 0, same_plain = 0;
        for (int l = 0; l < 64; ++l) { same += r[m * 64 + l] == r[(4 + m) * 64 + l]; same_plain += r[8 * 64 + l] == r[(4 + m) * 64 + l]; }
        printf("abid %d: cbsz=2 result == quad_splat<%d> result in %d / 64 lanes (== the un-broadcast result in %d / 64)\n", m, m, same, same_plain);
    }
    int neg = 0;
    for (int l = 0; l < 64; ++l) neg += r[9 * 64 + l] == r[10 * 64 + l];
    printf("blgp = 1 (neg A): == mfma(-a, b, c) in %d / 64 lanes\n", neg);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, grid = 256 * 8;
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) k_time<0><<<grid, 64>>>(d, o, iters); else k_time<1><<<grid, 64>>>(d, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("%s: %.3f ms for %d x 8 MFMAs per wave, 2 waves / SIMD: %.1f cycles per MFMA and SIMD at 2.4 GHz\n",
                            mode ? "cbsz / abid broadcast" : "ds_swizzle splat     ", ms, iters, ms * 1e-3 * 2.4e9 / (iters * 8.0 * 2));
        }
    return 0;
}
