// Profiling aid: rate of v_mfma_f64_4x4x4_4b_f64 vs v_mfma_f64_16x16x4_f64 (whole chip, 4 and 8 waves/SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(64) void k44(double *out, int iters)
{
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(64) void k16(double *out, int iters)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <typename F> float timeit(F launch)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main()
{
    double *out; hipMalloc(&out, sizeof(double) * 64 * 8192);
    const int iters = 4000;
    for (int wps : {1, 4, 8}) {
        const int grid = 1024 * wps;
        float a = timeit([&] { hipLaunchKernelGGL(k44<8>, dim3(grid), dim3(64), 0, 0, out, iters); });
        float b = timeit([&] { hipLaunchKernelGGL(k16<3>, dim3(grid), dim3(64), 0, 0, out, iters); });
        float c = timeit([&] { hipLaunchKernelGGL(k16<6>, dim3(grid), dim3(64), 0, 0, out, iters); });
        printf("%d waves/SIMD: 4x4x4_4b (8 acc): %.1f cycles/instr/SIMD @2.33GHz, %.1f TF | 16x16x4 (3 acc): %.1f cycles, %.1f TF | 16x16x4 (6 acc): %.1f cycles, %.1f TF\n", wps,
               a * 1e-3 * 2.33e9 / (iters * 8.0 * wps), (double)grid * iters * 8 * 512 / a / 1e9,
               b * 1e-3 * 2.33e9 / (iters * 3.0 * wps), (double)grid * iters * 3 * 2048 / b / 1e9,
               c * 1e-3 * 2.33e9 / (iters * 6.0 * wps), (double)grid * iters * 6 * 2048 / c / 1e9);
    }
    return 0;
}
