// Micro-benchmark (profiling aid, not part of the product): sustained rate of v_mfma_f64_16x16x4_f64
// and of v_fma_f64 on the whole chip, at 1..4 waves per SIMD, plus the shader clock seen under load.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(64) void k_mfma(double *out, int iters, unsigned long long *clk)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

__global__ __launch_bounds__(64) void k_fma(double *out, int iters)
{
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
    const double a = 1.0000001, b = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = fma(x[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("device %s CUs %d clock %d kHz\n", p.name, cus, p.clockRate);
    double *out; hipMalloc(&out, sizeof(double) * 64 * cus * 4 * 8);
    unsigned long long *clk; hipHostMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {
        const int grid = cus * 4 * wps;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mfma<3>, dim3(grid), dim3(64), 0, 0, out, iters, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n = (double)grid * iters * 3;
            if (rep) printf("mfma f64 16x16x4, 3 acc, %d waves/SIMD: %.3f ms, %.1f TF, %.1f shader cycles/MFMA/SIMD (clock %.0f MHz)\n", wps, ms,
                            n * 2048 / ms / 1e9, (double)clk[0] / (iters * 3.0 * wps), (double)clk[0] / ((double)clk[1] / 100.0));
        }
    }
    {
        const int grid = cus * 4 * 1;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma<1>, dim3(grid), dim3(64), 0, 0, out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mfma f64 dependent chain (1 acc), 1 wave/SIMD: %.1f shader cycles per MFMA\n", (double)clk[0] / iters);
    }
    for (int wps : {1, 2, 4}) {
        const int grid = cus * 4 * wps;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_fma, dim3(grid), dim3(64), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("v_fma_f64, 8 chains, %d waves/SIMD: %.3f ms, %.1f TF\n", wps, ms, (double)grid * iters * 8 * 64 * 2 / ms / 1e9);
    }
    return 0;
}
