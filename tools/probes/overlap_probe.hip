// Profiling aid (not part of the product): do f64 MFMA and VALU work overlap on one SIMD?
// 4 waves per SIMD; waves with (blockIdx / 1024) % 2 == 0 run MFMA chains, the others a VALU
// chain of the given flavour (mode), or every wave interleaves both (mix).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int VK>   // 0: v_fma_f64  1: v_fma_f32  2: v_add/xor u32   3: 64-bit int add/shift
__device__ __forceinline__ void valu_body(double (&x)[8], float (&f)[8], unsigned (&u)[8], unsigned long long (&q)[8])
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (VK == 0) x[i] = fma(x[i], 1.0000001, 1e-9);
        if (VK == 1) f[i] = fmaf(f[i], 1.0000001f, 1e-9f);
        if (VK == 2) u[i] = (u[i] ^ 0x9e3779b9u) + (u[i] >> 3);
        if (VK == 3) q[i] = (q[i] << 1) + (q[i] >> 7) + 12345ull;
    }
}

template <int VK>
__global__ __launch_bounds__(64) void k_split(double *out, int iters, int what)   // what: 1 = mfma waves only, 2 = valu waves only, 3 = both
{
    const bool mf = ((blockIdx.x >> 10) & 1) == 0;
    double x[8]; float f[8]; unsigned u[8]; unsigned long long q[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3 + i; f[i] = (float)x[i]; u[i] = threadIdx.x + i; q[i] = u[i]; }
    d4 acc[3];
    for (int i = 0; i < 3; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    if (mf) {
        if (what & 1)
            for (int it = 0; it < iters; ++it)
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    } else {
        if (what & 2)
            for (int it = 0; it < iters * 6; ++it) valu_body<VK>(x, f, u, q);
    }
    double s = 0;
    for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += x[i] + f[i] + u[i] + (double)q[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int VK>
__global__ __launch_bounds__(64) void k_mix(double *out, int iters, int what)     // every wave: 3 MFMA + 3*8 VALU per iteration
{
    double x[8]; float f[8]; unsigned u[8]; unsigned long long q[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3 + i; f[i] = (float)x[i]; u[i] = threadIdx.x + i; q[i] = u[i]; }
    d4 acc[3];
    for (int i = 0; i < 3; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (what & 1) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            if (what & 2) valu_body<VK>(x, f, u, q);
        }
    }
    double s = 0;
    for (int i = 0; i < 3; ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += x[i] + f[i] + u[i] + (double)q[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <typename F>
float timeit(F launch)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int VK>
void run(const char *name, double *out)
{
    const int iters = 4000, grid = 4096;
    float t[3], m[3];
    for (int w = 1; w <= 3; ++w) {
        t[w - 1] = timeit([&] { hipLaunchKernelGGL(k_split<VK>, dim3(grid), dim3(64), 0, 0, out, iters, w); });
        m[w - 1] = timeit([&] { hipLaunchKernelGGL(k_mix<VK>, dim3(grid), dim3(64), 0, 0, out, iters, w); });
    }
    printf("%-14s split waves (2 mfma + 2 valu per SIMD): mfma %.3f ms, valu %.3f ms, both %.3f ms | same wave: mfma %.3f, valu %.3f, both %.3f\n",
           name, t[0], t[1], t[2], m[0], m[1], m[2]);
}

int main()
{
    double *out; hipMalloc(&out, sizeof(double) * 64 * 4096);
    run<0>("v_fma_f64", out);
    run<1>("v_fma_f32", out);
    run<2>("u32 alu", out);
    run<3>("u64 alu", out);
    return 0;
}
