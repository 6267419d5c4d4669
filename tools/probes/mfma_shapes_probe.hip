// Micro-benchmark (profiling aid, not part of the product): sustained chip-wide rate of the f32 / f64 MFMA shapes a Gram could be
// built on, independent accumulators, 1 / 2 / 4 waves per SIMD.  (r01_probes.txt had found v_mfma_f64_16x16x4_f64 at ~104 cycles
// instead of its nominal 64 and the 4x4x4 shape at ~18 instead of 16: is v_mfma_f32_16x16x4_f32 -- 43-52 cycles instead of 32 --
// the best f32 shape?)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shapes_probe mfma_shapes_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f32v __attribute__((ext_vector_type(32)));
typedef double d4 __attribute__((ext_vector_type(4)));

struct S_f32_16x16x4 { typedef f4 acc; static constexpr double flops = 2.0 * 16 * 16 * 4; static constexpr const char *name = "f32 16x16x4";
    __device__ static acc op(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); } };
struct S_f32_32x32x2 { typedef f16v acc; static constexpr double flops = 2.0 * 32 * 32 * 2; static constexpr const char *name = "f32 32x32x2";
    __device__ static acc op(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); } };
struct S_f32_4x4x1 { typedef f4 acc; static constexpr double flops = 2.0 * 4 * 4 * 1 * 16; static constexpr const char *name = "f32 4x4x1 (16 blocks)";
    __device__ static acc op(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); } };
struct S_f32_16x16x1 { typedef f16v acc; static constexpr double flops = 2.0 * 16 * 16 * 1 * 4; static constexpr const char *name = "f32 16x16x1 (4 blocks)";
    __device__ static acc op(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, c, 0, 0, 0); } };
struct S_f32_32x32x1 { typedef f32v acc; static constexpr double flops = 2.0 * 32 * 32 * 1 * 2; static constexpr const char *name = "f32 32x32x1 (2 blocks)";
    __device__ static acc op(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, c, 0, 0, 0); } };
struct S_f64_16x16x4 { typedef d4 acc; static constexpr double flops = 2.0 * 16 * 16 * 4; static constexpr const char *name = "f64 16x16x4";
    __device__ static acc op(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); } };
struct S_f64_4x4x4 { typedef double acc; static constexpr double flops = 2.0 * 4 * 4 * 4 * 4; static constexpr const char *name = "f64 4x4x4 (4 blocks)";
    __device__ static acc op(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); } };

template <typename S, int NACC, typename E>
__global__ __launch_bounds__(64) void k_rate(E *out, int iters, unsigned long long *clk)
{
    typename S::acc acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = typename S::acc{};
    E a = (E)(threadIdx.x * 1e-3), b = (E)(1.0 + threadIdx.x * 1e-4);
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = S::op(a, b, acc[i]);
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    E s = 0;
    for (int i = 0; i < NACC; ++i) s += ((E *)&acc[i])[0];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <typename S, int NACC, typename E>
void run(int cus, E *out, unsigned long long *clk)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {1, 2, 4}) {
        const int grid = cus * 4 * wps;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k_rate<S, NACC, E>), dim3(grid), dim3(64), 0, 0, out, iters, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double n = (double)grid * iters * NACC;
        const double mhz = (double)clk[0] / ((double)clk[1] / 100.0);
        printf("%-24s %d acc, %d waves/SIMD: %8.3f ms  %6.1f TF  %6.1f cycles per instruction and SIMD (clock %.0f MHz)\n", S::name, NACC, wps, ms,
               n * S::flops / ms / 1e9, ms * 1e-3 * mhz * 1e6 / (iters * (double)NACC * wps), mhz);
    }
}

// cost of the cross-lane moves a 4x4x4 Gram needs for its splatted operand: ds_swizzle_b32 (LDS crossbar, no memory) and DPP moves
template <int MODE>
__global__ __launch_bounds__(64) void k_xlane(int *out, int iters)
{
    int x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) x[i] = __builtin_amdgcn_ds_swizzle(x[i], 0x13 | (1 << 7)) + 1;
            else if (MODE == 1) x[i] = __builtin_amdgcn_update_dpp(x[i], x[i], 0x124, 0xF, 0x2, false) + 1;   // row_ror:4, one bank
            else x[i] = __builtin_amdgcn_ds_bpermute((threadIdx.x ^ 4) * 4, x[i]) + 1;
        }
    }
    int s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int MODE>
void run_xlane(int cus, int *out, const char *name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps : {1, 2, 4}) {
        const int grid = cus * 4 * wps;
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k_xlane<MODE>), dim3(grid), dim3(64), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-24s 8 chains, %d waves/SIMD: %8.3f ms  %6.2f ns per instruction and CU = %5.1f cycles at 2.4 GHz (+ one v_add each)\n", name, wps, ms,
               ms * 1e6 / (iters * 8.0 * 4 * wps), ms * 1e6 / (iters * 8.0 * 4 * wps) * 2.4);
    }
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("device %s CUs %d\n", p.name, cus);
    void *out; hipMalloc(&out, 8 * 64 * cus * 4 * 4);
    unsigned long long *clk; hipHostMalloc(&clk, 16);
    run<S_f32_16x16x4, 8, float>(cus, (float *)out, clk);
    run<S_f32_32x32x2, 4, float>(cus, (float *)out, clk);
    run<S_f32_4x4x1, 8, float>(cus, (float *)out, clk);
    run<S_f32_16x16x1, 4, float>(cus, (float *)out, clk);
    run<S_f32_32x32x1, 2, float>(cus, (float *)out, clk);
    run<S_f64_16x16x4, 8, double>(cus, (double *)out, clk);
    run<S_f64_4x4x4, 8, double>(cus, (double *)out, clk);
    run_xlane<0>(cus, (int *)out, "ds_swizzle_b32");
    run_xlane<1>(cus, (int *)out, "v_mov_b32 dpp row_ror");
    run_xlane<2>(cus, (int *)out, "ds_bpermute_b32");
    return 0;
}
