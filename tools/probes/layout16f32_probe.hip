// Profiling aid: operand / result layout and rate of v_mfma_f32_16x16x4_f32 (brute force, like layout44_probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k(int *out)
{
    const int la = blockIdx.x, lb = blockIdx.y, l = threadIdx.x;
    const float a = (l == la) ? 1.f : 0.f, b = (l == lb) ? 1.f : 0.f;
    const f4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, f4{0, 0, 0, 0}, 0, 0, 0);
    for (int r = 0; r < 4; ++r) if (d[r] != 0.f) out[la * 64 + lb] = (l << 2 | r) + 1;
}
__global__ __launch_bounds__(64) void rate(float *out, int iters)
{
    f4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
int main()
{
    int *d; hipMalloc(&d, 64 * 64 * 4); hipMemset(d, 0, 64 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(64, 64), dim3(64), 0, 0, d);
    std::vector<int> h(64 * 64); hipMemcpy(h.data(), d, 64 * 64 * 4, hipMemcpyDeviceToHost);
    for (int la : {0, 1, 15, 16, 17, 32, 48, 63}) {
        printf("A lane %2d:", la);
        int n = 0;
        for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] && n++ < 6) printf(" (B %2d -> D lane %2d reg %d)", lb, (h[la * 64 + lb] - 1) >> 2, (h[la * 64 + lb] - 1) & 3);
        printf("\n");
    }
    float *o; hipMalloc(&o, 4 * 64 * 8192);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 4}) {
        const int grid = 1024 * wps, iters = 4000;
        hipLaunchKernelGGL(rate, dim3(grid), dim3(64), 0, 0, o, iters); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(rate, dim3(grid), dim3(64), 0, 0, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mfma f32 16x16x4, %d waves/SIMD: %.1f TF, %.1f cycles/instr/SIMD\n", wps, (double)grid * iters * 4 * 2048 / ms / 1e9, ms * 1e-3 * 2.33e9 / (iters * 4.0 * wps));
    }
    return 0;
}
