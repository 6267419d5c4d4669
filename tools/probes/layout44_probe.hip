// Profiling aid: operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 found by brute force:
// A = e_la, B = e_lb (unit vectors over the 64 lanes) -> which lane of D becomes 1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void k(int *out)
{
    const int la = blockIdx.x, lb = blockIdx.y, l = threadIdx.x;
    const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    if (d != 0.0) out[la * 64 + lb] = l + 1;
}
int main()
{
    int *d; hipMalloc(&d, 64 * 64 * 4); hipMemset(d, 0, 64 * 64 * 4);
    hipLaunchKernelGGL(k, dim3(64, 64), dim3(64), 0, 0, d);
    std::vector<int> h(64 * 64); hipMemcpy(h.data(), d, 64 * 64 * 4, hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb]) printf(" (B %2d -> D %2d)", lb, h[la * 64 + lb] - 1);
        printf("\n");
    }
    return 0;
}
