// On which SIMD does wave w of a multi-wave workgroup land?  k_sample_wg2 gives its waves different roles (wave 0: the chain of diagonal
// blocks, VALU; the others: the MFMA tile work): if wave w of EVERY workgroup lands on SIMD w (mod 4), the MFMA-heavy waves of a CU
// share SIMDs and the chain waves share the others.   hipcc --offload-arch=gfx950 -O3 tools/probes/simd_place_probe.hip -o tools/probes/simd_place_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NW, int LDS>
__global__ __launch_bounds__(64 * NW) void k(unsigned *out, int spin)
{
    __shared__ char pad[LDS];
    pad[threadIdx.x] = 1;
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // keep the workgroup resident for a while so that the CU fills up the way it does under a real launch
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(4);
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * NW + (threadIdx.x >> 6)] = hw + (unsigned)pad[0] - 1u;
}
template <int NW, int LDS>
void run(const char *what)
{
    const int grid = 4096;
    unsigned *d; hipMalloc(&d, grid * NW * 4);
    k<NW, LDS><<<grid, 64 * NW>>>(d, 200000);
    std::vector<unsigned> h(grid * NW);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int cnt[8][4] = {};
    for (int b = 0; b < grid; ++b) for (int w = 0; w < NW; ++w) cnt[w][(h[b * NW + w] >> 4) & 3]++;
    printf("%s: SIMD of wave w over %d workgroups\n", what, grid);
    for (int w = 0; w < NW; ++w) printf("   wave %d: SIMD0 %5d  SIMD1 %5d  SIMD2 %5d  SIMD3 %5d\n", w, cnt[w][0], cnt[w][1], cnt[w][2], cnt[w][3]);
    printf("   first workgroups (simd of waves 0..):");
    for (int b = 0; b < 12; ++b) { printf("  ["); for (int w = 0; w < NW; ++w) printf("%u", (h[b * NW + w] >> 4) & 3); printf("]"); }
    printf("\n");
    hipFree(d);
}
int main()
{
    run<2, 40000>("2 waves per workgroup, 40 KB LDS (k_sample_wg2<128, 2, float>)");
    run<4, 80000>("4 waves per workgroup, 80 KB LDS (k_sample_wg2<128, 4, double>)");
    run<8, 63000>("8 waves per workgroup, 63 KB LDS (k_sample_pf)");
    return 0;
}
