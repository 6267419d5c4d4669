// Profiling aid (not part of the product): where does the hardware dispatcher put single-wave
// workgroups?  Prints, for the first workgroups of a 6040-block launch, the XCC / SE / CU / SIMD
// they ran on.  build: hipcc --offload-arch=gfx950 -O3 -o dispatch_probe dispatch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>

__global__ __launch_bounds__(64, 4) void k_probe(unsigned *out, int spin)
{
    __shared__ double pad[1218];                       // 9744 B like k_sample1<32>
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = wall_clock64();
    double x = threadIdx.x;
    while (wall_clock64() - t0 < (unsigned long long)spin) x = x * 1.0000001 + 1e-9;
    pad[threadIdx.x] = x;
    if (threadIdx.x == 0) { out[blockIdx.x * 4] = hw; out[blockIdx.x * 4 + 1] = xcc; out[blockIdx.x * 4 + 2] = (unsigned)t0; out[blockIdx.x * 4 + 3] = (unsigned)pad[0]; }
}

int main()
{
    const int n = 6040;
    unsigned *d; hipMalloc(&d, n * 16);
    std::vector<unsigned> h(n * 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_probe, dim3(n), dim3(64), 0, 0, d, 2000);   // 20 us each
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, n * 16, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    std::map<unsigned, std::vector<int>> per_simd;
    unsigned tmin = ~0u;
    for (int i = 0; i < n; ++i) tmin = h[i * 4 + 2] < tmin ? h[i * 4 + 2] : tmin;
    for (int i = 0; i < n; ++i) {
        const unsigned hw = h[i * 4], xcc = h[i * 4 + 1] & 0xF;
        const unsigned wave = hw & 0xF, simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        if (i < 48 || (i % 1024) < 4) printf("wg %4d: xcc %u se %u sh %u cu %2u simd %u wave %u  t0 %+d\n", i, xcc, se, sh, cu, simd, wave, (int)(h[i * 4 + 2] - tmin));
        per_simd[(xcc << 16) | (se << 12) | (sh << 10) | (cu << 4) | simd].push_back(i);
    }
    printf("distinct SIMDs used: %zu\n", per_simd.size());
    int shown = 0;
    for (auto &kv : per_simd) {
        if (shown++ >= 6) break;
        printf("simd %05x:", kv.first);
        for (int w : kv.second) printf(" %d", w);
        printf("\n");
    }
    return 0;
}
