// Kernel-boundary cost on one stream: N dependent launches of a kernel shaped like the sampler
// (G single-wave workgroups, each busy for ~T cycles), in four flavours:
//   plain launches | + a satisfied cross-stream event wait before each | + a stop event riding on the
//   dispatch packet (hipExtLaunchKernelGGL) | + a marker (hipEventRecord) after each
// prints the per-launch wall time; the kernel's own duration (events around ONE launch) is the baseline.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64, 3) void busy(long long cycles, double *out)
{
    const long long t0 = wall_clock64();                 // 100 MHz
    double v = threadIdx.x;
    while (wall_clock64() - t0 < cycles) v = v * 1.0000001 + 1e-9;
    if (v == 12345.678) out[0] = v;
}

int main(int argc, char **argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 4873, N = 400;
    const long long ticks = argc > 2 ? atoll(argv[2]) : 1000;     // 10 us per workgroup at 100 MHz
    double *d; CK(hipMalloc(&d, 64));
    hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t a, b, done, other;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventCreateWithFlags(&done, hipEventDisableTiming | hipEventDisableSystemFence));
    CK(hipEventCreateWithFlags(&other, hipEventDisableTiming | hipEventDisableSystemFence));
    CK(hipEventRecord(other, s1)); CK(hipStreamSynchronize(s1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(busy, dim3(G), dim3(64), 0, s0, ticks, d);
    CK(hipStreamSynchronize(s0));
    CK(hipEventRecord(a, s0)); hipLaunchKernelGGL(busy, dim3(G), dim3(64), 0, s0, ticks, d); CK(hipEventRecord(b, s0));
    CK(hipStreamSynchronize(s0));
    float one = 0; CK(hipEventElapsedTime(&one, a, b));
    printf("G=%d ticks=%lld: one launch between events %.1f us\n", G, ticks, one * 1e3);
    for (int mode = 0; mode < 5; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipStreamSynchronize(s0));
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                if (mode == 1 || mode == 4) CK(hipStreamWaitEvent(s0, other, 0));
                if (mode == 2 || mode == 4) hipExtLaunchKernelGGL(busy, dim3(G), dim3(64), 0, s0, nullptr, done, 0, ticks, d);
                else hipLaunchKernelGGL(busy, dim3(G), dim3(64), 0, s0, ticks, d);
                if (mode == 3) CK(hipEventRecord(done, s0));
                if (mode == 4) CK(hipStreamWaitEvent(s1, done, 0));
            }
            CK(hipStreamSynchronize(s0));
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
            if (rep) printf("mode %d (%s): %.1f us per launch\n", mode,
                            mode == 0 ? "plain" : mode == 1 ? "+satisfied event wait" : mode == 2 ? "+ext stop event" : mode == 3 ? "+marker after" : "+wait, ext stop event, other stream waits on it",
                            us);
        }
    }
    return 0;
}
