// What sits between two dependent sampler-shaped launches?  Every workgroup records its start and end
// (wall_clock64, 100 MHz) so that, for a chain A -> B -> A -> ..., we can print per launch:
//   first start -> last start (dispatch ramp), last end of the previous launch -> first start (boundary).
// Variants: registers/LDS like k_sample1 (3 waves/SIMD) or tiny; a second stream that runs a small
// kernel behind every launch's stop event (like k_colstats); a cross-stream satisfied wait before it.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int FAT>
__global__ __launch_bounds__(64, 3) void busy(long long ticks, long long *stamps, double *sink, double *big, int traffic)
{
    __shared__ double lds[FAT ? 1280 : 8];
    const long long t0 = wall_clock64();
    double v[FAT ? 60 : 1];
#pragma unroll
    for (int i = 0; i < (FAT ? 60 : 1); ++i) v[i] = threadIdx.x + i;
    while (wall_clock64() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < (FAT ? 60 : 1); ++i) v[i] = v[i] * 1.0000001 + 1e-9;
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < (FAT ? 60 : 1); ++i) s += v[i];
    lds[threadIdx.x % 8] = s;
    if (s == 12345.678) sink[0] = lds[0];
    if (traffic) {                                                  // sampler-like traffic: gather 64 x 512 B from 16 MB, write 512 B
        double g = 0;
        for (int k = 0; k < 64; ++k) g += big[((size_t)((blockIdx.x * 977u + k * 7919u) % 32768u)) * 64 + threadIdx.x];
        big[(size_t)(32768 + blockIdx.x) * 64 + threadIdx.x] = g + s;
    }
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = wall_clock64(); }
}
template <int FATSIDE>
__global__ __launch_bounds__(64) void small(double *sink, long long ticks, long long *sstamp) { const long long t0 = wall_clock64(); if (threadIdx.x == 0) atomicMin((unsigned long long *)sstamp, (unsigned long long)t0); double w[FATSIDE ? 50 : 1]; for (int i = 0; i < (FATSIDE ? 50 : 1); ++i) w[i] = threadIdx.x * i; double v = threadIdx.x; if (ticks < 0) { for (int i = 0; i < (FATSIDE ? 50 : 1); ++i) v += w[i]; } while (wall_clock64() - t0 < ticks) v = v * 1.0000001 + 1e-9; if (v == 42.0) sink[2] = v; }

__global__ void gate(const volatile unsigned *flag, unsigned want) { if (threadIdx.x == 0) while (__hip_atomic_load((const unsigned *)flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < want) __builtin_amdgcn_s_sleep(4); }

int main(int argc, char **argv)
{
    const int G = 4873, N = 24;
    const long long ticks = 1500;                                   // 15 us per workgroup
    double *sink; CK(hipMalloc(&sink, 64)); CK(hipMemset(sink, 0, 64));
    long long *st; CK(hipMalloc(&st, (size_t)N * G * 2 * sizeof(long long)));
    hipStream_t s0, s1, s2; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi)); CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi));
    hipEvent_t done[2], other;
    for (auto &e : done) CK(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
    CK(hipEventCreateWithFlags(&other, hipEventDisableTiming | hipEventDisableSystemFence));
    CK(hipEventRecord(other, s1)); CK(hipStreamSynchronize(s1));
    std::vector<long long> h((size_t)N * G * 2);
    const int traffic = argc > 3 ? atoi(argv[3]) : 0; double *big; CK(hipMalloc(&big, (size_t)(32768 + 8192) * 64 * 8)); CK(hipMemset(big, 0, (size_t)(32768 + 8192) * 64 * 8));
    const int realdep = argc > 4 ? atoi(argv[4]) : 0; unsigned *hflag, *dflag; CK(hipHostMalloc((void **)&hflag, 64, hipHostMallocMapped)); CK(hipHostGetDevicePointer((void **)&dflag, hflag, 0)); *hflag = 0; unsigned gen = 0;
    hipEvent_t gev[2]; for (auto &e : gev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
    const int fatside = argc > 5 ? atoi(argv[5]) : 0; long long *sst; CK(hipMalloc(&sst, 64 * 8)); std::vector<long long> hs(64);
    const int side_wgs = argc > 1 ? atoi(argv[1]) : 64; const long long side_ticks = argc > 2 ? atoll(argv[2]) : 0;
    printf("side kernel: %d workgroups busy for %lld ticks (10 ns)\n", side_wgs, side_ticks);
    for (int variant = 2; variant < 4; ++variant) {
        const bool fat = variant & 1, side = variant & 2;
        CK(hipMemset(sst, 0x7f, 64 * 8));
        for (int i = 0; i < N; ++i) {
            long long *p = st + (size_t)i * G * 2;
            hipStream_t ss = (realdep && (i & 1)) ? s2 : s1;
            if (side && !realdep) CK(hipStreamWaitEvent(s0, other, 0));
            if (side && realdep) { ++gen; hipLaunchKernelGGL(gate, dim3(1), dim3(64), 0, ss, dflag, gen); CK(hipEventRecord(gev[i & 1], ss)); CK(hipStreamWaitEvent(s0, gev[i & 1], 0)); }
            if (fat) hipExtLaunchKernelGGL(busy<1>, dim3(G), dim3(64), 0, s0, nullptr, side ? done[i & 1] : nullptr, 0, ticks, p, sink, big, traffic);
            else hipExtLaunchKernelGGL(busy<0>, dim3(G), dim3(64), 0, s0, nullptr, side ? done[i & 1] : nullptr, 0, ticks, p, sink, big, traffic);
            if (side && realdep) __atomic_store_n(hflag, gen, __ATOMIC_RELEASE);      // the gate opens as soon as everything is enqueued
            if (side) { CK(hipStreamWaitEvent(ss, done[i & 1], 0)); if (fatside) hipLaunchKernelGGL(small<1>, dim3(side_wgs), dim3(64), 0, ss, sink, side_ticks, sst + i); else hipLaunchKernelGGL(small<0>, dim3(side_wgs), dim3(64), 0, ss, sink, side_ticks, sst + i); }
        }
        CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        CK(hipMemcpy(h.data(), st, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs.data(), sst, 64 * 8, hipMemcpyDeviceToHost));
        double ramp = 0, bound = 0, dur = 0, sdelay = 0; int n = 0;
        long long prev_end = 0;
        for (int i = 0; i < N; ++i) {
            long long fs = 1LL << 62, ls = 0, le = 0;
            for (int g = 0; g < G; ++g) { const long long a = h[((size_t)i * G + g) * 2], b = h[((size_t)i * G + g) * 2 + 1]; fs = std::min(fs, a); ls = std::max(ls, a); le = std::max(le, b); }
            if (i >= 4) { ramp += (ls - fs) / 100.0; bound += (fs - prev_end) / 100.0; dur += (le - fs) / 100.0; ++n; if (side && i > 4) sdelay += (hs[i - 1] - prev_end) / 100.0; }
            prev_end = le;
        }
        printf("%s kernel, %s: duration %.1f us (first start -> last end), dispatch ramp %.1f us, boundary (last end -> next first start) %.1f us; side kernel of the previous launch started %.1f us after that end\n",
               fat ? "fat (3 waves/SIMD, 10 KB LDS)" : "thin", side ? "stop event + side-stream kernel + satisfied wait" : "plain chain", dur / n, ramp / n, bound / n, sdelay / std::max(n - 1, 1));
    }
    return 0;
}
