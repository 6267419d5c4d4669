// kernels_common.h -- the kernels that do not depend on K (one translation unit: kcommon.hip).
#pragma once
#include "kernels.h"

namespace bpmf {

// hp.mu / hp.LambdaF blob: pinned host memory -> device memory (replaces a hipMemcpyAsync;
// the sampler re-reads LambdaF per column, so it must sit behind the L2)
__global__ __launch_bounds__(256) void k_stage(const double *__restrict__ src_host, double *__restrict__ dst, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src_host[i];
}

__global__ __launch_bounds__(64) void k_gate_stage(const unsigned *gate_host, unsigned want, const double *src_host,
                                                   double *__restrict__ dst, int n, unsigned long long *tmo, unsigned long long wait_ticks)
{
    gate_stage_body((int)blockIdx.x, (int)gridDim.x, gate_host, want, src_host, dst, n, nullptr, 0u, tmo, wait_ticks);
}

// multi-GPU: the all-reduced sums sit in device memory; copy them to the pinned result blob and
// publish the sequence number behind them.  fail_at >= 0: src[fail_at] is the summed "failed
// column + 1" word of k_colstats (0 = no rank failed; with several failing ranks the id is only a
// witness that something failed) and becomes the u64 word behind it.
__global__ __launch_bounds__(256) void k_publish(const double *__restrict__ src, double *__restrict__ dst_host, int n,
                                                 unsigned *flag_host, unsigned seq, int fail_at)
{
    for (int i = threadIdx.x; i < n; i += 256) dst_host[i] = src[i];
    if (fail_at >= 0 && threadIdx.x == 0) {
        const double d = src[fail_at];
        reinterpret_cast<unsigned long long *>(dst_host)[fail_at + 1] = (d == 0.0) ? ~0ull : (unsigned long long)(d - 1.0);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// test probe: the first n normals of stream `counter`
__global__ __launch_bounds__(64) void k_randn_probe(uint32_t counter, int n, double *out)
{
    __shared__ double z[128];
    draw_normals<128>(counter, n, z, threadIdx.x);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 64) out[i] = z[i];
}

}  // namespace bpmf
