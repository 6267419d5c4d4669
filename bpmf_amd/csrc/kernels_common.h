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

// fp32 path (K = 128): LambdaF as fp32 in the accumulator layout of the sampler's 16 x 16 tiles -- tile (I <= J) of the
// upper block triangle, lane (kq, li), register reg <-> LambdaF(16 J + li, 16 I + 4 kq + reg) (the lower-triangle entry:
// what LLT reads, c++/sample.cpp:306) -- so that a column adds its prior with one 16-byte load per tile and lane
// instead of four strided fp64 loads.  One single-wave workgroup per tile, once per half-iteration behind the staging.
__global__ __launch_bounds__(64) void k_lf32_tiles(const double *__restrict__ LF, float *__restrict__ out, int K)
{
    const int NT = K / 16;
    int t = blockIdx.x, I = 0;
    while (t >= NT - I) { t -= NT - I; ++I; }
    const int J = I + t, lane = threadIdx.x, kq = lane >> 4, li = lane & 15;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) v[reg] = (float)LF[(16 * J + li) + (size_t)(16 * I + 4 * kq + reg) * K];
    reinterpret_cast<f4 *>(out)[(size_t)blockIdx.x * 64 + lane] = v;
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

// ---------------------------------------------------------------------------
// Posterior aggregation for the -o outputs (c++/sample.cpp:364-368, c++/bpmf.cpp:281-295), on the device: the
// reference adds r and r r^T of every post-burn-in sample into aggrMu (K x N) / aggrLambda (K*K x N) and inverts one
// K x K covariance per column at the end.  One workgroup per column; K is a run-time argument.
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_aggr_add(const T *__restrict__ items, int ld, int K, int64_t c0, double *__restrict__ mu, double *__restrict__ lambda)
{
    // (ld: leading dimension of the factor matrix on the device -- the instantiated K; K: the caller's num_latent)
    const int64_t c = blockIdx.x;                                   // local column
    const T *x = items + (size_t)(c0 + c) * ld;
    double *l = lambda + (size_t)c * K * K;
    for (int e = threadIdx.x; e < K * K; e += 256) l[e] += (double)x[e % K] * (double)x[e / K];     // column-major K x K: (i, j) at i + j K
    if ((int)threadIdx.x < K) mu[(size_t)c * K + threadIdx.x] += (double)x[threadIdx.x];
    for (int e = 256 + (int)threadIdx.x; e < K; e += 256) mu[(size_t)c * K + e] += (double)x[e];
}

// cov = (prod - sum sum^T / n) / (n - 1); Lambda = cov^-1 (in-place Gauss-Jordan with partial pivoting, the matrix
// stays in global memory -- a column's K x K block is L2-resident while its workgroup works on it); mu = sum / n.
// A singular covariance (n <= K samples) gives NaN like the reference's inverse of a singular matrix gives inf / NaN.
__global__ __launch_bounds__(256) void k_aggr_finalize(int K, int nsamples, double *__restrict__ mu, double *__restrict__ lambda)
{
    __shared__ int piv[256];
    __shared__ double red_v[256];
    __shared__ int red_i[256];
    __shared__ int singular;
    const int64_t c = blockIdx.x;
    const int tid = threadIdx.x;
    double *a = lambda + (size_t)c * K * K;
    double *m = mu + (size_t)c * K;
    auto A = [&](int r, int cc) -> double & { return a[(size_t)cc * K + r]; };
    for (int e = tid; e < K * K; e += 256) {
        const int i = e % K, j = e / K;
        a[e] = (a[e] - (m[i] * m[j] / nsamples)) / (nsamples - 1);
    }
    if (tid == 0) singular = 0;
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        // pivot: largest |A(r, k)|, r >= k
        double best = -1.0; int bi = k;
        for (int r = k + tid; r < K; r += 256) { const double v = fabs(A(r, k)); if (v > best) { best = v; bi = r; } }
        red_v[tid] = best; red_i[tid] = bi;
        __syncthreads();
        for (int st = 128; st >= 1; st >>= 1) {
            if (tid < st && (red_v[tid + st] > red_v[tid] || (red_v[tid + st] == red_v[tid] && red_i[tid + st] < red_i[tid]))) { red_v[tid] = red_v[tid + st]; red_i[tid] = red_i[tid + st]; }
            __syncthreads();
        }
        const int p = red_i[0];
        if (tid == 0) { piv[k] = p; if (!(red_v[0] > 0.0)) singular = 1; }
        if (p != k) for (int j = tid; j < K; j += 256) { const double t = A(k, j); A(k, j) = A(p, j); A(p, j) = t; }
        __syncthreads();
        if (singular) break;
        const double d = A(k, k);
        __syncthreads();
        for (int j = tid; j < K; j += 256) A(k, j) = (j == k) ? 1.0 / d : A(k, j) / d;
        __syncthreads();
        // eliminate column k from the other rows: thread (r, j) tiles over the matrix; f = A(r, k) is read before it is overwritten
        for (int r0 = 0; r0 < K; r0 += 256 / 32) {                   // 8 rows at a time, 32 column lanes
            const int r = r0 + (tid >> 5);
            double f = 0.0;
            if (r < K && r != k) f = A(r, k);
            __syncthreads();
            if (r < K && r != k) {
                for (int j = (tid & 31); j < K; j += 32) A(r, j) = (j == k) ? -f * A(k, k) : A(r, j) - f * A(k, j);
            }
            __syncthreads();
        }
    }
    if (singular) {
        for (int e = tid; e < K * K; e += 256) a[e] = __builtin_nan("");
    } else {
        // undo the row interchanges as column interchanges, in reverse order
        for (int k = K - 1; k >= 0; --k) {
            const int p = piv[k];
            if (p != k) for (int r = tid; r < K; r += 256) { const double t = A(r, k); A(r, k) = A(r, p); A(r, p) = t; }
            __syncthreads();
        }
    }
    for (int e = tid; e < K; e += 256) m[e] /= nsamples;
}

}  // namespace bpmf
