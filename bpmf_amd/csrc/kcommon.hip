// kcommon.hip -- launchers of the kernels that do not depend on K (see launch.h).
#include "launch.h"
#include "kernels_common.h"

namespace bpmf_launch {

void stage(const double *src_host_dev, double *dst, int n, hipStream_t st)
{
    hipLaunchKernelGGL(bpmf::k_stage, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src_host_dev, dst, n);
}

void gate_stage(int nblocks, const unsigned *gate_host_dev, unsigned want, const double *src_host_dev, double *dst, int n,
                unsigned long long *tmo, unsigned long long ticks, hipStream_t st)
{
    hipLaunchKernelGGL(bpmf::k_gate_stage, dim3(nblocks), dim3(64), 0, st, gate_host_dev, want, src_host_dev, dst, n, tmo, ticks);
}

void lf32_tiles(const double *LambdaF_dev, float *out, int K, hipStream_t st)
{
    const int nt = K / 16;
    hipLaunchKernelGGL(bpmf::k_lf32_tiles, dim3((unsigned)(nt * (nt + 1) / 2)), dim3(64), 0, st, LambdaF_dev, out, K);
}

void publish(const double *src, double *dst_host_dev, int n, unsigned *flag_host_dev, unsigned seq, int fail_at, hipStream_t st)
{
    hipLaunchKernelGGL(bpmf::k_publish, dim3(1), dim3(256), 0, st, src, dst_host_dev, n, flag_host_dev, seq, fail_at);
}

void randn_probe(uint32_t counter, int n, double *out_dev, hipStream_t st)
{
    hipLaunchKernelGGL(bpmf::k_randn_probe, dim3(1), dim3(64), 0, st, counter, n, out_dev);
}

void aggr_add(const void *items, bool f32, int ld, int K, int64_t c0, int64_t ncols, double *mu, double *lambda, hipStream_t st)
{
    if (ncols <= 0) return;
    if (f32) hipLaunchKernelGGL(bpmf::k_aggr_add<float>, dim3((unsigned)ncols), dim3(256), 0, st, (const float *)items, ld, K, c0, mu, lambda);
    else hipLaunchKernelGGL(bpmf::k_aggr_add<double>, dim3((unsigned)ncols), dim3(256), 0, st, (const double *)items, ld, K, c0, mu, lambda);
}

void aggr_finalize(int K, int nsamples, int64_t ncols, double *mu, double *lambda, hipStream_t st)
{
    if (ncols > 0) hipLaunchKernelGGL(bpmf::k_aggr_finalize, dim3((unsigned)ncols), dim3(256), 0, st, K, nsamples, mu, lambda);
}

}  // namespace bpmf_launch
