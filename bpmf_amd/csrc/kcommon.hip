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

void publish(const double *src, double *dst_host_dev, int n, unsigned *flag_host_dev, unsigned seq, int fail_at, hipStream_t st)
{
    hipLaunchKernelGGL(bpmf::k_publish, dim3(1), dim3(256), 0, st, src, dst_host_dev, n, flag_host_dev, seq, fail_at);
}

void randn_probe(uint32_t counter, int n, double *out_dev, hipStream_t st)
{
    hipLaunchKernelGGL(bpmf::k_randn_probe, dim3(1), dim3(64), 0, st, counter, n, out_dev);
}

}  // namespace bpmf_launch
