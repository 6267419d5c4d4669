// k64_wg.hip -- K = 64 fp64, one workgroup (one wave) per column with the blocked factorisation (BPMF_HIP_MODE=2) (see launch.h)
#include "launch.h"
#include "kernels_f32.h"

namespace bpmf_launch {

template <typename Kern, typename Args>
static void go(Kern kernel, int grid, int block, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const Args &a)
{
    BPMF_LAUNCH(kernel, dim3(grid), dim3(block), st, e0, e1, a);
}

void k64_wg(int grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const bpmf::SampleArgsW<double> &a)
{
    go(bpmf::k_sample_wg<64, double, 1>, grid, 64, st, e0, e1, a);
}

}  // namespace bpmf_launch
