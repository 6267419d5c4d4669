// k64_persist.hip -- K = 64, persistent form of the sampler (BPMF_HIP_MODE=0) (see launch.h)
#include "launch.h"
#include "kernels.h"

namespace bpmf_launch {

template <typename Kern, typename Args>
static void go(Kern kernel, int grid, int block, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const Args &a)
{
    BPMF_LAUNCH(kernel, dim3(grid), dim3(block), st, e0, e1, a);
}

void k64_persistent(int grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const bpmf::SampleArgs &a)
{
    go(bpmf::k_sample<64>, grid, 64, st, e0, e1, a);
}

}  // namespace bpmf_launch
