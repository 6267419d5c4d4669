// k64_lr.hip -- K = 64, reflector sweeps for columns with a handful of ratings (see launch.h)
#include "launch.h"
#include "kernels_lr.h"

namespace bpmf_launch {

template <typename Kern, typename Args>
static void go(Kern kernel, int grid, int block, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const Args &a)
{
    BPMF_LAUNCH(kernel, dim3(grid), dim3(block), st, e0, e1, a);
}

void k64_lr(int width, int grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const bpmf::LrArgs &a)
{
    switch (width) {
    case 1: go(bpmf::k_sample_lr<64, 1>, grid, 64, st, e0, e1, a); break;
    case 2: go(bpmf::k_sample_lr<64, 2>, grid, 64, st, e0, e1, a); break;
    case 3: go(bpmf::k_sample_lr<64, 3>, grid, 64, st, e0, e1, a); break;
    default: go(bpmf::k_sample_lr<64, 4>, grid, 64, st, e0, e1, a); break;
    }
}

}  // namespace bpmf_launch
