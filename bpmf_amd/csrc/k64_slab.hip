// k64_slab.hip -- K = 64 fp64: one wave per column, slab Cholesky on the 4x4x4 f64 MFMA (the default form) (see launch.h)
#include "launch.h"
#include "kernels_slab.h"

namespace bpmf_launch {

void k64_slab(int grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const bpmf::SampleArgs &a)
{
    BPMF_LAUNCH((bpmf::k_sample_slab<64, double>), dim3(grid), dim3(64), st, e0, e1, a);
}

}  // namespace bpmf_launch
