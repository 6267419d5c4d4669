// k128_wg2.hip -- K = 128 fp32 factors: workgroup of two waves per item, diagonal blocks factored + inverted on the 4x4x4 f64 MFMA,
// panel / solves as products with the inverted blocks, heavy columns chunked (see launch.h)
#include "launch.h"
#include "kernels_wg2.h"

namespace bpmf_launch {

void k128_wg2(int grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const bpmf::SampleArgs &a, const bpmf::StatRiders &r)
{
    grid += r.nblocks;                                              // (riders: ahead of the items)
    BPMF_LAUNCH((bpmf::k_sample_wg2<128, 2>), dim3(grid), dim3(128), st, e0, e1, a, r);
}

}  // namespace bpmf_launch
