// k128_f64.hip -- the kernels and launch logic of num_latent = 128 in fp64 (the reference's arithmetic at num_latent 65 .. 128:
// bpmf-70 .. bpmf-128 of ci/multilatent.sh:5; see launch.h): k_sample_wg2<128, 4, double>, k_colstats_f32<128, double>, k_predict<128>
#include "launch_impl.h"
#include "kernels_wg2.h"

namespace bpmf_launch {

void k128_wg2_f64(int grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const bpmf::SampleArgs &a, const bpmf::StatRiders &r)
{
    grid += r.nblocks;                                              // (riders: ahead of the items)
    BPMF_LAUNCH((bpmf::k_sample_wg2<128, 4, double>), dim3(grid), dim3(256), st, e0, e1, a, r);
}

}  // namespace bpmf_launch

BPMF_INSTANTIATE_K(128, false)
