// k128_wg2.hip -- K = 128 fp32 factors: workgroup per item, diagonal blocks factored + inverted on the 4x4x4 f64 MFMA,
// panel / solves as products with the inverted blocks, heavy columns chunked (the default form; see launch.h)
#include "launch.h"
#include "kernels_wg2.h"

namespace bpmf_launch {

void k128_wg2(int grid, int nwaves, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const bpmf::SampleArgs &a, const bpmf::StatRiders &r)
{
    grid += r.nblocks;                                              // (riders: ahead of the items, or -- tail -- behind them)
    if (a.stamps) {                                                 // profiling: what the runtime says about residency
        static bool said = false;
        if (!said) {
            said = true;
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bpmf::k_sample_wg2<128, 2>, 128, 0);
            fprintf(stderr, "[bpmf_hip] k_sample_wg2<128,2>: %d workgroups per CU (runtime), grid %d\n", nb, grid);
        }
    }
    if (nwaves == 4) {
        BPMF_LAUNCH((bpmf::k_sample_wg2<128, 4>), dim3(grid), dim3(256), st, e0, e1, a, r);
    } else {
        BPMF_LAUNCH((bpmf::k_sample_wg2<128, 2>), dim3(grid), dim3(128), st, e0, e1, a, r);
    }
}

}  // namespace bpmf_launch
