// k64_pf.hip -- K = 64, product form for columns with <= 16 ratings (see launch.h)
#include "launch.h"
#include "kernels_lr.h"

namespace bpmf_launch {

void k64_pf(int cls, int grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const bpmf::LrArgs &a)
{
    switch (cls) {
    case 0: BPMF_LAUNCH((bpmf::k_sample_pf<64, 3>), dim3(grid), dim3(512), st, e0, e1, a); break;
    case 1: BPMF_LAUNCH((bpmf::k_sample_pf<64, 6>), dim3(grid), dim3(512), st, e0, e1, a); break;
    default: BPMF_LAUNCH((bpmf::k_sample_pf<64, 16>), dim3(grid), dim3(512), st, e0, e1, a); break;
    }
}

void k64_pf_prepare(int grid, hipStream_t st, hipEvent_t e0, const double *S0t, const double *other_items, int64_t nrows, double *Q)
{
    BPMF_LAUNCH(bpmf::k_pf_prepare<64>, dim3(grid), dim3(512), st, e0, (hipEvent_t) nullptr, S0t, other_items, nrows, Q);
}

}  // namespace bpmf_launch
