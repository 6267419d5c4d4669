// kreduce.hip -- the BPMF_REDUCE formulation (kernels_reduce.h), every fp64 num_latent (see launch.h)
#include "launch.h"
#include "kernels_reduce.h"

namespace bpmf_launch {

template <typename Kern, typename... Args>
static void go(Kern kernel, int grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, Args... a)
{
    BPMF_LAUNCH(kernel, dim3(grid), dim3(64), st, e0, e1, a...);
}

int reduce_part_words(int K)
{
    switch (K) {
    case 8: return bpmf::Geo<8>::PART;
    case 16: return bpmf::Geo<16>::PART;
    case 32: return bpmf::Geo<32>::PART;
    case 64: return bpmf::Geo<64>::PART;
    default: return 0;
    }
}

int reduce_waves_per_simd(int K) { return K <= 32 ? bpmf::Geo<32>::WPS : bpmf::Geo<64>::WPS; }

void reduce_precompute(int K, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const bpmf::PrecArgs &p)
{
    if (p.ncols <= 0) return;
    switch (K) {
    case 8: go(bpmf::k_precompute<8>, (int)p.ncols, st, e0, e1, p); break;
    case 16: go(bpmf::k_precompute<16>, (int)p.ncols, st, e0, e1, p); break;
    case 32: go(bpmf::k_precompute<32>, (int)p.ncols, st, e0, e1, p); break;
    case 64: go(bpmf::k_precompute<64>, (int)p.ncols, st, e0, e1, p); break;
    }
}

void reduce_sample(int K, int grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const bpmf::SampleArgs &a, const double *prec)
{
    if (a.nwork <= 0) return;
    switch (K) {
    case 8: go(bpmf::k_sample_prec<8>, grid, st, e0, e1, a, prec); break;
    case 16: go(bpmf::k_sample_prec<16>, grid, st, e0, e1, a, prec); break;
    case 32: go(bpmf::k_sample_prec<32>, grid, st, e0, e1, a, prec); break;
    case 64: go(bpmf::k_sample_prec<64>, grid, st, e0, e1, a, prec); break;
    }
}

}  // namespace bpmf_launch
