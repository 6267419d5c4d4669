// kernels_reduce.h -- the BPMF_REDUCE formulation of the reference (c++/sample.cpp:234-246 preComputeMuLambda,
// :289-291 the sampler reading precMu / precLambda, c++/mpi_reduce.h:24-47 the reduction onto the owners).
//
// Reference: after a side S has sampled its local columns, `other.preComputeMuLambda(S)` computes, for EVERY column j
// of the other side O, the part of its Gram and rhs that comes from the rows this rank owns of S
// (computeMuLambda(..., local_only = true)); before O is sampled the parts of all ranks are summed onto the owner of
// each column (MPI_Reduce per owner) and the sampler adds the prior to the sum instead of walking the ratings.
// No factor travels for the sampling; the price is K^2 + K doubles per column of both sides on every rank.
//
// Here: `prec` of a side is ncols x Geo<K>::PART doubles: per column the upper 16 x 16 tiles of sum u u^T in the
// accumulator layout of the 16x16x4 f64 MFMA followed by the rhs sums -- the very partial a chunk of a heavy column
// writes in k_sample (kernels.h), so the sampler below is k_sample with "load the partial" in place of the Gram.
//   k_precompute<K>:  one wave per column j of O, over the TRANSPOSE of this rank's block of S's ratings (built once
//                     by the host: capi_comm.hip set_reduce), longest columns first
//   k_sample_prec<K>: persistent waves, C = 64 / K columns factorised side by side (deposit_column + finish_slots)
#pragma once
#include "kernels.h"

namespace bpmf {

template <int K>
__global__ __launch_bounds__(64, Geo<K>::WPS) void k_precompute(PrecArgs p)
{
    constexpr int NT = Geo<K>::NT, NTRI = Geo<K>::NTRI, PART = Geo<K>::PART;
    const int lane = threadIdx.x;
    const int64_t j = p.order[blockIdx.x];
    const int64_t p0 = p.t_colptr[j];
    const int len = (int)(p.t_colptr[j + 1] - p0);
    d4 acc[NTRI];
    double r[NT];
#pragma unroll
    for (int t = 0; t < NTRI; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < NT; ++t) r[t] = 0.0;
    gram_chunk<K>(p.t_rowidx + p0, p.t_vals + p0, len, p.s_items, p.zero_row, p.mean_rating, p.alpha, acc, r, lane);
    double *out = p.prec + (size_t)j * PART;
#pragma unroll
    for (int t = 0; t < NTRI; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) out[(t * 4 + reg) * 64 + lane] = acc[t][reg];
    if (lane < 16) {
#pragma unroll
        for (int t = 0; t < NT; ++t) out[NTRI * 256 + t * 16 + lane] = r[t];
    }
}

// SampleArgs: items, col_from, LambdaF, Lmu, mu, prop_lambda, diag_only, fail, alpha, iter_plus_1 are read; nwork = local columns
template <int K>
__global__ __launch_bounds__(64, Geo<K>::WPS) void k_sample_prec(SampleArgs a, const double *__restrict__ prec)
{
    constexpr int NT = Geo<K>::NT, NTRI = Geo<K>::NTRI, PART = Geo<K>::PART, C = Geo<K>::C;
    __shared__ __attribute__((aligned(16))) double lds[Geo<K>::LDS_WORDS];
    int nfilled = 0;
    for (long long col = blockIdx.x; col < a.nwork; col += gridDim.x) {
        int lane = threadIdx.x;
        asm volatile("" : "+v"(lane));                               // (see k_sample: keeps lane-derived addresses out of the loop's live set)
        const int64_t idx = a.col_from + col;
        const double *pc = prec + (size_t)idx * PART;
        d4 acc[NTRI];
        double r[NT];
#pragma unroll
        for (int t = 0; t < NTRI; ++t)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) acc[t][reg] = pc[(t * 4 + reg) * 64 + lane];
#pragma unroll
        for (int t = 0; t < NT; ++t) r[t] = pc[NTRI * 256 + t * 16 + (lane & 15)];
        deposit_column<K>(a, idx, acc, r, lds, nfilled, lane);
        if (++nfilled == C) {
            finish_slots<K>(a, lds, C, lane);
            nfilled = 0;
        }
    }
    if (nfilled > 0) finish_slots<K>(a, lds, nfilled, threadIdx.x);
}

}  // namespace bpmf
