// launch.h -- what the C ABI (capi_*.hip) needs from the translation units that hold the kernels.  The templates
// are defined in launch_impl.h and explicitly instantiated one K per file (k8.hip ... k128.hip), so
// that the five instantiations of the sampler compile side by side (make -j).
#pragma once
#include "state.h"

// one kernel launch of a sampler sequence: events ride on the dispatch packet when given, the pending launch flags are consumed
// (EVERY kernel of a sampler sequence must be launched through this macro: bpmf_hip_side_kernel_resources runs the dispatch
//  logic over live factor pointers with a probe installed, and only this macro knows not to launch then)
#define BPMF_LAUNCH(kernel, grid, block, st, e0, e1, ...)                                                         \
    do {                                                                                                          \
        if (bpmf_launch::probe()) { bpmf_launch::probe()->record(reinterpret_cast<const void *>(kernel), #kernel, block); break; }   \
        const unsigned fl_ = bpmf_launch::take_flags();                                                           \
        hipEvent_t e0_ = (e0), e1_ = (e1);                                                                        \
        if (e0_ || e1_ || fl_) hipExtLaunchKernelGGL(kernel, grid, block, 0, st, e0_, e1_, fl_, __VA_ARGS__);    \
        else hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__);                                         \
    } while (0)

namespace bpmf_launch {

// "What would this side's sampler launch?" (bpmf_hip_side_kernel_resources): with a probe installed on the calling thread,
// BPMF_LAUNCH records the kernel's LDS / register budget and residency instead of launching it -- the dispatch logic of
// sampler_into answers for itself, no second copy of it to keep in step.
struct Probe {
    static constexpr int MAXK = 8;
    int n = 0;
    int64_t v[MAXK][4];          // LDS bytes per workgroup (static) | threads per workgroup | workgroups resident per CU | VGPRs (arch + acc)
    char name[MAXK][128];
    void record(const void *kernel, const char *text, dim3 block)
    {
        if (n >= MAXK) return;
        hipFuncAttributes a;
        int nb = 0;
        const int threads = (int)(block.x * block.y * block.z);
        if (hipFuncGetAttributes(&a, kernel) != hipSuccess) { (void)hipGetLastError(); return; }
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, 0) != hipSuccess) { (void)hipGetLastError(); nb = 0; }
        v[n][0] = (int64_t)a.sharedSizeBytes; v[n][1] = threads; v[n][2] = nb; v[n][3] = a.numRegs;
        snprintf(name[n], sizeof name[n], "%s", text);
        ++n;
    }
};
inline Probe *&probe() { static thread_local Probe *p = nullptr; return p; }

// hipExtAnyOrderLaunch for the NEXT sampler launch of this thread, consumed by the first kernel of its sequence: that
// launch may start while the packet ahead of it in the queue -- the statistics pass of the other side -- is still
// running (see bpmf_hip_sys_sample: "in-order head start").
inline unsigned &next_flags() { static thread_local unsigned f = 0; return f; }
inline unsigned take_flags() { unsigned &f = next_flags(); const unsigned v = f; f = 0; return v; }

// the per-column update of `self` into `out_items`, reading the parameter blob `d_in`
template <int K, bool F32>
int sampler_into(bpmf_hip_side *self, double *out_items, const bpmf_hip_side *other, int iter, double alpha, double *d_in, hipStream_t st,
                 hipEvent_t ev_start, hipEvent_t ev_stop);
// multi-GPU: every rank's fresh columns travel to the others, in place in the replicated factor matrix.
// sub < 0: the whole range of every rank; sub >= 0: sub-range `sub` of every rank (bpmf_hip_side_set_overlap:
// the exchange of one part of a side's columns runs on a stream of its own beside the sampling of the next part)
template <int K, bool F32>
int exchange(bpmf_hip_side *self, hipStream_t st, int sub);
// sum x / sum x x^T of this rank's columns (+ all-reduce), published to `out_host_dev`
template <int K, bool F32>
int stats(bpmf_hip_side *self, hipStream_t st, const double *d_in, double *out_host_dev, unsigned *flag, unsigned seq, unsigned *ticket,
          hipEvent_t ev_done = nullptr);     // ev_done: rides on the dispatch packet of the pass's last kernel (single GPU; no marker packet behind it)
template <int K, bool F32>
void predict(bpmf_hip_test *t, const bpmf_hip_side *self, const void *self_items, const void *other_items, int n, hipStream_t ps, bool beside);

// K = 64: every kernel family sits in a unit of its own (k64_*.hip) -- the instantiations
// of this size take minutes to compile.  e0 / e1: events riding on the dispatch packet, or NULL.
void k64_pf(int cls, int grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const bpmf::LrArgs &a);       // class 0..2: <= 3 | 6 | 16 ratings
void k64_pf_prepare(int grid, hipStream_t st, hipEvent_t e0, const double *S0t, const double *other_items, int64_t nrows, double *Q);

// slab form (kernels_slab.h): K = 64 fp64, one wave per work item
void k64_slab(int grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const bpmf::SampleArgs &a);
// K = 128 fp32: workgroup of two waves per item (kernels_wg2.h)
// (r: column statistics of another side as rider workgroups at the head of the grid, or r.nblocks == 0)
void k128_wg2(int grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const bpmf::SampleArgs &a, const bpmf::StatRiders &r);
// K = 128 fp64 (num_latent 65 .. 128 in the reference's arithmetic): the same form with fp64 factors, four waves per item (k128_f64.hip)
void k128_wg2_f64(int grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const bpmf::SampleArgs &a, const bpmf::StatRiders &r);

// BPMF_REDUCE formulation (kernels_reduce.h, kreduce.hip): fp64, K = 8 .. 64
int reduce_part_words(int K);                  // doubles per column of a side's `prec` array (0: K not supported)
int reduce_waves_per_simd(int K);
void reduce_precompute(int K, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const bpmf::PrecArgs &p);
void reduce_sample(int K, int grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, const bpmf::SampleArgs &a, const double *prec);

// kernels that do not depend on K (kcommon.hip)
void stage(const double *src_host_dev, double *dst, int n, hipStream_t st);
void gate_stage(int nblocks, const unsigned *gate_host_dev, unsigned want, const double *src_host_dev, double *dst, int n,
                unsigned long long *tmo, unsigned long long ticks, hipStream_t st);
void lf32_tiles(const double *LambdaF_dev, float *out, int K, hipStream_t st);    // fp32 path: LambdaF in tile layout behind the blob
void publish(const double *src, double *dst_host_dev, int n, unsigned *flag_host_dev, unsigned seq, int fail_at, hipStream_t st);
void randn_probe(uint32_t counter, int n, double *out_dev, hipStream_t st);
void aggr_add(const void *items, bool f32, int ld, int K, int64_t c0, int64_t ncols, double *mu, double *lambda, hipStream_t st);   // ld: device leading dimension, K: the caller's num_latent
void aggr_finalize(int K, int nsamples, int64_t ncols, double *mu, double *lambda, hipStream_t st);

}  // namespace bpmf_launch
