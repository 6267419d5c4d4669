// capi_internal.h -- what the translation units of the C ABI (capi_*.hip) share: the host-side helpers that cross the files.
// Round 6 cut the former capi.hip (2 400 lines) into
//   capi_context.hip   errors, RCCL entry points, host trace, bounded waits, context create / destroy / sync, normal stream
//   capi_side.hip      static work schedule, side create / destroy, factor storage, priors, launch reports
//   capi_sample.hip    sampler launches, stateless half-iteration, posterior aggregation, the stateful pipeline (bpmf_hip_sys_sample)
//   capi_comm.hip      communicator, ranges, parts, staleness, packed connectivity exchange, BPMF_REDUCE between ranks
//   capi_eval.hip      test sets and Sys::predict
// Everything here lives in namespace bpmf_capi with hidden visibility (-fvisibility=hidden): not part of the ABI.
#pragma once
#include <dlfcn.h>

#include "launch.h"

namespace bpmf_capi {

extern thread_local std::string g_err;                                 // the calling thread's last error (bpmf_hip_last_error)
extern const bool g_trace_on;                                          // BPMF_HIP_TRACE=1: host-side timeline, printed when a context dies
void trace(const char *tag, const bpmf_hip_side *s, int iter);
void trace_dump();

// bounded host-side waits (capi_context.hip): a stream / event that may carry a collective is polled with a deadline
double comm_timeout_s();
int comm_abort(bpmf_hip_ctx *c, const std::string &what);
int bounded_stream_sync(bpmf_hip_ctx *c, hipStream_t st, const char *what);
int bounded_event_sync(bpmf_hip_ctx *c, hipEvent_t ev, const char *what);
int wait_host(bpmf_hip_ctx *c);

// static work schedule of a side (capi_side.hip)
int build_schedule(bpmf_hip_side *s, const int64_t *colptr);
void free_schedule(bpmf_hip_side *s);
void pad_square(int Kt, int K, const double *src, double *dst, double diag);
void unpad_square(int Kt, int K, const double *src, double *dst);

// the stateful pipeline (capi_sample.hip)
int settle_async(bpmf_hip_side *s);                                    // waits until the worker is done with `s`; returns its deferred error
void predraw_stop(bpmf_hip_side *s);                                   // joins the side's pre-draw helper threads
int flush_pending_stats(bpmf_hip_ctx *c, bool on_main = false);        // statistics without a launch to ride in: a kernel of their own

// evaluation (capi_eval.hip)
void flush_deferred(bpmf_hip_test *t, bool on_main = false);           // enqueues an evaluation whose launch was put off

}  // namespace bpmf_capi

// KK: the instantiated num_latent; FF: the fp32 context (K = 128 only).  Uses the context `c` of the caller.
#define BPMF_DISPATCH_K(K_, ...)                                                     \
    [&]() -> int {                                                                   \
        switch (K_) {                                                                \
        case 8: { constexpr int KK = 8; constexpr bool FF = false; return __VA_ARGS__; }    \
        case 16: { constexpr int KK = 16; constexpr bool FF = false; return __VA_ARGS__; }  \
        case 32: { constexpr int KK = 32; constexpr bool FF = false; return __VA_ARGS__; }  \
        case 64: { constexpr int KK = 64; constexpr bool FF = false; return __VA_ARGS__; }  \
        case 128:                                                                    \
            if (c->dtype == BPMF_HIP_F32) { constexpr int KK = 128; constexpr bool FF = true; return __VA_ARGS__; } \
            else { constexpr int KK = 128; constexpr bool FF = false; return __VA_ARGS__; } \
        default: return fail(BPMF_HIP_EINVAL, "unsupported K");                     \
        }                                                                            \
    }()
