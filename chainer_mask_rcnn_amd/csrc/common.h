// Shared helpers for the gfx950 kernels of libmrcnn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>

#include "mrcnn_hip.h"

namespace mrcnn {

void set_error(const char *fmt, ...);

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return 1;
    }
    return 0;
}

#define MRCNN_HIP_TRY(expr)                                              \
    do {                                                                 \
        hipError_t e_ = (expr);                                          \
        if (e_ != hipSuccess) {                                          \
            mrcnn::set_error("%s: %s", #expr, hipGetErrorString(e_));    \
            return 1;                                                    \
        }                                                                \
    } while (0)

#define MRCNN_REQUIRE(cond, ...)                                         \
    do {                                                                 \
        if (!(cond)) {                                                   \
            mrcnn::set_error(__VA_ARGS__);                               \
            return 2;                                                    \
        }                                                                \
    } while (0)

static inline hipStream_t as_stream(void *s) { return (hipStream_t)s; }

// ---- in-library kernel timer (HIP events on the launch stream) -----------------
// bench.py enables it around the timed region to obtain, per kernel kind, the
// average launch duration and the algorithmic flops / bytes (roofline numbers).
enum ProfKind {
    PROF_CONV_FWD_128 = 0, PROF_CONV_FWD_64, PROF_CONV_DGRAD_128, PROF_CONV_DGRAD_64,
    PROF_CONV_WGRAD_128, PROF_CONV_WGRAD_64, PROF_ROI_ALIGN_FWD, PROF_ROI_ALIGN_BWD,
    PROF_NMS_MASK, PROF_NMS_SCAN, PROF_TOPK, PROF_SGD, PROF_ELEMENTWISE, PROF_WINO_TRANSFORM,
    PROF_CONV_FWD_W8,        // conv_gemm_kernel<2,2,FWD,...,W8>: 256x128 tiles on 512-thread workgroups
    PROF_NUM_KINDS
};
bool prof_enabled(int kind);
void prof_begin(int kind, double flops, double bytes, hipStream_t s);
void prof_end(hipStream_t s);
void prof_begin_ext(int kind, double flops, double bytes, hipEvent_t *start, hipEvent_t *stop);
struct ProfScope {
    hipStream_t s_;
    bool on_;
    ProfScope(int kind, double flops, double bytes, hipStream_t s) : s_(s), on_(prof_enabled(kind))
    {
        if (on_) prof_begin(kind, flops, bytes, s);
    }
    ~ProfScope()
    {
        if (on_) prof_end(s_);
    }
};

// Kernel-only timing of the launches issued inside the scope: launch sites call prof_take()
// and pass the two events to hipExtLaunchKernelGGL (start = first launch of the scope, stop =
// its `launches`-th); both are null when the kind is not timed.
struct ProfPending { hipEvent_t ev0 = nullptr, ev1 = nullptr; int remaining = 0; };
extern thread_local ProfPending g_prof_pending;
struct ProfKernelScope {
    ProfKernelScope(int kind, double flops, double bytes, int launches = 1)
    {
        prof_begin_ext(kind, flops, bytes, &g_prof_pending.ev0, &g_prof_pending.ev1);
        g_prof_pending.remaining = launches;
    }
    ~ProfKernelScope() { g_prof_pending = ProfPending(); }
};
static inline void prof_take(hipEvent_t *start, hipEvent_t *stop)
{
    *start = g_prof_pending.ev0;
    g_prof_pending.ev0 = nullptr;
    *stop = nullptr;
    if (--g_prof_pending.remaining <= 0) {
        *stop = g_prof_pending.ev1;
        g_prof_pending.ev1 = nullptr;
    }
}

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// roi_align.hip's share of mrcnn_set_tuning ("roi_fwd_lanes", "roi_bwd_lanes"); 0 = name handled
int roi_align_set_tuning(const char *name, int value);

}  // namespace mrcnn
