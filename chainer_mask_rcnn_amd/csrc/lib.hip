// Library-level entry points: error string, ABI version, device info.
#include <stdarg.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace mrcnn {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- kernel timer -------------------------------------------------------------------
thread_local ProfPending g_prof_pending;
namespace {
struct ProfRec { hipEvent_t start, stop; int kind; double flops, bytes; };
std::mutex g_prof_mu;
int g_prof_mode = 0;   // 0 off, 1 every kind timed, 2 only the dominant kind (+ ROIAlign) timed, 3 counting only
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_event_pool;
hipEvent_t get_event()
{
    if (!g_event_pool.empty()) {
        hipEvent_t e = g_event_pool.back();
        g_event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace

bool prof_timed(int kind)
{
    if (g_prof_mode == 1) return true;
    // mode 2: the forward-form 128x128 GEMM and its 256x128 / 512-thread form (W8) — together half
    // of the GPU time, profiles/*_kernel_stats.csv) — and the two ROIAlign launches of a step
    // (HBM-bound kernels the north star asks a GB/s figure for; two event pairs per step)
    return g_prof_mode == 2 && (kind == PROF_CONV_FWD_128 || kind == PROF_CONV_FWD_W8 ||
                                kind == PROF_ROI_ALIGN_FWD || kind == PROF_ROI_ALIGN_BWD);
}
bool prof_enabled(int) { return g_prof_mode != 0; }

void prof_begin(int kind, double flops, double bytes, hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    const bool timed = prof_timed(kind);   // untimed kinds still count launches / flops / bytes
    r.start = timed ? get_event() : nullptr;
    r.stop = timed ? get_event() : nullptr;
    r.kind = kind;
    r.flops = flops;
    r.bytes = bytes;
    if (timed) (void)hipEventRecord(r.start, s);
    g_prof_recs.push_back(r);
}

// Dispatch-timestamp timing: the record's events are handed to hipExtLaunchKernelGGL, which
// stamps them from the kernel's own dispatch packet — no marker packets on the stream (an
// hipEventRecord pair costs ~7 us of queue time per launch).  Null events when the kind is not
// timed in the current mode (the launch then behaves like hipLaunchKernelGGL).
void prof_begin_ext(int kind, double flops, double bytes, hipEvent_t *start, hipEvent_t *stop)
{
    *start = *stop = nullptr;
    if (g_prof_mode == 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    const bool timed = prof_timed(kind);
    r.start = timed ? get_event() : nullptr;
    r.stop = timed ? get_event() : nullptr;
    r.kind = kind;
    r.flops = flops;
    r.bytes = bytes;
    g_prof_recs.push_back(r);
    *start = r.start;
    *stop = r.stop;
}

void prof_end(hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_prof_recs.empty() && g_prof_recs.back().stop)
        (void)hipEventRecord(g_prof_recs.back().stop, s);
}
}  // namespace mrcnn

static const char *kProfNames[mrcnn::PROF_NUM_KINDS] = {
    "conv_gemm_kernel<2,2,FWD>", "conv_gemm_kernel<1,1,FWD>", "conv_gemm_kernel<2,2,DGRAD>",
    "conv_gemm_kernel<1,1,DGRAD>", "conv_gemm_kernel<2,2,WGRAD>", "conv_gemm_kernel<1,1,WGRAD>",
    "roi_align_fwd_kernel", "roi_align_bwd_kernel", "nms_mask_kernel", "nms_scan_kernel",
    "topk_rank_kernel", "sgd_kernel", "elementwise", "wino_transform_kernels",
    "conv_gemm_kernel<2,2,FWD,W8>"};

extern "C" int mrcnn_profile_enable(int on)
{
    MRCNN_REQUIRE(on >= 0 && on <= 3, "profile_enable: mode must be 0 (off), 1 (every kind timed), 2 (dominant "
                                      "kinds timed) or 3 (launches / flops / bytes counted, nothing timed), got %d", on);
    std::lock_guard<std::mutex> lk(mrcnn::g_prof_mu);
    for (auto &r : mrcnn::g_prof_recs) {
        if (r.start) mrcnn::g_event_pool.push_back(r.start);
        if (r.stop) mrcnn::g_event_pool.push_back(r.stop);
    }
    mrcnn::g_prof_recs.clear();
    mrcnn::g_prof_mode = on;
    return 0;
}

extern "C" int mrcnn_profile_num_kinds(void) { return mrcnn::PROF_NUM_KINDS; }

extern "C" const char *mrcnn_profile_kind_name(int kind)
{
    return (kind >= 0 && kind < mrcnn::PROF_NUM_KINDS) ? kProfNames[kind] : "";
}

// Sums over the launches of `kind` recorded since mrcnn_profile_enable(1).  The caller
// must have synchronised the stream(s).
extern "C" int mrcnn_profile_summary(int kind, double *total_ms, double *total_flops,
                                     double *total_bytes, int64_t *launches)
{
    std::lock_guard<std::mutex> lk(mrcnn::g_prof_mu);
    double ms = 0, fl = 0, by = 0;
    int64_t n = 0;
    for (auto &r : mrcnn::g_prof_recs) {
        if (r.kind != kind) continue;
        float t = 0.f;
        hipError_t e = r.start ? hipEventElapsedTime(&t, r.start, r.stop) : hipSuccess;
        if (e != hipSuccess) {
            mrcnn::set_error("profile_summary: %s", hipGetErrorString(e));
            return 1;
        }
        ms += t; fl += r.flops; by += r.bytes; ++n;
    }
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = fl;
    if (total_bytes) *total_bytes = by;
    if (launches) *launches = n;
    return 0;
}

extern "C" const char *mrcnn_last_error(void) { return mrcnn::g_err; }

extern "C" int mrcnn_abi_version(void) { return 1; }

extern "C" int mrcnn_device_info(int *n_cu, char *name, int name_len)
{
    int dev = 0;
    MRCNN_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    MRCNN_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (name && name_len > 0) {
        strncpy(name, prop.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    return 0;
}
