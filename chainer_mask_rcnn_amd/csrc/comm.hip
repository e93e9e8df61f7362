// Gradient exchange over RCCL / xGMI behind the C ABI ("mrcnn_allreduce_*
// thin wrappers over RCCL, ncclComm_t created from a unique id exchanged by the Python
// launcher").  Replaces ChainerMN's communicator as the reference uses it:
//   comm = chainermn.create_communicator('hierarchical')            examples/train_common.py:97-103
//   optimizer = chainermn.create_multi_node_optimizer(optimizer, comm)              :178
// i.e. bcast_data(model) once + allreduce_grad(model) before every update.
//
// MI355X-first: one communicator per process (one process per GPU), collectives run on the
// library's OWN high-priority HIP stream so that a bucket's all-reduce overlaps with the ResNet
// backward still running on the compute stream; ordering is by HIP events only (no host sync).
// RCCL is resolved at run time from the copy already loaded into the process (PyTorch-ROCm ships
// one) so that a process never holds two RCCL runtimes; there is no link-time dependency.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t,
                              hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    int (*GetVersion)(int *) = nullptr;
    char where[256] = "";
};

std::mutex g_api_mu;
RcclApi g_api;

int load_api()
{
    std::lock_guard<std::mutex> lk(g_api_mu);
    if (g_api.handle) return 0;
    // 1. a copy that is already mapped (PyTorch's own librccl.so, or one the host linked)
    const char *resident[] = {"librccl.so", "librccl.so.1"};
    void *h = nullptr;
    for (const char *n : resident) {
        h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (h) { snprintf(g_api.where, sizeof(g_api.where), "%s (already loaded)", n); break; }
    }
    // 2. the ROCm installation
    const char *fresh[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (int i = 0; !h && i < 3; ++i) {
        h = dlopen(fresh[i], RTLD_NOW | RTLD_GLOBAL);
        if (h) snprintf(g_api.where, sizeof(g_api.where), "%s", fresh[i]);
    }
    MRCNN_REQUIRE(h, "allreduce: cannot load RCCL (librccl.so): %s", dlerror());
#define SYM(field, name)                                                         \
    g_api.field = (decltype(g_api.field))dlsym(h, name);                          \
    MRCNN_REQUIRE(g_api.field, "allreduce: RCCL symbol %s missing", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce");
    SYM(Broadcast, "ncclBroadcast");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_api.GetVersion = (decltype(g_api.GetVersion))dlsym(h, "ncclGetVersion");
    g_api.handle = h;
    return 0;
}

#define MRCNN_NCCL_TRY(expr)                                                     \
    do {                                                                         \
        ncclResult_t r_ = (expr);                                                \
        if (r_ != ncclSuccess) {                                                 \
            mrcnn::set_error("%s: %s", #expr, g_api.GetErrorString(r_));         \
            return 3;                                                            \
        }                                                                        \
    } while (0)

struct TimedBucket { hipEvent_t start, stop; int id; int64_t bytes; };

struct Comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;      // the library's collective stream
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_done = nullptr;
    int rank = 0, world = 1, device = 0;
    bool timing = false;
    std::vector<TimedBucket> timed;
    std::vector<hipEvent_t> pool;
    hipEvent_t timing_event()
    {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};

// order the collective stream after what is queued on the given producer streams
int wait_producers(Comm *c, void *after_stream, void *after_stream2)
{
    void *prod[2] = {after_stream, after_stream2};
    for (int i = 0; i < 2; ++i) {
        if (i == 1 && !prod[i]) continue;            // the first may be the NULL (default) stream
        if (i == 1 && prod[1] == prod[0]) continue;
        MRCNN_HIP_TRY(hipEventRecord(c->ev_in[i], mrcnn::as_stream(prod[i])));
        MRCNN_HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_in[i], 0));
    }
    return 0;
}

}  // namespace

extern "C" int mrcnn_allreduce_unique_id(void *id128)
{
    MRCNN_REQUIRE(id128, "allreduce_unique_id: null buffer");
    if (int rc = load_api()) return rc;
    static_assert(sizeof(ncclUniqueId) == MRCNN_COMM_ID_BYTES, "unique id size");
    MRCNN_NCCL_TRY(g_api.GetUniqueId((ncclUniqueId *)id128));
    return 0;
}

extern "C" int mrcnn_allreduce_init(const void *id128, int rank, int world, void **comm_out)
{
    MRCNN_REQUIRE(id128 && comm_out, "allreduce_init: null pointer");
    MRCNN_REQUIRE(world >= 1 && rank >= 0 && rank < world, "allreduce_init: bad rank %d / world %d",
                  rank, world);
    if (int rc = load_api()) return rc;
    Comm *c = new Comm();
    c->rank = rank;
    c->world = world;
    MRCNN_HIP_TRY(hipGetDevice(&c->device));
    int lo = 0, hi = 0;
    MRCNN_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));   // hi = numerically lowest = highest
    MRCNN_HIP_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
    for (auto &e : c->ev_in) MRCNN_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    MRCNN_HIP_TRY(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    MRCNN_NCCL_TRY(g_api.CommInitRank(&c->comm, world, id, rank));
    *comm_out = c;
    return 0;
}

extern "C" int mrcnn_allreduce_destroy(void *comm)
{
    if (!comm) return 0;
    Comm *c = (Comm *)comm;
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) (void)g_api.CommDestroy(c->comm);
    for (auto &t : c->timed) { (void)hipEventDestroy(t.start); (void)hipEventDestroy(t.stop); }
    for (auto &e : c->pool) (void)hipEventDestroy(e);
    for (auto &e : c->ev_in) (void)hipEventDestroy(e);
    (void)hipEventDestroy(c->ev_done);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

extern "C" int mrcnn_allreduce_info(void *comm, int *rank, int *world, int *rccl_version,
                                    char *library, int library_len)
{
    MRCNN_REQUIRE(comm, "allreduce_info: null communicator");
    Comm *c = (Comm *)comm;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (rccl_version) {
        *rccl_version = 0;
        if (g_api.GetVersion) (void)g_api.GetVersion(rccl_version);
    }
    if (library && library_len > 0) {
        strncpy(library, g_api.where, library_len - 1);
        library[library_len - 1] = 0;
    }
    return 0;
}

// In-place SUM of buf[0..count) over all ranks, queued on the collective stream after
// everything queued so far on after_stream (and after_stream2 if non-NULL).
extern "C" int mrcnn_allreduce_bucket(void *comm, float *buf, int64_t count, int bucket_id,
                                      void *after_stream, void *after_stream2)
{
    MRCNN_REQUIRE(comm, "allreduce_bucket: null communicator");
    MRCNN_REQUIRE(count >= 0, "allreduce_bucket: count < 0");
    if (count == 0) return 0;
    MRCNN_REQUIRE(buf, "allreduce_bucket: null buffer");
    Comm *c = (Comm *)comm;
    if (int rc = wait_producers(c, after_stream, after_stream2)) return rc;
    TimedBucket t = {nullptr, nullptr, bucket_id, count * 4};
    if (c->timing) {
        t.start = c->timing_event();
        t.stop = c->timing_event();
        MRCNN_HIP_TRY(hipEventRecord(t.start, c->stream));
    }
    MRCNN_NCCL_TRY(g_api.AllReduce(buf, buf, (size_t)count, ncclFloat32, ncclSum, c->comm, c->stream));
    if (c->timing) {
        MRCNN_HIP_TRY(hipEventRecord(t.stop, c->stream));
        c->timed.push_back(t);
    }
    return 0;
}

// The given stream waits for every collective queued so far (call before the optimizer step).
extern "C" int mrcnn_allreduce_wait(void *comm, void *stream)
{
    MRCNN_REQUIRE(comm, "allreduce_wait: null communicator");
    Comm *c = (Comm *)comm;
    MRCNN_HIP_TRY(hipEventRecord(c->ev_done, c->stream));
    MRCNN_HIP_TRY(hipStreamWaitEvent(mrcnn::as_stream(stream), c->ev_done, 0));
    return 0;
}

// Rank `root`'s bytes to every rank (bcast_data of the model before the first update); queued
// on the collective stream after after_stream, and `after_stream` then waits for it.
extern "C" int mrcnn_allreduce_broadcast(void *comm, void *buf, int64_t bytes, int root,
                                         void *after_stream)
{
    MRCNN_REQUIRE(comm, "allreduce_broadcast: null communicator");
    MRCNN_REQUIRE(bytes >= 0, "allreduce_broadcast: bytes < 0");
    if (bytes == 0) return 0;
    MRCNN_REQUIRE(buf, "allreduce_broadcast: null buffer");
    Comm *c = (Comm *)comm;
    MRCNN_REQUIRE(root >= 0 && root < c->world, "allreduce_broadcast: bad root %d", root);
    if (int rc = wait_producers(c, after_stream, nullptr)) return rc;
    MRCNN_NCCL_TRY(g_api.Broadcast(buf, buf, (size_t)bytes, ncclUint8, root, c->comm, c->stream));
    return mrcnn_allreduce_wait(comm, after_stream);
}

// Per-bucket timing of the collectives (HIP events on the collective stream).
extern "C" int mrcnn_allreduce_timing(void *comm, int enable)
{
    MRCNN_REQUIRE(comm, "allreduce_timing: null communicator");
    Comm *c = (Comm *)comm;
    for (auto &t : c->timed) { c->pool.push_back(t.start); c->pool.push_back(t.stop); }
    c->timed.clear();
    c->timing = enable != 0;
    return 0;
}

// Sums over the timed collectives with the given bucket id (-1: all): total ms, bytes, count.
// The caller must have synchronised the device.
extern "C" int mrcnn_allreduce_bucket_times(void *comm, int bucket_id, double *total_ms,
                                            double *total_bytes, int64_t *launches)
{
    MRCNN_REQUIRE(comm, "allreduce_bucket_times: null communicator");
    Comm *c = (Comm *)comm;
    double ms = 0, by = 0;
    int64_t n = 0;
    for (auto &t : c->timed) {
        if (bucket_id >= 0 && t.id != bucket_id) continue;
        float f = 0.f;
        MRCNN_HIP_TRY(hipEventElapsedTime(&f, t.start, t.stop));
        ms += f; by += (double)t.bytes; ++n;
    }
    if (total_ms) *total_ms = ms;
    if (total_bytes) *total_bytes = by;
    if (launches) *launches = n;
    return 0;
}
