// Proposal path for gfx950: loc2bbox + clip, stable descending top-k, row gather.
//
// Replaces the host-side NumPy steps of chainercv's ProposalCreator (an
// un-vendored dependency of the reference; call site
// /root/reference/chainer_mask_rcnn/models/region_proposal_network.py:135-138,
// algorithm: SURVEY.md Appendix A.4), which copies loc/score/anchor to the CPU
// for every image.  Everything here stays on the device; integer results
// (order, counts) are bit-exact w.r.t. oracle/np_ref.py.
//
// Built with -ffp-contract=off: the box arithmetic is the same sequence of
// separately rounded fp32 operations NumPy performs.
#include <algorithm>

#include <type_traits>

#include "common.h"

namespace {

__global__ void decode_clip_kernel(const float4 *__restrict__ anchor,
                                   const float4 *__restrict__ loc, float4 *__restrict__ roi,
                                   uint8_t *__restrict__ valid, int n, float img_h, float img_w,
                                   float min_size)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = anchor[i];  // (y1, x1, y2, x2)
    const float4 l = loc[i];     // (dy, dx, dh, dw)
    const float h = a.z - a.x, w = a.w - a.y;
    const float cy = a.x + 0.5f * h, cx = a.y + 0.5f * w;
    const float ncy = l.x * h + cy, ncx = l.y * w + cx;
    // exp in double, rounded once to fp32 (matches oracle/np_ref.py loc2bbox)
    const float nh = (float)exp((double)l.z) * h;
    const float nw = (float)exp((double)l.w) * w;
    float4 r;
    r.x = ncy - 0.5f * nh;
    r.y = ncx - 0.5f * nw;
    r.z = ncy + 0.5f * nh;
    r.w = ncx + 0.5f * nw;
    r.x = fminf(fmaxf(r.x, 0.f), img_h);
    r.z = fminf(fmaxf(r.z, 0.f), img_h);
    r.y = fminf(fmaxf(r.y, 0.f), img_w);
    r.w = fminf(fmaxf(r.w, 0.f), img_w);
    roi[i] = r;
    if (valid) valid[i] = ((r.z - r.x) >= min_size) && ((r.w - r.y) >= min_size);
}

// Total-order key: larger key = earlier in `argsort(score)[::-1]` with the
// documented tie rule (equal scores: lower index first).  0 = invalid.
__device__ __forceinline__ uint64_t make_key(float s, int idx)
{
    s = s + 0.f;  // -0 -> +0
    uint32_t b = __float_as_uint(s);
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return ((uint64_t)b << 32) | (uint32_t)(~(uint32_t)idx);
}

// ---- exact stable top-k by bucketed rank ------------------------------------------------
// rank(key) = #{keys larger}.  Keys are bucketed by their top 16 bits (sign/exponent/7
// mantissa bits of the score): a histogram + a descending scan give every bucket its first
// rank, a counting-sort pass groups the keys of a bucket, and each key then only counts the
// larger keys inside its own bucket.  O(n + sum bucket^2) instead of O(n^2) compares (n =
// 64 260 anchors: 0.24 ms -> tens of microseconds); the result is a pure function of the keys
// (unique: the index is part of the key), so it does not depend on the atomics' order.
constexpr int kBuckets = 1 << 16;
constexpr int kRankSplits = 8;   // a bucket's compare loop is split 8 ways (degenerate inputs:
                                 // all scores equal -> one bucket of n keys)

// Several problems of the same size per launch: blockIdx.z = problem; workspace pointers
// advance by ws_stride bytes, score / valid by n elements.
template <typename T>
__device__ __forceinline__ T *grp(T *p, int64_t stride_bytes)
{
    return reinterpret_cast<T *>(reinterpret_cast<char *>(const_cast<
        typename std::remove_const<T>::type *>(p)) + (int64_t)blockIdx.z * stride_bytes);
}

__global__ void topk_keys_kernel(const float *__restrict__ score,
                                 const uint8_t *__restrict__ valid, int n,
                                 uint64_t *__restrict__ keys, int32_t *__restrict__ hist,
                                 int32_t *__restrict__ n_valid, int64_t ws_stride)
{
    score += (int64_t)blockIdx.z * n;
    if (valid) valid += (int64_t)blockIdx.z * n;
    keys = grp(keys, ws_stride); hist = grp(hist, ws_stride); n_valid = grp(n_valid, ws_stride);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool v = false;
    if (i < n) {
        v = valid ? (valid[i] != 0) : true;
        const uint64_t key = v ? make_key(score[i], i) : 0ull;
        keys[i] = key;
        atomicAdd(&hist[(int)(key >> 48)], 1);
    }
    const unsigned long long b = __ballot(v);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_valid, (int)__popcll(b));
}

// start[b] = number of keys in buckets above b (one workgroup of kScanThreads threads: four
// waves find room on a CU that a register-heavy GEMM of another stream shares, sixteen do not)
constexpr int kScanThreads = 256;
__global__ void __launch_bounds__(kScanThreads)
topk_scan_kernel(const int32_t *__restrict__ hist, int32_t *__restrict__ start, int64_t ws_stride)
{
    hist = grp(hist, ws_stride); start = grp(start, ws_stride);
    __shared__ int part[kScanThreads];
    const int t = threadIdx.x;
    constexpr int PER = kBuckets / kScanThreads;
    // thread t owns buckets [hi - PER + 1, hi], hi descending with t
    const int hi = kBuckets - 1 - t * PER;
    int sum = 0;
    for (int j = 0; j < PER; ++j) sum += hist[hi - j];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < kScanThreads; off <<= 1) {
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;       // exclusive prefix over the threads before t
    for (int j = 0; j < PER; ++j) {
        start[hi - j] = run;
        run += hist[hi - j];
    }
}

// counting sort by bucket (order inside a bucket is arbitrary); consumes hist
__global__ void topk_place_kernel(const uint64_t *__restrict__ keys, int n,
                                  int32_t *__restrict__ hist, const int32_t *__restrict__ start,
                                  uint64_t *__restrict__ sorted, int64_t ws_stride)
{
    keys = grp(keys, ws_stride); hist = grp(hist, ws_stride); start = grp(start, ws_stride);
    sorted = grp(sorted, ws_stride);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t key = keys[i];
    const int b = (int)(key >> 48);
    sorted[start[b] + atomicSub(&hist[b], 1) - 1] = key;
}

// grid (n / 256, kRankSplits): within-bucket count of larger keys, slice blockIdx.y
__global__ void __launch_bounds__(256)
topk_rank_kernel(const uint64_t *__restrict__ sorted, const int32_t *__restrict__ start, int n,
                 int32_t *__restrict__ rank, int64_t ws_stride)
{
    sorted = grp(sorted, ws_stride); start = grp(start, ws_stride); rank = grp(rank, ws_stride);
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint64_t key = sorted[p];
    const int b = (int)(key >> 48);
    const int s = start[b], e = b > 0 ? start[b - 1] : n;
    const int len = (e - s + kRankSplits - 1) / kRankSplits;
    const int q0 = s + (int)blockIdx.y * len, q1 = min(e, q0 + len);
    int cnt = 0;
    for (int q = q0; q < q1; ++q) cnt += sorted[q] > key;
    if (cnt) atomicAdd(&rank[p], cnt);
}

__global__ void topk_scatter_kernel(const uint64_t *__restrict__ sorted,
                                    const int32_t *__restrict__ start,
                                    const int32_t *__restrict__ rank, int n, int k,
                                    int32_t *__restrict__ order,
                                    const int32_t *__restrict__ n_valid,
                                    int32_t *__restrict__ n_out, int64_t ws_stride, int order_stride)
{
    sorted = grp(sorted, ws_stride); start = grp(start, ws_stride); rank = grp(rank, ws_stride);
    n_valid = grp(n_valid, ws_stride);
    order += (int64_t)blockIdx.z * order_stride;
    n_out += blockIdx.z;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p == 0) *n_out = min(k, *n_valid);
    if (p >= n) return;
    const uint64_t key = sorted[p];
    if (key == 0ull) return;
    const int r = start[(int)(key >> 48)] + rank[p];
    if (r < k) order[r] = (int32_t)(~(uint32_t)key);
}

__global__ void gather_rows_kernel(const float *__restrict__ src, const int32_t *__restrict__ idx,
                                   const int32_t *__restrict__ n_dev, int n_max, int cols,
                                   float *__restrict__ dst)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)n_max * cols) return;
    const int r = (int)(e / cols), c = (int)(e % cols);
    const int n = n_dev ? min(*n_dev, n_max) : n_max;
    dst[e] = r < n ? src[(int64_t)idx[r] * cols + c] : 0.f;
}

struct Norm4 { double v[4]; };

__global__ void decode_cls_boxes_kernel(const float4 *__restrict__ roi,
                                        const float *__restrict__ cls_loc, int ld_loc,
                                        float4 *__restrict__ cls_bbox, int R, int n_class,
                                        float inv_scale, Norm4 mean, Norm4 stdv, float size_h,
                                        float size_w, int use_div, float scale)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= R * n_class) return;
    const int r = e / n_class, l = e % n_class;
    float4 a = roi[r];
    // rois[keep] / scale  (models/mask_rcnn.py:220)
    if (use_div) { a.x = a.x / scale; a.y = a.y / scale; a.z = a.z / scale; a.w = a.w / scale; }
    else { a.x *= inv_scale; a.y *= inv_scale; a.z *= inv_scale; a.w *= inv_scale; }
    const float *lp = cls_loc + (int64_t)r * ld_loc + 4 * l;
    // (roi_cls_loc * std + mean).astype(float32) (:225-229): mean / std come from Python tuples,
    // i.e. float64 arrays — the product and sum are formed in double and rounded once
    const float dy = (float)((double)lp[0] * stdv.v[0] + mean.v[0]);
    const float dx = (float)((double)lp[1] * stdv.v[1] + mean.v[1]);
    const float dh = (float)((double)lp[2] * stdv.v[2] + mean.v[2]);
    const float dw = (float)((double)lp[3] * stdv.v[3] + mean.v[3]);
    const float h = a.z - a.x, w = a.w - a.y;
    const float cy = a.x + 0.5f * h, cx = a.y + 0.5f * w;
    const float ncy = dy * h + cy, ncx = dx * w + cx;
    const float nh = (float)exp((double)dh) * h;
    const float nw = (float)exp((double)dw) * w;
    float4 o;
    o.x = fminf(fmaxf(ncy - 0.5f * nh, 0.f), size_h);
    o.y = fminf(fmaxf(ncx - 0.5f * nw, 0.f), size_w);
    o.z = fminf(fmaxf(ncy + 0.5f * nh, 0.f), size_h);
    o.w = fminf(fmaxf(ncx + 0.5f * nw, 0.f), size_w);
    cls_bbox[e] = o;
}

}  // namespace

extern "C" int mrcnn_decode_clip(const float *anchor, const float *loc, float *roi,
                                 uint8_t *valid, int n, float img_h, float img_w,
                                 float min_size, void *stream)
{
    MRCNN_REQUIRE(n >= 0, "decode_clip: n < 0");
    if (n == 0) return 0;
    MRCNN_REQUIRE(anchor && loc && roi, "decode_clip: null pointer");
    MRCNN_REQUIRE(((uintptr_t)anchor | (uintptr_t)loc | (uintptr_t)roi) % 16 == 0,
                  "decode_clip: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(decode_clip_kernel, dim3(mrcnn::ceil_div(n, 256)), dim3(256), 0,
                       mrcnn::as_stream(stream), (const float4 *)anchor, (const float4 *)loc,
                       (float4 *)roi, valid, n, img_h, img_w, min_size);
    return mrcnn::check_launch("decode_clip");
}

extern "C" int64_t mrcnn_topk_workspace_bytes(int n)
{
    return (int64_t)n * 20 + 2 * (int64_t)kBuckets * 4 + 256;
}

extern "C" int mrcnn_topk_desc_batched(const float *score, const uint8_t *valid, int groups, int n,
                                       int k, int32_t *order, int32_t *n_out, void *ws,
                                       void *stream)
{
    MRCNN_REQUIRE(n >= 0 && k >= 0 && groups >= 0 && groups <= 65535, "topk_desc: bad n/k/groups");
    if (groups == 0) return 0;
    MRCNN_REQUIRE(n == 0 || (score && order && n_out && ws), "topk_desc: null pointer");
    hipStream_t s = mrcnn::as_stream(stream);
    if (n == 0) {
        if (n_out) MRCNN_HIP_TRY(hipMemsetAsync(n_out, 0, 4 * (size_t)groups, s));
        return 0;
    }
    // per-problem workspace: [n_valid | pad to 64 B][rank n x i32][pad][hist 64K x i32] (zeroed)
    // then [start 64K x i32][keys n x u64][sorted n x u64]; problems ws_stride bytes apart
    const int64_t ws_stride = ((mrcnn_topk_workspace_bytes(n) + 63) / 64) * 64;
    char *w = (char *)ws;
    int32_t *n_valid = (int32_t *)w;
    int32_t *rank = (int32_t *)(w + 64);
    const size_t hist_off = 64 + (((size_t)n * 4 + 63) / 64) * 64;
    int32_t *hist = (int32_t *)(w + hist_off);
    const size_t start_off = hist_off + (size_t)kBuckets * 4;
    int32_t *start = (int32_t *)(w + start_off);
    uint64_t *keys = (uint64_t *)(w + start_off + (size_t)kBuckets * 4);
    uint64_t *sorted = keys + n;
    for (int g = 0; g < groups; ++g)
        MRCNN_HIP_TRY(hipMemsetAsync(w + g * ws_stride, 0, start_off, s));
    const unsigned blocks = (unsigned)mrcnn::ceil_div(n, 256), G = (unsigned)groups;
    mrcnn::ProfScope prof(mrcnn::PROF_TOPK, 0., groups * (40.0 * n + 3.0 * 4 * kBuckets), s);
    hipLaunchKernelGGL(topk_keys_kernel, dim3(blocks, 1, G), dim3(256), 0, s, score, valid, n, keys,
                       hist, n_valid, ws_stride);
    hipLaunchKernelGGL(topk_scan_kernel, dim3(1, 1, G), dim3(kScanThreads), 0, s, hist, start, ws_stride);
    hipLaunchKernelGGL(topk_place_kernel, dim3(blocks, 1, G), dim3(256), 0, s, keys, n, hist, start,
                       sorted, ws_stride);
    hipLaunchKernelGGL(topk_rank_kernel, dim3(blocks, kRankSplits, G), dim3(256), 0, s, sorted,
                       start, n, rank, ws_stride);
    hipLaunchKernelGGL(topk_scatter_kernel, dim3(blocks, 1, G), dim3(256), 0, s, sorted, start, rank,
                       n, k, order, n_valid, n_out, ws_stride, k);
    return mrcnn::check_launch("topk_desc");
}

extern "C" int mrcnn_topk_desc(const float *score, const uint8_t *valid, int n, int k,
                               int32_t *order, int32_t *n_out, void *ws, void *stream)
{
    return mrcnn_topk_desc_batched(score, valid, 1, n, k, order, n_out, ws, stream);
}

// ---- per-class score threshold + stable sort + gather for MaskRCNN._suppress ----------------
// (models/mask_rcnn.py:178-202 does this per class in a Python loop on CPU copies.)
namespace {
__global__ void detect_keys_kernel(const float *__restrict__ prob, int R, int n_class, float thresh,
                                   uint64_t *__restrict__ keys, int32_t *__restrict__ counts)
{
    const int g = blockIdx.y;                       // foreground class g -> column g + 1
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool v = false;
    if (i < R) {
        const float p = prob[(int64_t)i * n_class + g + 1];
        v = p > thresh;
        keys[(int64_t)g * R + i] = v ? make_key(p, i) : 0ull;
    }
    const unsigned long long b = __ballot(v);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&counts[g], (int)__popcll(b));
}

__global__ void __launch_bounds__(256)
detect_rank_gather_kernel(const uint64_t *__restrict__ keys, const float *__restrict__ prob,
                          const float4 *__restrict__ cls_bbox, int R, int n_class,
                          float4 *__restrict__ sorted_boxes, float *__restrict__ sorted_prob)
{
    const int g = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t *__restrict__ kg = keys + (int64_t)g * R;
    const uint64_t ki = i < R ? kg[i] : 0ull;
    int cnt = 0;
    for (int j = 0; j < R; ++j) cnt += kg[j] > ki;
    if (i < R && ki != 0ull) {
        sorted_boxes[(int64_t)g * R + cnt] = cls_bbox[(int64_t)i * n_class + g + 1];
        sorted_prob[(int64_t)g * R + cnt] = prob[(int64_t)i * n_class + g + 1];
    }
}
}  // namespace

extern "C" int64_t mrcnn_detect_sort_workspace_bytes(int R, int n_class)
{
    return (int64_t)R * (n_class - 1) * 8 + 64;
}

extern "C" int mrcnn_detect_sort(const float *prob, const float *cls_bbox, int R, int n_class,
                                 float thresh, float *sorted_boxes, float *sorted_prob,
                                 int32_t *counts, void *ws, void *stream)
{
    MRCNN_REQUIRE(R >= 0 && n_class >= 2, "detect_sort: bad shape");
    MRCNN_REQUIRE(counts, "detect_sort: null counts");
    hipStream_t s = mrcnn::as_stream(stream);
    const int G = n_class - 1;
    MRCNN_HIP_TRY(hipMemsetAsync(counts, 0, 4 * (size_t)G, s));
    if (R == 0) return 0;
    MRCNN_REQUIRE(prob && cls_bbox && sorted_boxes && sorted_prob && ws, "detect_sort: null pointer");
    MRCNN_REQUIRE(((uintptr_t)cls_bbox | (uintptr_t)sorted_boxes) % 16 == 0,
                  "detect_sort: boxes must be 16-byte aligned");
    const int blocks = (int)mrcnn::ceil_div(R, 256);
    hipLaunchKernelGGL(detect_keys_kernel, dim3(blocks, G), dim3(256), 0, s, prob, R, n_class, thresh,
                       (uint64_t *)ws, counts);
    hipLaunchKernelGGL(detect_rank_gather_kernel, dim3(blocks, G), dim3(256), 0, s,
                       (const uint64_t *)ws, prob, (const float4 *)cls_bbox, R, n_class,
                       (float4 *)sorted_boxes, sorted_prob);
    return mrcnn::check_launch("detect_sort");
}

namespace {
// Second half of MaskRCNN._suppress (models/mask_rcnn.py:195-202) on the device: the kept rows
// of every class, class after class (ascending label) and in NMS keep order inside a class,
// packed into dense arrays.  One workgroup: a serial prefix sum over the <= 1024 classes in
// LDS, then the rows are copied cooperatively.
__global__ void __launch_bounds__(256)
detect_compact_kernel(const int32_t *__restrict__ keep, const int32_t *__restrict__ n_keep,
                      const float4 *__restrict__ sorted_boxes, const float *__restrict__ sorted_prob,
                      int G, int R, float4 *__restrict__ bbox, int32_t *__restrict__ label,
                      float *__restrict__ score, int32_t *__restrict__ total)
{
    extern __shared__ int32_t offs[];             // G + 1 offsets
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int l = 0; l < G; ++l) { offs[l] = acc; acc += min(max(n_keep[l], 0), R); }
        offs[G] = acc;
        *total = acc;
    }
    __syncthreads();
    for (int l = 0; l < G; ++l) {
        const int n = offs[l + 1] - offs[l];
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int k = keep[(int64_t)l * R + i];
            const int o = offs[l] + i;
            bbox[o] = sorted_boxes[(int64_t)l * R + k];
            score[o] = sorted_prob[(int64_t)l * R + k];
            label[o] = l;
        }
    }
}
}  // namespace

extern "C" int mrcnn_detect_compact(const int32_t *keep, const int32_t *n_keep,
                                    const float *sorted_boxes, const float *sorted_prob, int G,
                                    int R, float *bbox, int32_t *label, float *score,
                                    int32_t *total, void *stream)
{
    MRCNN_REQUIRE(G > 0 && G <= 4096 && R >= 0, "detect_compact: bad shape");
    MRCNN_REQUIRE(total, "detect_compact: null total");
    hipStream_t s = mrcnn::as_stream(stream);
    if (R == 0) {
        MRCNN_HIP_TRY(hipMemsetAsync(total, 0, 4, s));
        return 0;
    }
    MRCNN_REQUIRE(keep && n_keep && sorted_boxes && sorted_prob && bbox && label && score,
                  "detect_compact: null pointer");
    MRCNN_REQUIRE(((uintptr_t)sorted_boxes | (uintptr_t)bbox) % 16 == 0,
                  "detect_compact: boxes must be 16-byte aligned");
    hipLaunchKernelGGL(detect_compact_kernel, dim3(1), dim3(256), (size_t)(G + 1) * 4, s, keep, n_keep,
                       (const float4 *)sorted_boxes, sorted_prob, G, R, (float4 *)bbox, label, score,
                       total);
    return mrcnn::check_launch("detect_compact");
}

extern "C" int mrcnn_gather_rows(const float *src, const int32_t *idx, const int32_t *n_dev,
                                 int n_max, int cols, float *dst, void *stream)
{
    MRCNN_REQUIRE(n_max >= 0 && cols > 0, "gather_rows: bad shape");
    if (n_max == 0) return 0;
    MRCNN_REQUIRE(src && idx && dst, "gather_rows: null pointer");
    const int64_t total = (int64_t)n_max * cols;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(mrcnn::ceil_div(total, 256)), dim3(256), 0,
                       mrcnn::as_stream(stream), src, idx, n_dev, n_max, cols, dst);
    return mrcnn::check_launch("gather_rows");
}

extern "C" int mrcnn_decode_cls_boxes(const float *roi, const float *cls_loc, int ld_loc,
                                      float *cls_bbox, int R, int n_class, float scale,
                                      const double *mean4, const double *std4, float size_h,
                                      float size_w, void *stream)
{
    MRCNN_REQUIRE(R >= 0 && n_class > 0, "decode_cls_boxes: bad shape");
    if (R == 0) return 0;
    MRCNN_REQUIRE(roi && cls_loc && cls_bbox && mean4 && std4, "decode_cls_boxes: null pointer");
    Norm4 mean, stdv;
    for (int i = 0; i < 4; ++i) { mean.v[i] = mean4[i]; stdv.v[i] = std4[i]; }
    hipLaunchKernelGGL(decode_cls_boxes_kernel, dim3(mrcnn::ceil_div((int64_t)R * n_class, 256)),
                       dim3(256), 0, mrcnn::as_stream(stream), (const float4 *)roi, cls_loc, ld_loc,
                       (float4 *)cls_bbox, R, n_class, 1.f / scale, mean, stdv, size_h, size_w, 1,
                       scale);
    return mrcnn::check_launch("decode_cls_boxes");
}
