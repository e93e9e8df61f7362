// Device half of the target creators: the IoU matrices, the label
// rules and the 14x14 mask-target resampling of
//   ProposalTargetCreator.__call__   /root/reference/chainer_mask_rcnn/models/utils/proposal_target_creator.py:121-177
//   chainercv AnchorTargetCreator    (call site models/mask_rcnn_train_chain.py:153-158)
// as HIP kernels.  The RANDOM part — which candidates are kept — stays on the host so that the
// global np.random stream is consumed exactly as the reference consumes it (the parity
// contract): the host reads back one small vector per image (max IoU per candidate / anchor
// labels), draws with np.random.choice, and uploads the chosen indices; the heavy arrays
// (candidate boxes, regression targets, anchor targets, mask targets) never leave the device.
//
// Arithmetic follows the host NumPy versions operation by operation in fp32
// (-ffp-contract=off), so integer results (argmax, labels, mask targets) are identical; the
// two logs of bbox2loc are evaluated in double and rounded once (<= 1 ulp from np.log).
#include <math.h>

#include "common.h"

namespace {

// chainercv bbox_iou for one pair, fp32 exactly as NumPy evaluates it
__device__ __forceinline__ float iou_pair(const float *a, const float *b)
{
    const float tl0 = fmaxf(a[0], b[0]), tl1 = fmaxf(a[1], b[1]);
    const float br0 = fminf(a[2], b[2]), br1 = fminf(a[3], b[3]);
    const float inter = (tl0 < br0 && tl1 < br1) ? (br0 - tl0) * (br1 - tl1) : 0.f * ((br0 - tl0) * (br1 - tl1));
    const float area_a = (a[2] - a[0]) * (a[3] - a[1]);
    const float area_b = (b[2] - b[0]) * (b[3] - b[1]);
    return inter / (area_a + area_b - inter);
}

// per row of `a`: max IoU over the G boxes of `b` and its first argmax (np.argmax / np.max:
// a NaN wins and propagates).  Optionally writes the whole (na, g) matrix.
__global__ void iou_argmax_kernel(const float *__restrict__ a, int na, const float *__restrict__ b,
                                  int g, float *__restrict__ iou_out, float *__restrict__ max_out,
                                  int *__restrict__ argmax_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na) return;
    float box[4] = {a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]};
    float best = 0.f;
    int arg = 0;
    bool have = false, nan_seen = false;
    for (int j = 0; j < g; ++j) {
        const float v = iou_pair(box, b + 4 * j);
        if (iou_out) iou_out[(int64_t)i * g + j] = v;
        if (nan_seen) continue;
        if (v != v) { best = v; arg = j; nan_seen = true; continue; }
        if (!have || v > best) { best = v; arg = j; have = true; }
    }
    max_out[i] = best;
    argmax_out[i] = arg;
}

// column maxima of the (na, g) IoU matrix: one workgroup per ground-truth box
__global__ void iou_colmax_kernel(const float *__restrict__ iou, int na, int g,
                                  float *__restrict__ colmax)
{
    __shared__ float red[256];
    const int j = blockIdx.x;
    float m = -INFINITY;
    bool nan_seen = false;
    for (int i = threadIdx.x; i < na; i += blockDim.x) {
        const float v = iou[(int64_t)i * g + j];
        if (v != v) nan_seen = true;
        else m = fmaxf(m, v);
    }
    red[threadIdx.x] = nan_seen ? NAN : m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            const float x = red[threadIdx.x], y = red[threadIdx.x + s];
            red[threadIdx.x] = (x != x || y != y) ? NAN : fmaxf(x, y);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) colmax[j] = red[0];
}

// AnchorTargetCreator._create_label before the random subsampling (A.5):
//   label = -1; label[max < neg] = 0; label[any_j iou[i,j] == gt_max[j]] = 1; label[max >= pos] = 1
__global__ void anchor_label_kernel(const float *__restrict__ iou, const float *__restrict__ max_iou,
                                    const float *__restrict__ gt_max, int na, int g, float neg_thresh,
                                    float pos_thresh, int *__restrict__ label)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na) return;
    int l = -1;
    const float m = max_iou[i];
    if (m < neg_thresh) l = 0;
    for (int j = 0; j < g; ++j)
        if (iou[(int64_t)i * g + j] == gt_max[j]) { l = 1; break; }
    if (m >= pos_thresh) l = 1;
    label[i] = l;
}

__device__ __forceinline__ void bbox2loc_one(const float *src, const float *dst, float *out)
{
    // chainercv bbox2loc (A.2), fp32; log in double, rounded once
    const float eps = 1.1920929e-07f;
    float sh = src[2] - src[0], sw = src[3] - src[1];
    const float scy = src[0] + 0.5f * sh, scx = src[1] + 0.5f * sw;
    const float dh = dst[2] - dst[0], dw = dst[3] - dst[1];
    const float dcy = dst[0] + 0.5f * dh, dcx = dst[1] + 0.5f * dw;
    sh = fmaxf(sh, eps);
    sw = fmaxf(sw, eps);
    out[0] = (dcy - scy) / sh;
    out[1] = (dcx - scx) / sw;
    out[2] = (float)log((double)(dh / sh));
    out[3] = (float)log((double)(dw / sw));
}

// full-size anchor targets: loc = 0 / label = -1 outside the image, the inside anchors get
// their label (minus the host's `disabled` draws) and bbox2loc(anchor, bbox[argmax])
__global__ void anchor_fill_kernel(float *__restrict__ loc, int *__restrict__ label, int n_anchor)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_anchor) return;
    label[i] = -1;
    reinterpret_cast<float4 *>(loc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

__global__ void anchor_scatter_kernel(const float *__restrict__ anchor_inside,
                                      const int *__restrict__ inside_index,
                                      const int *__restrict__ label_inside,
                                      const int *__restrict__ argmax, const float *__restrict__ bbox,
                                      int n_inside, float *__restrict__ loc, int *__restrict__ label)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_inside) return;
    const int dst = inside_index[i];
    label[dst] = label_inside[i];
    float out[4];
    bbox2loc_one(anchor_inside + 4 * i, bbox + 4 * argmax[i], out);
    reinterpret_cast<float4 *>(loc)[dst] = make_float4(out[0], out[1], out[2], out[3]);
}

__global__ void set_label_kernel(int *__restrict__ label, const int *__restrict__ index, int n, int value)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) label[index[i]] = value;
}

// ProposalTargetCreator after the host's draws (:148-158): gather the chosen candidates,
// labels (+1, background rows 0) and normalised regression targets
__global__ void proposal_gather_kernel(const float *__restrict__ cand, const float *__restrict__ bbox,
                                       const int *__restrict__ gt_label, const int *__restrict__ assigned,
                                       const int *__restrict__ chosen, int n_sample, int n_fg,
                                       float m0, float m1, float m2, float m3, float s0, float s1,
                                       float s2, float s3, float *__restrict__ sample_roi,
                                       float *__restrict__ loc, int *__restrict__ label,
                                       int *__restrict__ gt_index)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sample) return;
    const int c = chosen[i];
    const int a = assigned[c];
    const float4 box = reinterpret_cast<const float4 *>(cand)[c];
    reinterpret_cast<float4 *>(sample_roi)[i] = box;
    const float src[4] = {box.x, box.y, box.z, box.w};
    float out[4];
    bbox2loc_one(src, bbox + 4 * a, out);
    reinterpret_cast<float4 *>(loc)[i] =
        make_float4((out[0] - m0) / s0, (out[1] - m1) / s1, (out[2] - m2) / s2, (out[3] - m3) / s3);
    label[i] = i < n_fg ? gt_label[a] + 1 : 0;
    gt_index[i] = a;
}

// one axis of the cv2 INTER_LINEAR rule as models/utils/proposal_target_creator._mask_targets
// evaluates it: position in double, weight rounded to fp32, clamped at the crop's border
struct Tap { int i0, i1; float t; };
__device__ __forceinline__ Tap mask_axis(int d, int n_in, int start, int M, int limit)
{
    const double n = (double)max(n_in, 1);
    const double pos = ((double)d + 0.5) * (n / (double)M) - 0.5;
    long long i0 = (long long)floor(pos);
    float t = (float)(pos - (double)i0);
    const long long last = (long long)n - 1;
    if (i0 < 0 || i0 >= last) t = 0.f;
    i0 = i0 < 0 ? 0 : (i0 > last ? last : i0);
    const long long i1 = i0 + 1 > last ? last : i0 + 1;
    Tap r;
    r.i0 = min(max((int)i0 + start, 0), limit - 1);
    r.i1 = min(max((int)i1 + start, 0), limit - 1);
    r.t = t;
    return r;
}

// mask targets (:160-177): rows [0, n_fg) = bilinear resize of the {0,1} crop
// mask[gt_index][y0:y1, x0:x1] (box = round-half-even of the sampled RoI) to M x M,
// thresholded at 0.5 (the reference's one-hot / resize / argmax); rows [n_fg, n) = -1
__global__ void mask_targets_kernel(const uint8_t *__restrict__ masks, int H, int W,
                                    const float *__restrict__ sample_roi,
                                    const int *__restrict__ gt_index, int n, int n_fg, int M,
                                    int *__restrict__ out)
{
    const int r = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= M * M) return;
    int *o = out + ((int64_t)r * M * M + p);
    if (r >= n_fg) { *o = -1; return; }
    const float *b = sample_roi + 4 * r;
    const int y0 = min(max((int)rintf(b[0]), 0), H), x0 = min(max((int)rintf(b[1]), 0), W);
    const int y1 = min(max((int)rintf(b[2]), 0), H), x1 = min(max((int)rintf(b[3]), 0), W);
    const int h = max(y1 - y0, 0), w = max(x1 - x0, 0);
    if (h == 0 || w == 0) { *o = 0; return; }
    const int py = p / M, px = p - py * M;
    const Tap ty = mask_axis(py, h, y0, M, H), tx = mask_axis(px, w, x0, M, W);
    const uint8_t *m = masks + (int64_t)gt_index[r] * H * W;
    auto f = [&](int yy, int xx) -> float { return m[(int64_t)yy * W + xx] > 0 ? 1.f : 0.f; };
    const float top = f(ty.i0, tx.i0) * (1.f - tx.t) + f(ty.i0, tx.i1) * tx.t;
    const float bot = f(ty.i1, tx.i0) * (1.f - tx.t) + f(ty.i1, tx.i1) * tx.t;
    const float prob = top * (1.f - ty.t) + bot * ty.t;
    *o = prob > 0.5f ? 1 : 0;
}

inline int blocks_for(int64_t n, int threads = 256) { return (int)((n + threads - 1) / threads); }

}  // namespace

extern "C" int mrcnn_bbox_iou_argmax(const float *boxes_a, int na, const float *boxes_b, int g,
                                     float *iou, float *max_iou, int32_t *argmax, float *col_max,
                                     void *stream)
{
    MRCNN_REQUIRE(na >= 0 && g > 0, "bbox_iou_argmax: bad sizes (na=%d, g=%d)", na, g);
    if (na == 0) return 0;
    MRCNN_REQUIRE(boxes_a && boxes_b && max_iou && argmax, "bbox_iou_argmax: null pointer");
    MRCNN_REQUIRE(!col_max || iou, "bbox_iou_argmax: column maxima need the IoU matrix");
    hipStream_t s = mrcnn::as_stream(stream);
    hipLaunchKernelGGL(iou_argmax_kernel, dim3(blocks_for(na)), dim3(256), 0, s, boxes_a, na, boxes_b,
                       g, iou, max_iou, argmax);
    if (col_max)
        hipLaunchKernelGGL(iou_colmax_kernel, dim3(g), dim3(256), 0, s, iou, na, g, col_max);
    return mrcnn::check_launch("bbox_iou_argmax");
}

extern "C" int mrcnn_anchor_labels(const float *iou, const float *max_iou, const float *gt_max,
                                   int na, int g, float neg_iou_thresh, float pos_iou_thresh,
                                   int32_t *label, void *stream)
{
    MRCNN_REQUIRE(na >= 0 && g > 0, "anchor_labels: bad sizes");
    if (na == 0) return 0;
    MRCNN_REQUIRE(iou && max_iou && gt_max && label, "anchor_labels: null pointer");
    hipLaunchKernelGGL(anchor_label_kernel, dim3(blocks_for(na)), dim3(256), 0, mrcnn::as_stream(stream),
                       iou, max_iou, gt_max, na, g, neg_iou_thresh, pos_iou_thresh, label);
    return mrcnn::check_launch("anchor_labels");
}

extern "C" int mrcnn_anchor_targets_finish(const float *anchor_inside, const int32_t *inside_index,
                                           int32_t *label_inside, const int32_t *argmax,
                                           const float *bbox, int n_inside, const int32_t *disabled,
                                           int n_disabled, int n_anchor, float *loc, int32_t *label,
                                           void *stream)
{
    MRCNN_REQUIRE(n_inside >= 0 && n_disabled >= 0 && n_anchor >= n_inside, "anchor_targets_finish: bad sizes");
    MRCNN_REQUIRE(loc && label, "anchor_targets_finish: null output");
    MRCNN_REQUIRE(((uintptr_t)loc % 16) == 0, "anchor_targets_finish: loc must be 16-byte aligned");
    hipStream_t s = mrcnn::as_stream(stream);
    if (n_anchor > 0)
        hipLaunchKernelGGL(anchor_fill_kernel, dim3(blocks_for(n_anchor)), dim3(256), 0, s, loc, label,
                           n_anchor);
    if (n_inside > 0) {
        MRCNN_REQUIRE(anchor_inside && inside_index && label_inside && argmax && bbox,
                      "anchor_targets_finish: null pointer");
        if (n_disabled > 0) {
            MRCNN_REQUIRE(disabled, "anchor_targets_finish: null disabled list");
            hipLaunchKernelGGL(set_label_kernel, dim3(blocks_for(n_disabled)), dim3(256), 0, s,
                               label_inside, disabled, n_disabled, -1);
        }
        hipLaunchKernelGGL(anchor_scatter_kernel, dim3(blocks_for(n_inside)), dim3(256), 0, s,
                           anchor_inside, inside_index, label_inside, argmax, bbox, n_inside, loc, label);
    }
    return mrcnn::check_launch("anchor_targets_finish");
}

extern "C" int mrcnn_proposal_targets_gather(const float *cand, const float *bbox,
                                             const int32_t *gt_label, const int32_t *assigned,
                                             const int32_t *chosen, int n_sample, int n_fg,
                                             const float *mean4_host, const float *std4_host,
                                             float *sample_roi, float *gt_roi_loc,
                                             int32_t *gt_roi_label, int32_t *gt_index, void *stream)
{
    MRCNN_REQUIRE(n_sample >= 0 && n_fg >= 0 && n_fg <= n_sample, "proposal_targets_gather: bad sizes");
    if (n_sample == 0) return 0;
    MRCNN_REQUIRE(cand && bbox && gt_label && assigned && chosen && mean4_host && std4_host &&
                      sample_roi && gt_roi_loc && gt_roi_label && gt_index,
                  "proposal_targets_gather: null pointer");
    MRCNN_REQUIRE(((uintptr_t)cand % 16) == 0 && ((uintptr_t)sample_roi % 16) == 0 &&
                      ((uintptr_t)gt_roi_loc % 16) == 0,
                  "proposal_targets_gather: box arrays must be 16-byte aligned");
    hipLaunchKernelGGL(proposal_gather_kernel, dim3(blocks_for(n_sample)), dim3(256), 0,
                       mrcnn::as_stream(stream), cand, bbox, gt_label, assigned, chosen, n_sample, n_fg,
                       mean4_host[0], mean4_host[1], mean4_host[2], mean4_host[3], std4_host[0],
                       std4_host[1], std4_host[2], std4_host[3], sample_roi, gt_roi_loc, gt_roi_label,
                       gt_index);
    return mrcnn::check_launch("proposal_targets_gather");
}

extern "C" int mrcnn_mask_targets(const uint8_t *masks, int G, int H, int W, const float *sample_roi,
                                  const int32_t *gt_index, int n, int n_fg, int M, int32_t *out,
                                  void *stream)
{
    MRCNN_REQUIRE(n >= 0 && n_fg >= 0 && n_fg <= n && M > 0 && G > 0 && H > 0 && W > 0,
                  "mask_targets: bad sizes");
    if (n == 0) return 0;
    MRCNN_REQUIRE(masks && sample_roi && gt_index && out, "mask_targets: null pointer");
    MRCNN_REQUIRE(n <= 65535, "mask_targets: too many rows");
    hipLaunchKernelGGL(mask_targets_kernel, dim3(blocks_for(M * M), n), dim3(256), 0,
                       mrcnn::as_stream(stream), masks, H, W, sample_roi, gt_index, n, n_fg, M, out);
    return mrcnn::check_launch("mask_targets");
}
