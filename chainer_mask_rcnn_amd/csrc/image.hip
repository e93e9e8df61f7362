// Device-side image I/O at both ends of MaskRCNN.predict:
//   prepare      — cv2.resize(img, None, fx=scale, fy=scale) + mean subtraction + zero-padded
//                  batch assembly (/root/reference/chainer_mask_rcnn/models/mask_rcnn.py:152-176,
//                  datasets/concat_examples.py:20-26)
//   paste_masks  — segm_results / expand_boxes (models/mask_rcnn.py:44-107): pad the 14x14
//                  mask by one pixel, expand the box by (M+2)/M, truncate to int, bilinear
//                  resize to the box, threshold at 0.5, paste into the image.
// cv2 is not installable in the build container; both kernels restate OpenCV's INTER_LINEAR
// float path (src = (float)((dst + 0.5) * scale - 0.5), floor, clamp to the border, horizontal
// then vertical blend) exactly as oracle/np_infer.py does; uint8 sources take OpenCV's 8-bit
// fixed-point path (11-bit coefficients, result rounded to uint8 before the mean is subtracted),
// which is what cv2.resize runs for the decoded images MaskRCNNTransform feeds.
// Built with -ffp-contract=off.
#include "common.h"

namespace {

struct Lin { int i0, i1; float t; };

// OpenCV resizeLinear coordinate rule for one axis
__device__ __forceinline__ Lin lin_coord(int d, double scale, int n_in)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= n_in - 1) { s = n_in - 1; f = 0.f; }
    Lin l;
    l.i0 = s;
    l.i1 = min(s + 1, n_in - 1);
    l.t = f;
    return l;
}

// src (C,H,W) fp32 or uint8 -> dst image n of an (N, dstH, dstW, C) NHWC batch, rows/cols
// beyond (outH, outW) are left untouched (the caller zero-fills the batch).  flip_x writes
// the resized image mirrored left-right (chainercv random_flip after the resize).
template <typename T>
__global__ void prepare_kernel(const T *__restrict__ src, int C, int H, int W, double inv_scale,
                               float m0, float m1, float m2, float *__restrict__ dst, int dstH,
                               int dstW, int outH, int outW, int n, int flip_x)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= outW || y >= outH) return;
    const Lin ly = lin_coord(y, inv_scale, H);
    const Lin lx = lin_coord(flip_x ? outW - 1 - x : x, inv_scale, W);
    float *o = dst + (((int64_t)n * dstH + y) * dstW + x) * C;
    if constexpr (sizeof(T) == 1) {
        // OpenCV 8-bit path (imgproc/resize.cpp: HResizeLinear<uchar,int,short> +
        // VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>), see
        // oracle/np_infer.py:cv_resize_linear_u8.  Integer arithmetic: bit-exact vs the oracle.
        float fy = (float)(((double)y + 0.5) * inv_scale - 0.5);
        const int sy = (int)floorf(fy);
        fy -= (float)sy;                                     // vertical weight is NOT clamped
        const int y0 = min(max(sy, 0), H - 1), y1 = min(max(sy + 1, 0), H - 1);
        const int a0 = __float2int_rn((1.f - lx.t) * 2048.f), a1 = __float2int_rn(lx.t * 2048.f);
        const int b0 = __float2int_rn((1.f - fy) * 2048.f), b1 = __float2int_rn(fy * 2048.f);
        for (int c = 0; c < C; ++c) {
            const T *p = src + (int64_t)c * H * W;
            const int d0 = (int)p[y0 * W + lx.i0] * a0 + (int)p[y0 * W + lx.i1] * a1;
            const int d1 = (int)p[y1 * W + lx.i0] * a0 + (int)p[y1 * W + lx.i1] * a1;
            const int v = (((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2;
            o[c] = (float)min(max(v, 0), 255) - (c == 0 ? m0 : (c == 1 ? m1 : m2));
        }
        return;
    }
    for (int c = 0; c < C; ++c) {
        const T *p = src + (int64_t)c * H * W;
        const float top = (float)p[ly.i0 * W + lx.i0] * (1.f - lx.t) + (float)p[ly.i0 * W + lx.i1] * lx.t;
        const float bot = (float)p[ly.i1 * W + lx.i0] * (1.f - lx.t) + (float)p[ly.i1 * W + lx.i1] * lx.t;
        const float v = top * (1.f - ly.t) + bot * ly.t;
        o[c] = v - (c == 0 ? m0 : (c == 1 ? m1 : m2));
    }
}

// one workgroup row per (detection, image row)
__global__ void paste_masks_kernel(const float *__restrict__ logits, const int32_t *__restrict__ label,
                                   const float *__restrict__ bbox, int D, int M, int Kc, int im_h,
                                   int im_w, uint8_t *__restrict__ out)
{
    const int d = blockIdx.z;
    const int y = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= im_w) return;
    // expand_boxes on (x1, y1, x2, y2) = bbox[:, [1,0,3,2]], scale = (M + 2) / M, fp32
    const float *b = bbox + 4 * d;
    const float scale = (float)(((double)M + 2.0) / (double)M);
    float w_half = (b[3] - b[1]) * .5f, h_half = (b[2] - b[0]) * .5f;
    const float x_c = (b[3] + b[1]) * .5f, y_c = (b[2] + b[0]) * .5f;
    w_half *= scale;
    h_half *= scale;
    const int rx1 = (int)(x_c - w_half), rx2 = (int)(x_c + w_half);   // astype(int32): truncate
    const int ry1 = (int)(y_c - h_half), ry2 = (int)(y_c + h_half);
    const int w = max(rx2 - rx1 + 1, 1), h = max(ry2 - ry1 + 1, 1);
    const int x_0 = max(rx1, 0), x_1 = min(rx2 + 1, im_w);
    const int y_0 = max(ry1, 0), y_1 = min(ry2 + 1, im_h);
    uint8_t v = 0;
    if (x >= x_0 && x < x_1 && y >= y_0 && y < y_1) {
        const int P = M + 2;
        const Lin ly = lin_coord(y - ry1, (double)P / (double)h, P);
        const Lin lx = lin_coord(x - rx1, (double)P / (double)w, P);
        const int ch = label[d];
        // padded_mask[1:-1, 1:-1] = sigmoid(logits[d, ch]); logits are (D, M, M, Kc) NHWC
        auto pm = [&](int py, int px) -> float {
            if (py < 1 || py > M || px < 1 || px > M) return 0.f;
            const float z = logits[(((int64_t)d * M + (py - 1)) * M + (px - 1)) * Kc + ch];
            // sigmoid in double, rounded once (as oracle/np_infer.py): the 0.5 threshold below
            // must not depend on the last ulp of a device expf
            return (float)(1.0 / (1.0 + exp(-(double)z)));
        };
        const float top = pm(ly.i0, lx.i0) * (1.f - lx.t) + pm(ly.i0, lx.i1) * lx.t;
        const float bot = pm(ly.i1, lx.i0) * (1.f - lx.t) + pm(ly.i1, lx.i1) * lx.t;
        v = (top * (1.f - ly.t) + bot * ly.t) > 0.5f ? 1 : 0;
    }
    out[((int64_t)d * im_h + y) * im_w + x] = v;
}

}  // namespace

extern "C" int mrcnn_prepare_image(const void *src_chw, int src_is_u8, int C, int H, int W,
                                   double scale, const float *mean_host, float *dst_nhwc, int dstH,
                                   int dstW, int outH, int outW, int n, int flip_x, void *stream)
{
    MRCNN_REQUIRE(src_chw && dst_nhwc && mean_host, "prepare_image: null pointer");
    MRCNN_REQUIRE(C == 3 && H > 0 && W > 0 && scale > 0., "prepare_image: expects a 3-channel image");
    MRCNN_REQUIRE(outH <= dstH && outW <= dstW && outH > 0 && outW > 0, "prepare_image: bad sizes");
    if (src_is_u8)
        hipLaunchKernelGGL(prepare_kernel<uint8_t>, dim3((outW + 255) / 256, outH), dim3(256), 0,
                           mrcnn::as_stream(stream), (const uint8_t *)src_chw, C, H, W, 1.0 / scale,
                           mean_host[0], mean_host[1], mean_host[2], dst_nhwc, dstH, dstW, outH,
                           outW, n, flip_x);
    else
        hipLaunchKernelGGL(prepare_kernel<float>, dim3((outW + 255) / 256, outH), dim3(256), 0,
                           mrcnn::as_stream(stream), (const float *)src_chw, C, H, W, 1.0 / scale,
                           mean_host[0], mean_host[1], mean_host[2], dst_nhwc, dstH, dstW, outH,
                           outW, n, flip_x);
    return mrcnn::check_launch("prepare_image");
}

extern "C" int mrcnn_paste_masks(const float *mask_logits, const int32_t *label, const float *bbox,
                                 int D, int M, int Kc, int im_h, int im_w, uint8_t *out,
                                 void *stream)
{
    MRCNN_REQUIRE(D >= 0 && M > 0 && Kc > 0 && im_h > 0 && im_w > 0, "paste_masks: bad shape");
    if (D == 0) return 0;
    MRCNN_REQUIRE(mask_logits && label && bbox && out, "paste_masks: null pointer");
    MRCNN_REQUIRE(D <= 65535 && im_h <= 65535, "paste_masks: grid too large");
    hipLaunchKernelGGL(paste_masks_kernel, dim3((im_w + 255) / 256, im_h, D), dim3(256), 0,
                       mrcnn::as_stream(stream), mask_logits, label, bbox, D, M, Kc, im_h, im_w, out);
    return mrcnn::check_launch("paste_masks");
}
