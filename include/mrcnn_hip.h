/*
 * mrcnn_hip.h — C ABI of libmrcnn_hip.so, the MI355X (gfx950) implementation of
 * the Mask R-CNN ResNet-C4 hot path of wkentaro/chainer-mask-rcnn.
 *
 * The reference has no FFI of its own (it is 100 % Python; its device code is
 * CuPy kernel strings and third-party cuDNN/chainercv calls), so each entry
 * point below cites the reference interface it replaces (file:line under
 * /root/reference) and is what a ctypes binding on the reference side would
 * bind (INTEGRATION.md shows that binding).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error (never throws,
 *     never aborts); mrcnn_last_error() returns a message for the calling thread.
 *   - all tensor pointers are DEVICE pointers owned by the caller (allocated by
 *     PyTorch-ROCm); the library neither frees nor retains them.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work
 *     is enqueued asynchronously on it, no implicit device synchronisation.
 *   - activations are NHWC fp32 ("channels-last": the physical layout of a
 *     torch channels_last tensor whose logical shape is the reference's NCHW);
 *     conv filters are KRSC = (out, kh, kw, in) fp32 — the channels_last image
 *     of chainer's (out, in, kh, kw).
 *   - boxes are (y_min, x_min, y_max, x_max) fp32 as everywhere in the reference
 *     model (models/mask_rcnn.py:69); RoI rows for pooling are
 *     (batch_index, x1, y1, x2, y2) as functions/roi_align_2d.py:540-541.
 */
#ifndef MRCNN_HIP_H_
#define MRCNN_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ----------------------------------------------------------- */
const char *mrcnn_last_error(void);
int mrcnn_abi_version(void);
/* Number of compute units / device name of the current device (diagnostics). */
int mrcnn_device_info(int *n_cu, char *name, int name_len);

/* Kernel timer: HIP events recorded on the launch stream around every hot kernel while
 * enabled.  bench.py uses it to report roofline numbers (average launch duration and
 * algorithmic flops / bytes per kernel kind; kinds follow the kernel symbols rocprofv3
 * reports).  mrcnn_profile_enable(mode) clears the records; mode 0 = off, 1 = every kind,
 * 2 = only the forward-form 128x128 conv GEMM, the dominant kernel symbol (~48 instead of
 * ~500 event pairs per train step, so the timed region is barely perturbed; the other kinds
 * still count launches / flops / bytes), 3 = count launches / flops / bytes of every kind and
 * time nothing (no events on the stream at all); any other value is an error.
 * mrcnn_profile_summary sums the records (synchronise the stream first). */
int mrcnn_profile_enable(int on);
int mrcnn_profile_num_kinds(void);
const char *mrcnn_profile_kind_name(int kind);
int mrcnn_profile_summary(int kind, double *total_ms, double *total_flops,
                          double *total_bytes, int64_t *launches);

/* ---- ROIAlign ---------------------------------------------------------- */
/* Replaces ROIAlign2D.forward_gpu (functions/roi_align_2d.py:162-290).
 * x (N,H,W,C) NHWC, rois (R,5) = (batch, x1, y1, x2, y2), y (R,PH,PW,C). */
int mrcnn_roi_align_fwd(const float *x, const float *rois, float *y,
                        int N, int H, int W, int C, int R, int PH, int PW,
                        float spatial_scale, int sampling_ratio, void *stream);
/* Replaces ROIAlign2D.backward_gpu (functions/roi_align_2d.py:391-524).
 * gy (R,PH,PW,C) -> gx (N,H,W,C); gx is zero-filled by the call. */
int mrcnn_roi_align_bwd(const float *gy, const float *rois, float *gx,
                        int N, int H, int W, int C, int R, int PH, int PW,
                        float spatial_scale, int sampling_ratio, void *stream);

/* Strided-bin variants: only the bins (oh*bin_stride, ow*bin_stride) of the PH x PW grid are
 * produced / consumed, y and gy are (R, ceil(PH/bs), ceil(PW/bs), C).  res5's first block
 * reads the 14x14 RoI features through 1x1 stride-2 convolutions only
 * (models/mask_rcnn_resnet.py:131-133 with roi_size // 7 == 2), i.e. it uses just the even
 * bins; bin_stride = 2 skips the three quarters of the ROIAlign output nobody reads
 * (identical values for the bins that are read).
 *
 * order (may be NULL): a permutation of 0..R-1 on the device, the sequence in which the RoIs are
 * PROCESSED (results are written to their own rows and do not depend on it).  An XCD works on a
 * contiguous run of that sequence, so an order that keeps neighbouring RoIs together
 * (functions.roi_align_2d.spatial_order: image, 6-row band, x centre) lets it read its region of
 * the feature map from HBM once: 65 -> 52 us at the C2 shape. */
int mrcnn_roi_align_fwd_ex(const float *x, const float *rois, float *y,
                           int N, int H, int W, int C, int R, int PH, int PW, int bin_stride,
                           float spatial_scale, int sampling_ratio, const int *order,
                           void *stream);
/* ROIAlign with a fused per-channel epilogue: y = relu?(roi_align(x) * scale[c] + shift[c]).
 * Used where the head pools the OUTPUT of res5.a's 1x1 convolutions instead of their input
 * (models/mask_rcnn_resnet.py:131-133, 168-176: ROIAlign, then conv1 / the shortcut conv4, each 1x1,
 * followed by AffineChannel2D (+ ReLU)).  ROIAlign is linear over positions per channel and has no
 * constant term (skipped samples contribute zero, functions/roi_align_2d.py:228-236), a 1x1
 * convolution without bias is linear over channels per position, so
 *     affine(conv1x1(roi_align(x))) == affine(roi_align(conv1x1(x)))
 * in exact arithmetic: the convolution runs on the N*H*W map pixels instead of the R*7*7 pooled
 * ones (8 568 instead of 50 176 rows at the C2 shape) and this call applies the AffineChannel2D
 * (scale, shift: C floats each, both required) and the ReLU to the pooled values. */
int mrcnn_roi_align_fwd_affine(const float *x, const float *rois, float *y,
                               int N, int H, int W, int C, int R, int PH, int PW, int bin_stride,
                               float spatial_scale, int sampling_ratio, const int *order,
                               const float *scale, const float *shift, int relu, void *stream);
/* Backward, two forms.  ws = NULL (and mrcnn_roi_align_bwd): gather form with one atomic add
 * per (RoI patch pixel, channel) into a zero-filled gx — the order of the fp32 additions, like
 * the reference's atomicAdd kernel (:508-515), varies from run to run.  ws = a 16-byte aligned
 * workspace of mrcnn_roi_align_bwd_workspace_bytes(N, H, W, R, PH, PW, bin_stride): pixel-owner
 * form — per RoI the patch extent and the separable bilinear weights of every feature row and
 * column are tabulated once, then every gx pixel is summed in registers over the RoIs that cover
 * it (in RoI order, bins in row-major order) and stored once: no atomics, no zero-fill,
 * bit-reproducible, ~10x faster at the C2 shape (H, W <= 65535; otherwise the first form runs).
 * A bin is applied to the 8-pixel row tile it touches with weight 0 on the pixels it does not
 * touch: identical sums for finite gy; a non-finite gy element reaches those neighbours too. */
int64_t mrcnn_roi_align_bwd_workspace_bytes(int N, int H, int W, int R, int PH, int PW,
                                            int bin_stride);
int mrcnn_roi_align_bwd_ex(const float *gy, const float *rois, float *gx,
                           int N, int H, int W, int C, int R, int PH, int PW, int bin_stride,
                           float spatial_scale, int sampling_ratio, void *ws, void *stream);
/* Same call with the workspace's extent: fails (returns non-zero, mrcnn_last_error) when ws is
 * non-NULL and ws_bytes < mrcnn_roi_align_bwd_workspace_bytes(N, H, W, R, PH, PW, bin_stride) — a
 * caller that sized ws by an older formula gets an error instead of a table write past its buffer.
 * mrcnn_roi_align_bwd_ex is this call with the extent taken on trust. */
int mrcnn_roi_align_bwd_ws(const float *gy, const float *rois, float *gx,
                           int N, int H, int W, int C, int R, int PH, int PW, int bin_stride,
                           float spatial_scale, int sampling_ratio, void *ws, int64_t ws_bytes,
                           void *stream);

/* ---- row-sparse backward of a 3x3 / stride 1 / pad 1 convolution --------------------------
 * The reference back-propagates the RPN losses through conv1 of
 * models/region_proposal_network.py:75-80 densely, although the losses ignore every anchor but
 * the <= 256 sampled ones per image (models/mask_rcnn_train_chain.py:150-166): the gradient of
 * conv1's output is exactly zero outside their map positions.  With the list of those positions
 * (`rows`, sorted indices into N*H*W, and `lookup`, position -> row or -1; both built where the
 * anchor targets are) the backward is: gather the rows' 3x3 input patches and gradient rows,
 * run weight and data gradient as the 1x1 problem (N = n_rows, C = 9*C_in) through
 * mrcnn_conv2d_wgrad / mrcnn_conv2d_dgrad_wt, and sum the patch gradients back per map pixel in
 * a fixed tap order (every gx element written once, no atomics).  x (N,H,W,C), g (N,H,W,K) NHWC,
 * patches (n_rows, 3, 3, C), g_rows (n_rows, K), g_patches (n_rows, 3, 3, C), gx (N,H,W,C). */
int mrcnn_sparse3x3_gather(const float *x, const float *g, const int32_t *rows, int n_rows,
                           int N, int H, int W, int C, int K, float *patches, float *g_rows,
                           void *stream);
int mrcnn_sparse3x3_scatter(const float *g_patches, const int32_t *lookup, int N, int H, int W,
                            int C, float *gx, void *stream);

/* ---- AffineChannel2D ---------------------------------------------------- */
/* Replaces AffineChannel2DFunction.forward / backward
 * (functions/affine_channel_2d.py:10-22, :38-56).  x,y (M,C) NHWC rows,
 * W,b (C).  Inside the model the affine is fused into the conv epilogue
 * (mrcnn_conv2d_fwd); these stand-alone entry points serve the public
 * `affine_channel_2d` function. gW/gb may be NULL (skipped). */
int mrcnn_affine_fwd(const float *x, const float *W, const float *b, float *y,
                     int64_t M, int C, void *stream);
int64_t mrcnn_colsum_workspace_bytes(int C);
int mrcnn_affine_bwd(const float *x, const float *W, const float *gy,
                     float *gx, float *gW, float *gb, int64_t M, int C,
                     void *ws /* mrcnn_colsum_workspace_bytes(C) */, void *stream);

/* ---- Proposal path (integer results) ------------------------------------ */
/* loc2bbox + clip + min-size validity: chainercv ProposalCreator steps 1-3
 * (call site models/region_proposal_network.py:135-138).
 * anchor (n,4), loc (n,4) -> roi (n,4); valid[i] = h>=min_size && w>=min_size. */
int mrcnn_decode_clip(const float *anchor, const float *loc, float *roi,
                      uint8_t *valid, int n, float img_h, float img_w,
                      float min_size, void *stream);
/* Stable descending top-k over valid entries: `argsort(score)[::-1][:k]` of
 * ProposalCreator (tie rule: lower index first).  order[j] = index of the j-th
 * best valid score (j < *n_out), n_out (device int32) = min(k, #valid).
 * valid may be NULL (all valid). */
int64_t mrcnn_topk_workspace_bytes(int n);
int mrcnn_topk_desc(const float *score, const uint8_t *valid, int n, int k,
                    int32_t *order, int32_t *n_out, void *ws, void *stream);
/* `groups` problems of the same size in one set of launches (a batch's images): score and
 * valid are (groups, n), order (groups, k), n_out (groups); ws holds
 * groups x round_up(mrcnn_topk_workspace_bytes(n), 64) bytes. */
int mrcnn_topk_desc_batched(const float *score, const uint8_t *valid, int groups, int n, int k,
                            int32_t *order, int32_t *n_out, void *ws, void *stream);
/* dst[j,:] = src[idx[j],:] for j < *n_dev (n_dev device int32, rows of `cols`
 * fp32); rows j >= *n_dev up to n_max are zero-filled. */
int mrcnn_gather_rows(const float *src, const int32_t *idx, const int32_t *n_dev,
                      int n_max, int cols, float *dst, void *stream);
/* chainercv non_maximum_suppression on score-sorted boxes (call
 * sites models/mask_rcnn.py:193-194 and ProposalCreator).  bbox (n_max,4)
 * sorted by descending score, *n_dev boxes valid.  Writes keep[0..*n_keep) =
 * indices into bbox (ascending = descending score), at most `limit` (<=0: no
 * limit).  mask_ws: caller-owned workspace of mrcnn_nms_workspace_bytes(n_max, 1). */
int64_t mrcnn_nms_workspace_bytes(int n_max, int groups);
int mrcnn_nms_sorted(const float *bbox, const int32_t *n_dev, int n_max,
                     float thresh, int limit, int32_t *keep, int32_t *n_keep,
                     void *mask_ws, void *stream);
/* Batched form for per-class NMS (MaskRCNN._suppress, models/mask_rcnn.py:178-202):
 * `groups` independent problems, group g has n_dev[g] boxes at bbox + g*n_max*4,
 * writes keep + g*n_max and n_keep[g]; workspace mrcnn_nms_workspace_bytes(n_max, groups). */
int mrcnn_nms_sorted_batched(const float *bbox, const int32_t *n_dev, int groups,
                             int n_max, float thresh, int limit, int32_t *keep,
                             int32_t *n_keep, void *mask_ws, void *stream);

/* ---- Convolution family (fp32 implicit GEMM on MFMA) ---------------------- */
/* Epilogue flags */
#define MRCNN_EPI_BIAS     1   /* y += bias[c]                                  */
#define MRCNN_EPI_AFFINE   2   /* y = y*scale[c] + shift[c]  (AffineChannel2D)  */
#define MRCNN_EPI_RESIDUAL 4   /* y += residual[m,c]                            */
#define MRCNN_EPI_RELU     8   /* y = max(y,0)                                  */
#define MRCNN_EPI_ACCUM    16  /* y += previous contents of y (dgrad fan-in)    */
#define MRCNN_EPI_EXACT_SIGNS 32 /* mrcnn_conv3x3_wino_fwd only, see there        */

typedef struct {
    int N, H, W, C;        /* input  (N,H,W,C)  NHWC                         */
    int K, R, S;           /* filter (K,R,S,C)  KRSC                         */
    int stride, pad;
    int P, Q;              /* output (N,P,Q,K)  NHWC                         */
} mrcnn_conv_desc;

/* Replaces chainer L.Convolution2D forward (cuDNN) + following AffineChannel2D,
 * residual add and ReLU of chainer's BottleneckA/B (call sites
 * models/region_proposal_network.py:75-80,124-131, models/mask_rcnn_resnet.py:131-143,
 * chainer ResNet50Layers via models/resnet_extractor.py:93).
 * y = epi( conv(x, w) ).  scale/shift/bias/residual may be NULL when unused.
 * split_ws (forward and backward-data entry points): NULL or a scratch buffer of
 * mrcnn_conv2d_split_workspace_bytes() bytes.  With it, a small-M problem whose 64x64 tiles
 * do not divide evenly over the 256 CUs (e.g. 536 tiles: the busiest CU would run 3, the
 * average 2.09) runs its leftover rows split along K into many short workgroups whose partial
 * sums are combined, in a fixed order, by a small epilogue kernel. */
int64_t mrcnn_conv2d_split_workspace_bytes(void);
/* Developer switches for A/B measurements (results are identical either way):
 *   "position_major_rows" (default 1): forward-form convolutions over many small maps (RoI
 *   features, 3x3 / pad 1 on 7x7) order their GEMM rows position-major so that the K slices of
 *   filter taps that fall into the zero padding for every row of a tile are skipped.
 *   "fused_tail" (default 512 = target number of pieces, 0 = off): the rows beyond the last full
 *   round of resident workgroups run as K-split pieces INSIDE the main launch (dispatched last,
 *   they fill the CUs while the final round drains) instead of a separate remainder launch;
 *   summation order of those rows differs between the two settings (both deterministic).
 * Arithmetic selection (NOT results-identical: same accuracy class, different rounding):
 *   "split_bf16" (default 3 since round 4; 0 = fp32 MFMA everywhere; bit 0: the 128x128
 *   forward-form and weight-gradient kernels, bit 1: the 64x64 forward-form kernel): operands are
 *   staged as three bf16 planes whose sum is the fp32 element exactly, six bf16 MFMAs per K step,
 *   fp32 accumulation.  Error against float64 is at (measured: below) the fp32-MFMA kernels' level
 *   (tests/test_gpu_split_bf16.py, tests/test_split_arithmetic_cpu.py).
 *   "big_min_tiles" (default 384): fewest 128x128 tiles for which the 128x128 kernels are used
 *   (1 forces them; lets small test problems exercise the kernels of the full-size step).
 *   "big_split_k" (default -1 since round 6): with -1, a small-M, K-deep forward-form problem whose
 *   128x128 tiles cut along K fill one round of the resident workgroups (the batch-2 res4 3x3 layers)
 *   runs that way, slabs summed in order by the epilogue kernel, instead of as 64x64 tiles; 0 = off;
 *   k > 0 = aim at k workgroups (probe); "big_split_min_slices" (default 16): fewest K slices per slab.
 *   "tiny_split" (default 1): launches of <= 128 tiles and >= 32 K slices are cut along K.
 *   Summation order differs between these settings (each deterministic).
 * Kernel selection, results bit-identical either way (round 6):
 *   "w8" (default 1): large pointwise forward-form launches of the split arithmetic (K >= "w8_min_k",
 *   default 256, and at least 768 tiles) on 256x128 tiles / 512-thread workgroups.
 *   "pw" (default 3): bit 0 = pointwise (1x1 / stride 1) forward-form launches, bit 1 = 3x3 / stride 1 /
 *   pad 1 ones run instantiations with that geometry as compile-time constants.
 *   "roi_fwd_lanes" / "roi_bwd_lanes" (default 256): lanes per ROIAlign workgroup. */
int mrcnn_set_tuning(const char *name, int value);
int mrcnn_conv2d_fwd(const mrcnn_conv_desc *d, const float *x, const float *w,
                     const float *bias, const float *scale, const float *shift,
                     const float *residual, float *y, int epi_flags, void *split_ws,
                     void *stream);
/* Gradient w.r.t. the input.  gy (N,P,Q,K) -> gx (N,H,W,C).  With
 * MRCNN_EPI_ACCUM gx += result (fan-in of several consumers). */
int mrcnn_conv2d_dgrad(const mrcnn_conv_desc *d, const float *gy, const float *w,
                       float *gx, int epi_flags, void *stream);
/* Gradient w.r.t. the filter, gw (K,R,S,C) (bias gradient: mrcnn_colsum of gy).
 * ws: split-K workspace of mrcnn_conv2d_wgrad_workspace_bytes(d) (NULL: no split). */
int64_t mrcnn_conv2d_wgrad_workspace_bytes(const mrcnn_conv_desc *d);
int mrcnn_conv2d_wgrad(const mrcnn_conv_desc *d, const float *x, const float *gy,
                       float *gw, void *ws, void *stream);
/* Extended backward entry points.  The backward of a conv's fused epilogue (AffineChannel2D
 * scale s[k], then ReLU with output y) can be applied at either end of the gradient tensor:
 *
 *  consumer side — while gy is staged into LDS: g = gy * (mask_y > 0) * in_scale[k]
 *      (mask_y: output of the ReLU that followed THIS conv, same shape as gy; in_scale: its
 *      affine scale (K); either may be NULL).  Costs 16 more registers per lane, so these
 *      variants run two workgroups per CU.
 *  producer side — in the dgrad epilogue that WRITES the gradient:
 *      gx = (acc * out_scale[c] + res_g * (res_y > 0)) * (out_mask_y > 0)
 *      out_mask_y / out_scale: ReLU output / affine scale of the conv that produced this
 *      conv's INPUT ((N,H,W,C) / (C); either may be NULL).  The next dgrad / wgrad down the
 *      chain then need no mask and run the plain, three-workgroups-per-CU kernels.
 *      res_g (+ optional res_y): identity-shortcut gradient of a bottleneck, (N,H,W,C);
 *      stride 1 only.  With MRCNN_EPI_ACCUM the previous gx is added before the mask.
 *  A per-output-channel scale that cannot go to the producer (the block-top gradient feeds
 *  conv3, conv4 and the shortcut with different scales) is folded into the filter for dgrad
 *  (row_scale of mrcnn_filter_flip_transpose) and applied to the rows of gw in the wgrad
 *  epilogue (out_row_scale): gw[k] = s[k] * sum_m gy[m,k] x[m].
 * Used by functions/conv.py:_StageFn (a whole ResNet stage as one autograd node) and
 * _BottleneckFn; chainer runs each of these as separate elementwise kernels. */
int mrcnn_conv2d_dgrad_ex(const mrcnn_conv_desc *d, const float *gy, const float *w,
                          float *gx, int epi_flags, const float *mask_y,
                          const float *in_scale, const float *res_g, const float *res_y,
                          const float *out_mask_y, const float *out_scale, void *split_ws,
                          void *stream);
int mrcnn_conv2d_wgrad_ex(const mrcnn_conv_desc *d, const float *x, const float *gy,
                          float *gw, void *ws, const float *mask_y, const float *in_scale,
                          const float *out_row_scale, void *stream);
/* Winograd F(4x4,3x3) path (csrc/conv_winograd.h) for 3x3 / stride 1 / pad 1 convolutions —
 * the algorithm cuDNN selects for the same layers in the reference (chainer autotune off:
 * cudnnGetConvolutionForwardAlgorithm; call sites as mrcnn_conv2d_fwd).  A third of the MFMA
 * work of the direct form on the RoI head's 7x7 maps.  Interpolation points 0, +-1, 1/2, -2:
 * fp32 error max 3.4e-6 / rms 3.7e-7 of the tensor scale against an fp64 direct convolution
 * (direct fp32: 3.5e-7 / 5.7e-8), inside the 1e-4 parity tolerance.
 *   fwd:   y = epi(conv(x, w)), epi_flags: MRCNN_EPI_AFFINE (scale, shift) or MRCNN_EPI_BIAS
 *          (the bias in `shift`, scale NULL), and MRCNN_EPI_RELU.  With MRCNN_EPI_EXACT_SIGNS
 *          every output whose pre-activation lies within the propagated Winograd rounding
 *          bound of zero (a few in 10^5) is recomputed as a direct fp32 dot product, so the
 *          ReLU decisions — what a recorded graph's backward masks depend on — have the
 *          accuracy of the direct kernel (w must be given).  u: NULL (the filter is
 *          transformed per call) or the output of mrcnn_conv3x3_wino_filter for this w
 *          (mrcnn_conv3x3_wino_u_bytes(d) bytes; inference keeps it while w is unchanged, w may
 *          then be NULL).  v: NULL or a buffer of mrcnn_conv3x3_wino_v_bytes(d) that receives the
 *          transformed input (36 x tiles x C), which mrcnn_conv3x3_wino_wgrad consumes.
 *   dgrad: gx = (dgrad(gy * w_row_scale[k]) * out_scale[c]) masked by (out_mask_y > 0)
 *          (any of the three may be NULL; same meaning as in mrcnn_conv2d_dgrad_wt).
 *   wgrad: gw (K,3,3,C) from gy and exactly one of x (the raw input, transformed into the
 *          scratch) and v (the forward's kept transform); out_row_scale[k] (or NULL)
 *          multiplies gw's rows.
 * ws: scratch of mrcnn_conv3x3_wino_workspace_bytes(d), private to the stream. */
int64_t mrcnn_conv3x3_wino_v_bytes(const mrcnn_conv_desc *d);
int64_t mrcnn_conv3x3_wino_workspace_bytes(const mrcnn_conv_desc *d);
int64_t mrcnn_conv3x3_wino_u_bytes(const mrcnn_conv_desc *d);
int mrcnn_conv3x3_wino_filter(const mrcnn_conv_desc *d, const float *w, float *u, void *stream);
int mrcnn_conv3x3_wino_fwd(const mrcnn_conv_desc *d, const float *x, const float *w,
                           const float *u, const float *scale, const float *shift, float *y,
                           int epi_flags, float *v, void *ws, void *stream);
int mrcnn_conv3x3_wino_fixup_count(const mrcnn_conv_desc *d, const void *ws, void *stream,
                                   int *count);   /* diagnostics: outputs recomputed by the last
                                                     EXACT_SIGNS forward on ws (synchronises) */
int mrcnn_conv3x3_wino_dgrad(const mrcnn_conv_desc *d, const float *gy, const float *w,
                             const float *w_row_scale, float *gx, const float *out_scale,
                             const float *out_mask_y, void *ws, void *stream);
int mrcnn_conv3x3_wino_wgrad(const mrcnn_conv_desc *d, const float *x, const float *v,
                             const float *gy, float *gw, const float *out_row_scale, void *ws,
                             void *stream);
/* Stride-1 dgrad expressed as a forward-form convolution of gy with the flipped, transposed
 * filter wT[c][R-1-r][S-1-s][k] = w[k][r][s][c] * row_scale[k] (both GEMM operands
 * K-contiguous).  mrcnn_filter_flip_transpose builds wT (C,R,S,K) from w (K,R,S,C)
 * (row_scale (K) or NULL); it moves R*S*K*C*8 bytes, negligible next to the dgrad.
 * Same extra arguments as _ex.  Round 6: mrcnn_conv2d_dgrad_wt also takes stride > 1 for 1x1 / pad 0
 * filters (the strided projections of res3.a / res4.a): gy is gathered densely, the rows are
 * scattered to the strided pixels of gx, which the call zero-fills unless MRCNN_EPI_ACCUM is set
 * (residual gradient / output mask arguments must be NULL there). */
int mrcnn_filter_flip_transpose(const float *w, float *wT, int K, int R, int S, int C,
                                const float *row_scale, void *stream);
/* The same for n filters in one launch (host arrays of n device pointers / sizes; row_scale
 * or its entries may be NULL): a ResNet stage's backward needs ~14 of them. */
int mrcnn_filter_flip_transpose_batched(int n, const void *const *w, void *const *wT,
                                        const int *K, const int *R, const int *S, const int *C,
                                        const void *const *row_scale, void *stream);
int mrcnn_conv2d_dgrad_wt(const mrcnn_conv_desc *d, const float *gy, const float *wT,
                          float *gx, int epi_flags, const float *mask_y,
                          const float *in_scale, const float *res_g, const float *res_y,
                          const float *out_mask_y, const float *out_scale, void *split_ws,
                          void *stream);
/* Stem: conv1 7x7/2 pad 3 with bias of chainer ResNet50Layers (SURVEY.md A.1;
 * models/resnet_extractor.py:65-67) fused with bn1-as-affine and ReLU.  x4 is
 * the image padded to 4 channels (N,H,W,4); w784 is the filter laid out
 * (K,7,8,4) with zeros at s=7 and c=3.  Forward only (frozen, :86-87). */
int mrcnn_conv_stem_fwd(const float *x4, const float *w784, const float *bias,
                        const float *scale, const float *shift, float *y, int N,
                        int H, int W, int K, int epi_flags, void *stream);

/* Replaces chainer L.Deconvolution2D(2048,256,2,stride=2) (models/mask_rcnn_resnet.py:138-139,193).
 * x (N,H,W,C) -> y (N,2H,2W,K); filter w (C,2,2,K) = channels_last image of
 * chainer's (in,out,kh,kw).  Epilogue: bias + optional ReLU. */
int mrcnn_deconv2x2s2_fwd(const float *x, const float *w, const float *bias,
                          float *y, int N, int H, int W, int C, int K,
                          int epi_flags, void *stream);
/* The same forward on the TRANSPOSED filter wT (4K, C) = mrcnn_filter_flip_transpose(w, wT, C, 1, 1, 4K)
 * — a 1x1 convolution (both operands K-contiguous: the split-operand arithmetic) whose output columns
 * (a, b, o) the epilogue scatters to y[n, 2y + a, 2x + b, o].  Same values up to fp32 rounding. */
int mrcnn_deconv2x2s2_fwd_wt(const float *x, const float *wT, const float *bias,
                             float *y, int N, int H, int W, int C, int K, int epi_flags,
                             void *stream);
int mrcnn_deconv2x2s2_dgrad(const float *gy, const float *w, float *gx,
                            int N, int H, int W, int C, int K, void *stream);
int64_t mrcnn_deconv2x2s2_wgrad_workspace_bytes(int N, int H, int W, int C, int K);
int mrcnn_deconv2x2s2_wgrad(const float *x, const float *gy, float *gw,
                            int N, int H, int W, int C, int K, void *ws,
                            void *stream);

/* ReLU / affine backward helper: g[m,c] = gy[m,c] * (y[m,c] > 0) * scale[c]
 * (scale NULL = 1; y NULL = no mask).  Backward of the fused conv epilogue. */
int mrcnn_epilogue_bwd(const float *gy, const float *y, const float *scale,
                       float *g, int64_t M, int C, void *stream);
/* Column sums: out[c] = sum_m g[m,c] (bias gradients); ws of
 * mrcnn_colsum_workspace_bytes(C). Deterministic (two-level fixed-order sum). */
int mrcnn_colsum(const float *g, float *out, int64_t M, int C, void *ws, void *stream);

/* ---- Pooling ------------------------------------------------------------ */
/* F.max_pooling_2d(x,3,stride=2,pad=1), cover_all=True (models/resnet_extractor.py:69).
 * x (N,H,W,C) -> y (N,P,Q,C), P = (H+2-3+1)/2+1. Forward only: the stem is
 * frozen (unchain_backward at res2, models/resnet_extractor.py:86-87). */
int mrcnn_maxpool3x3s2p1_fwd(const float *x, float *y, int N, int H, int W, int C,
                             int P, int Q, void *stream);
/* F.average_pooling_2d(res5, 7, stride=7) on a 7x7 map (models/mask_rcnn_resnet.py:188):
 * x (R,HW,C) -> y (R,C) mean over HW; backward broadcasts gy/HW. */
int mrcnn_avgpool_fwd(const float *x, float *y, int R, int HW, int C, void *stream);
int mrcnn_avgpool_bwd(const float *gy, float *gx, int R, int HW, int C,
                      int accumulate, void *stream);
/* Gradient entering res5's last block from the two consumers of its output y (R,HW,C)
 * (models/mask_rcnn_resnet.py:186-195: average pooling -> cls_loc / score, and deconv6 on the
 * foreground rows), already through y's ReLU — one pass instead of avgpool backward + row
 * scatter-add + ReLU mask (three passes over the 411 MB tensor):
 *   g[r,p,c] = (g_pool[r,c] / HW + (slot[r] >= 0 ? g_rows[slot[r],p,c] : 0)) * (y[r,p,c] > 0)
 * g_rows (F,HW,C) / slot (R) int32 may both be NULL (no row consumer).  Same operation order
 * as the three separate kernels, so the result is bit-identical to them. */
int mrcnn_head_tail_bwd(const float *g_pool, const float *g_rows, const int32_t *slot,
                        const float *y, float *g, int R, int HW, int C, void *stream);

/* ---- Losses (models/mask_rcnn_train_chain.py:163-181,192-213) -------------- */
/* All loss kernels write loss[0] (device scalar, already normalised) and, when gx
 * is non-NULL, the gradient of that scalar w.r.t. x (the total loss is a plain
 * sum, :180).  ws: workspace of mrcnn_loss_workspace_bytes(rows) bytes, rows = R
 * for softmax_ce, 0 otherwise.  No host synchronisation. */
int64_t mrcnn_loss_workspace_bytes(int rows);
/* F.sigmoid_cross_entropy(x, t), t in {-1,0,1}; x element i at x[i*x_stride + x_off(i)]:
 * plain form: x (n) contiguous. */
int mrcnn_sigmoid_ce(const float *x, const int32_t *t, int64_t n, float *loss,
                     float *gx, void *ws, void *stream);
/* mask form: x (R, Kc, HW) NHWC rows = (R, HW, Kc); selects channel label[r]-1
 * for row r (models/mask_rcnn_train_chain.py:176-178); t (R,HW) in {-1,0,1};
 * gx (R,HW,Kc) fully written (zeros elsewhere). */
int mrcnn_mask_sigmoid_ce(const float *x, const int32_t *label, const int32_t *t,
                          int R, int HW, int Kc, float *loss, float *gx,
                          void *ws, void *stream);
/* F.softmax_cross_entropy(x (R,ncls) with row stride ldx, t) ignore -1. */
int mrcnn_softmax_ce(const float *x, int ldx, const int32_t *t, int R, int ncls,
                     float *loss, float *gx, int ldg, void *ws, void *stream);
/* _fast_rcnn_loc_loss: pred (n,4) [row r at pred + r*ld + 4*cls[r] when cls != NULL],
 * gt_loc (n,4), gt_label (n); in_weight = label>0; normaliser = #(label>=0).
 * gx written with the same addressing (caller zero-fills the rest). */
int mrcnn_smooth_l1(const float *pred, int ld, const int32_t *cls,
                    const float *gt_loc, const int32_t *gt_label, int n,
                    float sigma, float *loss, float *gx, void *ws, void *stream);
/* F.softmax(x) row-wise (models/mask_rcnn.py:208) */
int mrcnn_softmax(const float *x, int ldx, float *y, int ldy, int R, int ncls,
                  void *stream);

/* ---- Optimizer (chainer MomentumSGD + WeightDecay, examples/train_common.py:176-180) */
/* g += wd*p; v = momentum*v - lr*g; p += v, over one flat arena of n floats.
 * grad_scale multiplies g first (1/world_size after an all-reduce sum). */
int mrcnn_sgd_momentum_wd(float *p, const float *g, float *v, int64_t n, float lr,
                          float momentum, float wd, float grad_scale,
                          void *stream);
/* Same update; with zero_grad != 0 the gradient arena is cleared in the same pass (chainer's
 * cleargrads() before the next backward, examples/train_common.py:226-231 via
 * StandardUpdater.update_core): a parameter whose backward does not run in the next step then
 * contributes a zero gradient instead of a stale one. */
int mrcnn_sgd_momentum_wd_ex(float *p, float *g, float *v, int64_t n, float lr,
                             float momentum, float wd, float grad_scale, int zero_grad,
                             void *stream);

/* Per-class `prob > thresh` filter + stable descending sort + gather, all foreground
 * classes in one launch: the first half of MaskRCNN._suppress (models/mask_rcnn.py:178-202).
 * prob (R, n_class), cls_bbox (R, n_class, 4) -> sorted_boxes (n_class-1, R, 4),
 * sorted_prob (n_class-1, R), counts (n_class-1); feed to mrcnn_nms_sorted_batched. */
int64_t mrcnn_detect_sort_workspace_bytes(int R, int n_class);
int mrcnn_detect_sort(const float *prob, const float *cls_bbox, int R, int n_class,
                      float thresh, float *sorted_boxes, float *sorted_prob,
                      int32_t *counts, void *ws, void *stream);

/* ---- Image I/O at both ends of predict (device restatement of the reference's cv2 calls) ---- */
/* MaskRCNN.prepare (models/mask_rcnn.py:152-176) for one image + its slot in the zero-padded
 * batch (datasets/concat_examples.py:20-26): bilinear resize by `scale` (OpenCV INTER_LINEAR
 * float rule) of src (C=3,H,W) — fp32, or uint8 as decoded when src_is_u8 — into image n of
 * dst (N,dstH,dstW,3) NHWC, minus the RGB mean (mean_host: HOST pointer to 3 floats).
 * (outH,outW) = rounded scaled size.  flip_x != 0 mirrors the resized image left-right: the
 * random_flip of datasets/transforms.py:38-40 folded into the same pass. */
int mrcnn_prepare_image(const void *src_chw, int src_is_u8, int C, int H, int W, double scale,
                        const float *mean_host, float *dst_nhwc, int dstH, int dstW,
                        int outH, int outW, int n, int flip_x, void *stream);
/* segm_results / expand_boxes (models/mask_rcnn.py:44-107): mask_logits (D,M,M,Kc) NHWC head
 * outputs, label (D) foreground class per detection, bbox (D,4) yx in image coordinates ->
 * out (D,im_h,im_w) uint8 {0,1}: sigmoid, 1-pixel zero pad, box expanded by (M+2)/M and
 * truncated to int, INTER_LINEAR resize, threshold 0.5, paste. */
int mrcnn_paste_masks(const float *mask_logits, const int32_t *label, const float *bbox,
                      int D, int M, int Kc, int im_h, int im_w, uint8_t *out, void *stream);

/* Second half of MaskRCNN._suppress (models/mask_rcnn.py:195-202): the rows kept by
 * mrcnn_nms_sorted_batched (keep (G,R), n_keep (G)) of every class packed densely, class after
 * class and in keep order: bbox (<= G*R, 4), label, score, *total = number of rows. */
int mrcnn_detect_compact(const int32_t *keep, const int32_t *n_keep, const float *sorted_boxes,
                         const float *sorted_prob, int G, int R, float *bbox, int32_t *label,
                         float *score, int32_t *total, void *stream);

/* ---- Inference post-processing (models/mask_rcnn.py:204-265) ---------------- */
/* Per-class decode: cls_bbox[r,l,:] = clip(loc2bbox(roi[r]/scale,
 * cls_loc[r,l,:]*std+mean), 0, size) for all classes (:225-240). */
/* mean4/std4 are HOST pointers to 4 doubles (loc_normalize_mean/std): the reference builds
 * them from Python tuples, i.e. as float64 arrays, so `loc * std + mean` is evaluated in double
 * and rounded to fp32 once (pinned by tests/golden/to_bboxes.npz). */
int mrcnn_decode_cls_boxes(const float *roi, const float *cls_loc, int ld_loc,
                           float *cls_bbox, int R, int n_class, float scale,
                           const double *mean4, const double *std4, float size_h,
                           float size_w, void *stream);

/* ---- Target creators: the device half ------------------------- */
/* The deterministic arithmetic of ProposalTargetCreator.__call__
 * (models/utils/proposal_target_creator.py:121-177) and chainercv's AnchorTargetCreator
 * (call site models/mask_rcnn_train_chain.py:153-158): IoU matrices, label rules, regression
 * targets, 14x14 mask targets.  The np.random draws stay with the caller (host), which reads
 * max_iou / the anchor labels back, draws, and passes the chosen indices in. */
/* chainercv bbox_iou(boxes_a (na,4), boxes_b (g,4)) reduced per row: max_iou (na), first
 * argmax (na); optional full matrix iou (na,g) and its column maxima col_max (g). */
int mrcnn_bbox_iou_argmax(const float *boxes_a, int na, const float *boxes_b, int g,
                          float *iou, float *max_iou, int32_t *argmax, float *col_max,
                          void *stream);
/* AnchorTargetCreator label rule before subsampling: -1 / 0 (max < neg) / 1 (row holds a
 * column maximum, or max >= pos). */
int mrcnn_anchor_labels(const float *iou, const float *max_iou, const float *gt_max, int na,
                        int g, float neg_iou_thresh, float pos_iou_thresh, int32_t *label,
                        void *stream);
/* label_inside[disabled[.]] = -1 (the host's draws), then the full-size targets: label (-1
 * outside the image) and loc = bbox2loc(anchor, bbox[argmax]) (0 outside). */
int mrcnn_anchor_targets_finish(const float *anchor_inside, const int32_t *inside_index,
                                int32_t *label_inside, const int32_t *argmax, const float *bbox,
                                int n_inside, const int32_t *disabled, int n_disabled,
                                int n_anchor, float *loc, int32_t *label, void *stream);
/* Rows chosen[0..n_sample) of the candidates (the first n_fg are foreground): sample_roi,
 * gt_roi_loc = (bbox2loc(roi, bbox[assigned]) - mean) / std, gt_roi_label (class + 1 | 0),
 * gt_index = assigned ground-truth box.  mean4 / std4 are HOST pointers to 4 floats. */
int mrcnn_proposal_targets_gather(const float *cand, const float *bbox, const int32_t *gt_label,
                                  const int32_t *assigned, const int32_t *chosen, int n_sample,
                                  int n_fg, const float *mean4_host, const float *std4_host,
                                  float *sample_roi, float *gt_roi_loc, int32_t *gt_roi_label,
                                  int32_t *gt_index, void *stream);
/* (n, M, M) int32 mask targets: rows < n_fg = crop of masks[gt_index] (uint8 (G,H,W)) at the
 * rounded RoI, cv2 INTER_LINEAR to M x M, > 0.5; other rows -1. */
int mrcnn_mask_targets(const uint8_t *masks, int G, int H, int W, const float *sample_roi,
                       const int32_t *gt_index, int n, int n_fg, int M, int32_t *out,
                       void *stream);

/* ---- Gradient exchange over RCCL / xGMI ---------------------------------------------- */
/* Replaces ChainerMN's communicator as the reference uses it
 * (examples/train_common.py:97-103 `chainermn.create_communicator('hierarchical')`, :178
 * `create_multi_node_optimizer`): bcast_data of the model before the first update and
 * allreduce_grad before every update.  One communicator per process (= per GPU).  RCCL
 * (ncclComm_t) is resolved at run time from the librccl.so already loaded into the process.
 * Collectives run on the communicator's OWN high-priority HIP stream, ordered against the
 * caller's streams with events only, so a bucket overlaps with the backward still running.
 *
 *   rank 0:   mrcnn_allreduce_unique_id(id)  -> 128 bytes, handed to every rank by the launcher
 *   all:      mrcnn_allreduce_init(id, rank, world, &comm)       (collective)
 *   per step: mrcnn_allreduce_bucket(comm, grads + lo, hi - lo, bucket, compute, side) ...
 *             mrcnn_allreduce_wait(comm, compute)  then the SGD launch (grad_scale = 1/world)
 */
#define MRCNN_COMM_ID_BYTES 128
int mrcnn_allreduce_unique_id(void *id128);
int mrcnn_allreduce_init(const void *id128, int rank, int world, void **comm);
int mrcnn_allreduce_destroy(void *comm);
/* rank / world of the communicator, RCCL version code, and which librccl was bound. */
int mrcnn_allreduce_info(void *comm, int *rank, int *world, int *rccl_version,
                         char *library, int library_len);
/* In-place SUM over ranks of buf[0..count) fp32, queued on the collective stream AFTER
 * everything queued so far on after_stream and (if non-NULL) after_stream2 — the compute
 * stream and the weight-gradient side stream.  bucket_id only labels the timing records. */
int mrcnn_allreduce_bucket(void *comm, float *buf, int64_t count, int bucket_id,
                           void *after_stream, void *after_stream2);
/* `stream` waits (device side) for every collective queued so far. */
int mrcnn_allreduce_wait(void *comm, void *stream);
/* rank `root`'s bytes to all ranks; ordered after and before `after_stream`. */
int mrcnn_allreduce_broadcast(void *comm, void *buf, int64_t bytes, int root,
                              void *after_stream);
/* HIP-event timing of the collectives: enable clears the records; bucket_times sums them
 * (bucket_id -1 = all) after the caller synchronised the device. */
int mrcnn_allreduce_timing(void *comm, int enable);
int mrcnn_allreduce_bucket_times(void *comm, int bucket_id, double *total_ms,
                                 double *total_bytes, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* MRCNN_HIP_H_ */
