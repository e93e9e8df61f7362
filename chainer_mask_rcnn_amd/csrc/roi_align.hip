// ROIAlign forward / backward for gfx950, NHWC.
//
// Replaces the two CuPy kernel strings of the reference
// (/root/reference/chainer_mask_rcnn/functions/roi_align_2d.py:171-288 fwd,
// :395-522 bwd), which run one thread per NCHW output element with
// uncoalesced 4-tap gathers.  Here one workgroup owns one output row of a RoI
// (roi, ph, all pw): all sample geometry is wave-uniform arithmetic derived
// from blockIdx, and the lanes run across channels, so every tap is a
// contiguous 16 B/lane read of the NHWC feature row and every result one
// contiguous 16 B/lane store.  HBM-bound: the forward writes R*PH*PW*C*4 bytes.
//
// Built with -ffp-contract=off and the reference's operation order so that the
// forward is bit-identical to the fp32 CPU restatement (oracle/roi_align_ref.c).
#include <algorithm>
#include <cstring>

#include "common.h"

namespace {

struct RoiGeom {
    int batch;
    float start_w, start_h, bin_h, bin_w;
    int grid_h, grid_w;
    float count;
};

// roi_align_2d.py:184-211
__device__ __forceinline__ RoiGeom roi_geom(const float *__restrict__ roi,
                                            float spatial_scale, int PH, int PW,
                                            int sampling_ratio)
{
    RoiGeom g;
    g.batch = (int)roi[0];
    g.start_w = roi[1] * spatial_scale;
    g.start_h = roi[2] * spatial_scale;
    float end_w = roi[3] * spatial_scale;
    float end_h = roi[4] * spatial_scale;
    float roi_w = fmaxf(end_w - g.start_w, 1.f);
    float roi_h = fmaxf(end_h - g.start_h, 1.f);
    g.bin_h = roi_h / (float)PH;
    g.bin_w = roi_w / (float)PW;
    g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_h / (float)PH);
    g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_w / (float)PW);
    g.count = (float)(g.grid_h * g.grid_w);
    return g;
}

struct Tap1D {
    int lo, hi;
    float l, h;  // weight of hi / lo
    bool valid;
};

// one axis of roi_align_2d.py:228-262
__device__ __forceinline__ Tap1D tap1d(float p, int size)
{
    Tap1D t;
    t.valid = !(p < -1.f || p > (float)size);
    if (p <= 0.f) p = 0.f;
    int lo = (int)p;
    int hi;
    if (lo >= size - 1) {
        hi = lo = size - 1;
        p = (float)lo;
    } else {
        hi = lo + 1;
    }
    t.lo = lo;
    t.hi = hi;
    t.l = p - (float)lo;
    t.h = 1.f - t.l;
    return t;
}

// Streaming stores for the pooled output (written once, read by the next kernel after 200 MB of
// other traffic): keeps the feature map's taps resident in L2.
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_stream(float *p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void store_stream(float4 *p, float4 v)
{
    __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4 *>(p));
}

// Optional fused epilogue of the forward: y = relu?(v * scale[c] + shift[c]) — the AffineChannel2D
// (+ ReLU) that follows a 1x1 convolution whose output is pooled (mrcnn_roi_align_fwd_affine).
template <typename V> struct FwdEpi { V scale, shift; bool on, relu; };
__device__ __forceinline__ float epi_apply(const FwdEpi<float> &e, float v)
{
    if (!e.on) return v;
    v = v * e.scale + e.shift;
    return e.relu ? fmaxf(v, 0.f) : v;
}
__device__ __forceinline__ float4 epi_apply(const FwdEpi<float4> &e, float4 v)
{
    if (!e.on) return v;
    v = make_float4(v.x * e.scale.x + e.shift.x, v.y * e.scale.y + e.shift.y,
                    v.z * e.scale.z + e.shift.z, v.w * e.scale.w + e.shift.w);
    if (e.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    return v;
}

// acc / count exactly as the reference divides, without the ~10-instruction fp32 division per
// component where it is not needed: count == 1 leaves acc, a power of two multiplies by its
// (exact) reciprocal — the correctly rounded quotient and product of the same real number.
// Workgroup-uniform choice.  (The geometry arithmetic and this division run on the vector ALU
// for every lane — gfx950 has no scalar float unit — and bounded the forward kernel once its
// loads were batched.)
template <typename V> struct VecOps;
template <typename V>
__device__ __forceinline__ V finish_mean(const V acc, int count_i, float count);

template <> struct VecOps<float> {
    static __device__ __forceinline__ float zero() { return 0.f; }
    static __device__ __forceinline__ float mad4(float acc, float w1, float v1, float w2, float v2,
                                                 float w3, float v3, float w4, float v4)
    {
        return acc + (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
    }
    static __device__ __forceinline__ float div(float a, float c) { return a / c; }
};
template <> struct VecOps<float4> {
    static __device__ __forceinline__ float4 zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ float4 mad4(float4 acc, float w1, float4 v1, float w2,
                                                  float4 v2, float w3, float4 v3, float w4,
                                                  float4 v4)
    {
        acc.x += (w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x);
        acc.y += (w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y);
        acc.z += (w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z);
        acc.w += (w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w);
        return acc;
    }
    static __device__ __forceinline__ float4 div(float4 a, float c)
    {
        return make_float4(a.x / c, a.y / c, a.z / c, a.w / c);
    }
};

template <>
__device__ __forceinline__ float finish_mean<float>(const float acc, int count_i, float count)
{
    if (count_i == 1) return acc;
    if ((count_i & (count_i - 1)) == 0) return acc * (1.f / count);
    return acc / count;
}
template <>
__device__ __forceinline__ float4 finish_mean<float4>(const float4 acc, int count_i, float count)
{
    if (count_i == 1) return acc;
    if ((count_i & (count_i - 1)) == 0) {
        const float r = 1.f / count;
        return make_float4(acc.x * r, acc.y * r, acc.z * r, acc.w * r);
    }
    return make_float4(acc.x / count, acc.y / count, acc.z / count, acc.w / count);
}

// Tap reuse.  Neighbouring samples of a bin are at most one pixel apart (the adaptive grid is
// ceil(bin size)), so a sample's columns are its left neighbour's (same lo) or start at the
// neighbour's right column (lo == previous hi): those values are taken from the neighbour's
// registers instead of being loaded again — 4 or 6 loads per pair of samples instead of 8.
// Workgroup-uniform decisions (the geometry is); the accumulation is unchanged.

// GH x GW samples of NB consecutive bins of one output row, all taps loaded before the first
// is used (NB * GH * GW * 4 independent 16 B loads in flight per lane), then accumulated in the
// reference's order.  Tap coordinates are clamped into the map even for samples the reference
// skips (tap1d), so the loads need no branch; the skip guards the accumulation only.
template <typename V, int GH, int GW, int NB>
__device__ __forceinline__ void fwd_bins(const V *__restrict__ img, V *__restrict__ out,
                                         const RoiGeom &g, int ph, int ow0, int OW, int BS, int H,
                                         int W, int CV, const FwdEpi<V> &epi)
{
    constexpr int S = GH * GW;
    V v[NB][S][4];
    float wt[NB][S][4];
    bool ok[NB][S];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (ow0 + b >= OW) break;
        const int pw = (ow0 + b) * BS;
#pragma unroll
        for (int iy = 0; iy < GH; ++iy) {
            const float yy = g.start_h + ph * g.bin_h + (float)(iy + .5f) * g.bin_h / (float)GH;
            const Tap1D ty = tap1d(yy, H);
            int plo = -1, phi = -1;
#pragma unroll
            for (int ix = 0; ix < GW; ++ix) {
                const float xx = g.start_w + pw * g.bin_w +
                                 (float)(ix + .5f) * g.bin_w / (float)GW;
                const Tap1D tx = tap1d(xx, W);
                const int s = iy * GW + ix;
                ok[b][s] = ty.valid && tx.valid;
                wt[b][s][0] = ty.h * tx.h;
                wt[b][s][1] = ty.h * tx.l;
                wt[b][s][2] = ty.l * tx.h;
                wt[b][s][3] = ty.l * tx.l;
                const int sp = ix > 0 ? s - 1 : s;      // the left neighbour's registers
                if (ix > 0 && tx.lo == plo && tx.hi == phi) {
                    v[b][s][0] = v[b][sp][0]; v[b][s][1] = v[b][sp][1];
                    v[b][s][2] = v[b][sp][2]; v[b][s][3] = v[b][sp][3];
                } else if (ix > 0 && tx.lo == phi) {
                    v[b][s][0] = v[b][sp][1];
                    v[b][s][2] = v[b][sp][3];
                    v[b][s][1] = img[((int64_t)ty.lo * W + tx.hi) * CV];
                    v[b][s][3] = img[((int64_t)ty.hi * W + tx.hi) * CV];
                } else {
                    v[b][s][0] = img[((int64_t)ty.lo * W + tx.lo) * CV];
                    v[b][s][1] = img[((int64_t)ty.lo * W + tx.hi) * CV];
                    v[b][s][2] = img[((int64_t)ty.hi * W + tx.lo) * CV];
                    v[b][s][3] = img[((int64_t)ty.hi * W + tx.hi) * CV];
                }
                plo = tx.lo;
                phi = tx.hi;
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        if (ow0 + b >= OW) break;
        V acc = VecOps<V>::zero();
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (ok[b][s])
                acc = VecOps<V>::mad4(acc, wt[b][s][0], v[b][s][0], wt[b][s][1], v[b][s][1],
                                      wt[b][s][2], v[b][s][2], wt[b][s][3], v[b][s][3]);
        store_stream(&out[(int64_t)(ow0 + b) * CV],
                     epi_apply(epi, finish_mean<V>(acc, g.grid_h * g.grid_w, g.count)));
    }
}

// One bin with any sampling grid: the 4 * GW taps of one sample row in flight together (GW = 0:
// run-time grid width, one sample at a time), sample rows in sequence — the reference's order.
template <typename V, int GW>
__device__ __forceinline__ void fwd_bin_rows(const V *__restrict__ img, V *__restrict__ out,
                                             const RoiGeom &g, int ph, int ow_, int BS, int H, int W,
                                             int CV, const FwdEpi<V> &epi)
{
    const int pw = ow_ * BS;
    V acc = VecOps<V>::zero();
    for (int iy = 0; iy < g.grid_h; ++iy) {
        const float yy = g.start_h + ph * g.bin_h + (float)(iy + .5f) * g.bin_h / (float)g.grid_h;
        const Tap1D ty = tap1d(yy, H);
        if constexpr (GW > 0) {
            V v[GW][4];
            float wt[GW][4];
            bool ok[GW];
            int plo = -1, phi = -1;
#pragma unroll
            for (int ix = 0; ix < GW; ++ix) {
                const float xx = g.start_w + pw * g.bin_w +
                                 (float)(ix + .5f) * g.bin_w / (float)GW;
                const Tap1D tx = tap1d(xx, W);
                ok[ix] = ty.valid && tx.valid;
                wt[ix][0] = ty.h * tx.h;
                wt[ix][1] = ty.h * tx.l;
                wt[ix][2] = ty.l * tx.h;
                wt[ix][3] = ty.l * tx.l;
                const int sp = ix > 0 ? ix - 1 : ix;    // the left neighbour's registers
                if (ix > 0 && tx.lo == plo && tx.hi == phi) {
                    v[ix][0] = v[sp][0]; v[ix][1] = v[sp][1]; v[ix][2] = v[sp][2]; v[ix][3] = v[sp][3];
                } else if (ix > 0 && tx.lo == phi) {
                    v[ix][0] = v[sp][1];
                    v[ix][2] = v[sp][3];
                    v[ix][1] = img[((int64_t)ty.lo * W + tx.hi) * CV];
                    v[ix][3] = img[((int64_t)ty.hi * W + tx.hi) * CV];
                } else {
                    v[ix][0] = img[((int64_t)ty.lo * W + tx.lo) * CV];
                    v[ix][1] = img[((int64_t)ty.lo * W + tx.hi) * CV];
                    v[ix][2] = img[((int64_t)ty.hi * W + tx.lo) * CV];
                    v[ix][3] = img[((int64_t)ty.hi * W + tx.hi) * CV];
                }
                plo = tx.lo;
                phi = tx.hi;
            }
#pragma unroll
            for (int ix = 0; ix < GW; ++ix)
                if (ok[ix])
                    acc = VecOps<V>::mad4(acc, wt[ix][0], v[ix][0], wt[ix][1], v[ix][1], wt[ix][2],
                                          v[ix][2], wt[ix][3], v[ix][3]);
        } else {
            for (int ix = 0; ix < g.grid_w; ++ix) {
                const float xx = g.start_w + pw * g.bin_w +
                                 (float)(ix + .5f) * g.bin_w / (float)g.grid_w;
                const Tap1D tx = tap1d(xx, W);
                if (!(ty.valid && tx.valid)) continue;
                const float w1 = ty.h * tx.h, w2 = ty.h * tx.l;
                const float w3 = ty.l * tx.h, w4 = ty.l * tx.l;
                const V v1 = img[((int64_t)ty.lo * W + tx.lo) * CV];
                const V v2 = img[((int64_t)ty.lo * W + tx.hi) * CV];
                const V v3 = img[((int64_t)ty.hi * W + tx.lo) * CV];
                const V v4 = img[((int64_t)ty.hi * W + tx.hi) * CV];
                acc = VecOps<V>::mad4(acc, w1, v1, w2, v2, w3, v3, w4, v4);
            }
        }
    }
    store_stream(&out[(int64_t)ow_ * CV], epi_apply(epi, finish_mean<V>(acc, g.grid_h * g.grid_w, g.count)));
}

// V = float4 when C % 4 == 0 (CV = C/4 vectors per pixel), else float.
// One workgroup owns one output ROW of a RoI (OW bins): a workgroup per bin (the first version)
// lived for one RoI read + four dependent-latency loads + one store and was bound by that
// turnaround (0.24 of the HBM peak with 4 taps per bin); here the taps of up to four bins are in
// flight together.  The sampling grid is a property of the RoI, so the choice among the
// unrolled bodies is workgroup-uniform.
template <typename V, bool EPI = false>
__global__ void __launch_bounds__(256)
roi_align_fwd_kernel(const V *__restrict__ x, const float *__restrict__ rois, V *__restrict__ y,
                     int H, int W, int CV, int PH, int PW, float spatial_scale, int sampling_ratio,
                     int OH, int OW, int BS, int rows, const int *__restrict__ order,
                     const V *__restrict__ scale = nullptr, const V *__restrict__ shift = nullptr,
                     int relu = 0)
{
    // output bin (oh, ow) is bin (oh*BS, ow*BS) of the PH x PW grid (BS = 1: every bin)
    // an XCD (workgroup id mod 8) owns a contiguous run of rows (n*OH + oh): the OH rows of a
    // RoI, which re-read each other's taps, share one L2
    const int per = ((int)gridDim.x + 7) / 8;
    const int row = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
    if (row >= rows) return;
    // `order` (optional): the RoI processed at position row / OH — a permutation that puts
    // neighbouring RoIs next to each other, so that an XCD's run of rows covers one region of
    // the map; results do not depend on it
    const int oh = row % OH;
    const int ph = oh * BS;
    const int n = order ? order[row / OH] : row / OH;
    const RoiGeom g = roi_geom(rois + 5 * n, spatial_scale, PH, PW, sampling_ratio);

    // channel chunks of blockDim.x lanes: blockIdx.y, the SLOW grid index — every output row of chunk
    // 0 is dispatched before chunk 1, so an XCD's run of rows touches its region of the map one
    // channel chunk (0.5 - 1 KB per pixel instead of 8 KB at 2048 channels) at a time
    for (int c = (int)(blockIdx.y * blockDim.x + threadIdx.x); c < CV; c += (int)(gridDim.y * blockDim.x)) {
        const V *__restrict__ img = x + (int64_t)g.batch * H * W * CV + c;
        V *__restrict__ out = y + ((int64_t)n * OH + oh) * OW * CV + c;
        FwdEpi<V> epi;
        epi.on = EPI;
        epi.relu = EPI && relu != 0;
        if constexpr (EPI) { epi.scale = scale[c]; epi.shift = shift[c]; }
        if (g.grid_h == 1 && g.grid_w == 1) {
            for (int ow0 = 0; ow0 < OW; ow0 += 4) fwd_bins<V, 1, 1, 4>(img, out, g, ph, ow0, OW, BS, H, W, CV, epi);
        } else if (g.grid_h == 1 && g.grid_w == 2) {
            for (int ow0 = 0; ow0 < OW; ow0 += 2) fwd_bins<V, 1, 2, 2>(img, out, g, ph, ow0, OW, BS, H, W, CV, epi);
        } else if (g.grid_h == 2 && g.grid_w == 1) {
            for (int ow0 = 0; ow0 < OW; ow0 += 2) fwd_bins<V, 2, 1, 2>(img, out, g, ph, ow0, OW, BS, H, W, CV, epi);
        } else if (g.grid_h == 2 && g.grid_w == 2) {
            for (int ow0 = 0; ow0 < OW; ++ow0) fwd_bins<V, 2, 2, 1>(img, out, g, ph, ow0, OW, BS, H, W, CV, epi);
        } else if (g.grid_w == 1) {
            for (int ow_ = 0; ow_ < OW; ++ow_) fwd_bin_rows<V, 1>(img, out, g, ph, ow_, BS, H, W, CV, epi);
        } else if (g.grid_w == 2) {
            for (int ow_ = 0; ow_ < OW; ++ow_) fwd_bin_rows<V, 2>(img, out, g, ph, ow_, BS, H, W, CV, epi);
        } else if (g.grid_w == 3) {
            for (int ow_ = 0; ow_ < OW; ++ow_) fwd_bin_rows<V, 3>(img, out, g, ph, ow_, BS, H, W, CV, epi);
        } else if (g.grid_w == 4) {
            for (int ow_ = 0; ow_ < OW; ++ow_) fwd_bin_rows<V, 4>(img, out, g, ph, ow_, BS, H, W, CV, epi);
        } else {
            for (int ow_ = 0; ow_ < OW; ++ow_) fwd_bin_rows<V, 0>(img, out, g, ph, ow_, BS, H, W, CV, epi);
        }
    }
}

__device__ __forceinline__ void atomic_add_vec(float *p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_vec(float4 *p, float4 v)
{
    float *q = reinterpret_cast<float *>(p);
    unsafeAtomicAdd(q + 0, v.x);
    unsafeAtomicAdd(q + 1, v.y);
    unsafeAtomicAdd(q + 2, v.z);
    unsafeAtomicAdd(q + 3, v.w);
}
__device__ __forceinline__ float vfma(float acc, float w, float v) { return acc + w * v; }
__device__ __forceinline__ float4 vfma(float4 acc, float w, float4 v)
{
    return make_float4(acc.x + w * v.x, acc.y + w * v.y, acc.z + w * v.z, acc.w + w * v.w);
}
__device__ __forceinline__ float scale_div(float d, float w, float c) { return d * w / c; }
__device__ __forceinline__ float4 scale_div(float4 d, float w, float c)
{
    return make_float4(d.x * w / c, d.y * w / c, d.z * w / c, d.w * w / c);
}

// roi_align_2d.py:395-522: g_k = top_diff * w_k / count, scatter-add.
template <typename V>
__global__ void roi_align_bwd_kernel(const V *__restrict__ gy, const float *__restrict__ rois,
                                     V *__restrict__ gx, int H, int W, int CV, int PH, int PW,
                                     float spatial_scale, int sampling_ratio, int OH, int OW,
                                     int BS)
{
    const int bin = blockIdx.x;
    const int pw = (bin % OW) * BS;
    const int ph = ((bin / OW) % OH) * BS;
    const int n = bin / (OW * OH);
    const RoiGeom g = roi_geom(rois + 5 * n, spatial_scale, PH, PW, sampling_ratio);
    V *__restrict__ img = gx + (int64_t)g.batch * H * W * CV;
    const V *__restrict__ top = gy + (int64_t)bin * CV;

    for (int c = threadIdx.x; c < CV; c += blockDim.x) {
        const V d = top[c];
        for (int iy = 0; iy < g.grid_h; ++iy) {
            const float yy = g.start_h + ph * g.bin_h +
                             (float)(iy + .5f) * g.bin_h / (float)g.grid_h;
            const Tap1D ty = tap1d(yy, H);
            for (int ix = 0; ix < g.grid_w; ++ix) {
                const float xx = g.start_w + pw * g.bin_w +
                                 (float)(ix + .5f) * g.bin_w / (float)g.grid_w;
                const Tap1D tx = tap1d(xx, W);
                if (!(ty.valid && tx.valid)) continue;
                const float w1 = ty.h * tx.h, w2 = ty.h * tx.l;
                const float w3 = ty.l * tx.h, w4 = ty.l * tx.l;
                atomic_add_vec(&img[((int64_t)ty.lo * W + tx.lo) * CV + c], scale_div(d, w1, g.count));
                atomic_add_vec(&img[((int64_t)ty.lo * W + tx.hi) * CV + c], scale_div(d, w2, g.count));
                atomic_add_vec(&img[((int64_t)ty.hi * W + tx.lo) * CV + c], scale_div(d, w3, g.count));
                atomic_add_vec(&img[((int64_t)ty.hi * W + tx.hi) * CV + c], scale_div(d, w4, g.count));
            }
        }
    }
}

// ---- gather-form backward ------------------------------------------------------------
// The scatter form above issues 4*grid_h*grid_w float atomics per output element
// (~2e9 at the C2 shape: atomic-rate bound, 13 ms).  Bilinear weights are separable,
//   w(sample -> pixel (y,x)) = wy(y; sample row) * wx(x; sample col),
// so the gradient a RoI sends to pixel (y,x) is
//   sum_ph sum_pw  Ay[y][ph] * Bx[x][pw] * gy[roi, ph, pw, :] / count,
// with Ay[y][ph] = sum over the bin's valid sample rows of their weight on row y (and Bx
// likewise).  One workgroup owns one (roi, feature row y): it builds Ay (PH floats) and
// Bx (patch width x PW floats) in LDS, then every lane (a float4 of channels) gathers the
// <= 3x3 contributing bins per pixel with coalesced reads and issues ONE atomic add per
// (pixel, channel) — the RoI's patch area instead of 4x its sample count.
template <typename V>
__global__ void __launch_bounds__(256)
roi_align_bwd_gather_kernel(const V *__restrict__ gy, const float *__restrict__ rois,
                            V *__restrict__ gx, int H, int W, int CV, int PH, int PW,
                            float spatial_scale, int sampling_ratio, int OH, int OW, int BS)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n = blockIdx.y;
    const RoiGeom g = roi_geom(rois + 5 * n, spatial_scale, PH, PW, sampling_ratio);

    // rows/cols any sample of this RoI can touch
    const float y_first = g.start_h + .5f * g.bin_h / (float)g.grid_h;
    const float y_last = g.start_h + (PH - 1) * g.bin_h + (g.grid_h - .5f) * g.bin_h / (float)g.grid_h;
    const float x_first = g.start_w + .5f * g.bin_w / (float)g.grid_w;
    const float x_last = g.start_w + (PW - 1) * g.bin_w + (g.grid_w - .5f) * g.bin_w / (float)g.grid_w;
    const int ylo = max(0, min(H - 1, (int)floorf(fmaxf(y_first, 0.f)) - 1));
    const int yhi = max(0, min(H - 1, (int)floorf(fmaxf(y_last, 0.f)) + 2));
    const int xlo = max(0, min(W - 1, (int)floorf(fmaxf(x_first, 0.f)) - 1));
    const int xhi = max(0, min(W - 1, (int)floorf(fmaxf(x_last, 0.f)) + 2));
    const int y = ylo + blockIdx.x;
    if (y > yhi) return;
    const int PX = xhi - xlo + 1;

    float *Ay = lds;                       // [PH]
    float *Bx = lds + PH;                  // [PX][PW]
    int *pwlo = reinterpret_cast<int *>(Bx + (W + 4) * PW);  // [PX]
    int *pwhi = pwlo + (W + 4);
    for (int oh = threadIdx.x; oh < OH; oh += blockDim.x) {
        const int ph = oh * BS;
        float a = 0.f;
        for (int iy = 0; iy < g.grid_h; ++iy) {
            const float yy = g.start_h + ph * g.bin_h + (float)(iy + .5f) * g.bin_h / (float)g.grid_h;
            const Tap1D t = tap1d(yy, H);
            if (!t.valid) continue;
            if (t.lo == y) a += t.h;
            if (t.hi == y) a += t.l;
        }
        Ay[oh] = a;
    }
    for (int e = threadIdx.x; e < PX * OW; e += blockDim.x) {
        const int xi = e / OW, ow_ = e - xi * OW;
        const int pw = ow_ * BS;
        const int x = xlo + xi;
        float b = 0.f;
        for (int ix = 0; ix < g.grid_w; ++ix) {
            const float xx = g.start_w + pw * g.bin_w + (float)(ix + .5f) * g.bin_w / (float)g.grid_w;
            const Tap1D t = tap1d(xx, W);
            if (!t.valid) continue;
            if (t.lo == x) b += t.h;
            if (t.hi == x) b += t.l;
        }
        Bx[e] = b;
    }
    __syncthreads();
    for (int xi = threadIdx.x; xi < PX; xi += blockDim.x) {
        int lo = OW, hi = -1;
        for (int pw = 0; pw < OW; ++pw)
            if (Bx[xi * OW + pw] != 0.f) { lo = min(lo, pw); hi = pw; }
        pwlo[xi] = lo;
        pwhi[xi] = hi;
    }
    __syncthreads();
    int phlo = OH, phhi = -1;
    for (int ph = 0; ph < OH; ++ph)
        if (Ay[ph] != 0.f) { phlo = min(phlo, ph); phhi = ph; }
    if (phhi < 0) return;

    const float inv_count = 1.f / g.count;
    const V *__restrict__ top = gy + (int64_t)n * OH * OW * CV;
    V *__restrict__ row = gx + (((int64_t)g.batch * H + y) * W + xlo) * CV;
    for (int c = threadIdx.x; c < CV; c += blockDim.x) {
        for (int xi = 0; xi < PX; ++xi) {
            const int l = pwlo[xi], h = pwhi[xi];
            if (h < 0) continue;
            V acc = VecOps<V>::zero();
            for (int ph = phlo; ph <= phhi; ++ph) {
                const float ay = Ay[ph] * inv_count;
                if (ay == 0.f) continue;
                for (int pw = l; pw <= h; ++pw)
                    acc = vfma(acc, ay * Bx[xi * OW + pw], top[((int64_t)ph * OW + pw) * CV + c]);
            }
            atomic_add_vec(&row[(int64_t)xi * CV + c], acc);
        }
    }
}

// ---- pixel-owner backward (no atomics, deterministic) -------------------------------------
// The gather form above still issues one atomic per (RoI patch pixel, channel): ~1.7e8 float
// atomics at the C2 shape, which bound it at 1.35 ms.  Here every feature-map pixel is OWNED
// by one workgroup, which sums the contributions of all RoIs that cover it in registers and
// stores the result once (no atomics, no zero-fill, summation order = RoI order, then bin
// order: bit-reproducible run to run):
//   roi_bwd_tables_kernel  (grid R)    per RoI: its patch extent and the separable weights
//                                      Ay[y][oh] (already / count) and Bx[ow][x] of every
//                                      feature row / column, written once to the workspace
//   roi_align_bwd_owner    (grid = 8-pixel row tiles, XCD-contiguous)
//                                      one workgroup owns XT = 8 pixels of one row x all
//                                      channels.  It walks the RoIs in index order, a
//                                      workgroup-wide chunk at a time:
//                                       A  ordered compaction of the chunk's RoIs whose patch
//                                          meets the tile;
//                                       B  their (RoI, bin) candidates, two per thread, read the
//                                          tables; the bins with a non-zero weight on the tile
//                                          are compacted IN ORDER into an LDS list
//                                          (gy offset, 8 weights ay*bx);
//                                       C  every lane (one float4 of channels) streams the
//                                          list four entries at a time — four independent 16 B
//                                          loads in flight per lane, weights broadcast from LDS,
//                                          packed fp32 FMAs.
// The first version of this form rebuilt the weights of four RoIs per barrier pair inside
// every tile (tap arithmetic with divisions, for all RoIs of the ROW) and issued one
// dependent gy load per bin: latency-bound at 0.12 of the HBM peak.
struct RoiExtent { int ylo, yhi, xlo, xhi; };

__device__ __forceinline__ RoiExtent roi_extent(const RoiGeom &g, int H, int W, int PH, int PW)
{
    const float y_first = g.start_h + .5f * g.bin_h / (float)g.grid_h;
    const float y_last = g.start_h + (PH - 1) * g.bin_h + (g.grid_h - .5f) * g.bin_h / (float)g.grid_h;
    const float x_first = g.start_w + .5f * g.bin_w / (float)g.grid_w;
    const float x_last = g.start_w + (PW - 1) * g.bin_w + (g.grid_w - .5f) * g.bin_w / (float)g.grid_w;
    RoiExtent e;
    e.ylo = max(0, min(H - 1, (int)floorf(fmaxf(y_first, 0.f)) - 1));
    e.yhi = max(0, min(H - 1, (int)floorf(fmaxf(y_last, 0.f)) + 2));
    e.xlo = max(0, min(W - 1, (int)floorf(fmaxf(x_first, 0.f)) - 1));
    e.xhi = max(0, min(W - 1, (int)floorf(fmaxf(x_last, 0.f)) + 2));
    return e;
}

constexpr int kOwnXT = 8;        // pixels of a row per workgroup
constexpr int kOwnThreads = 256; // upper bound of the workgroup size (LDS list capacities)
constexpr int kOwnScan = 4;      // RoIs tested per thread and chunk
#ifndef MRCNN_ROI_DEPTH
#define MRCNN_ROI_DEPTH 4    // measured: 4 at five waves per SIMD = 8 at four; 16 spills
#endif
constexpr int kOwnDepth = MRCNN_ROI_DEPTH;     // gy loads in flight per lane

// ext[r] = (ylo | yhi << 16, xlo | xhi << 16, batch, 0);  Ay[r][y][oh];  Bx[r][ow][x] (row
// stride Wp, a multiple of 8, zero beyond the patch)
__global__ void __launch_bounds__(256)
roi_bwd_tables_kernel(const float *__restrict__ rois, int H, int W, int Wp, int PH, int PW, int OH,
                      int OW, int BS, float spatial_scale, int sampling_ratio,
                      int4 *__restrict__ ext, float *__restrict__ Ay, float *__restrict__ Bx)
{
    const int r = blockIdx.x;
    const RoiGeom g = roi_geom(rois + 5 * r, spatial_scale, PH, PW, sampling_ratio);
    const RoiExtent e = roi_extent(g, H, W, PH, PW);
    if (threadIdx.x == 0)
        ext[r] = make_int4((int)((unsigned)e.ylo | ((unsigned)e.yhi << 16)),
                           (int)((unsigned)e.xlo | ((unsigned)e.xhi << 16)), g.batch, 0);
    float *__restrict__ ay = Ay + (int64_t)r * H * OH;
    for (int i = threadIdx.x; i < H * OH; i += blockDim.x) {
        const int y = i / OH, oh = i - y * OH;
        float a = 0.f;
        if (y >= e.ylo && y <= e.yhi) {
            const int ph = oh * BS;
            for (int iy = 0; iy < g.grid_h; ++iy) {
                const float yy = g.start_h + ph * g.bin_h +
                                 (float)(iy + .5f) * g.bin_h / (float)g.grid_h;
                const Tap1D t = tap1d(yy, H);
                if (!t.valid) continue;
                if (t.lo == y) a += t.h;
                if (t.hi == y) a += t.l;
            }
            a = a / g.count;
        }
        ay[i] = a;
    }
    float *__restrict__ bx = Bx + (int64_t)r * OW * Wp;
    for (int i = threadIdx.x; i < OW * Wp; i += blockDim.x) {
        const int ow_ = i / Wp, x = i - ow_ * Wp;
        float b = 0.f;
        if (x >= e.xlo && x <= e.xhi) {
            const int pw = ow_ * BS;
            for (int ix = 0; ix < g.grid_w; ++ix) {
                const float xx = g.start_w + pw * g.bin_w +
                                 (float)(ix + .5f) * g.bin_w / (float)g.grid_w;
                const Tap1D t = tap1d(xx, W);
                if (!t.valid) continue;
                if (t.lo == x) b += t.h;
                if (t.hi == x) b += t.l;
            }
        }
        bx[i] = b;
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float fma_vec(float acc, float w, float v) { return __builtin_fmaf(w, v, acc); }
__device__ __forceinline__ float4 fma_vec(float4 acc, float w, float4 v)
{
#if defined(MRCNN_ROI_BWD_SCALAR_FMA) && defined(MRCNN_EXPERIMENT_BUILD)   // experiment build (tools/build_variant.sh): four v_fma_f32
    float4 r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r.x) : "v"(v.x), "v"(w), "v"(acc.x));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r.y) : "v"(v.y), "v"(w), "v"(acc.y));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r.z) : "v"(v.z), "v"(w), "v"(acc.z));
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r.w) : "v"(v.w), "v"(w), "v"(acc.w));
    return r;
#endif
    // two v_pk_fma_f32
    const f32x2 ww = {w, w};
    const f32x2 lo = __builtin_elementwise_fma((f32x2){v.x, v.y}, ww, (f32x2){acc.x, acc.y});
    const f32x2 hi = __builtin_elementwise_fma((f32x2){v.z, v.w}, ww, (f32x2){acc.z, acc.w});
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}

// acc[x] += w[x] * v for the tile's 8 pixels.  Pixels the bin does not touch carry w = 0 and take
// the FMA as well (adds 0 for finite gy): per-pixel tests cost four selects per FMA pair here
// and made this loop, not the loads, the bound of the kernel.
template <typename V>
__device__ __forceinline__ void owner_consume(V (&acc)[kOwnXT], const V v, const float *w)
{
    const float4 w0 = *reinterpret_cast<const float4 *>(w);
    const float4 w1 = *reinterpret_cast<const float4 *>(w + 4);
    const float ww[kOwnXT] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
    for (int x = 0; x < kOwnXT; ++x) acc[x] = fma_vec(acc[x], ww[x], v);
}

#ifndef MRCNN_ROI_WAVES
#define MRCNN_ROI_WAVES 5
#endif
template <typename V>
__global__ void __launch_bounds__(kOwnThreads) __attribute__((amdgpu_waves_per_eu(MRCNN_ROI_WAVES, 8)))
roi_align_bwd_owner_kernel(const V *__restrict__ gy, const int4 *__restrict__ ext,
                           const float *__restrict__ Ay, const float *__restrict__ Bx,
                           V *__restrict__ gx, int R, int N, int H, int W, int Wp, int CV, int OH,
                           int OW)
{
    constexpr int XT = kOwnXT, CAP = 2 * kOwnThreads;
    __shared__ int sList[kOwnScan * kOwnThreads];
    __shared__ int sWave[kOwnThreads / 64];
    __shared__ int64_t sOff[CAP];        // gy element offset of the bin's first channel vector
    __shared__ __attribute__((aligned(16))) float sW[CAP][XT];

    // tile id: an XCD (workgroup id mod 8) owns a contiguous run of (image, row, tile) ids, so
    // the rows that re-read a RoI's gy bins share one L2
    const int tiles_x = (W + XT - 1) / XT;
    const int total = tiles_x * H * N;
    const int per = (total + 7) / 8;
    const int logical = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
    if (logical >= total) return;
    const int x0 = (logical % tiles_x) * XT;
    const int y = (logical / tiles_x) % H;
    const int n = logical / (tiles_x * H);

    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = nthr >> 6;
    const int nb = OH * OW;
    const int cap = 2 * nthr;

    // One channel chunk of nthr lanes per workgroup (blockIdx.y; grid-stride form): the 2048-channel
    // gradients of the projected head would otherwise build the tile's lists once per chunk in
    // sequence.  KEEP THIS KERNEL FREE OF SCRATCH (tests/test_build_cpu.py): written as a plain
    // block instead of this loop the compiler spilled three registers at the 96-register budget, and
    // with the weight-gradient GEMMs of the side stream running beside it single dwords of single
    // lanes came back wrong from the spill slots (a handful of gx elements per launch, run to run) —
    // private-segment memory of two queues' concurrent kernels is not something to rely on here.
    for (int c0 = (int)blockIdx.y * nthr; c0 < CV; c0 += (int)gridDim.y * nthr) {
        const int c = c0 + tid;
        const bool cok = c < CV;
        const V *__restrict__ top = gy + (cok ? c : 0);
        V acc[XT];
#pragma unroll
        for (int x = 0; x < XT; ++x) acc[x] = VecOps<V>::zero();

        for (int r0 = 0; r0 < R; r0 += kOwnScan * nthr) {
            // A: the chunk's RoIs whose patch meets this tile, in index order (kOwnScan
            // consecutive RoIs per thread)
            bool in[kOwnScan];
            int lcnt = 0;
#pragma unroll
            for (int j = 0; j < kOwnScan; ++j) {
                const int r = r0 + tid * kOwnScan + j;
                in[j] = false;
                if (r < R) {
                    const int4 e = ext[r];
                    const int ylo = e.x & 0xffff, yhi = (int)((unsigned)e.x >> 16);
                    const int xlo = e.y & 0xffff, xhi = (int)((unsigned)e.y >> 16);
                    in[j] = e.z == n && y >= ylo && y <= yhi && xhi >= x0 && xlo <= x0 + XT - 1;
                }
                lcnt += in[j] ? 1 : 0;
            }
            int linc = lcnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int t = __shfl_up(linc, d);
                if (lane >= d) linc += t;
            }
            __syncthreads();          // the previous chunk's list and wave counts are consumed
            if (lane == 63) sWave[wave] = linc;
            __syncthreads();
            int off = 0, m = 0;
            for (int w = 0; w < nwaves; ++w) {
                const int cw = sWave[w];
                if (w < wave) off += cw;
                m += cw;
            }
            off += linc - lcnt;
#pragma unroll
            for (int j = 0; j < kOwnScan; ++j)
                if (in[j]) sList[off++] = r0 + tid * kOwnScan + j;
            __syncthreads();

            for (int w0 = 0; w0 < m * nb; w0 += cap) {
                // B: two (RoI, bin) candidates per thread, kept in candidate order
                float wv[2][XT];
                int64_t of[2];
                int mk[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ci = w0 + tid * 2 + j;
                    mk[j] = 0;
                    of[j] = 0;
#pragma unroll
                    for (int x = 0; x < XT; ++x) wv[j][x] = 0.f;
                    if (ci < m * nb) {
                        const int gi = ci / nb, b = ci - gi * nb;
                        const int oh = b / OW, ow_ = b - oh * OW;
                        const int rr = sList[gi];
                        const float ay = Ay[((int64_t)rr * H + y) * OH + oh];
                        const float4 *bp = reinterpret_cast<const float4 *>(
                            Bx + ((int64_t)rr * OW + ow_) * Wp + x0);
                        const float4 b0 = bp[0], b1 = bp[1];     // independent of ay: in flight together
                        if (ay != 0.f) {
                            const float bb[XT] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                            for (int x = 0; x < XT; ++x) {
                                wv[j][x] = ay * bb[x];
                                if (bb[x] != 0.f) mk[j] |= 1 << x;
                            }
                            of[j] = ((int64_t)rr * nb + b) * CV;
                        }
                    }
                }
                const int lc = (mk[0] != 0) + (mk[1] != 0);
                int inc = lc;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int t = __shfl_up(inc, d);
                    if (lane >= d) inc += t;
                }
                __syncthreads();      // the previous window's list has been consumed
                if (lane == 63) sWave[wave] = inc;
                __syncthreads();
                int base = 0, nent = 0;
                for (int w = 0; w < nwaves; ++w) {
                    const int cw = sWave[w];
                    if (w < wave) base += cw;
                    nent += cw;
                }
                int pos = base + inc - lc;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (mk[j]) {
                        sOff[pos] = of[j];
                        *reinterpret_cast<float4 *>(&sW[pos][0]) =
                            make_float4(wv[j][0], wv[j][1], wv[j][2], wv[j][3]);
                        *reinterpret_cast<float4 *>(&sW[pos][4]) =
                            make_float4(wv[j][4], wv[j][5], wv[j][6], wv[j][7]);
                        ++pos;
                    }
                }
                __syncthreads();

                // C: stream the list, kOwnDepth independent loads in flight per lane.  The
                // compiler fences keep the weight reads of an entry next to its FMAs (hoisted above
                // the loads they would hold 8 registers per entry in flight and halve the occupancy)
                if (cok) {
                    for (int e = 0; e < nent; e += kOwnDepth) {
                        V v[kOwnDepth];
#pragma unroll
                        for (int i = 0; i < kOwnDepth; ++i)
                            if (e + i < nent) v[i] = top[sOff[e + i]];
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int i = 0; i < kOwnDepth; ++i) {
                            if (e + i < nent) owner_consume(acc, v[i], sW[e + i]);
                            asm volatile("" ::: "memory");
                        }
                    }
                }
            }
        }
        if (cok) {
            V *__restrict__ row = gx + (((int64_t)n * H + y) * W + x0) * CV + c;
#pragma unroll
            for (int x = 0; x < XT; ++x)
                if (x0 + x < W) row[(int64_t)x * CV] = acc[x];
        }
    }
}

int g_roi_fwd_lanes = 0;     // mrcnn_set_tuning("roi_fwd_lanes"): cap of the forward's lanes per workgroup (0 = 256)
int g_roi_bwd_lanes = 0;     // mrcnn_set_tuning("roi_bwd_lanes"): same for the pixel-owner backward

inline int pick_threads(int cv)
{
    int t = ((cv + 63) / 64) * 64;
    return t > 256 ? 256 : (t < 64 ? 64 : t);
}

int check_args(const void *a, const void *b, const void *c, int N, int H, int W, int C, int R,
               int PH, int PW, int sampling_ratio)
{
    MRCNN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && R >= 0 && PH > 0 && PW > 0,
                  "roi_align: bad shape N=%d H=%d W=%d C=%d R=%d PH=%d PW=%d", N, H, W, C, R, PH,
                  PW);
    MRCNN_REQUIRE(sampling_ratio >= 0, "roi_align: sampling_ratio must be >= 0");
    MRCNN_REQUIRE(R == 0 || (a && b && c), "roi_align: null pointer");
    MRCNN_REQUIRE((int64_t)R * PH * PW < (int64_t)INT32_MAX, "roi_align: too many bins");
    return 0;
}

}  // namespace

namespace {
int roi_align_fwd_launch(const float *x, const float *rois, float *y, int N, int H, int W, int C, int R,
                         int PH, int PW, int bin_stride, float spatial_scale, int sampling_ratio,
                         const int *order, const float *scale, const float *shift, int relu,
                         void *stream)
{
    if (int rc = check_args(x, rois, y, N, H, W, C, R, PH, PW, sampling_ratio)) return rc;
    MRCNN_REQUIRE(bin_stride >= 1, "roi_align: bin_stride must be >= 1");
    const bool epi = scale != nullptr || shift != nullptr;
    MRCNN_REQUIRE(!epi || (scale && shift), "roi_align_fwd_affine: scale and shift go together");
    if (R == 0) return 0;
    const int OH = (PH + bin_stride - 1) / bin_stride, OW = (PW + bin_stride - 1) / bin_stride;
    const int bins = R * OH * OW;
    hipStream_t s = mrcnn::as_stream(stream);
    // algorithmic bytes: write R*OH*OW*C, read the feature maps once
    hipEvent_t ev0, ev1;      // kernel-only timing from the dispatch packet (bench.py roofline)
    mrcnn::prof_begin_ext(mrcnn::PROF_ROI_ALIGN_FWD, 0.,
                          4.0 * ((double)bins * C + (double)N * H * W * C), &ev0, &ev1);
    const int cvl = (C % 4 == 0) ? C / 4 : C;
    int nthr = pick_threads(cvl);
    if (g_roi_fwd_lanes > 0) nthr = std::min(nthr, g_roi_fwd_lanes);
    const dim3 grid((R * OH + 7) / 8 * 8, (cvl + nthr - 1) / nthr);
    const bool vec = C % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0) &&
                     (!epi || (((uintptr_t)scale % 16 == 0) && ((uintptr_t)shift % 16 == 0)));
    if (vec) {
        const int cv = C / 4;
        if (epi)
            hipExtLaunchKernelGGL((roi_align_fwd_kernel<float4, true>), grid, dim3(nthr), 0, s,
                                  ev0, ev1, 0, (const float4 *)x, rois, (float4 *)y, H, W, cv, PH, PW,
                                  spatial_scale, sampling_ratio, OH, OW, bin_stride, R * OH, order,
                                  (const float4 *)scale, (const float4 *)shift, relu);
        else
            hipExtLaunchKernelGGL((roi_align_fwd_kernel<float4, false>), grid, dim3(nthr), 0, s,
                                  ev0, ev1, 0, (const float4 *)x, rois, (float4 *)y, H, W, cv, PH, PW,
                                  spatial_scale, sampling_ratio, OH, OW, bin_stride, R * OH, order,
                                  (const float4 *)nullptr, (const float4 *)nullptr, 0);
    } else if (epi) {
        hipExtLaunchKernelGGL((roi_align_fwd_kernel<float, true>), grid, dim3(nthr), 0, s, ev0,
                              ev1, 0, x, rois, y, H, W, C, PH, PW, spatial_scale, sampling_ratio, OH, OW,
                              bin_stride, R * OH, order, scale, shift, relu);
    } else {
        hipExtLaunchKernelGGL((roi_align_fwd_kernel<float, false>), grid, dim3(nthr), 0, s, ev0,
                              ev1, 0, x, rois, y, H, W, C, PH, PW, spatial_scale, sampling_ratio, OH, OW,
                              bin_stride, R * OH, order, (const float *)nullptr, (const float *)nullptr, 0);
    }
    return mrcnn::check_launch("roi_align_fwd");
}
}  // namespace

int mrcnn::roi_align_set_tuning(const char *name, int value)     // behind mrcnn_set_tuning (conv_gemm.hip)
{
    if (strcmp(name, "roi_fwd_lanes") == 0) { g_roi_fwd_lanes = value; return 0; }
    if (strcmp(name, "roi_bwd_lanes") == 0) { g_roi_bwd_lanes = value; return 0; }
    return 1;
}

extern "C" int mrcnn_roi_align_fwd_ex(const float *x, const float *rois, float *y, int N, int H,
                                      int W, int C, int R, int PH, int PW, int bin_stride,
                                      float spatial_scale, int sampling_ratio, const int *order,
                                      void *stream)
{
    return roi_align_fwd_launch(x, rois, y, N, H, W, C, R, PH, PW, bin_stride, spatial_scale,
                                sampling_ratio, order, nullptr, nullptr, 0, stream);
}

extern "C" int mrcnn_roi_align_fwd_affine(const float *x, const float *rois, float *y, int N, int H,
                                          int W, int C, int R, int PH, int PW, int bin_stride,
                                          float spatial_scale, int sampling_ratio, const int *order,
                                          const float *scale, const float *shift, int relu,
                                          void *stream)
{
    MRCNN_REQUIRE(scale && shift, "roi_align_fwd_affine: null scale / shift");
    return roi_align_fwd_launch(x, rois, y, N, H, W, C, R, PH, PW, bin_stride, spatial_scale,
                                sampling_ratio, order, scale, shift, relu, stream);
}

extern "C" int mrcnn_roi_align_fwd(const float *x, const float *rois, float *y, int N, int H,
                                   int W, int C, int R, int PH, int PW, float spatial_scale,
                                   int sampling_ratio, void *stream)
{
    return mrcnn_roi_align_fwd_ex(x, rois, y, N, H, W, C, R, PH, PW, 1, spatial_scale,
                                  sampling_ratio, nullptr, stream);
}

namespace {
// workspace sections of the pixel-owner backward, 256-byte aligned
struct OwnerWs {
    int64_t ext, ay, bx, total;
    int Wp;
};
inline OwnerWs owner_ws(int H, int W, int R, int OH, int OW)
{
    OwnerWs w;
    auto up = [](int64_t v) { return (v + 255) / 256 * 256; };
    w.Wp = (W + kOwnXT - 1) / kOwnXT * kOwnXT;
    w.ext = 0;
    w.ay = up((int64_t)R * 16);
    w.bx = w.ay + up((int64_t)R * H * OH * 4);
    w.total = w.bx + up((int64_t)R * OW * w.Wp * 4);
    return w;
}
}  // namespace

extern "C" int64_t mrcnn_roi_align_bwd_workspace_bytes(int N, int H, int W, int R, int PH, int PW,
                                                       int bin_stride)
{
    if (N <= 0 || H <= 0 || W <= 0 || R <= 0 || PH <= 0 || PW <= 0 || bin_stride < 1) return 0;
    const int OH = (PH + bin_stride - 1) / bin_stride, OW = (PW + bin_stride - 1) / bin_stride;
    return owner_ws(H, W, R, OH, OW).total;
}

extern "C" int mrcnn_roi_align_bwd_ex(const float *gy, const float *rois, float *gx, int N, int H,
                                      int W, int C, int R, int PH, int PW, int bin_stride,
                                      float spatial_scale, int sampling_ratio, void *ws,
                                      void *stream)
{
    // (no extent given: the caller vouches for mrcnn_roi_align_bwd_workspace_bytes of THIS header)
    return mrcnn_roi_align_bwd_ws(gy, rois, gx, N, H, W, C, R, PH, PW, bin_stride, spatial_scale,
                                  sampling_ratio, ws, ws ? INT64_MAX : 0, stream);
}

extern "C" int mrcnn_roi_align_bwd_ws(const float *gy, const float *rois, float *gx, int N, int H,
                                      int W, int C, int R, int PH, int PW, int bin_stride,
                                      float spatial_scale, int sampling_ratio, void *ws,
                                      int64_t ws_bytes, void *stream)
{
    if (int rc = check_args(gy, rois, gx, N, H, W, C, R, PH, PW, sampling_ratio)) return rc;
    MRCNN_REQUIRE(bin_stride >= 1, "roi_align: bin_stride must be >= 1");
    MRCNN_REQUIRE(ws == nullptr || ws_bytes >= mrcnn_roi_align_bwd_workspace_bytes(N, H, W, R, PH, PW, bin_stride),
                  "roi_align_bwd: workspace smaller than mrcnn_roi_align_bwd_workspace_bytes(N, H, W, R, PH, PW, bin_stride)");
    hipStream_t s = mrcnn::as_stream(stream);
    MRCNN_REQUIRE(gx != nullptr, "roi_align_bwd: null gx");
    const int OH = (PH + bin_stride - 1) / bin_stride, OW = (PW + bin_stride - 1) / bin_stride;
    const int bins = R * OH * OW;
    const bool vec = C % 4 == 0 && ((uintptr_t)gx % 16 == 0) && ((uintptr_t)gy % 16 == 0);
    const int64_t tiles = (int64_t)((W + kOwnXT - 1) / kOwnXT) * H * N;
    if (ws && R > 0 && H <= 65535 && W <= 65535 && tiles + 8 < (int64_t)INT32_MAX &&
        (uintptr_t)ws % 16 == 0) {
        // pixel-owner form: every gx element is written exactly once, no zero-fill.
        // algorithmic bytes: read R*OH*OW*C, write the feature-map gradient
        hipEvent_t ev0, ev1;  // start of the table kernel .. end of the owner kernel, dispatch timestamps
        mrcnn::prof_begin_ext(mrcnn::PROF_ROI_ALIGN_BWD, 0.,
                              4.0 * ((double)bins * C + (double)N * H * W * C), &ev0, &ev1);
        const OwnerWs w = owner_ws(H, W, R, OH, OW);
        int4 *ext = (int4 *)((char *)ws + w.ext);
        float *Ay = (float *)((char *)ws + w.ay);
        float *Bx = (float *)((char *)ws + w.bx);
        hipExtLaunchKernelGGL(roi_bwd_tables_kernel, dim3(R), dim3(256), 0, s, ev0, nullptr, 0, rois, H,
                              W, w.Wp, PH, PW, OH, OW, bin_stride, spatial_scale, sampling_ratio, ext,
                              Ay, Bx);
        int nthr = pick_threads(vec ? C / 4 : C);
        if (g_roi_bwd_lanes > 0) nthr = std::min(nthr, g_roi_bwd_lanes);
        const dim3 grid((unsigned)((tiles + 7) / 8 * 8), (unsigned)(((vec ? C / 4 : C) + nthr - 1) / nthr));
        if (vec)
            hipExtLaunchKernelGGL(roi_align_bwd_owner_kernel<float4>, grid, dim3(nthr),
                                  0, s, nullptr, ev1, 0, (const float4 *)gy, ext, Ay, Bx, (float4 *)gx,
                                  R, N, H, W, w.Wp, C / 4, OH, OW);
        else
            hipExtLaunchKernelGGL(roi_align_bwd_owner_kernel<float>, grid, dim3(nthr), 0, s,
                                  nullptr, ev1, 0, gy, ext, Ay, Bx, gx, R, N, H, W, w.Wp, C, OH, OW);
        return mrcnn::check_launch("roi_align_bwd");
    }
    MRCNN_HIP_TRY(hipMemsetAsync(gx, 0, sizeof(float) * (size_t)N * H * W * C, s));
    if (R == 0) return 0;
    // algorithmic bytes: read R*OH*OW*C, read-modify-write the feature-map gradient
    mrcnn::ProfScope prof(mrcnn::PROF_ROI_ALIGN_BWD, 0.,
                          4.0 * ((double)bins * C + 2.0 * (double)N * H * W * C), s);
    const size_t lds = sizeof(float) * ((size_t)PH + (size_t)(W + 4) * PW) + sizeof(int) * 2 * (W + 4);
    if (lds <= 48 * 1024 && H <= 65535 && R <= 65535) {
        // gather form: one workgroup per (roi, feature row), one atomic per (pixel, channel)
        if (vec)
            hipLaunchKernelGGL(roi_align_bwd_gather_kernel<float4>, dim3(H, R),
                               dim3(pick_threads(C / 4)), lds, s, (const float4 *)gy, rois,
                               (float4 *)gx, H, W, C / 4, PH, PW, spatial_scale, sampling_ratio,
                               OH, OW, bin_stride);
        else
            hipLaunchKernelGGL(roi_align_bwd_gather_kernel<float>, dim3(H, R), dim3(pick_threads(C)),
                               lds, s, gy, rois, gx, H, W, C, PH, PW, spatial_scale, sampling_ratio,
                               OH, OW, bin_stride);
    } else if (vec) {
        hipLaunchKernelGGL(roi_align_bwd_kernel<float4>, dim3(bins), dim3(pick_threads(C / 4)), 0, s,
                           (const float4 *)gy, rois, (float4 *)gx, H, W, C / 4, PH, PW,
                           spatial_scale, sampling_ratio, OH, OW, bin_stride);
    } else {
        hipLaunchKernelGGL(roi_align_bwd_kernel<float>, dim3(bins), dim3(pick_threads(C)), 0, s, gy,
                           rois, gx, H, W, C, PH, PW, spatial_scale, sampling_ratio, OH, OW,
                           bin_stride);
    }
    return mrcnn::check_launch("roi_align_bwd");
}

extern "C" int mrcnn_roi_align_bwd(const float *gy, const float *rois, float *gx, int N, int H,
                                   int W, int C, int R, int PH, int PW, float spatial_scale,
                                   int sampling_ratio, void *stream)
{
    return mrcnn_roi_align_bwd_ex(gy, rois, gx, N, H, W, C, R, PH, PW, 1, spatial_scale,
                                  sampling_ratio, nullptr, stream);
}
