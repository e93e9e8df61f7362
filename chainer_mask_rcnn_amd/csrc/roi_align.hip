// ROIAlign forward / backward for gfx950, NHWC.
//
// Replaces the two CuPy kernel strings of the reference
// (/root/reference/chainer_mask_rcnn/functions/roi_align_2d.py:171-288 fwd,
// :395-522 bwd), which run one thread per NCHW output element with
// uncoalesced 4-tap gathers.  Here one workgroup owns one output bin
// (roi, ph, pw): all sample geometry is wave-uniform (SGPR) arithmetic derived
// from blockIdx, and the lanes run across channels, so every tap is a
// contiguous 16 B/lane read of the NHWC feature row and the result is one
// contiguous 16 B/lane store.  HBM-bound: the forward writes R*PH*PW*C*4 bytes.
//
// Built with -ffp-contract=off and the reference's operation order so that the
// forward is bit-identical to the fp32 CPU restatement (oracle/roi_align_ref.c).
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

template <typename V> struct VecOps;
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

// V = float4 when C % 4 == 0 (CV = C/4 vectors per pixel), else float.
template <typename V>
__global__ void roi_align_fwd_kernel(const V *__restrict__ x, const float *__restrict__ rois,
                                     V *__restrict__ y, int H, int W, int CV, int PH, int PW,
                                     float spatial_scale, int sampling_ratio, int OH, int OW,
                                     int BS)
{
    // output bin (oh, ow) is bin (oh*BS, ow*BS) of the PH x PW grid (BS = 1: every bin)
    const int bin = blockIdx.x;  // ((n*OH)+oh)*OW+ow
    const int pw = (bin % OW) * BS;
    const int ph = ((bin / OW) % OH) * BS;
    const int n = bin / (OW * OH);
    const RoiGeom g = roi_geom(rois + 5 * n, spatial_scale, PH, PW, sampling_ratio);
    const V *__restrict__ img = x + (int64_t)g.batch * H * W * CV;
    V *__restrict__ out = y + (int64_t)bin * CV;

    for (int c = threadIdx.x; c < CV; c += blockDim.x) {
        V acc = VecOps<V>::zero();
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
                const V v1 = img[((int64_t)ty.lo * W + tx.lo) * CV + c];
                const V v2 = img[((int64_t)ty.lo * W + tx.hi) * CV + c];
                const V v3 = img[((int64_t)ty.hi * W + tx.lo) * CV + c];
                const V v4 = img[((int64_t)ty.hi * W + tx.hi) * CV + c];
                acc = VecOps<V>::mad4(acc, w1, v1, w2, v2, w3, v3, w4, v4);
            }
        }
        out[c] = VecOps<V>::div(acc, g.count);
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
// stores the result once (no atomics, no zero-fill, summation order = RoI order: bit-
// reproducible run to run):
//   roi_rows_kernel      (grid H x N)            per (image, row): the ordered list of RoIs
//                                                whose patch contains the row
//   roi_align_bwd_rows   (grid W/XT x H x N)     XT pixels of one row x all channels; RoIs of
//                                                the row's list are processed four at a time:
//                                                their separable weights Ay[ph], Bx[pw][x] are
//                                                built cooperatively in LDS, then each lane
//                                                (one float4 of channels) streams the
//                                                contributing gy bins with coalesced reads.
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

// list[(n*H + y)*R + k] = k-th RoI (ascending index) of image n whose patch contains row y
__global__ void __launch_bounds__(256)
roi_rows_kernel(const float *__restrict__ rois, int R, int H, int W, int PH, int PW,
                float spatial_scale, int sampling_ratio, uint16_t *__restrict__ lists,
                int32_t *__restrict__ counts)
{
    __shared__ int wave_cnt[4];
    const int y = blockIdx.x, n = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint16_t *__restrict__ list = lists + ((int64_t)n * H + y) * R;
    int base = 0;
    for (int r0 = 0; r0 < R; r0 += 256) {
        const int r = r0 + tid;
        bool in = false;
        if (r < R) {
            const RoiGeom g = roi_geom(rois + 5 * r, spatial_scale, PH, PW, sampling_ratio);
            if (g.batch == n) {
                const RoiExtent e = roi_extent(g, H, W, PH, PW);
                in = y >= e.ylo && y <= e.yhi;
            }
        }
        const unsigned long long b = __ballot(in);
        if (lane == 0) wave_cnt[wave] = (int)__popcll(b);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        if (in) list[off + (int)__popcll(b & ((1ull << lane) - 1ull))] = (uint16_t)r;
        base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (tid == 0) counts[n * H + y] = base;
}

constexpr int kRowsXT = 8;      // pixels of a row per workgroup
constexpr int kRowsGroup = 4;   // RoIs whose weights are built per barrier pair
constexpr int kRowsMaxBins = 16;

template <typename V>
__global__ void __launch_bounds__(256)
roi_align_bwd_rows_kernel(const V *__restrict__ gy, const float *__restrict__ rois,
                          const uint16_t *__restrict__ lists, const int32_t *__restrict__ counts,
                          V *__restrict__ gx, int R, int H, int W, int CV, int PH, int PW,
                          float spatial_scale, int sampling_ratio, int OH, int OW, int BS)
{
    constexpr int XT = kRowsXT, GR = kRowsGroup, MB = kRowsMaxBins;
    __shared__ __attribute__((aligned(16))) float sBx[GR][MB][XT];   // [roi][pw][x]
    __shared__ float sAy[GR][MB];                                     // already / count
    __shared__ int sRoi[GR];                                          // RoI index or -1 (skip)
    const int x0 = blockIdx.x * XT, y = blockIdx.y, n = blockIdx.z;
    const int tid = threadIdx.x;
    const int cnt = counts[n * H + y];
    const uint16_t *__restrict__ list = lists + ((int64_t)n * H + y) * R;
    const int per_roi = OH + OW * XT;            // weight entries of one RoI
    const int nthreads = blockDim.x;

    for (int c0 = 0; c0 < CV; c0 += nthreads) {
        const int c = c0 + tid;
        V acc[XT];
#pragma unroll
        for (int x = 0; x < XT; ++x) acc[x] = VecOps<V>::zero();

        for (int k0 = 0; k0 < cnt; k0 += GR) {
            __syncthreads();                     // the previous group's weights are consumed
            for (int e = tid; e < GR * per_roi; e += nthreads) {
                const int gi = e / per_roi, j = e - gi * per_roi;
                const int k = k0 + gi;
                if (k >= cnt) {
                    if (j == 0) sRoi[gi] = -1;
                    continue;
                }
                const int r = list[k];
                const RoiGeom g = roi_geom(rois + 5 * r, spatial_scale, PH, PW, sampling_ratio);
                if (j == 0) {
                    const RoiExtent ex = roi_extent(g, H, W, PH, PW);
                    sRoi[gi] = (ex.xhi < x0 || ex.xlo > x0 + XT - 1) ? -1 : r;
                }
                if (j < OH) {
                    const int ph = j * BS;
                    float a = 0.f;
                    for (int iy = 0; iy < g.grid_h; ++iy) {
                        const float yy = g.start_h + ph * g.bin_h +
                                         (float)(iy + .5f) * g.bin_h / (float)g.grid_h;
                        const Tap1D t = tap1d(yy, H);
                        if (!t.valid) continue;
                        if (t.lo == y) a += t.h;
                        if (t.hi == y) a += t.l;
                    }
                    sAy[gi][j] = a / g.count;
                } else {
                    const int q = j - OH, ow_ = q / XT, xi = q - ow_ * XT;
                    const int pw = ow_ * BS, x = x0 + xi;
                    float b = 0.f;
                    for (int ix = 0; ix < g.grid_w; ++ix) {
                        const float xx = g.start_w + pw * g.bin_w +
                                         (float)(ix + .5f) * g.bin_w / (float)g.grid_w;
                        const Tap1D t = tap1d(xx, W);
                        if (!t.valid) continue;
                        if (t.lo == x) b += t.h;
                        if (t.hi == x) b += t.l;
                    }
                    sBx[gi][ow_][xi] = b;
                }
            }
            __syncthreads();
            if (c < CV) {
                for (int gi = 0; gi < GR; ++gi) {
                    const int r = sRoi[gi];
                    if (r < 0) continue;
                    const V *__restrict__ top = gy + (int64_t)r * OH * OW * CV + c;
                    for (int oh = 0; oh < OH; ++oh) {
                        const float ay = sAy[gi][oh];
                        if (ay == 0.f) continue;
                        for (int ow_ = 0; ow_ < OW; ++ow_) {
                            float w[XT];
                            bool any = false;
#pragma unroll
                            for (int x = 0; x < XT; ++x) {
                                w[x] = sBx[gi][ow_][x];
                                any |= w[x] != 0.f;
                            }
                            if (!any) continue;
                            const V v = top[(int64_t)(oh * OW + ow_) * CV];
#pragma unroll
                            for (int x = 0; x < XT; ++x)
                                if (w[x] != 0.f) acc[x] = vfma(acc[x], ay * w[x], v);
                        }
                    }
                }
            }
        }
        if (c < CV) {
            V *__restrict__ row = gx + (((int64_t)n * H + y) * W + x0) * CV + c;
#pragma unroll
            for (int x = 0; x < XT; ++x)
                if (x0 + x < W) row[(int64_t)x * CV] = acc[x];
        }
    }
}

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

extern "C" int mrcnn_roi_align_fwd_ex(const float *x, const float *rois, float *y, int N, int H,
                                      int W, int C, int R, int PH, int PW, int bin_stride,
                                      float spatial_scale, int sampling_ratio, void *stream)
{
    if (int rc = check_args(x, rois, y, N, H, W, C, R, PH, PW, sampling_ratio)) return rc;
    MRCNN_REQUIRE(bin_stride >= 1, "roi_align: bin_stride must be >= 1");
    if (R == 0) return 0;
    const int OH = (PH + bin_stride - 1) / bin_stride, OW = (PW + bin_stride - 1) / bin_stride;
    const int bins = R * OH * OW;
    hipStream_t s = mrcnn::as_stream(stream);
    // algorithmic bytes: write R*OH*OW*C, read the feature maps once
    mrcnn::ProfScope prof(mrcnn::PROF_ROI_ALIGN_FWD, 0.,
                          4.0 * ((double)bins * C + (double)N * H * W * C), s);
    if (C % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)y % 16 == 0)) {
        const int cv = C / 4;
        hipLaunchKernelGGL(roi_align_fwd_kernel<float4>, dim3(bins), dim3(pick_threads(cv)), 0, s,
                           (const float4 *)x, rois, (float4 *)y, H, W, cv, PH, PW, spatial_scale,
                           sampling_ratio, OH, OW, bin_stride);
    } else {
        hipLaunchKernelGGL(roi_align_fwd_kernel<float>, dim3(bins), dim3(pick_threads(C)), 0, s, x,
                           rois, y, H, W, C, PH, PW, spatial_scale, sampling_ratio, OH, OW,
                           bin_stride);
    }
    return mrcnn::check_launch("roi_align_fwd");
}

extern "C" int mrcnn_roi_align_fwd(const float *x, const float *rois, float *y, int N, int H,
                                   int W, int C, int R, int PH, int PW, float spatial_scale,
                                   int sampling_ratio, void *stream)
{
    return mrcnn_roi_align_fwd_ex(x, rois, y, N, H, W, C, R, PH, PW, 1, spatial_scale,
                                  sampling_ratio, stream);
}

extern "C" int64_t mrcnn_roi_align_bwd_workspace_bytes(int N, int H, int R)
{
    if (N <= 0 || H <= 0 || R < 0) return 0;
    // per (image, row): RoI list (uint16) + count (int32), 64-byte aligned sections
    const int64_t lists = (((int64_t)N * H * R * 2 + 63) / 64) * 64;
    return lists + (int64_t)N * H * 4 + 64;
}

extern "C" int mrcnn_roi_align_bwd_ex(const float *gy, const float *rois, float *gx, int N, int H,
                                      int W, int C, int R, int PH, int PW, int bin_stride,
                                      float spatial_scale, int sampling_ratio, void *ws,
                                      void *stream)
{
    if (int rc = check_args(gy, rois, gx, N, H, W, C, R, PH, PW, sampling_ratio)) return rc;
    MRCNN_REQUIRE(bin_stride >= 1, "roi_align: bin_stride must be >= 1");
    hipStream_t s = mrcnn::as_stream(stream);
    MRCNN_REQUIRE(gx != nullptr, "roi_align_bwd: null gx");
    const int OH = (PH + bin_stride - 1) / bin_stride, OW = (PW + bin_stride - 1) / bin_stride;
    const int bins = R * OH * OW;
    const bool vec = C % 4 == 0 && ((uintptr_t)gx % 16 == 0) && ((uintptr_t)gy % 16 == 0);
    if (ws && R > 0 && R <= 65535 && H <= 65535 && N <= 65535 && OH <= kRowsMaxBins &&
        OW <= kRowsMaxBins) {
        // pixel-owner form: every gx element is written exactly once, no zero-fill.
        // algorithmic bytes: read R*OH*OW*C, write the feature-map gradient
        mrcnn::ProfScope prof(mrcnn::PROF_ROI_ALIGN_BWD, 0.,
                              4.0 * ((double)bins * C + (double)N * H * W * C), s);
        uint16_t *lists = (uint16_t *)ws;
        int32_t *counts = (int32_t *)((char *)ws + (((int64_t)N * H * R * 2 + 63) / 64) * 64);
        hipLaunchKernelGGL(roi_rows_kernel, dim3(H, N), dim3(256), 0, s, rois, R, H, W, PH, PW,
                           spatial_scale, sampling_ratio, lists, counts);
        const dim3 grid((W + kRowsXT - 1) / kRowsXT, H, N);
        if (vec)
            hipLaunchKernelGGL(roi_align_bwd_rows_kernel<float4>, grid, dim3(pick_threads(C / 4)), 0,
                               s, (const float4 *)gy, rois, lists, counts, (float4 *)gx, R, H, W,
                               C / 4, PH, PW, spatial_scale, sampling_ratio, OH, OW, bin_stride);
        else
            hipLaunchKernelGGL(roi_align_bwd_rows_kernel<float>, grid, dim3(pick_threads(C)), 0, s,
                               gy, rois, lists, counts, gx, R, H, W, C, PH, PW, spatial_scale,
                               sampling_ratio, OH, OW, bin_stride);
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
