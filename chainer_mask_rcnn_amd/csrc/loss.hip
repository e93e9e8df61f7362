// The five Mask R-CNN losses (+ inference softmax) for gfx950.
//
// Replaces chainer's F.sigmoid_cross_entropy / F.softmax_cross_entropy and the
// reference's _smooth_l1_loss / _fast_rcnn_loc_loss
// (/root/reference/chainer_mask_rcnn/models/mask_rcnn_train_chain.py:163-181,
// :192-213; formulas SURVEY.md Appendix A.1).  Each loss is two launches:
//   pass 1  per-workgroup partial (sum, count) -> workspace
//   pass 2  every workgroup re-reduces the (<=256) partials, block 0 writes the
//           normalised scalar, all write the gradient (already divided by count).
// No host synchronisation (the reference syncs on `xp.sum(gt_label >= 0)` :212).
#include "common.h"

namespace {

constexpr int kMaxParts = 256;

struct Partials {
    double sum[kMaxParts];
    int count[kMaxParts];
};

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// block (256 threads) reduce -> partial slot
__device__ __forceinline__ void block_store_partial(float s, int c, Partials *__restrict__ ws)
{
    __shared__ float ss[4];
    __shared__ int sc[4];
    s = wave_sum(s);
    c = wave_sum_i(c);
    if ((threadIdx.x & 63) == 0) { ss[threadIdx.x >> 6] = s; sc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ws->sum[blockIdx.x] = ((double)ss[0] + (double)ss[1]) + ((double)ss[2] + (double)ss[3]);
        ws->count[blockIdx.x] = (sc[0] + sc[1]) + (sc[2] + sc[3]);
    }
}

// every block: total over `parts` partials (serial, fixed order => deterministic)
__device__ __forceinline__ void load_totals(const Partials *__restrict__ ws, int parts,
                                            double *sum, int *count)
{
    __shared__ double tsum;
    __shared__ int tcount;
    if (threadIdx.x == 0) {
        double s = 0.;
        int c = 0;
        for (int i = 0; i < parts; ++i) { s += ws->sum[i]; c += ws->count[i]; }
        tsum = s;
        tcount = c;
    }
    __syncthreads();
    *sum = tsum;
    *count = tcount;
}

__device__ __forceinline__ float sce_loss_el(float x, int t)
{
    // -(x*(t - (x>=0)) - log1p(exp(-|x|)))
    return -(x * ((float)t - (x >= 0.f ? 1.f : 0.f)) - log1pf(expf(-fabsf(x))));
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// ---- sigmoid cross entropy, flat --------------------------------------------------
__global__ void __launch_bounds__(256)
sce_partial_kernel(const float *__restrict__ x, const int32_t *__restrict__ t, int64_t n,
                   Partials *__restrict__ ws)
{
    float s = 0.f;
    int c = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ti = t[i];
        if (ti != -1) { s += sce_loss_el(x[i], ti); ++c; }
    }
    block_store_partial(s, c, ws);
}

__global__ void __launch_bounds__(256)
sce_grad_kernel(const float *__restrict__ x, const int32_t *__restrict__ t, int64_t n,
                const Partials *__restrict__ ws, int parts, float *__restrict__ loss,
                float *__restrict__ gx)
{
    double sum; int count;
    load_totals(ws, parts, &sum, &count);
    const float inv = 1.f / (float)max(count, 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) *loss = (float)(sum / (double)max(count, 1));
    if (!gx) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ti = t[i];
        gx[i] = ti != -1 ? (sigmoidf(x[i]) - (float)ti) * inv : 0.f;
    }
}

// ---- mask sigmoid cross entropy with per-row channel selection --------------------
// x (R,HW,Kc): row r uses channel (label[r]-1) mod Kc (NumPy negative index for
// background rows, whose targets are all -1).
__global__ void __launch_bounds__(256)
msce_partial_kernel(const float *__restrict__ x, const int32_t *__restrict__ label,
                    const int32_t *__restrict__ t, int R, int HW, int Kc,
                    Partials *__restrict__ ws)
{
    float s = 0.f;
    int c = 0;
    const int64_t n = (int64_t)R * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int ti = t[i];
        if (ti != -1) {
            const int r = (int)(i / HW);
            int ch = label[r] - 1;
            if (ch < 0) ch += Kc;
            s += sce_loss_el(x[i * Kc + ch], ti);
            ++c;
        }
    }
    block_store_partial(s, c, ws);
}

__global__ void __launch_bounds__(256)
msce_grad_kernel(const float *__restrict__ x, const int32_t *__restrict__ label,
                 const int32_t *__restrict__ t, int R, int HW, int Kc,
                 const Partials *__restrict__ ws, int parts, float *__restrict__ loss,
                 float *__restrict__ gx)
{
    double sum; int count;
    load_totals(ws, parts, &sum, &count);
    const float inv = 1.f / (float)max(count, 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) *loss = (float)(sum / (double)max(count, 1));
    if (!gx) return;
    const int64_t n = (int64_t)R * HW * Kc;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / Kc;
        const int k = (int)(e % Kc);
        const int r = (int)(i / HW);
        int ch = label[r] - 1;
        if (ch < 0) ch += Kc;
        float g = 0.f;
        if (k == ch) {
            const int ti = t[i];
            if (ti != -1) g = (sigmoidf(x[e]) - (float)ti) * inv;
        }
        gx[e] = g;
    }
}

// ---- softmax cross entropy: one wave per row ---------------------------------------
__global__ void __launch_bounds__(256)
smce_partial_kernel(const float *__restrict__ x, int ldx, const int32_t *__restrict__ t, int R,
                    int ncls, Partials *__restrict__ ws, float *__restrict__ lse)
{
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    float s = 0.f;
    int c = 0;
    for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < R; r += gridDim.x * wpb) {
        const float *row = x + (int64_t)r * ldx;
        float m = -INFINITY;
        for (int k = lane; k < ncls; k += 64) m = fmaxf(m, row[k]);
        m = wave_max(m);
        float z = 0.f;
        for (int k = lane; k < ncls; k += 64) z += expf(row[k] - m);
        z = wave_sum(z);
        const float l = m + logf(z);
        if (lane == 0) {
            lse[r] = l;
            const int ti = t[r];
            if (ti != -1) { s += l - row[ti]; ++c; }
        }
    }
    block_store_partial(s, c, ws);
}

__global__ void __launch_bounds__(256)
smce_grad_kernel(const float *__restrict__ x, int ldx, const int32_t *__restrict__ t, int R,
                 int ncls, const Partials *__restrict__ ws, int parts,
                 const float *__restrict__ lse, float *__restrict__ loss, float *__restrict__ gx,
                 int ldg)
{
    double sum; int count;
    load_totals(ws, parts, &sum, &count);
    const float inv = 1.f / (float)max(count, 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) *loss = (float)(sum / (double)max(count, 1));
    if (!gx) return;
    const int64_t n = (int64_t)R * ncls;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / ncls), k = (int)(e % ncls);
        const int ti = t[r];
        float g = 0.f;
        if (ti != -1) g = (expf(x[(int64_t)r * ldx + k] - lse[r]) - (k == ti ? 1.f : 0.f)) * inv;
        gx[(int64_t)r * ldg + k] = g;
    }
}

__global__ void __launch_bounds__(256)
softmax_kernel(const float *__restrict__ x, int ldx, float *__restrict__ y, int ldy, int R,
               int ncls)
{
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    for (int r = blockIdx.x * wpb + (threadIdx.x >> 6); r < R; r += gridDim.x * wpb) {
        const float *row = x + (int64_t)r * ldx;
        float m = -INFINITY;
        for (int k = lane; k < ncls; k += 64) m = fmaxf(m, row[k]);
        m = wave_max(m);
        float z = 0.f;
        for (int k = lane; k < ncls; k += 64) z += expf(row[k] - m);
        z = wave_sum(z);
        for (int k = lane; k < ncls; k += 64) y[(int64_t)r * ldy + k] = expf(row[k] - m) / z;
    }
}

// ---- smooth L1 (models/mask_rcnn_train_chain.py:192-213) ---------------------------
__global__ void __launch_bounds__(256)
sl1_partial_kernel(const float *__restrict__ pred, int ld, const int32_t *__restrict__ cls,
                   const float *__restrict__ gt_loc, const int32_t *__restrict__ gt_label, int n,
                   float sigma2, Partials *__restrict__ ws)
{
    float s = 0.f;
    int c = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int lab = gt_label[i];
        if (lab >= 0) ++c;
        if (lab > 0) {
            const float *p = pred + (int64_t)i * ld + (cls ? 4 * cls[i] : 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = p[k] - gt_loc[(int64_t)i * 4 + k];
                const float a = fabsf(d);
                s += a < (1.f / sigma2) ? (sigma2 * 0.5f) * d * d : a - 0.5f / sigma2;
            }
        }
    }
    block_store_partial(s, c, ws);
}

__global__ void __launch_bounds__(256)
sl1_grad_kernel(const float *__restrict__ pred, int ld, const int32_t *__restrict__ cls,
                const float *__restrict__ gt_loc, const int32_t *__restrict__ gt_label, int n,
                float sigma2, const Partials *__restrict__ ws, int parts,
                float *__restrict__ loss, float *__restrict__ gx)
{
    double sum; int count;
    load_totals(ws, parts, &sum, &count);
    // the reference divides by xp.sum(gt_label >= 0) without a guard (:212)
    const float inv = 1.f / (float)count;
    if (blockIdx.x == 0 && threadIdx.x == 0) *loss = (float)(sum / (double)count);
    if (!gx) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int lab = gt_label[i];
        const int64_t off = (int64_t)i * ld + (cls ? 4 * cls[i] : 0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float g = 0.f;
            if (lab > 0) {
                const float d = pred[off + k] - gt_loc[(int64_t)i * 4 + k];
                const float a = fabsf(d);
                g = a < (1.f / sigma2) ? sigma2 * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
                g *= inv;
            }
            gx[off + k] = g;
        }
    }
}

inline int parts_for(int64_t n)
{
    int64_t b = mrcnn::ceil_div(n, 256 * 4);
    if (b > kMaxParts) b = kMaxParts;
    return (int)(b < 1 ? 1 : b);
}
inline int grid_for(int64_t work)
{
    int64_t b = mrcnn::ceil_div(work, 256);
    if (b > 2048) b = 2048;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" int64_t mrcnn_loss_workspace_bytes(int rows) { return (int64_t)sizeof(Partials) + 4 * (int64_t)rows + 64; }

extern "C" int mrcnn_sigmoid_ce(const float *x, const int32_t *t, int64_t n, float *loss,
                                float *gx, void *ws, void *stream)
{
    MRCNN_REQUIRE(n >= 0, "sigmoid_ce: n < 0");
    MRCNN_REQUIRE(loss && ws && (n == 0 || (x && t)), "sigmoid_ce: null pointer");
    hipStream_t s = mrcnn::as_stream(stream);
    const int parts = parts_for(n);
    hipLaunchKernelGGL(sce_partial_kernel, dim3(parts), dim3(256), 0, s, x, t, n, (Partials *)ws);
    hipLaunchKernelGGL(sce_grad_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, t, n,
                       (const Partials *)ws, parts, loss, gx);
    return mrcnn::check_launch("sigmoid_ce");
}

extern "C" int mrcnn_mask_sigmoid_ce(const float *x, const int32_t *label, const int32_t *t, int R,
                                     int HW, int Kc, float *loss, float *gx, void *ws,
                                     void *stream)
{
    MRCNN_REQUIRE(R >= 0 && HW > 0 && Kc > 0, "mask_sigmoid_ce: bad shape");
    MRCNN_REQUIRE(loss && ws && (R == 0 || (x && t && label)), "mask_sigmoid_ce: null pointer");
    hipStream_t s = mrcnn::as_stream(stream);
    const int parts = parts_for((int64_t)R * HW);
    hipLaunchKernelGGL(msce_partial_kernel, dim3(parts), dim3(256), 0, s, x, label, t, R, HW, Kc,
                       (Partials *)ws);
    hipLaunchKernelGGL(msce_grad_kernel, dim3(grid_for((int64_t)R * HW * Kc)), dim3(256), 0, s, x,
                       label, t, R, HW, Kc, (const Partials *)ws, parts, loss, gx);
    return mrcnn::check_launch("mask_sigmoid_ce");
}

extern "C" int mrcnn_softmax_ce(const float *x, int ldx, const int32_t *t, int R, int ncls,
                                float *loss, float *gx, int ldg, void *ws, void *stream)
{
    MRCNN_REQUIRE(R >= 0 && ncls > 0 && ldx >= ncls, "softmax_ce: bad shape");
    MRCNN_REQUIRE(loss && ws && (R == 0 || (x && t)), "softmax_ce: null pointer");
    hipStream_t s = mrcnn::as_stream(stream);
    int parts = (int)mrcnn::ceil_div(R, 4);
    if (parts > kMaxParts) parts = kMaxParts;
    if (parts < 1) parts = 1;
    float *lse = (float *)((char *)ws + sizeof(Partials));
    hipLaunchKernelGGL(smce_partial_kernel, dim3(parts), dim3(256), 0, s, x, ldx, t, R, ncls,
                       (Partials *)ws, lse);
    hipLaunchKernelGGL(smce_grad_kernel, dim3(grid_for((int64_t)R * ncls)), dim3(256), 0, s, x, ldx,
                       t, R, ncls, (const Partials *)ws, parts, lse, loss, gx, ldg);
    return mrcnn::check_launch("softmax_ce");
}

extern "C" int mrcnn_softmax(const float *x, int ldx, float *y, int ldy, int R, int ncls,
                             void *stream)
{
    MRCNN_REQUIRE(R >= 0 && ncls > 0, "softmax: bad shape");
    if (R == 0) return 0;
    MRCNN_REQUIRE(x && y, "softmax: null pointer");
    int blocks = (int)mrcnn::ceil_div(R, 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(softmax_kernel, dim3(blocks), dim3(256), 0, mrcnn::as_stream(stream), x, ldx,
                       y, ldy, R, ncls);
    return mrcnn::check_launch("softmax");
}

extern "C" int mrcnn_smooth_l1(const float *pred, int ld, const int32_t *cls, const float *gt_loc,
                               const int32_t *gt_label, int n, float sigma, float *loss,
                               float *gx, void *ws, void *stream)
{
    MRCNN_REQUIRE(n >= 0 && ld >= 4, "smooth_l1: bad shape");
    MRCNN_REQUIRE(loss && ws && (n == 0 || (pred && gt_loc && gt_label)), "smooth_l1: null pointer");
    hipStream_t s = mrcnn::as_stream(stream);
    const int parts = parts_for(n);
    const float sigma2 = sigma * sigma;
    hipLaunchKernelGGL(sl1_partial_kernel, dim3(parts), dim3(256), 0, s, pred, ld, cls, gt_loc,
                       gt_label, n, sigma2, (Partials *)ws);
    hipLaunchKernelGGL(sl1_grad_kernel, dim3(grid_for(n)), dim3(256), 0, s, pred, ld, cls, gt_loc,
                       gt_label, n, sigma2, (const Partials *)ws, parts, loss, gx);
    return mrcnn::check_launch("smooth_l1");
}
