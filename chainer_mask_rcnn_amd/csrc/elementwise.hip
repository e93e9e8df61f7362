// HBM-bound elementwise / pooling / optimizer kernels for gfx950 (NHWC, fp32).
// Each kernel moves 16 B per lane where the channel count allows.
#include <algorithm>

#include "common.h"

namespace {

constexpr int kColsumSplits = 128;

// ---- AffineChannel2D (functions/affine_channel_2d.py:10-22, :38-56) --------
__global__ void affine_fwd_kernel(const float *__restrict__ x, const float *__restrict__ W,
                                  const float *__restrict__ b, float *__restrict__ y,
                                  int64_t total, int C)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        y[i] = W[c] * x[i] + b[c];
    }
}

// g = gy * (y > 0 ? 1 : 0) * scale[c]   (y == NULL: no mask, scale == NULL: 1)
template <bool VEC>
__global__ void epilogue_bwd_kernel(const float *__restrict__ gy, const float *__restrict__ y,
                                    const float *__restrict__ scale, float *__restrict__ g,
                                    int64_t total, int C)
{
    if (VEC) {
        const int64_t tv = total / 4;
        const int cv = C / 4;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tv;
             i += (int64_t)gridDim.x * blockDim.x) {
            float4 v = reinterpret_cast<const float4 *>(gy)[i];
            if (y) {
                const float4 m = reinterpret_cast<const float4 *>(y)[i];
                v.x = m.x > 0.f ? v.x : 0.f;
                v.y = m.y > 0.f ? v.y : 0.f;
                v.z = m.z > 0.f ? v.z : 0.f;
                v.w = m.w > 0.f ? v.w : 0.f;
            }
            if (scale) {
                const float4 s = reinterpret_cast<const float4 *>(scale)[i % cv];
                v.x *= s.x; v.y *= s.y; v.z *= s.z; v.w *= s.w;
            }
            reinterpret_cast<float4 *>(g)[i] = v;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
             i += (int64_t)gridDim.x * blockDim.x) {
            float v = gy[i];
            if (y) v = y[i] > 0.f ? v : 0.f;
            if (scale) v *= scale[i % C];
            g[i] = v;
        }
    }
}

// partial[split][c] = sum over the split's rows of a[m,c] * (b ? b[m,c] : 1)
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t M, int C,
                      float *__restrict__ partial)
{
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int split = blockIdx.y;
    const int64_t rows_per = (M + gridDim.y - 1) / gridDim.y;
    const int64_t m0 = split * rows_per;
    const int64_t m1 = min(M, m0 + rows_per);
    float acc = 0.f;
    if (c < C) {
        for (int64_t m = m0 + ry; m < m1; m += 4) {
            const float v = a[m * C + c];
            acc += b ? v * b[m * C + c] : v;
        }
    }
    red[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && c < C)
        partial[(int64_t)split * C + c] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
}

// 64 columns x 4 lanes of splits per workgroup: lane group g sums the slabs s = g, g + 4, ... with
// four loads in flight, then the four partial sums are combined in a fixed order (deterministic).
// One thread per column walking all 128 slabs was a chain of 128 dependent L2 round trips: 30 us
// on the critical path of every bias gradient.
__global__ void __launch_bounds__(256)
colsum_final_kernel(const float *__restrict__ partial, int splits, int C, float *__restrict__ out)
{
    __shared__ float red[4][64];
    const int cx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (c < C) {
        int s = g;
        for (; s + 12 < splits; s += 16) {
            const float v0 = partial[(int64_t)s * C + c], v1 = partial[(int64_t)(s + 4) * C + c];
            const float v2 = partial[(int64_t)(s + 8) * C + c], v3 = partial[(int64_t)(s + 12) * C + c];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; s < splits; s += 4) a0 += partial[(int64_t)s * C + c];
    }
    red[g][cx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (g == 0 && c < C) out[c] = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
}

// ---- F.max_pooling_2d(x,3,2,1) cover_all, -inf padding ------------------------
template <typename V>
__device__ __forceinline__ V vmax(V a, V b);
template <> __device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }
template <> __device__ __forceinline__ float4 vmax(float4 a, float4 b)
{
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
template <typename V> __device__ __forceinline__ V vneginf();
template <> __device__ __forceinline__ float vneginf() { return -INFINITY; }
template <> __device__ __forceinline__ float4 vneginf()
{
    return make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
}

template <typename V>
__global__ void maxpool_kernel(const V *__restrict__ x, V *__restrict__ y, int N, int H, int W,
                               int CV, int P, int Q)
{
    const int64_t total = (int64_t)N * P * Q * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV);
        int64_t t = i / CV;
        const int q = (int)(t % Q); t /= Q;
        const int p = (int)(t % P);
        const int n = (int)(t / P);
        V m = vneginf<V>();
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = 2 * p - 1 + r;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int ix = 2 * q - 1 + s;
                if (ix < 0 || ix >= W) continue;
                m = vmax(m, x[(((int64_t)n * H + iy) * W + ix) * CV + c]);
            }
        }
        y[i] = m;
    }
}

// ---- F.average_pooling_2d over the whole HW map --------------------------------
template <bool VEC>
__global__ void avgpool_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int R,
                                   int HW, int C)
{
    const int w = VEC ? 4 : 1;
    const int cv = C / w;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)R * cv) return;
    const int r = (int)(i / cv), c = (int)(i % cv);
    const float inv = 1.f / (float)HW;
    if (VEC) {
        const float4 *p = reinterpret_cast<const float4 *>(x) + (int64_t)r * HW * cv + c;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < HW; ++k) {
            const float4 v = p[(int64_t)k * cv];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        reinterpret_cast<float4 *>(y)[i] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
    } else {
        const float *p = x + (int64_t)r * HW * C + c;
        float a = 0.f;
        for (int k = 0; k < HW; ++k) a += p[(int64_t)k * C];
        y[i] = a * inv;
    }
}

template <bool VEC>
__global__ void avgpool_bwd_kernel(const float *__restrict__ gy, float *__restrict__ gx, int R,
                                   int HW, int C, int accumulate)
{
    const int w = VEC ? 4 : 1;
    const int cv = C / w;
    const int64_t total = (int64_t)R * HW * cv;
    const float inv = 1.f / (float)HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        const int r = (int)(i / ((int64_t)HW * cv));
        if (VEC) {
            float4 g = reinterpret_cast<const float4 *>(gy)[(int64_t)r * cv + c];
            g.x *= inv; g.y *= inv; g.z *= inv; g.w *= inv;
            float4 *o = reinterpret_cast<float4 *>(gx) + i;
            if (accumulate) { const float4 p = *o; g.x += p.x; g.y += p.y; g.z += p.z; g.w += p.w; }
            *o = g;
        } else {
            float g = gy[(int64_t)r * C + c] * inv;
            if (accumulate) g += gx[i];
            gx[i] = g;
        }
    }
}

// ---- MomentumSGD + WeightDecay -----------------------------------
template <bool ZERO_GRAD>
__global__ void sgd_kernel(float *__restrict__ p, float *__restrict__ g,
                           float *__restrict__ v, int64_t n, float lr, float momentum, float wd,
                           float grad_scale)
{
    const int64_t nv = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv;
         i += (int64_t)gridDim.x * blockDim.x) {
        float4 pp = reinterpret_cast<float4 *>(p)[i];
        const float4 gg = reinterpret_cast<const float4 *>(g)[i];
        float4 vv = reinterpret_cast<float4 *>(v)[i];
        vv.x = momentum * vv.x - lr * (gg.x * grad_scale + wd * pp.x);
        vv.y = momentum * vv.y - lr * (gg.y * grad_scale + wd * pp.y);
        vv.z = momentum * vv.z - lr * (gg.z * grad_scale + wd * pp.z);
        vv.w = momentum * vv.w - lr * (gg.w * grad_scale + wd * pp.w);
        pp.x += vv.x; pp.y += vv.y; pp.z += vv.z; pp.w += vv.w;
        reinterpret_cast<float4 *>(v)[i] = vv;
        reinterpret_cast<float4 *>(p)[i] = pp;
        if (ZERO_GRAD) reinterpret_cast<float4 *>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int64_t i = nv * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float vv = momentum * v[i] - lr * (g[i] * grad_scale + wd * p[i]);
        v[i] = vv;
        p[i] += vv;
        if (ZERO_GRAD) g[i] = 0.f;
    }
}

inline int grid_for(int64_t work, int threads = 256)
{
    int64_t b = mrcnn::ceil_div(work, threads);
    if (b > 256 * 16) b = 256 * 16;
    return (int)(b < 1 ? 1 : b);
}
inline bool aligned16(const void *p) { return ((uintptr_t)p % 16) == 0; }

}  // namespace

extern "C" int mrcnn_affine_fwd(const float *x, const float *W, const float *b, float *y,
                                int64_t M, int C, void *stream)
{
    MRCNN_REQUIRE(M >= 0 && C > 0, "affine_fwd: bad shape");
    if (M == 0) return 0;
    MRCNN_REQUIRE(x && W && b && y, "affine_fwd: null pointer");
    const int64_t total = M * C;
    hipLaunchKernelGGL(affine_fwd_kernel, dim3(grid_for(total)), dim3(256), 0,
                       mrcnn::as_stream(stream), x, W, b, y, total, C);
    return mrcnn::check_launch("affine_fwd");
}

extern "C" int mrcnn_epilogue_bwd(const float *gy, const float *y, const float *scale, float *g,
                                  int64_t M, int C, void *stream)
{
    MRCNN_REQUIRE(M >= 0 && C > 0, "epilogue_bwd: bad shape");
    if (M == 0) return 0;
    MRCNN_REQUIRE(gy && g, "epilogue_bwd: null pointer");
    const int64_t total = M * C;
    const bool vec = (C % 4 == 0) && aligned16(gy) && aligned16(g) && (!y || aligned16(y)) &&
                     (!scale || aligned16(scale));
    if (vec)
        hipLaunchKernelGGL(epilogue_bwd_kernel<true>, dim3(grid_for(total / 4)), dim3(256), 0,
                           mrcnn::as_stream(stream), gy, y, scale, g, total, C);
    else
        hipLaunchKernelGGL(epilogue_bwd_kernel<false>, dim3(grid_for(total)), dim3(256), 0,
                           mrcnn::as_stream(stream), gy, y, scale, g, total, C);
    return mrcnn::check_launch("epilogue_bwd");
}

extern "C" int64_t mrcnn_colsum_workspace_bytes(int C) { return (int64_t)kColsumSplits * C * 4; }

static int colsum_impl(const float *a, const float *b, float *out, int64_t M, int C, void *ws,
                       hipStream_t s)
{
    int splits = (int)std::min<int64_t>(kColsumSplits, mrcnn::ceil_div(M, 64));
    if (splits < 1) splits = 1;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((C + 63) / 64, splits), dim3(256), 0, s, a, b, M,
                       C, (float *)ws);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 63) / 64), dim3(256), 0, s,
                       (const float *)ws, splits, C, out);
    return mrcnn::check_launch("colsum");
}

extern "C" int mrcnn_colsum(const float *g, float *out, int64_t M, int C, void *ws, void *stream)
{
    MRCNN_REQUIRE(M >= 0 && C > 0, "colsum: bad shape");
    MRCNN_REQUIRE(g && out && ws, "colsum: null pointer");
    return colsum_impl(g, nullptr, out, M, C, ws, mrcnn::as_stream(stream));
}

extern "C" int mrcnn_affine_bwd(const float *x, const float *W, const float *gy, float *gx,
                                float *gW, float *gb, int64_t M, int C, void *ws, void *stream)
{
    MRCNN_REQUIRE(M >= 0 && C > 0, "affine_bwd: bad shape");
    MRCNN_REQUIRE(gy && W, "affine_bwd: null pointer");
    hipStream_t s = mrcnn::as_stream(stream);
    if (gx)
        if (int rc = mrcnn_epilogue_bwd(gy, nullptr, W, gx, M, C, stream)) return rc;
    if (gW) {
        MRCNN_REQUIRE(x && ws, "affine_bwd: gW needs x and workspace");
        if (int rc = colsum_impl(gy, x, gW, M, C, ws, s)) return rc;
    }
    if (gb) {
        MRCNN_REQUIRE(ws, "affine_bwd: gb needs workspace");
        if (int rc = colsum_impl(gy, nullptr, gb, M, C, ws, s)) return rc;
    }
    return 0;
}

extern "C" int mrcnn_maxpool3x3s2p1_fwd(const float *x, float *y, int N, int H, int W, int C,
                                        int P, int Q, void *stream)
{
    MRCNN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, "maxpool: bad shape");
    // chainer get_conv_outsize(size,3,2,1,cover_all=True) = (size + 2 - 3 + 2 - 1)//2 + 1
    MRCNN_REQUIRE(P == (H + 2 - 3 + 1) / 2 + 1 && Q == (W + 2 - 3 + 1) / 2 + 1,
                  "maxpool: output size must be cover_all (%d,%d) for input (%d,%d)",
                  (H + 2 - 3 + 1) / 2 + 1, (W + 2 - 3 + 1) / 2 + 1, H, W);
    MRCNN_REQUIRE(x && y, "maxpool: null pointer");
    hipStream_t s = mrcnn::as_stream(stream);
    if (C % 4 == 0 && aligned16(x) && aligned16(y)) {
        const int64_t total = (int64_t)N * P * Q * (C / 4);
        hipLaunchKernelGGL(maxpool_kernel<float4>, dim3(grid_for(total)), dim3(256), 0, s,
                           (const float4 *)x, (float4 *)y, N, H, W, C / 4, P, Q);
    } else {
        const int64_t total = (int64_t)N * P * Q * C;
        hipLaunchKernelGGL(maxpool_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, x, y, N, H,
                           W, C, P, Q);
    }
    return mrcnn::check_launch("maxpool");
}

extern "C" int mrcnn_avgpool_fwd(const float *x, float *y, int R, int HW, int C, void *stream)
{
    MRCNN_REQUIRE(R >= 0 && HW > 0 && C > 0, "avgpool_fwd: bad shape");
    if (R == 0) return 0;
    MRCNN_REQUIRE(x && y, "avgpool_fwd: null pointer");
    hipStream_t s = mrcnn::as_stream(stream);
    if (C % 4 == 0 && aligned16(x) && aligned16(y))
        hipLaunchKernelGGL(avgpool_fwd_kernel<true>, dim3(mrcnn::ceil_div((int64_t)R * C / 4, 256)),
                           dim3(256), 0, s, x, y, R, HW, C);
    else
        hipLaunchKernelGGL(avgpool_fwd_kernel<false>, dim3(mrcnn::ceil_div((int64_t)R * C, 256)),
                           dim3(256), 0, s, x, y, R, HW, C);
    return mrcnn::check_launch("avgpool_fwd");
}

extern "C" int mrcnn_avgpool_bwd(const float *gy, float *gx, int R, int HW, int C, int accumulate,
                                 void *stream)
{
    MRCNN_REQUIRE(R >= 0 && HW > 0 && C > 0, "avgpool_bwd: bad shape");
    if (R == 0) return 0;
    MRCNN_REQUIRE(gy && gx, "avgpool_bwd: null pointer");
    hipStream_t s = mrcnn::as_stream(stream);
    if (C % 4 == 0 && aligned16(gy) && aligned16(gx))
        hipLaunchKernelGGL(avgpool_bwd_kernel<true>, dim3(grid_for((int64_t)R * HW * C / 4)),
                           dim3(256), 0, s, gy, gx, R, HW, C, accumulate);
    else
        hipLaunchKernelGGL(avgpool_bwd_kernel<false>, dim3(grid_for((int64_t)R * HW * C)), dim3(256),
                           0, s, gy, gx, R, HW, C, accumulate);
    return mrcnn::check_launch("avgpool_bwd");
}

namespace {
__global__ void __launch_bounds__(256)
head_tail_bwd_kernel(const float4 *__restrict__ g_pool, const float4 *__restrict__ g_rows,
                     const int *__restrict__ slot, const float4 *__restrict__ y,
                     float4 *__restrict__ g, int R, int HW, int cv)
{
    const int64_t total = (int64_t)R * HW * cv;
    const float inv = 1.f / (float)HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv);
        const int64_t rp = i / cv;
        const int r = (int)(rp / HW), p = (int)(rp - (int64_t)r * HW);
        float4 v = g_pool[(int64_t)r * cv + c];
        v.x *= inv; v.y *= inv; v.z *= inv; v.w *= inv;
        const int sl = slot ? slot[r] : -1;
        if (sl >= 0) {
            const float4 a = g_rows[((int64_t)sl * HW + p) * cv + c];
            v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        }
        const float4 m = y[i];
        v.x = m.x > 0.f ? v.x : 0.f;
        v.y = m.y > 0.f ? v.y : 0.f;
        v.z = m.z > 0.f ? v.z : 0.f;
        v.w = m.w > 0.f ? v.w : 0.f;
        g[i] = v;
    }
}
}  // namespace

extern "C" int mrcnn_head_tail_bwd(const float *g_pool, const float *g_rows, const int32_t *slot,
                                   const float *y, float *g, int R, int HW, int C, void *stream)
{
    MRCNN_REQUIRE(R >= 0 && HW > 0 && C > 0 && C % 4 == 0, "head_tail_bwd: bad shape (C must be a multiple of 4)");
    if (R == 0) return 0;
    MRCNN_REQUIRE(g_pool && y && g, "head_tail_bwd: null pointer");
    MRCNN_REQUIRE((g_rows == nullptr) == (slot == nullptr), "head_tail_bwd: g_rows and slot go together");
    MRCNN_REQUIRE(aligned16(g_pool) && aligned16(y) && aligned16(g) && (!g_rows || aligned16(g_rows)),
                  "head_tail_bwd: pointers must be 16-byte aligned");
    const int64_t total = (int64_t)R * HW * (C / 4);
    mrcnn::ProfScope prof(mrcnn::PROF_ELEMENTWISE, 0., 8.0 * (double)R * HW * C, mrcnn::as_stream(stream));
    hipLaunchKernelGGL(head_tail_bwd_kernel, dim3(grid_for(total)), dim3(256), 0,
                       mrcnn::as_stream(stream), (const float4 *)g_pool, (const float4 *)g_rows, slot,
                       (const float4 *)y, (float4 *)g, R, HW, C / 4);
    return mrcnn::check_launch("head_tail_bwd");
}

// ---- row-sparse backward of a 3x3 / stride 1 / pad 1 convolution ---------------------------
// The RPN's losses ignore all but the <= 256 sampled anchors of an image
// (models/mask_rcnn_train_chain.py:150-166, AnchorTargetCreator label -1), so the gradient that
// reaches conv1's output (models/region_proposal_network.py:75-80) is exactly zero outside the
// <= 512 map positions of those anchors.  The backward then needs only those rows: their 3x3
// input patches are gathered, weight and data gradients are ONE 1x1-shaped GEMM each over
// (rows x 9C) operands (6 % of the dense work), and the patch gradients are summed back per map
// pixel in a fixed tap order (pixel-owner form, no atomics).
namespace {
// grid (rows, 10): y < 9: tap (r, s) = (y / 3, y % 3) of the row's patch; y == 9: its gradient row
__global__ void __launch_bounds__(256)
sparse3x3_gather_kernel(const float4 *__restrict__ x, const float4 *__restrict__ g,
                        const int32_t *__restrict__ rows, int H, int W, int C4, int K4,
                        float4 *__restrict__ patches, float4 *__restrict__ g_rows)
{
    const int i = blockIdx.x, t = blockIdx.y;
    const int p = rows[i];
    if (t == 9) {
        const float4 *src = g + (int64_t)p * K4;
        float4 *dst = g_rows + (int64_t)i * K4;
        for (int c = threadIdx.x; c < K4; c += blockDim.x) dst[c] = src[c];
        return;
    }
    const int px = p % W, py = (p / W) % H, n = p / (W * H);
    const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
    const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    const float4 *src = x + (((int64_t)n * H + yy) * W + xx) * C4;
    float4 *dst = patches + ((int64_t)i * 9 + t) * C4;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = threadIdx.x; c < C4; c += blockDim.x) dst[c] = in ? src[c] : zero;
}

// one workgroup = one map pixel q: gx[q] = sum over taps (r, s), in that order, of the patch
// gradient of the sampled position q - (r - 1, s - 1) (lookup: position -> row or -1)
__global__ void __launch_bounds__(256)
sparse3x3_scatter_kernel(const float4 *__restrict__ gp, const int32_t *__restrict__ lookup, int H,
                         int W, int C4, float4 *__restrict__ gx)
{
    const int q = blockIdx.x;
    const int qx = q % W, qy = (q / W) % H, n = q / (W * H);
    int src[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int yy = qy - (t / 3 - 1), xx = qx - (t % 3 - 1);
        src[t] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                     ? lookup[((int64_t)n * H + yy) * W + xx] : -1;
    }
    for (int c = threadIdx.x; c < C4; c += blockDim.x) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (src[t] < 0) continue;
            const float4 v = gp[((int64_t)src[t] * 9 + t) * C4 + c];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        gx[(int64_t)q * C4 + c] = acc;
    }
}
}  // namespace

extern "C" int mrcnn_sparse3x3_gather(const float *x, const float *g, const int32_t *rows,
                                      int n_rows, int N, int H, int W, int C, int K,
                                      float *patches, float *g_rows, void *stream)
{
    MRCNN_REQUIRE(n_rows >= 0 && N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && C % 4 == 0 &&
                      K % 4 == 0, "sparse3x3_gather: bad shape");
    if (n_rows == 0) return 0;
    MRCNN_REQUIRE(x && g && rows && patches && g_rows, "sparse3x3_gather: null pointer");
    mrcnn::ProfScope prof(mrcnn::PROF_ELEMENTWISE, 0.,
                          8.0 * (double)n_rows * (9.0 * C + K), mrcnn::as_stream(stream));
    hipLaunchKernelGGL(sparse3x3_gather_kernel, dim3(n_rows, 10), dim3(256), 0,
                       mrcnn::as_stream(stream), (const float4 *)x, (const float4 *)g, rows, H, W,
                       C / 4, K / 4, (float4 *)patches, (float4 *)g_rows);
    return mrcnn::check_launch("sparse3x3_gather");
}

extern "C" int mrcnn_sparse3x3_scatter(const float *g_patches, const int32_t *lookup, int N, int H,
                                       int W, int C, float *gx, void *stream)
{
    MRCNN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "sparse3x3_scatter: bad shape");
    MRCNN_REQUIRE((int64_t)N * H * W < (int64_t)INT32_MAX, "sparse3x3_scatter: map too large");
    MRCNN_REQUIRE(g_patches && lookup && gx, "sparse3x3_scatter: null pointer");
    hipLaunchKernelGGL(sparse3x3_scatter_kernel, dim3(N * H * W), dim3(256), 0,
                       mrcnn::as_stream(stream), (const float4 *)g_patches, lookup, H, W, C / 4,
                       (float4 *)gx);
    return mrcnn::check_launch("sparse3x3_scatter");
}

extern "C" int mrcnn_sgd_momentum_wd_ex(float *p, float *g, float *v, int64_t n, float lr,
                                        float momentum, float wd, float grad_scale, int zero_grad,
                                        void *stream)
{
    MRCNN_REQUIRE(n >= 0, "sgd: n < 0");
    if (n == 0) return 0;
    MRCNN_REQUIRE(p && g && v, "sgd: null pointer");
    MRCNN_REQUIRE(aligned16(p) && aligned16(g) && aligned16(v), "sgd: arenas must be 16-byte aligned");
    mrcnn::ProfScope prof(mrcnn::PROF_SGD, 0., (zero_grad ? 24.0 : 20.0) * (double)n,
                          mrcnn::as_stream(stream));
    if (zero_grad)
        hipLaunchKernelGGL(sgd_kernel<true>, dim3(grid_for(n / 4 + 1)), dim3(256), 0,
                           mrcnn::as_stream(stream), p, g, v, n, lr, momentum, wd, grad_scale);
    else
        hipLaunchKernelGGL(sgd_kernel<false>, dim3(grid_for(n / 4 + 1)), dim3(256), 0,
                           mrcnn::as_stream(stream), p, g, v, n, lr, momentum, wd, grad_scale);
    return mrcnn::check_launch("sgd");
}

extern "C" int mrcnn_sgd_momentum_wd(float *p, const float *g, float *v, int64_t n, float lr,
                                     float momentum, float wd, float grad_scale, void *stream)
{
    return mrcnn_sgd_momentum_wd_ex(p, const_cast<float *>(g), v, n, lr, momentum, wd, grad_scale, 0,
                                    stream);
}
