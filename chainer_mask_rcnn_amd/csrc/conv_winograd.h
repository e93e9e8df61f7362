// Winograd F(4x4, 3x3) path for the 3x3 / stride 1 / pad 1 convolutions (fp32) — included at the
// end of conv_gemm.hip, whose GemmParams and launchers it drives.
//
// Replaces, for the layers where it is selected (functions/conv.py: the res5 3x3 convolutions of
// the RoI head — chainer ResNet50Layers.res5 applied per RoI, /root/reference/chainer_mask_rcnn/
// models/mask_rcnn_resnet.py:131-143), the same cuDNN calls as conv_gemm.hip: cuDNN itself
// picks its fp32 Winograd algorithms for exactly these shapes.  On 7x7 maps the direct form
// needs 441 multiply-adds per (RoI, c, k) — 361 with the tap skipping of the position-major
// kernel — and F(4x4,3x3) on the map padded to 8x8 needs 4 tiles x 36 = 144: a third of the
// MFMA work, for three extra HBM-bound passes (data transform, output transform, and the
// per-step filter transform).
//
//   forward      V = B^T d B  (6x6 patches)     M_xi = V_xi U_xi^T      y = A^T M A (+ epilogue)
//   backward-data: the same with g and the flipped, transposed filter
//   backward-filter: dU_xi = Gy_xi^T V_xi  with Gy = G' g G'^T (4x4 tiles of g, F(3x3,4x4)),
//                    gw = A'^T dU A'      (V is the forward's transformed input, kept)
//
// Interpolation points (0, 1, -1, 1/2, -2, inf): B^T and A^T are dyadic (exact in fp32); the
// filter transform G runs in double.  Measured fp32 error against an fp64 direct convolution
// (C = K = 512, 7x7 maps): max 3.4e-6, rms 3.7e-7 of the tensor scale (direct fp32 MFMA:
// 3.5e-7 / 5.7e-8; the textbook points (0, +-1, +-2) give 1.1e-5 / 4.7e-7) — inside the 1e-4
// parity tolerance of BASELINE.json with a margin of 30x; tests/test_gpu_winograd.py holds the
// kernels to the same per-element criterion as the direct kernels.
//
// Layouts (floats): V [36][T][C], U [36][K][C] (forward) / [36][C][K] (backward-data),
// M [36][T][K], T = N * ceil(H/4) * ceil(W/4) tiles, xi = 6 * a + b.
#pragma once

namespace {

constexpr int kXi = 36;

struct F2 { float x, y; };
__device__ __forceinline__ F2 operator+(F2 a, F2 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ F2 operator-(F2 a, F2 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ F2 operator*(float s, F2 a) { return {s * a.x, s * a.y}; }

// t = B^T d
template <typename T>
__device__ __forceinline__ void wino_bt(const T (&d)[6], T (&t)[6])
{
    t[0] = d[0] - 1.5f * d[1] - 2.f * d[2] + 1.5f * d[3] + d[4];
    t[1] = 2.5f * d[3] + 0.5f * d[2] - d[1] + d[4];
    t[2] = d[1] - 2.5f * d[2] + 0.5f * d[3] + d[4];
    t[3] = 2.f * (d[3] - d[1]) - d[2] + d[4];
    t[4] = 0.5f * (d[1] - d[3]) - d[2] + d[4];
    t[5] = d[1] - 1.5f * d[2] - 2.f * d[3] + 1.5f * d[4] + d[5];
}
// y = A^T m
template <typename T>
__device__ __forceinline__ void wino_at(const T (&m)[6], T (&y)[4])
{
    const T s12 = m[1] + m[2], d12 = m[1] - m[2];
    y[0] = m[0] + s12 + m[3] + m[4];
    y[1] = d12 + 0.5f * m[3] - 2.f * m[4];
    y[2] = s12 + 0.25f * m[3] + 4.f * m[4];
    y[3] = d12 + 0.125f * m[3] - 8.f * m[4] + m[5];
}
// |A^T| |m|: bound of the rounding-error propagation through y = A^T m
template <typename T>
__device__ __forceinline__ void wino_at_abs(const T (&m)[6], T (&y)[4])
{
    const T s12 = m[1] + m[2];
    y[0] = m[0] + s12 + m[3] + m[4];
    y[1] = s12 + 0.5f * m[3] + 2.f * m[4];
    y[2] = s12 + 0.25f * m[3] + 4.f * m[4];
    y[3] = s12 + 0.125f * m[3] + 8.f * m[4] + m[5];
}
// t = G' g  (F(3,4) filter-side matrix with its rows scaled to integers; the inverse scales
// 1, 1/3, 1/3, 1/15, 1/15, 1 sit in A'^T, kWgradAT)
template <typename T>
__device__ __forceinline__ void wino_g4(const T (&g)[4], T (&t)[6])
{
    t[0] = g[0];
    t[1] = g[0] + g[1] + g[2] + g[3];
    t[2] = g[1] - g[0] - g[2] + g[3];
    t[3] = -16.f * g[0] - 8.f * g[1] - 4.f * g[2] - 2.f * g[3];
    t[4] = g[0] - 2.f * g[1] + 4.f * g[2] - 8.f * g[3];
    t[5] = g[3];
}

// filter transform G (6x3) and the backward-filter output transform A'^T (3x6), double
__constant__ double kFilterG[6][3] = {{1., 0., 0.},
                                      {1. / 3, 1. / 3, 1. / 3},
                                      {-1. / 3, 1. / 3, -1. / 3},
                                      {-16. / 15, -8. / 15, -4. / 15},
                                      {1. / 15, -2. / 15, 4. / 15},
                                      {0., 0., 1.}};
__constant__ double kWgradAT[3][6] = {{1., 1. / 3, 1. / 3, 1. / 15, 1. / 15, 0.},
                                      {0., 1. / 3, -1. / 3, 1. / 30, -2. / 15, 0.},
                                      {0., 1. / 3, 1. / 3, 1. / 60, 4. / 15, 1.}};

struct WinoGeom {
    int N, H, W, TH, TW;
    int64_t T;
};
inline WinoGeom wino_geom(const mrcnn_conv_desc *d)
{
    WinoGeom g;
    g.N = d->N; g.H = d->H; g.W = d->W;
    g.TH = (d->H + 3) / 4; g.TW = (d->W + 3) / 4;
    g.T = (int64_t)d->N * g.TH * g.TW;
    return g;
}

// ---- data transform: one thread = one tile x two channels ------------------------------------
// GRAD_TILE = false: v = B^T d B over the 6x6 patch at (4ty-1, 4tx-1) (zero outside the map)
// GRAD_TILE = true : v = G' g G'^T over the 4x4 tile at (4ty, 4tx)
template <bool GRAD_TILE>
__global__ void __launch_bounds__(256) wino_data_transform_kernel(
    const float *__restrict__ x, float *__restrict__ v, int H, int W, int C, int TH, int TW,
    int64_t T)
{
    constexpr int D = GRAD_TILE ? 4 : 6;
    const int c2n = C >> 1;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= T * c2n) return;
    const int64_t tile = idx / c2n;
    const int c = (int)(idx - tile * c2n) * 2;
    const int tx = (int)(tile % TW);
    const int64_t q = tile / TW;
    const int ty = (int)(q % TH);
    const int64_t n = q / TH;
    const int y0 = 4 * ty - (GRAD_TILE ? 0 : 1), x0 = 4 * tx - (GRAD_TILE ? 0 : 1);
    const float *base = x + (n * H * W) * C + c;
    F2 d[D][D];
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int yy = y0 + i, xx = x0 + j;
            const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            d[i][j] = ok ? *reinterpret_cast<const F2 *>(base + ((int64_t)yy * W + xx) * C)
                         : F2{0.f, 0.f};
        }
    F2 t[6][D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        F2 col[D], o[6];
#pragma unroll
        for (int i = 0; i < D; ++i) col[i] = d[i][j];
        if constexpr (GRAD_TILE) wino_g4(col, o);
        else wino_bt(col, o);
#pragma unroll
        for (int a = 0; a < 6; ++a) t[a][j] = o[a];
    }
    float *out = v + tile * C + c;
    const int64_t xi_stride = T * C;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        F2 o[6];
        if constexpr (GRAD_TILE) wino_g4(t[a], o);
        else wino_bt(t[a], o);
#pragma unroll
        for (int b = 0; b < 6; ++b)
            *reinterpret_cast<F2 *>(out + (a * 6 + b) * xi_stride) = o[b];
    }
}

// ---- output transform + epilogue: one thread = one tile x two channels ------------------------
//   y = A^T m A;  y = y * scale[k] + shift[k] (AFFINE) or y += shift[k] (BIAS);  relu (RELU);
//   y = mask > 0 ? y : 0
struct WinoOutParams {
    const float *m;
    float *y;
    const float *scale, *shift, *mask;
    int H, W, K, TH, TW, flags;
    int64_t T;
    // FIXUP: outputs whose pre-activation lies within the propagated rounding bound of zero are
    // appended to a list ((pixel, channel) pairs) and recomputed directly by wino_fixup_kernel
    int *fix_count;
    int2 *fix_list;
    int fix_cap;
    float ambiguity;
};
// Relative rounding per stage (input transform, fp32 GEMM over C channels, output transform)
// propagated as |A^T| |M| |A|; the factor is ~8x the largest ratio |y - y_fp64| / (|A^T||M||A|)
// measured over the head / RPN shapes (tests/test_gpu_winograd.py pins it).
float g_wino_ambiguity = 4e-6f;      // mrcnn_set_tuning("wino_ambiguity_ppb", parts per 1e9)

template <bool FIXUP>
__global__ void __launch_bounds__(256) wino_output_transform_kernel(const WinoOutParams p)
{
    const int k2n = p.K >> 1;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.T * k2n) return;
    const int64_t tile = idx / k2n;
    const int k = (int)(idx - tile * k2n) * 2;
    const int tx = (int)(tile % p.TW);
    const int64_t q = tile / p.TW;
    const int ty = (int)(q % p.TH);
    const int64_t n = q / p.TH;
    const float *in = p.m + tile * p.K + k;
    const int64_t xi_stride = p.T * p.K;
    F2 m[6][6];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b)
            m[a][b] = *reinterpret_cast<const F2 *>(in + (a * 6 + b) * xi_stride);
    const int y0 = 4 * ty, x0 = 4 * tx;
    const int64_t obase = (n * p.H * p.W) * p.K + k;
    F2 mk[4][4];
    if (p.mask) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = y0 + i < p.H && x0 + j < p.W;
                mk[i][j] = ok ? *reinterpret_cast<const F2 *>(
                                    p.mask + obase + ((int64_t)(y0 + i) * p.W + x0 + j) * p.K)
                              : F2{0.f, 0.f};
            }
    }
    F2 s[4][6];
    F2 sa[FIXUP ? 4 : 1][6];
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        F2 col[6], o[4];
#pragma unroll
        for (int a = 0; a < 6; ++a) col[a] = m[a][b];
        wino_at(col, o);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i][b] = o[i];
        if constexpr (FIXUP) {
#pragma unroll
            for (int a = 0; a < 6; ++a) col[a] = F2{fabsf(col[a].x), fabsf(col[a].y)};
            wino_at_abs(col, o);
#pragma unroll
            for (int i = 0; i < 4; ++i) sa[i][b] = o[i];
        }
    }
    F2 sc = {1.f, 1.f}, sh = {0.f, 0.f};
    const bool aff = (p.flags & (MRCNN_EPI_AFFINE | MRCNN_EPI_BIAS)) != 0;
    const bool relu = (p.flags & MRCNN_EPI_RELU) != 0;
    if (aff) {
        if (p.scale) sc = *reinterpret_cast<const F2 *>(p.scale + k);
        if (p.shift) sh = *reinterpret_cast<const F2 *>(p.shift + k);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        F2 o[4], ob[4];
        wino_at(s[i], o);
        if constexpr (FIXUP) wino_at_abs(sa[i], ob);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (y0 + i >= p.H || x0 + j >= p.W) continue;
            F2 v = o[j];
            if (aff) { v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; }
            if constexpr (FIXUP) {
                const float bx = p.ambiguity * ob[j].x * fabsf(sc.x);
                const float by = p.ambiguity * ob[j].y * fabsf(sc.y);
                const int pix = (int)(n * p.H * p.W) + (y0 + i) * p.W + x0 + j;
                if (fabsf(v.x) < bx) {
                    const int at = atomicAdd(p.fix_count, 1);
                    if (at < p.fix_cap) p.fix_list[at] = make_int2(pix, k);
                }
                if (fabsf(v.y) < by) {
                    const int at = atomicAdd(p.fix_count, 1);
                    if (at < p.fix_cap) p.fix_list[at] = make_int2(pix, k + 1);
                }
            }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); }
            if (p.mask) {
                v.x = mk[i][j].x > 0.f ? v.x : 0.f;
                v.y = mk[i][j].y > 0.f ? v.y : 0.f;
            }
            *reinterpret_cast<F2 *>(p.y + obase + ((int64_t)(y0 + i) * p.W + x0 + j) * p.K) = v;
        }
    }
}

// ---- direct recomputation of the listed outputs: one wave per (pixel, channel) ------------------
// y = relu?(dot(x patch, w[k]) * scale[k] + shift[k]); lanes stride over channels (16 B each),
// pairwise wave reduction: the rounding of a direct fp32 dot product.
struct WinoFixParams {
    const float *x, *w, *scale, *shift;
    float *y;
    const int *count;
    const int2 *list;
    int cap, H, W, C, K, flags;
};
__global__ void __launch_bounds__(256) wino_fixup_kernel(const WinoFixParams p)
{
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * 256 + threadIdx.x) >> 6), nwaves = (int)(gridDim.x * 4);
    const int n_items = min(*p.count, p.cap);
    for (int it = wave; it < n_items; it += nwaves) {
        const int2 e = p.list[it];
        const int pix = e.x, k = e.y;
        const int n = pix / (p.H * p.W), rem = pix - n * (p.H * p.W);
        const int yy = rem / p.W, xx = rem - yy * p.W;
        float acc = 0.f;
        for (int r = 0; r < 3; ++r) {
            const int iy = yy + r - 1;
            if ((unsigned)iy >= (unsigned)p.H) continue;
            for (int s = 0; s < 3; ++s) {
                const int ix = xx + s - 1;
                if ((unsigned)ix >= (unsigned)p.W) continue;
                const float *xp = p.x + ((int64_t)(n * p.H + iy) * p.W + ix) * p.C;
                const float *wp = p.w + ((int64_t)k * 9 + r * 3 + s) * p.C;
                for (int c = lane * 4; c < p.C; c += 256) {
                    const float4 a = *reinterpret_cast<const float4 *>(xp + c);
                    const float4 b = *reinterpret_cast<const float4 *>(wp + c);
                    acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0) {
            float v = acc;
            if (p.flags & (MRCNN_EPI_AFFINE | MRCNN_EPI_BIAS))
                v = v * (p.scale ? p.scale[k] : 1.f) + (p.shift ? p.shift[k] : 0.f);
            if (p.flags & MRCNN_EPI_RELU) v = fmaxf(v, 0.f);
            p.y[(int64_t)pix * p.K + k] = v;
        }
    }
}

// ---- filter transform: u[xi][k][c] = (G w[k,:,:,c] G^T)[xi]  (KRSC filter) --------------------
// TRANSPOSED (backward-data): u[xi][c][k] = (G flip(w[k,:,:,c]) G^T)[xi] * row_scale[k]
template <bool TRANSPOSED>
__global__ void __launch_bounds__(256) wino_filter_transform_kernel(
    const float *__restrict__ w, float *__restrict__ u, int K, int C,
    const float *__restrict__ row_scale)
{
    __shared__ float tile[32][33];
    const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    double f[4][9];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = k0 + ty + 8 * j, c = c0 + tx;
        const bool ok = k < K && c < C;
        const double rsc = (TRANSPOSED && row_scale && ok) ? (double)row_scale[k] : 1.0;
#pragma unroll
        for (int rs = 0; rs < 9; ++rs) {
            const int src = TRANSPOSED ? 8 - rs : rs;
            f[j][rs] = ok ? (double)w[((int64_t)k * 9 + src) * C + c] * rsc : 0.0;
        }
    }
    for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) {
            float val[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double acc = 0.0;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double row = kFilterG[b][0] * f[j][r * 3] + kFilterG[b][1] * f[j][r * 3 + 1] +
                                       kFilterG[b][2] * f[j][r * 3 + 2];
                    acc += kFilterG[a][r] * row;
                }
                val[j] = (float)acc;
            }
            float *dst = u + (int64_t)(a * 6 + b) * K * C;
            if constexpr (!TRANSPOSED) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = k0 + ty + 8 * j, c = c0 + tx;
                    if (k < K && c < C) dst[(int64_t)k * C + c] = val[j];
                }
            } else {
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 4; ++j) tile[ty + 8 * j][tx] = val[j];
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = c0 + ty + 8 * j, k = k0 + tx;
                    if (c < C && k < K) dst[(int64_t)c * K + k] = tile[tx][ty + 8 * j];
                }
            }
        }
}

// ---- backward-filter finish: ordered slab sum, gw[k][r][s][c] = (A'^T dU[:, k, c] A')[r][s] -----
__global__ void __launch_bounds__(256) wino_wgrad_finish_kernel(
    const float *__restrict__ slabs, int splits, int64_t split_stride, float *__restrict__ gw,
    const float *__restrict__ row_scale, int K, int C)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)K * C) return;
    const int k = (int)(idx / C), c = (int)(idx - (int64_t)k * C);
    const int64_t kc = (int64_t)K * C;
    float du[kXi];
#pragma unroll
    for (int xi = 0; xi < kXi; ++xi) du[xi] = slabs[xi * kc + idx];
    for (int s = 1; s < splits; ++s) {       // slab order: deterministic
        const float *sl = slabs + s * split_stride;
#pragma unroll
        for (int xi = 0; xi < kXi; ++xi) du[xi] += sl[xi * kc + idx];
    }
    const double rsc = row_scale ? (double)row_scale[k] : 1.0;
    double t[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            double acc = 0.0;
#pragma unroll
            for (int a = 0; a < 6; ++a) acc += kWgradAT[r][a] * (double)du[a * 6 + b];
            t[r][b] = acc;
        }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            double acc = 0.0;
#pragma unroll
            for (int b = 0; b < 6; ++b) acc += kWgradAT[s][b] * t[r][b];
            gw[((int64_t)k * 9 + r * 3 + s) * C + c] = (float)(acc * rsc);
        }
}

int wino_check(const mrcnn_conv_desc *d, const char *who)
{
    if (int rc = check_desc(d)) return rc;
    MRCNN_REQUIRE(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1,
                  "%s: the Winograd path serves 3x3 / stride 1 / pad 1 only", who);
    MRCNN_REQUIRE(d->C % 4 == 0 && d->K % 4 == 0, "%s: channels must be multiples of 4", who);
    const WinoGeom g = wino_geom(d);
    MRCNN_REQUIRE(g.T * std::max(d->C, d->K) < ((int64_t)1 << 29),
                  "%s: one frequency plane exceeds 2 GiB; split the batch", who);
    return 0;
}

inline int64_t wino_u_floats(const mrcnn_conv_desc *d) { return (int64_t)kXi * d->K * d->C; }
inline int64_t wino_plane_floats(const mrcnn_conv_desc *d, int ch)
{
    return (int64_t)kXi * wino_geom(d).T * ch;
}
constexpr int kWinoMaxSplits = 8;

template <bool GRAD_TILE>
void wino_launch_data_transform(const float *x, float *v, const WinoGeom &g, int C, hipStream_t s)
{
    const int64_t n = g.T * (C / 2);
    const double in_px = (double)g.N * g.H * g.W, tiles = (double)g.T;
    mrcnn::ProfScope prof(mrcnn::PROF_WINO_TRANSFORM, 0., 4.0 * C * (in_px + 36.0 * tiles), s);
    hipLaunchKernelGGL((wino_data_transform_kernel<GRAD_TILE>),
                       dim3((unsigned)mrcnn::ceil_div(n, 256)), dim3(256), 0, s, x, v, g.H, g.W, C,
                       g.TH, g.TW, g.T);
}

// the 36 per-frequency GEMMs  out[xi] (T x N) = a[xi] (T x Kc) . b[xi]^T (N x Kc)
int wino_batched_gemm(const float *a, const float *b, float *out, int64_t T, int N, int Kc,
                      hipStream_t s)
{
    GemmParams p = {};
    p.A = a; p.B = b; p.C = out;
    p.M = (int)T; p.N = N; p.Kc = Kc;
    p.gp = p.gq = p.sh = p.sw = 1;
    p.R = p.S = 1; p.stride = 1; p.pad = 0;
    p.lda = Kc; p.ldb = Kc; p.ldc = N;
    p.out_mode = OUT_PLAIN;
    p.batch_a = T * Kc; p.batch_b = (int64_t)N * Kc; p.batch_c = T * N;
    if (int rc = set_extents(p, T * Kc, (int64_t)N * Kc, T * N)) return rc;
    const bool big = p.M > 64 && p.N > 64;
    const int64_t tiles = big ? mrcnn::ceil_div(p.M, 128) * mrcnn::ceil_div(p.N, 128)
                              : mrcnn::ceil_div(p.M, 64) * mrcnn::ceil_div(p.N, 64);
    const double flops = 2.0 * kXi * (double)T * N * Kc;
    const double bytes = 4.0 * kXi * ((double)T * N + (double)T * Kc + (double)N * Kc);
    const int rem = p.M % 128;
    const bool two_launches = big && rem > 0 && rem <= 64 && p.M >= 256;
    const bool w8 = big && w8_ok(p, kXi);
    mrcnn::ProfKernelScope prof(w8 ? mrcnn::PROF_CONV_FWD_W8 : big ? mrcnn::PROF_CONV_FWD_128 : mrcnn::PROF_CONV_FWD_64,
                                flops, bytes, two_launches && !w8 ? 2 : 1);
    if (w8) {
        // 256 x 128 tiles on 512-thread workgroups (conv_gemm.hip, W8): the 36 problems in one grid
        launch_w8_kernel(p, mrcnn::ceil_div(p.M, kW8BM) * mrcnn::ceil_div(p.N, kW8BN), kXi, s);
    } else if (two_launches) {
        // a last row tile that is at most half full (the RPN's 546 tiles of a 2 x 51 x 84 map:
        // 4 x 128 + 34) runs as 64-row tiles instead of a whole 128-row tile of mostly padding
        const int full = p.M - rem;
        GemmParams q = p;
        q.M = full;
        launch_kernel<2, 2, FWD>(q, (int64_t)(full / 128) * mrcnn::ceil_div(p.N, 128), 1, s, kXi);
        q.M = p.M;
        q.m_lo = full;
        launch_kernel<1, 1, FWD>(q, mrcnn::ceil_div(p.N, 64), 1, s, kXi);
    } else if (big) {
        launch_kernel<2, 2, FWD>(p, tiles, 1, s, kXi);
    } else {
        launch_kernel<1, 1, FWD>(p, tiles, 1, s, kXi);
    }
    return mrcnn::check_launch("wino_batched_gemm");
}

}  // namespace

// transformed-input bytes (the forward writes it, the backward-filter reads it)
extern "C" int64_t mrcnn_conv3x3_wino_v_bytes(const mrcnn_conv_desc *d)
{
    return d ? 4 * wino_plane_floats(d, d->C) : 0;
}

// scratch for any of the three passes
extern "C" int64_t mrcnn_conv3x3_wino_workspace_bytes(const mrcnn_conv_desc *d)
{
    if (!d) return 0;
    const int64_t ch = std::max(d->C, d->K);
    return 4 * ((1 + kWinoMaxSplits) * wino_u_floats(d) + 2 * wino_plane_floats(d, (int)ch));
}

// u_bytes: size of a transformed filter (36 x K x C floats)
extern "C" int64_t mrcnn_conv3x3_wino_u_bytes(const mrcnn_conv_desc *d)
{
    return d ? 4 * wino_u_floats(d) : 0;
}

// u (36, K, C) = G w G^T of a KRSC filter — what mrcnn_conv3x3_wino_fwd builds per call when it
// is not handed one (inference: the filter does not change between calls)
extern "C" int mrcnn_conv3x3_wino_filter(const mrcnn_conv_desc *d, const float *w, float *u,
                                         void *stream)
{
    if (int rc = wino_check(d, "conv3x3_wino_filter")) return rc;
    MRCNN_REQUIRE(w && u, "conv3x3_wino_filter: null pointer");
    hipLaunchKernelGGL((wino_filter_transform_kernel<false>),
                       dim3((d->C + 31) / 32, (d->K + 31) / 32), dim3(256), 0,
                       mrcnn::as_stream(stream), w, u, d->K, d->C, (const float *)nullptr);
    return mrcnn::check_launch("conv3x3_wino_filter");
}

extern "C" int mrcnn_conv3x3_wino_fwd(const mrcnn_conv_desc *d, const float *x, const float *w,
                                      const float *u_pre, const float *scale, const float *shift,
                                      float *y, int epi_flags, float *v, void *ws, void *stream)
{
    if (int rc = wino_check(d, "conv3x3_wino_fwd")) return rc;
    MRCNN_REQUIRE(x && (w || u_pre) && y && ws, "conv3x3_wino_fwd: null pointer");
    MRCNN_REQUIRE((epi_flags & ~(MRCNN_EPI_AFFINE | MRCNN_EPI_BIAS | MRCNN_EPI_RELU |
                                 MRCNN_EPI_EXACT_SIGNS)) == 0,
                  "conv3x3_wino_fwd: only AFFINE or BIAS, RELU and EXACT_SIGNS");
    MRCNN_REQUIRE(!(epi_flags & MRCNN_EPI_EXACT_SIGNS) || w,
                  "conv3x3_wino_fwd: EXACT_SIGNS needs the filter itself");
    MRCNN_REQUIRE(!(epi_flags & MRCNN_EPI_AFFINE) || scale, "conv3x3_wino_fwd: affine flag without scale");
    MRCNN_REQUIRE(!(epi_flags & MRCNN_EPI_BIAS) || (shift && !scale && !(epi_flags & MRCNN_EPI_AFFINE)),
                  "conv3x3_wino_fwd: the bias flag takes the bias in `shift` and excludes AFFINE");
    hipStream_t s = mrcnn::as_stream(stream);
    const WinoGeom g = wino_geom(d);
    float *u_ws = (float *)ws;
    float *m = u_ws + wino_u_floats(d);
    float *vbuf = v ? v : m + wino_plane_floats(d, d->K);
    const float *u = u_pre;
    if (!u) {
        hipLaunchKernelGGL((wino_filter_transform_kernel<false>),
                           dim3((d->C + 31) / 32, (d->K + 31) / 32), dim3(256), 0, s, w, u_ws, d->K,
                           d->C, (const float *)nullptr);
        u = u_ws;
    }
    wino_launch_data_transform<false>(x, vbuf, g, d->C, s);
    if (int rc = wino_batched_gemm(vbuf, u, m, g.T, d->K, d->C, s)) return rc;
    WinoOutParams o = {};
    o.m = m; o.y = y; o.scale = scale; o.shift = shift;
    o.H = d->H; o.W = d->W; o.K = d->K; o.TH = g.TH; o.TW = g.TW;
    o.flags = epi_flags & ~MRCNN_EPI_EXACT_SIGNS; o.T = g.T;
    const bool fixup = (epi_flags & MRCNN_EPI_EXACT_SIGNS) != 0;
    if (fixup) {
        // behind everything the forward uses: [count (16 B) | list]
        float *fix = (float *)ws + wino_u_floats(d) + 2 * wino_plane_floats(d, std::max(d->C, d->K));
        o.fix_count = (int *)fix;
        o.fix_list = (int2 *)(fix + 4);
        o.fix_cap = (int)std::min<int64_t>(1 << 20, (8 * wino_u_floats(d) - 4) / 2);
        o.ambiguity = g_wino_ambiguity;
        MRCNN_HIP_TRY(hipMemsetAsync(o.fix_count, 0, 16, s));
    }
    {
        mrcnn::ProfScope prof(mrcnn::PROF_WINO_TRANSFORM, 0.,
                              4.0 * d->K * (36.0 * g.T + (double)d->N * d->H * d->W), s);
        const dim3 grid((unsigned)mrcnn::ceil_div(g.T * (d->K / 2), 256));
        if (fixup) hipLaunchKernelGGL(wino_output_transform_kernel<true>, grid, dim3(256), 0, s, o);
        else hipLaunchKernelGGL(wino_output_transform_kernel<false>, grid, dim3(256), 0, s, o);
    }
    if (fixup) {
        WinoFixParams f = {};
        f.x = x; f.w = w; f.scale = scale; f.shift = shift; f.y = y;
        f.count = o.fix_count; f.list = o.fix_list; f.cap = o.fix_cap;
        f.H = d->H; f.W = d->W; f.C = d->C; f.K = d->K; f.flags = o.flags;
        hipLaunchKernelGGL(wino_fixup_kernel, dim3(1024), dim3(256), 0, s, f);
    }
    return mrcnn::check_launch("conv3x3_wino_fwd");
}

// gx = (dgrad(gy) * out_scale[c]) masked by (out_mask_y > 0); w_row_scale[k] multiplies gy's
// channels (folded into the transformed filter)
extern "C" int mrcnn_conv3x3_wino_dgrad(const mrcnn_conv_desc *d, const float *gy, const float *w,
                                        const float *w_row_scale, float *gx,
                                        const float *out_scale, const float *out_mask_y, void *ws,
                                        void *stream)
{
    if (int rc = wino_check(d, "conv3x3_wino_dgrad")) return rc;
    MRCNN_REQUIRE(gy && w && gx && ws, "conv3x3_wino_dgrad: null pointer");
    hipStream_t s = mrcnn::as_stream(stream);
    const WinoGeom g = wino_geom(d);
    const int64_t ch = std::max(d->C, d->K);
    float *u = (float *)ws;
    float *gt = u + (1 + kWinoMaxSplits) * wino_u_floats(d);
    float *m = gt + wino_plane_floats(d, (int)ch);
    hipLaunchKernelGGL((wino_filter_transform_kernel<true>),
                       dim3((d->C + 31) / 32, (d->K + 31) / 32), dim3(256), 0, s, w, u, d->K, d->C,
                       w_row_scale);
    wino_launch_data_transform<false>(gy, gt, g, d->K, s);
    if (int rc = wino_batched_gemm(gt, u, m, g.T, d->C, d->K, s)) return rc;
    WinoOutParams o = {};
    o.m = m; o.y = gx; o.scale = out_scale; o.mask = out_mask_y;
    o.H = d->H; o.W = d->W; o.K = d->C; o.TH = g.TH; o.TW = g.TW;
    o.flags = out_scale ? MRCNN_EPI_AFFINE : 0; o.T = g.T;
    {
        mrcnn::ProfScope prof(mrcnn::PROF_WINO_TRANSFORM, 0.,
                              4.0 * d->C * (36.0 * g.T + (out_mask_y ? 2.0 : 1.0) * d->N * d->H * d->W), s);
        hipLaunchKernelGGL(wino_output_transform_kernel<false>,
                           dim3((unsigned)mrcnn::ceil_div(g.T * (d->C / 2), 256)), dim3(256), 0, s, o);
    }
    return mrcnn::check_launch("conv3x3_wino_dgrad");
}

// gw (K,3,3,C) = backward-filter from gy and EITHER the forward's kept transformed input v OR the
// raw input x (transformed here into the scratch); out_row_scale[k] multiplies gw's rows
extern "C" int mrcnn_conv3x3_wino_wgrad(const mrcnn_conv_desc *d, const float *x, const float *v,
                                        const float *gy, float *gw, const float *out_row_scale,
                                        void *ws, void *stream)
{
    if (int rc = wino_check(d, "conv3x3_wino_wgrad")) return rc;
    MRCNN_REQUIRE((x != nullptr) != (v != nullptr), "conv3x3_wino_wgrad: exactly one of x / v");
    MRCNN_REQUIRE(gy && gw && ws, "conv3x3_wino_wgrad: null pointer");
    hipStream_t s = mrcnn::as_stream(stream);
    const WinoGeom g = wino_geom(d);
    float *slabs = (float *)ws + wino_u_floats(d);
    float *gt = (float *)ws + (1 + kWinoMaxSplits) * wino_u_floats(d);
    if (!v) {
        float *vx = gt + wino_plane_floats(d, std::max(d->C, d->K));
        wino_launch_data_transform<false>(x, vx, g, d->C, s);
        v = vx;
    }
    wino_launch_data_transform<true>(gy, gt, g, d->K, s);
    GemmParams p = {};
    p.A = gt; p.B = v;
    p.M = d->K; p.N = d->C; p.Kc = (int)g.T;
    p.gp = p.gq = p.sh = p.sw = 1;
    p.R = p.S = 1; p.stride = 1; p.pad = 0;
    p.lda = d->C; p.ldg = d->K; p.cin = d->C; p.ldc = d->C;
    const int64_t kc = (int64_t)d->K * d->C;
    const bool big = p.M > 64 && p.N > 64;
    const int64_t tiles = big ? mrcnn::ceil_div(p.M, 128) * mrcnn::ceil_div(p.N, 128)
                              : mrcnn::ceil_div(p.M, 64) * mrcnn::ceil_div(p.N, 64);
    const int64_t slots = big ? (single_buffered(2, WGRAD, false) ? 768 : kSlotsBig) : kSlotsSmall;
    int splits = std::min(kWinoMaxSplits, wgrad_splits(tiles * kXi, g.T, slots));
    p.split_len = (int)(mrcnn::ceil_div(mrcnn::ceil_div(g.T, splits), BK) * BK);
    splits = (int)mrcnn::ceil_div(g.T, p.split_len);
    p.split_stride = kXi * kc;
    p.batch_a = g.T * d->K; p.batch_b = g.T * d->C; p.batch_c = kc;
    if (int rc = set_extents(p, g.T * d->K, g.T * d->C, kc)) return rc;
    p.C = slabs;
    {
        mrcnn::ProfKernelScope prof(big ? mrcnn::PROF_CONV_WGRAD_128 : mrcnn::PROF_CONV_WGRAD_64,
                                    2.0 * kXi * (double)kc * (double)g.T,
                                    4.0 * kXi * ((double)kc * splits + (double)g.T * (d->K + d->C)));
        if (big) launch_kernel<2, 2, WGRAD>(p, tiles, splits, s, kXi);
        else launch_kernel<1, 1, WGRAD>(p, tiles, splits, s, kXi);
    }
    hipLaunchKernelGGL(wino_wgrad_finish_kernel, dim3((unsigned)mrcnn::ceil_div(kc, 256)), dim3(256),
                       0, s, (const float *)slabs, splits, p.split_stride, gw, out_row_scale, d->K,
                       d->C);
    return mrcnn::check_launch("conv3x3_wino_wgrad");
}

// number of outputs the last EXACT_SIGNS forward on this workspace recomputed (test / diagnostics;
// synchronises the stream)
extern "C" int mrcnn_conv3x3_wino_fixup_count(const mrcnn_conv_desc *d, const void *ws, void *stream,
                                              int *count)
{
    MRCNN_REQUIRE(d && ws && count, "conv3x3_wino_fixup_count: null pointer");
    const float *fix = (const float *)ws + wino_u_floats(d) +
                       2 * wino_plane_floats(d, std::max(d->C, d->K));
    MRCNN_HIP_TRY(hipMemcpyAsync(count, fix, sizeof(int), hipMemcpyDeviceToHost, mrcnn::as_stream(stream)));
    MRCNN_HIP_TRY(hipStreamSynchronize(mrcnn::as_stream(stream)));
    return 0;
}
