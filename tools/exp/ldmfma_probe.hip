// Developer experiment (not product code): do 16-byte-per-lane global loads and bf16 MFMAs of the
// same wave / the same CU overlap, or do their times add up?  A barrier-free, LDS-free loop:
// per iteration NL buffer loads (consumed at the end of the iteration, as a GEMM's staging
// registers are) and NM independent-accumulator MFMAs.
//   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ldmfma tools/exp/ldmfma_probe.hip ; run: /tmp/ldmfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NL, int NM, bool SPREAD, int ND = 0, int NW = 0>
__global__ void __launch_bounds__(256, 2) probe(const char *buf, unsigned bytes, float *out, int iters, unsigned rs)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(buf), 0, bytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    // a wave reads 8 rows of 128 bytes per load (8 lanes x 16 B per row), rows 8 KB apart, like a GEMM tile
    unsigned base = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (8u * rs) + (lane >> 3) * rs + (lane & 7) * 16u;
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    unsigned sink = 0;
    __shared__ __attribute__((aligned(16))) unsigned lds[12288];      // 48 KB like the GEMM's stage
    for (int i = threadIdx.x; i < 12288; i += 256) lds[i] = i;
    __syncthreads();
    const unsigned rd_off = (threadIdx.x & 63) * 20u % 3000u * 4u;   // 80-byte rows, 16-byte aligned
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    for (int it = 0; it < iters; ++it) {
        u32x4 v[NL > 0 ? NL : 1];
        if constexpr (!SPREAD) {
#pragma unroll
            for (int i = 0; i < NL; ++i)
                v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (base + (unsigned)(it * NL + i) * 128u) % bytes, 0, 0);
#pragma unroll
            for (int m = 0; m < NM; ++m)
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
        } else {
            constexpr int G = NL > 0 ? NL : 1;
#pragma unroll
            for (int i = 0; i < G; ++i) {
                if (NL > 0)
                    v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (base + (unsigned)(it * NL + i) * 128u) % bytes, 0, 0);
#pragma unroll
                for (int m = NM * i / G; m < NM * (i + 1) / G; ++m)
                    acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
                for (int dd = ND * i / G; dd < ND * (i + 1) / G; ++dd) {
                    const u32x4 t = *reinterpret_cast<const u32x4 *>(&lds[rd_off + ((dd * 256 + it * 16) & 8191)]);
                    asm volatile("" ::"v"(t));
                    sink ^= t.y;
                }
#pragma unroll
                for (int ww = NW * i / G; ww < NW * (i + 1) / G; ++ww) {
                    u32x2 t2; t2.x = sink; t2.y = (unsigned)ww;
                    *reinterpret_cast<u32x2 *>(&lds[(threadIdx.x * 2 + ww * 512) & 8191]) = t2;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) { asm volatile("" ::"v"(v[i])); sink ^= v[i].x ^ v[i].w; }
    }
    float s = (float)sink;
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NL, int NM, bool SPREAD, int ND = 0, int NW = 0>
void run(const char *buf, unsigned bytes, float *out, const char *what)
{
    const int iters = 400, blocks = 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<NL, NM, SPREAD, ND, NW>), dim3(blocks), dim3(256), 0, 0, buf, bytes, out, iters, bytes > (64u << 20) ? 131072u : 8192u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // per CU: 2 workgroups x 4 waves; cycles per iteration at 2.4 GHz
    const double cyc = ms * 1e-3 * 2.4e9 / iters;
    printf("%-10s NL=%2d NM=%2d ND=%2d NW=%2d %s: %.3f ms, %6.0f cycles per iteration (MFMA alone would be %d, loads at 64 B/clk/CU %d)\n",
           what, NL, NM, ND, NW, SPREAD ? "spread" : "burst ", ms, cyc, NM * 32 * 2, NL * 8 * 16);
}

int main()
{
    char *buf; float *out;
    const unsigned small = 2u << 20, big = 24u << 20;
    hipMalloc(&buf, 1u << 30); hipMemset(buf, 1, 1u << 30);
    hipMalloc(&out, 512 * 256 * 4);
    for (int pass = 0; pass < 2; ++pass) {
        const unsigned bytes = pass == 0 ? small : big;
        const char *what = pass == 0 ? "2MB(L2)" : "24MB(MALL)";
        run<0, 48, false>(buf, bytes, out, what);
        run<8, 0, false>(buf, bytes, out, what);
        run<8, 48, false>(buf, bytes, out, what);
        run<8, 48, true>(buf, bytes, out, what);
        run<12, 48, true>(buf, bytes, out, what);
        run<4, 48, true>(buf, bytes, out, what);
        run<16, 0, false>(buf, bytes, out, what);
        run<0, 48, true, 24, 0>(buf, bytes, out, what);
        run<8, 48, true, 24, 0>(buf, bytes, out, what);
        run<0, 48, true, 24, 24>(buf, bytes, out, what);
        run<8, 48, true, 24, 24>(buf, bytes, out, what);
        run<8, 48, true, 0, 24>(buf, bytes, out, what);
    }
    return 0;
}
