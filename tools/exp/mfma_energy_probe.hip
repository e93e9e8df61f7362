// Developer experiment (not product code): matrix-pipe throughput AT THE POWER CAP for the product
// mixes an fp32-equivalent multiply-add can be built from — random operands read from LDS, no global
// traffic, 2 workgroups of 4 waves per CU, ~3 s per variant so that the package power settles:
//   bf16x3 : six  v_mfma_f32_32x32x16_bf16 per 16-deep K step (the shipped split arithmetic)
//   i8x4   : ten  v_mfma_i32_32x32x32_i8   per 32-deep K step (four 7-bit slices per operand, i+j<=3)
//   f32    : eight v_mfma_f32_32x32x2_f32  per 16-deep K step (fp32 MFMA)
// Reported: fp32-equivalent TFLOP/s (2 * 32*32*K per accumulator tile and K step).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/mfma_energy_probe tools/exp/mfma_energy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// per iteration: every wave owns 2x2 accumulator tiles and walks a 32-deep K slice
// STAGE bits (bf16x3 only): 1 = 8 global loads per slice (L2-resident), 2 = split3 + 24 plane writes,
// 4 = the kernel's two workgroup barriers per slice
template <int KIND, int STAGE = 0>
__global__ void __launch_bounds__(256, 2) probe(const unsigned *rnd, float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned lds[12288];
    for (int i = threadIdx.x; i < 12288; i += 256) lds[i] = rnd[(blockIdx.x * 12288 + i) & ((1 << 22) - 1)];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned *my = lds + wave * 3072;
    f32x16 acc[2][2];
    i32x16 iacc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; iacc[i][j][e] = 0; }
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(rnd), 0, 2u << 20, 0x00020000);
    const unsigned gbase = (blockIdx.x * 4 + wave) * 65536u + (lane >> 3) * 8192u + (lane & 7) * 16u;
    u32x4 v[8];
    for (int i = 0; i < 8; ++i) v[i] = (u32x4){lds[lane + i], lds[lane + 64 + i], lds[lane + 128 + i], lds[lane + 192 + i]};
    unsigned *wr = lds + 8192 + wave * 1024;        // (writes go to a region the fragment reads do not touch)
    for (int it = 0; it < iters; ++it) {
        const unsigned o = (unsigned)(it * 64) & 1023u;
        if constexpr ((STAGE & 2) != 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float a0 = __uint_as_float(v[i].x), a1 = __uint_as_float(v[i].y), a2 = __uint_as_float(v[i].z), a3 = __uint_as_float(v[i].w);
                unsigned w[6];
                {   // split3 of two pairs
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    auto pk = [](float p, float q) { const f32x2 t = {p, q}; return __builtin_bit_cast(unsigned, __builtin_convertvector(t, bf16x2)); };
                    w[0] = pk(a0, a1); a0 -= __uint_as_float(w[0] << 16); a1 -= __uint_as_float(w[0] & 0xffff0000u);
                    w[1] = pk(a0, a1); a0 -= __uint_as_float(w[1] << 16); a1 -= __uint_as_float(w[1] & 0xffff0000u);
                    w[2] = pk(a0, a1);
                    w[3] = pk(a2, a3); a2 -= __uint_as_float(w[3] << 16); a3 -= __uint_as_float(w[3] & 0xffff0000u);
                    w[4] = pk(a2, a3); a2 -= __uint_as_float(w[4] << 16); a3 -= __uint_as_float(w[4] & 0xffff0000u);
                    w[5] = pk(a2, a3);
                }
                *reinterpret_cast<uint2 *>(&wr[(lane * 2 + i * 128) & 1023]) = make_uint2(w[0], w[3]);
                *reinterpret_cast<uint2 *>(&wr[(lane * 2 + i * 128 + 256) & 1023]) = make_uint2(w[1], w[4]);
                *reinterpret_cast<uint2 *>(&wr[(lane * 2 + i * 128 + 512) & 1023]) = make_uint2(w[2], w[5]);
            }
        }
        if constexpr ((STAGE & 4) != 0) __syncthreads();
        if constexpr ((STAGE & 1) != 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (gbase + (unsigned)(it * 8 + i) * 128u) & ((2u << 20) - 1), 0, 0);
        }
        if constexpr (KIND == 0) {            // bf16x3: 2 K steps x (2+2) tiles x 3 planes of fragments
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[2][3], fb[2][3];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        fa[t][q] = *reinterpret_cast<const bf16x8 *>(&my[((lane * 4 + (ks * 6 + t * 3 + q) * 256) + o) & 2047]);
                        fb[t][q] = *reinterpret_cast<const bf16x8 *>(&my[((lane * 4 + (ks * 6 + t * 3 + q) * 256 + 128) + o) & 2047] + 1024 - 1024);
                    }
                constexpr int QA[6] = {0, 2, 1, 1, 0, 0}, QB[6] = {2, 0, 1, 0, 1, 0};
#pragma unroll
                for (int c = 0; c < 6; ++c)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][QA[c]], fb[j][QB[c]], acc[i][j], 0, 0, 0);
            }
        } else if constexpr (KIND == 1) {     // i8x4: one 32-deep K step, (2+2) tiles x 4 slices
            i32x4 fa[2][4], fb[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    fa[t][q] = *reinterpret_cast<const i32x4 *>(&my[((lane * 4 + (t * 4 + q) * 256) + o) & 2047]);
                    fb[t][q] = *reinterpret_cast<const i32x4 *>(&my[((lane * 4 + (t * 4 + q) * 256 + 128) + o) & 2047]);
                }
#pragma unroll
            for (int qa = 0; qa < 4; ++qa)
#pragma unroll
                for (int qb = 0; qb + qa < 4; ++qb)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            iacc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i][qa], fb[j][qb], iacc[i][j], 0, 0, 0);
        } else {                               // fp32 MFMA: 16 x (K = 2) per tile and 32-deep slice
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                float fa[2][4], fb[2][4];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float4 va = *reinterpret_cast<const float4 *>(&my[((lane * 4 + (kb * 2 + t) * 256) + o) & 2047]);
                    const float4 vb = *reinterpret_cast<const float4 *>(&my[((lane * 4 + (kb * 2 + t) * 256 + 128) + o) & 2047]);
                    fa[t][0] = va.x; fa[t][1] = va.y; fa[t][2] = va.z; fa[t][3] = va.w;
                    fb[t][0] = vb.x; fb[t][1] = vb.y; fb[t][2] = vb.z; fb[t][3] = vb.w;
                }
#pragma unroll
                for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t4], fb[j][t4], acc[i][j], 0, 0, 0);
            }
        }
        if constexpr ((STAGE & 4) != 0) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += (float)v[i].x;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int e = 0; e < 16; ++e) s += acc[i][j][e] + (float)iacc[i][j][e];
    if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int STAGE = 0>
void run(const unsigned *rnd, float *out, const char *what, int kdepth_per_iter)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 512;
    int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {       // the last repetition (~1 s) is the measurement
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<KIND, STAGE>), dim3(blocks), dim3(256), 0, 0, rnd, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 0) iters = (int)(iters * 1200.0 / ms);
        if (rep == 2) {
            // per iteration and wave: 4 tiles x 32x32 x kdepth MACs
            const double flops = 2.0 * 4 * 32 * 32 * kdepth_per_iter * (double)iters * blocks * 4;
            printf("%-7s %8.1f fp32-equivalent TFLOP/s  (%.0f ms, %s)\n", what, flops / ms / 1e9, ms,
                   hipGetErrorString(hipGetLastError()));
            fflush(stdout);
        }
    }
}

int main(int argc, char **argv)
{
    unsigned *rnd; float *out;
    const size_t n = 1 << 22;
    unsigned *h = (unsigned *)malloc(n * 4);
    srand(1);
    const bool zeros = argc > 1 && argv[1][0] == 'z';
    for (size_t i = 0; i < n; ++i) {
        // bf16 pairs / fp32 words with sane exponents: random mantissas, exponent bits around 1.0
        unsigned m = (unsigned)rand() ^ ((unsigned)rand() << 11);
        h[i] = zeros ? 0u : ((m & 0x807F807Fu) | 0x3F003F00u | ((m >> 3) & 0x00800080u));
    }
    hipMalloc(&rnd, n * 4); hipMemcpy(rnd, h, n * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 512 * 256 * 4);
    const char *only = argc > 2 ? argv[2] : "";
    if (!only[0] || only[0] == 'b') run<0>(rnd, out, "bf16x3", 32);
    if (only[0] == 's') {      // staging components beside the six-product MFMAs
        run<0, 0>(rnd, out, "mfma", 32);
        run<0, 1>(rnd, out, "+loads", 32);
        run<0, 2>(rnd, out, "+split", 32);
        run<0, 4>(rnd, out, "+barr", 32);
        run<0, 3>(rnd, out, "+ld+sp", 32);
        run<0, 7>(rnd, out, "+all", 32);
    }
    if (!only[0] || only[0] == 'i') run<1>(rnd, out, "i8x4", 32);
    if (!only[0] || only[0] == 'f') run<2>(rnd, out, "f32", 32);
    return 0;
}
