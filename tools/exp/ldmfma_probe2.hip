// Developer experiment (not product code): the split forward-form GEMM's slice loop rebuilt from
// its parts, to see which part turns the free-running 5560 cycles per CU round of
// ldmfma_probe.hip into the kernel's 7500 - 8700: per iteration and wave 8 global loads consumed by
// the NEXT iteration's staging (split into three bf16 planes or not, 24 ds_write_b64), 24
// ds_read_b128 and 48 bf16 MFMAs, with or without the two workgroup barriers.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/ldmfma_probe2 tools/exp/ldmfma_probe2.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float sub_f32(float a, float b)
{
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split3(float a, float b, unsigned &h, unsigned &m, unsigned &l)
{
    h = pack_bf16(a, b);
    a = sub_f32(a, __uint_as_float(h << 16));
    b = sub_f32(b, __uint_as_float(h & 0xffff0000u));
    m = pack_bf16(a, b);
    a = sub_f32(a, __uint_as_float(m << 16));
    b = sub_f32(b, __uint_as_float(m & 0xffff0000u));
    l = pack_bf16(a, b);
}

// BAR: 0 none, 1 the kernel's two barriers per slice;  CONV: split3 + plane writes, else raw writes
// WAVES: workgroup = WAVES waves (4 = the kernel's), LOADS in {0, 8}
template <int BAR, bool CONV, int LOADS, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 8 / WAVES) probe(const char *buf, unsigned bytes, float *out, int iters)
{
    constexpr int NL = 8, NM = 48, ND = 24;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(buf), 0, bytes, 0x00020000);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned base = (blockIdx.x * WAVES + wave) * 65536u + (lane >> 3) * 8192u + (lane & 7) * 16u;
    __shared__ __attribute__((aligned(16))) unsigned lds[WAVES * 3072];      // 12 KB per wave
    unsigned *my = lds + wave * 3072;
    for (int i = lane; i < 3072; i += 64) my[i] = i;
    __syncthreads();
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(lane + e); b[e] = (__bf16)(float)(lane - e); }
    unsigned sink = 0;
    u32x4 v[NL];
    for (int i = 0; i < NL; ++i) v[i] = (u32x4){(unsigned)lane, 1u, 2u, 3u};
    const unsigned rd = (lane * 20u % 700u) * 4u;
    for (int it = 0; it < iters; ++it) {
        // ---- staging phase: the loads of the previous iteration -> LDS
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            unsigned w0, w1, w2, w3, w4, w5;
            if (CONV) {
                split3(__uint_as_float(v[i].x), __uint_as_float(v[i].y), w0, w1, w2);
                split3(__uint_as_float(v[i].z), __uint_as_float(v[i].w), w3, w4, w5);
            } else {
                w0 = v[i].x; w1 = v[i].y; w2 = v[i].z; w3 = v[i].w; w4 = v[i].x ^ 1u; w5 = v[i].y ^ 1u;
            }
            *reinterpret_cast<uint2 *>(&my[(lane * 2 + i * 128) & 2047]) = make_uint2(w0, w3);
            *reinterpret_cast<uint2 *>(&my[(lane * 2 + i * 128 + 1024) & 2047]) = make_uint2(w1, w4);
            *reinterpret_cast<uint2 *>(&my[(lane * 2 + i * 128 + 512) & 2047 + 1024]) = make_uint2(w2, w5);
        }
        if (BAR) __syncthreads();
        // ---- compute phase: next loads dealt out over the MFMAs, fragment reads
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (LOADS)
                v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, (base + (unsigned)(it * NL + i) * 128u) % bytes, 0, 0);
#pragma unroll
            for (int m = NM * i / NL; m < NM * (i + 1) / NL; ++m)
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
            for (int dd = ND * i / NL; dd < ND * (i + 1) / NL; ++dd) {
                const u32x4 t = *reinterpret_cast<const u32x4 *>(&my[(rd + dd * 64) & 2047]);
                asm volatile("" ::"v"(t));
                sink ^= t.y;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR) __syncthreads();
    }
    float s = (float)sink;
    for (int i = 0; i < NL; ++i) s += (float)v[i].x;
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (s == 123.456f) out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

template <int BAR, bool CONV, int LOADS, int WAVES>
void run(const char *buf, float *out)
{
    const int iters = 400, blocks = 2048 / WAVES;       // 8 waves per CU in every configuration
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((probe<BAR, CONV, LOADS, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, buf, 2u << 20, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("waves/workgroup %d  barriers %d  split %d  loads %d : %.3f ms, %6.0f cycles per CU round (%s)\n", WAVES, BAR,
           (int)CONV, LOADS, ms, ms * 1e-3 * 2.4e9 / iters, hipGetErrorString(hipGetLastError()));
}

int main()
{
    char *buf; float *out;
    hipMalloc(&buf, 1u << 28); hipMemset(buf, 1, 1u << 28);
    hipMalloc(&out, 2048 * 64 * 4);
    run<0, false, 0, 4>(buf, out);
    run<0, false, 8, 4>(buf, out);
    run<0, true, 0, 4>(buf, out);
    run<0, true, 8, 4>(buf, out);
    run<1, false, 0, 4>(buf, out);
    run<1, false, 8, 4>(buf, out);
    run<1, true, 0, 4>(buf, out);
    run<1, true, 8, 4>(buf, out);
    run<1, true, 8, 2>(buf, out);
    run<1, true, 8, 1>(buf, out);
    run<1, true, 8, 8>(buf, out);
    return 0;
}
