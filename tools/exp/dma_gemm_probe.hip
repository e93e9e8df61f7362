// Developer prototype (not product code): the split-operand GEMM with BOTH operands as plane images
// (include/mrcnn_hip.h "operand planes") moved by LDS-DMA (buffer_load_dwordx4 ... lds) into two LDS
// stages — no VGPR round trip, no conversion, no ds_write — to measure what that structure reaches
// at the power cap.  C[M][N] = A[M][K] . B[N][K]^T, K % 32 == 0, fp32 output, 128x128 tiles,
// 4 waves (2x2, 64x64 each), one barrier per 32-deep K slice.  Same product / K order as the
// library's SPLIT kernel: results are bit-identical to mrcnn_conv2d_fwd of the 1x1 problem.
//   build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/exp/libdma_gemm_probe.so tools/exp/dma_gemm_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int SROW = 32;                           // ushorts per plane row (64 bytes)
constexpr int PLA = BM * SROW, PLB = BN * SROW;    // ushorts per plane
constexpr int STAGE = 3 * (PLA + PLB);             // ushorts per stage (48 KB)
constexpr unsigned kOOB = 0x80000000u;

template <int STAGES>
__global__ void __launch_bounds__(256, STAGES == 2 ? 1 : 3)
dma_gemm_kernel(const unsigned short *__restrict__ Apl, const unsigned short *__restrict__ Bpl,
                float *__restrict__ C, int M, int N, int K, unsigned a_bytes, unsigned b_bytes)
{
    __shared__ __attribute__((aligned(16))) unsigned short smem[STAGES * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lk = lane >> 5;
    // XCD-contiguous tile order, N fastest (tiles of one A row panel share an L2)
    const int ntn = N / BN, T = gridDim.x;
    const int q8 = T >> 3, r8 = T & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(Apl), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short *>(Bpl), 0, b_bytes, 0x00020000);
    // DMA piece (wave, i): plane q = idx >> 3, 16 rows rb * 16 .. + 15; lane -> row rb*16 + lane/4,
    // LDS slot lane % 4 of that row, which holds the logical 16-byte slot (lane % 4) ^ ((row >> 2) & 3)
    unsigned offA[6], offB[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int id = wave * 6 + i, q = id >> 3, rb = id & 7;
        const int row = rb * 16 + (lane >> 2);
        const int slot = (lane & 3) ^ ((row >> 2) & 3);
        offA[i] = m0 + row < M ? (unsigned)(m0 + row) * (unsigned)(K * 6) + q * 64u + slot * 16u : kOOB;
        offB[i] = n0 + row < N ? (unsigned)(n0 + row) * (unsigned)(K * 6) + q * 64u + slot * 16u : kOOB;
    }
    auto issue = [&](int kt, int stage) {
        const unsigned so = (unsigned)kt * 192u;          // scalar offset: K slice kt of every row
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int id = __builtin_amdgcn_readfirstlane(wave * 6 + i), q = id >> 3, rb = id & 7;
            unsigned short *dstA = smem + stage * STAGE + q * PLA + rb * 16 * SROW;
            unsigned short *dstB = smem + stage * STAGE + 3 * PLA + q * PLB + rb * 16 * SROW;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void *)dstA, 16, offA[i], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void *)dstB, 16, offB[i], so, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute = [&](int stage) {
        const unsigned short *pa = smem + stage * STAGE;
        const unsigned short *pb = pa + 3 * PLA;
        bf16x8 fa[2][2][3], fb[2][2][3];
        auto frag = [&](int ks, bf16x8 (&a)[2][3], bf16x8 (&b)[2][3]) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int row = wm * 64 + i * 32 + li;
                    a[i][q] = *reinterpret_cast<const bf16x8 *>(pa + q * PLA + row * SROW + (((ks * 2 + lk) ^ ((row >> 2) & 3)) << 3));
                }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int row = wn * 64 + j * 32 + li;
                    b[j][q] = *reinterpret_cast<const bf16x8 *>(pb + q * PLB + row * SROW + (((ks * 2 + lk) ^ ((row >> 2) & 3)) << 3));
                }
        };
        frag(0, fa[0], fb[0]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (ks == 0) frag(1, fa[1], fb[1]);
            constexpr int QA[6] = {0, 2, 1, 1, 0, 0}, QB[6] = {2, 0, 1, 0, 1, 0};
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i][QA[c]], fb[ks][j][QB[c]], acc[i][j], 0, 0, 0);
        }
    };

    const int nslices = K / BK;
    issue(0, 0);
    if constexpr (STAGES == 2) {
        for (int kt = 0; kt < nslices; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of slice kt have landed
            __syncthreads();                                    // ... everybody's; and compute(kt-1) is over
            if (kt + 1 < nslices) issue(kt + 1, (kt + 1) & 1);
            compute(kt & 1);
        }
    } else {        // one stage, three workgroups per CU: the DMA of a workgroup lands under the others' MFMAs
        for (int kt = 0; kt < nslices; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            compute(0);
            __syncthreads();
            if (kt + 1 < nslices) issue(kt + 1, 0);
        }
    }
    // plain per-element stores (the accumulator layout of v_mfma_f32_32x32x16)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                if (row < M && col < N) C[(int64_t)row * N + col] = acc[i][j][e];
            }
        }
}

extern "C" int dma_gemm(const void *Apl, const void *Bpl, float *C, int M, int N, int K, void *stream)
{
    const int tiles = ((M + BM - 1) / BM) * (N / BN);
    static const bool one = getenv("DMA_STAGES") && atoi(getenv("DMA_STAGES")) == 1;
    if (one) {
        hipLaunchKernelGGL(dma_gemm_kernel<1>, dim3(tiles), dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short *)Apl, (const unsigned short *)Bpl, C, M, N, K,
                           (unsigned)((int64_t)M * K * 6), (unsigned)((int64_t)N * K * 6));
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(dma_gemm_kernel<2>, dim3(tiles), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short *)Apl, (const unsigned short *)Bpl, C, M, N, K,
                       (unsigned)((int64_t)M * K * 6), (unsigned)((int64_t)N * K * 6));
    return (int)hipGetLastError();
}
