// Developer experiment (not product code): a register-only fp32 FMA kernel, used to measure
// how much vector-ALU work can run beside the MFMA GEMM on the same CUs (MI355X_MICROARCH.md:
// "MFMA and VALU pipes are separate").
#include <hip/hip_runtime.h>

extern "C" __global__ void __launch_bounds__(256) valu_fma_kernel(float *out, int iters, float a, float b)
{
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaf(acc[i], a, b);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 123.456f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" int exp_launch_valu(float *out, int blocks, int iters, void *stream)
{
    hipLaunchKernelGGL(valu_fma_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 1.0001f, 0.5f);
    return (int)hipGetLastError();
}
