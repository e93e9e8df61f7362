// Developer probe: semantics of `buffer_load_dwordx4 ... lds` (LDS-DMA) on gfx950 —
// lane-linear destination, and what an out-of-range lane writes.
#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) void *lds_ptr_t;
extern "C" __global__ void __launch_bounds__(256) dma_probe_kernel(const float *src, unsigned src_bytes, float *out, int oob_lane)
{
    __shared__ __attribute__((aligned(16))) float buf[256 * 4];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src), 0, src_bytes, 0x00020000);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 1024; i += 256) buf[i] = -7.f;
    __syncthreads();
    unsigned off = (unsigned)(tid * 16);
    if (lane == oob_lane) off = 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(buf + wave * 256), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) out[i] = buf[i];
}
extern "C" int dma_probe(const float *src, unsigned src_bytes, float *out, int oob_lane, void *stream)
{
    hipLaunchKernelGGL(dma_probe_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, src, src_bytes, out, oob_lane);
    return (int)hipGetLastError();
}
