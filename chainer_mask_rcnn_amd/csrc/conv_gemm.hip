// Implicit-GEMM convolution family on MFMA for gfx950 (NHWC / KRSC), fp32 in / out.
//
// Replaces every cuDNN call the reference makes through chainer's
// L.Convolution2D / L.Deconvolution2D / L.Linear and their backward passes
// (call sites /root/reference/chainer_mask_rcnn/models/region_proposal_network.py:75-80,
// models/mask_rcnn_resnet.py:131-143, chainer ResNet50Layers via
// models/resnet_extractor.py:93), plus the AffineChannel2D + residual + ReLU that
// follow each conv inside chainer's BottleneckA/B (fused into the epilogue here).
//
// One kernel body, three operand-gather modes:
//   FWD    y[m, k]   = sum_{r,s,c} x[pix(m,r,s), c] * w[k, r, s, c]
//   DGRAD  gx[m, c]  = sum_{r,s,k} gy[pix'(m,r,s), k] * w[k, r, s, c]
//   WGRAD  gw[k, rsc] = sum_m gy[m, k] * x[pix(m,r,s), c]      (split over m)
// Two arithmetics, both fp32 in / fp32 out / fp32 accumulate and fp32-roundoff-close to a CPU
// fp32 GEMM (there is no TF32 / xf32 on gfx950):
//   SPLIT (default, "split_bf16" knob): every operand element is staged as three bf16 values
//     whose sum is the element exactly, six v_mfma_f32_32x32x16_bf16 per 16-deep K step (see
//     SPLIT below); forward form 128x128 / 64x64 and the 128x128 weight gradient.
//   fp32 MFMA: v_mfma_f32_32x32x2_f32 (157 TF/s peak) — every other instantiation, and all of
//     them with split_bf16 = 0.
// The description below is the fp32-MFMA body; the SPLIT instantiations replace its stage
// layout and its inner product loop only.
//
// Tiling (wave64): 256 threads = 2x2 waves, each wave owns TM x TN MFMA tiles of
// 32x32 (TM=TN=2 -> 128x128 block tile; TM=TN=1 -> 64x64 for small problems).
// K is consumed in 32-deep slices staged global -> registers -> LDS
// (double-buffered, one barrier per slice).  Operands whose K index is
// contiguous in memory sit in LDS as [row][32+4] and are fetched with one
// ds_read_b128 per 4 MFMAs; operands whose K index is the strided one sit as
// [k][row] and are fetched with conflict-free ds_read_b32.  Inside an 8-deep
// K block MFMA step t pairs k = 4*half + t of both operands (any pairing is
// valid as long as A and B agree).
#include <algorithm>
#include <cstring>
#include <type_traits>

#include "common.h"

// Ablation / instrumentation switches (MRCNN_DBG_*, MRCNN_GEMM_TRACE, MRCNN_GEMM_CLOCKPROBE: several
// of them produce garbage results by design) compile only in an experiment build
// (tools/build_variant.sh passes -DMRCNN_EXPERIMENT_BUILD and writes the library under
// csrc/variants/): a stray EXTRA= on the product build must not ship a silently wrong library.
#if !defined(MRCNN_EXPERIMENT_BUILD) &&                                                              \
    (defined(MRCNN_DBG_NOLOAD) || defined(MRCNN_DBG_NOLOAD_A) || defined(MRCNN_DBG_NOLOAD_B) ||      \
     defined(MRCNN_DBG_NOSTORE) || defined(MRCNN_DBG_NOSPLITVALU) || defined(MRCNN_DBG_NOSPLIT_B) || \
     defined(MRCNN_DBG_NOSTAGE) || defined(MRCNN_DBG_NOGLOBAL) || defined(MRCNN_DBG_PLAIN_EPI) ||    \
     defined(MRCNN_DBG_AMOD) || defined(MRCNN_DBG_PITCH) || defined(MRCNN_GEMM_TRACE) ||             \
     defined(MRCNN_GEMM_CLOCKPROBE) || defined(MRCNN_GEMM_BIGBLOCKS) || defined(MRCNN_SPLIT_PK_SUB))
#error "MRCNN_DBG_* / trace / probe switches need -DMRCNN_EXPERIMENT_BUILD (tools/build_variant.sh): such a library must never be the product build"
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef MRCNN_GEMM_BK
#define MRCNN_GEMM_BK 32
#endif
constexpr int BK = MRCNN_GEMM_BK;
#ifndef MRCNN_GEMM_LOAD_AUX       // cache policy bits of the staged operand loads (experiment)
#define MRCNN_GEMM_LOAD_AUX 0
#endif
#ifndef MRCNN_GEMM_SETPRIO
#define MRCNN_GEMM_SETPRIO 0
#endif
// The forward-form kernel without mask staging fits 168 registers, so it runs with ONE LDS stage
// (37 KB) and three workgroups per CU: a wave spends ~40 % of a K slice issuing MFMAs and
// ~60 % staging (measured with s_memtime), so three interleaved waves per SIMD keep the pipe
// fuller than two (res5 3x3: 129 vs 122 TFLOP/s).  The variants that stage a mask (16 more
// registers) and the K-strided DGRAD gather would spill at 168 and keep two stages / two
// workgroups per CU.
#ifndef MRCNN_GEMM_SINGLEBUF
#define MRCNN_GEMM_SINGLEBUF 1
#endif
#ifndef MRCNN_GEMM_SINGLEBUF_WGRAD
#define MRCNN_GEMM_SINGLEBUF_WGRAD 1
#endif
#ifndef MRCNN_GEMM_MINWAVES
#define MRCNN_GEMM_MINWAVES 1
#endif
#ifndef MRCNN_GEMM_SINGLEBUF_SMALL
#define MRCNN_GEMM_SINGLEBUF_SMALL 1
#endif
constexpr bool single_buffered(int tm, int mode, bool masked)
{
    if (tm == 1) return MRCNN_GEMM_SINGLEBUF_SMALL != 0 && !masked && mode == 0;
    return MRCNN_GEMM_SINGLEBUF != 0 && tm >= 2 && !masked &&
           (mode == 0 || (mode == 2 && MRCNN_GEMM_SINGLEBUF_WGRAD != 0));
}
// minimum workgroups per CU the register allocation must allow (256-thread workgroups: one
// wave per SIMD each)
constexpr int min_blocks(int tm, int mode, bool masked)
{
    if (!single_buffered(tm, mode, masked)) return MRCNN_GEMM_MINWAVES;
#ifdef MRCNN_GEMM_BIGBLOCKS     // experiment: resident workgroups per CU of the 128x128 kernels
    return tm == 2 ? MRCNN_GEMM_BIGBLOCKS : 6;
#else
    return tm == 4 ? 2 : (tm == 2 ? 3 : 6);
#endif
}
// Experiment, off by default: the last workgroup of a wgrad tile to arrive sums the split-K
// slabs inside the GEMM kernel instead of a separate reduce launch.  Correct (tools/
// check_wgrad_reduce.py: no stale slab under load, bit-repeatable), but the agent-scope release
// every workgroup needs before it signals writes back its XCD's whole dirty L2 — measured
// 55.6 vs 51.6 ms per train step, i.e. 4 ms SLOWER than the 0.67 ms reduce launches it removes.
#ifndef MRCNN_WGRAD_INKERNEL_REDUCE
#define MRCNN_WGRAD_INKERNEL_REDUCE 0
#endif
constexpr int64_t kWgradCounterBytes = 1 << 20;   // >= 4 B x tiles for any supported filter
#ifndef MRCNN_SPLIT_ILV          // SPLIT forward form: next slice's loads between the MFMAs (see ILV)
#define MRCNN_SPLIT_ILV 1
#endif
#ifndef MRCNN_GEMM_WIDE_EPILOGUE
#define MRCNN_GEMM_WIDE_EPILOGUE 1
#endif
constexpr int KPAD = 4;  // K-contiguous LDS rows are 36 floats (conflict-free b128)

// The stride-1 dgrad is also run in FWD mode: a forward convolution of gy with the flipped,
// transposed filter (both operands K-contiguous).  MASKED (template flag of the kernel) adds
// the fused epilogue-backward of the producing conv to the A staging: g = gy * (mask_y > 0) *
// in_scale[k].
enum Mode { FWD = 0, DGRAD = 1, WGRAD = 2 };
constexpr bool is_fwd(int m) { return m == FWD; }
enum OutMode { OUT_PLAIN = 0, OUT_STRIDED = 1, OUT_DECONV = 2 };

struct GemmParams {
    const float *A;      // FWD: x      DGRAD: gy      WGRAD: gy
    const float *B;      // FWD: w      DGRAD: w       WGRAD: x
    float *C;            // FWD: y      DGRAD: gx      WGRAD: gw / split slabs
    const float *bias, *scale, *shift, *residual;
    int M, N;            // output tile space (rows, cols)
    int m_lo;            // first row handled by this launch (tail-split launches)
    int Kc;              // FWD: C_in   DGRAD: K_out   WGRAD: #pixels
    int gp, gq;          // pixel grid the rows (FWD/DGRAD) or K index (WGRAD) run over
    int sh, sw;          // spatial dims of the gathered tensor
    int R, S, stride, pad;
    int lda;             // channel stride (floats) of a pixel of the gathered tensor
    int ldb;             // FWD: row stride of w (R*S*C); DGRAD: R*S*C_in; WGRAD: unused
    int ldg;             // WGRAD: row stride of gy
    int cin;             // DGRAD/WGRAD: C_in
    int ldc;             // output row stride
    int flags;
    int out_mode;
    int oh, ow, ko;      // OUT_STRIDED: gx spatial dims; OUT_DECONV: ko = out channels
    int ostride;         // OUT_STRIDED: stride of the OUTPUT rows (0: `stride`; the forward-form strided dgrad gathers densely)
    int stem;            // FWD: stem gather (8 pixels x 4 ch per K slice)
    int split_len;       // WGRAD: pixels per split; FWD/DGRAD: K slices per split (0: no split)
    int64_t split_stride;// floats between split slabs
    int out_row0;        // FWD/DGRAD split launches: first row of the slab (subtracted)
    // Forward-form launches on many small maps (RoI features): GEMM rows ordered
    // (block of kPermBlock images, position, image in block): a 128-row tile holds ONE pixel
    // position of 128 images, so a filter tap that falls into the zero padding for that
    // position does so for EVERY row of the tile and its K slices are skipped (a 3x3 / pad 1
    // convolution on 7x7 maps: 18 % of all slices); consecutive tiles walk the positions of
    // the same image block, whose pixels therefore stay in the XCD's L2.  perm_n = number of
    // images (M is padded to whole blocks), 0 = natural (image, y, x) order.
    int perm_n;
    // WGRAD in-kernel slab reduction: per-tile arrival counters (zeroed by the host) and the
    // final gradient; the last workgroup of a tile to arrive sums the slabs in slab order
    int *tile_counters;
    float *reduce_out;
    // Fused backward of the producing conv's epilogue, applied while gy is staged
    // (DGRAD A operand / WGRAD A' operand):  g = gy * (mask_y > 0) * in_scale[k]
    const float *mask_y;   // output of the ReLU that followed the conv (same shape as gy) or NULL
    const float *in_scale; // AffineChannel2D scale of that conv (K_out) or NULL
    // backward-data epilogue: gx = (acc * scale[c] + res_g * (res_y > 0)) * (out_mask_y > 0)
    //   res_g / res_y : identity-shortcut gradient of a bottleneck (res_y NULL: res_g is added as is)
    //   out_mask_y    : output of the ReLU that produced this conv's INPUT, i.e. the epilogue-
    //                   backward of the NEXT conv down the chain applied where its incoming
    //                   gradient is produced (then that conv needs no mask staging at all)
    // WGRAD epilogue: gw[k, :] *= scale[k]
    const float *res_g, *res_y, *out_mask_y;
    float *split_ws;     // host only: caller's split-K workspace (kSplitWsBytes) or NULL
    unsigned a_bytes, b_bytes, c_bytes;  // buffer extents (bytes) of A (and mask_y), B, C
    // Fused tail (FWD / DGRAD, gridDim.y == 1): workgroups [0, tail_first) run whole tiles as
    // usual; the workgroups behind them run the tiles of the LEFTOVER rows — the rows beyond the
    // last full round of resident workgroups — each cut along K into tail_splits pieces that
    // write raw partial sums into slabs of tail_ws (summed, in order, by splitk_epilogue_kernel).
    // Dispatched last, the short pieces fill the CUs as the final round of whole tiles drains,
    // instead of a separate remainder launch that waits for the main launch to finish.
    int tail_first, tail_splits, tail_split_len, tail_row0;
    unsigned tail_bytes;
    float *tail_ws;
    int64_t tail_stride;
    // Batched launches (gridDim.z > 1; the 36 per-frequency GEMMs of the Winograd path,
    // winograd.hip): floats between consecutive problems of A, B and C.  The extents above
    // are then per problem.
    int64_t batch_a, batch_b, batch_c;
    // Start-up stagger (launch_kernel_m): the dispatcher places workgroup b, b + 256, b + 512 on
    // the same CU, and co-resident workgroups that start together stay in lockstep for hundreds
    // of microseconds — they stage, hit their barriers and run their epilogues at the same time,
    // and the matrix pipe idles meanwhile.  Workgroup b of the first stagger_slots x 256 sleeps
    // (b / 256) x stagger_cycles shader cycles before its first load, so the phases interleave.
    // Placement only decides how well this works, never the result.
    int stagger_slots, stagger_cycles;
};

#ifdef MRCNN_GEMM_TRACE
__device__ unsigned long long g_trace[64 * 4 * 64 * 5];
#endif
#ifdef MRCNN_GEMM_CLOCKPROBE
// developer instrumentation: per workgroup (shader-clock, 100 MHz reference clock) stamps at the
// start and the end of the kernel body + the XCC / CU it ran on (tools/exp/clock_probe.py)
constexpr int kProbeSlots = 16384;
__device__ unsigned long long g_probe[kProbeSlots * 5];
__device__ unsigned long long g_probe2[kProbeSlots * 8];   // main loop entry / exit (s_memrealtime)
#endif

// GEMM row -> (image, position) under the block-position-major order of GemmParams::perm_n
constexpr int kPermBlock = 128;
struct PermRow { int n, pos; };
__host__ __device__ __forceinline__ PermRow perm_row(int m, int pq)
{
    const int q = m / kPermBlock, img = m - q * kPermBlock;
    const int blk = q / pq;
    PermRow r;
    r.pos = q - blk * pq;
    r.n = blk * kPermBlock + img;
    return r;
}

template <int TM, int TN, int MODE>
struct Cfg {
    static constexpr int BM = 64 * TM, BN = 64 * TN;
    static constexpr bool A_KC = (MODE != WGRAD);  // A K-contiguous?
    static constexpr bool B_KC = is_fwd(MODE);
    static constexpr int A_FLOATS = A_KC ? BM * (BK + KPAD) : BK * BM;
    static constexpr int B_FLOATS = B_KC ? BN * (BK + KPAD) : BK * BN;
    static constexpr int A_V4 = BM * BK / 4 / 256;  // float4 per thread per slice
    static constexpr int B_V4 = BN * BK / 4 / 256;
};

// ---- buffer addressing -----------------------------------------------------------------
// Every global access of the kernel goes through a raw buffer descriptor: an element that
// must read as zero (image border of the im2col gather, tile tails in M / N / K) is given
// the byte offset kOOB, which lies beyond num_records, so the hardware returns 0 for loads
// and drops stores.  No per-element branch or select touches a loaded value, so every load
// of a K slice stays in flight across the slice's MFMAs.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0x80000000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, p ? bytes : 0u, 0x00020000);
}
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned off)
{
#ifdef MRCNN_DBG_NOLOAD   // experiment: every staged load hits the out-of-range path
    off = kOOB;
#endif
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, MRCNN_GEMM_LOAD_AUX);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z),
                       __uint_as_float(v.w));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
__device__ __forceinline__ void bstore1(__amdgpu_buffer_rsrc_t r, unsigned off, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, 0, 0);
}
#ifndef MRCNN_GEMM_STORE_AUX
#define MRCNN_GEMM_STORE_AUX 0
#endif
__device__ __forceinline__ void bstore4(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v)
{
    u32x4 u;
    u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y);
    u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
#ifdef MRCNN_DBG_NOSTORE      // ablation: every wide store takes the dropped (out-of-range) path
    off = kOOB;
#endif
    __builtin_amdgcn_raw_buffer_store_b128(u, r, off, 0, MRCNN_GEMM_STORE_AUX);
}
__device__ __forceinline__ void bstore8(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned a, unsigned b)
{
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 u;
    u.x = a; u.y = b;
    __builtin_amdgcn_raw_buffer_store_b64(u, r, off, 0, 0);
}
__device__ __forceinline__ float4 relu_mask(float4 v, float4 y)
{
    return make_float4(y.x > 0.f ? v.x : 0.f, y.y > 0.f ? v.y : 0.f, y.z > 0.f ? v.z : 0.f,
                       y.w > 0.f ? v.w : 0.f);
}
__device__ __forceinline__ float4 mul4(float4 v, float4 s)
{
    return make_float4(v.x * s.x, v.y * s.y, v.z * s.z, v.w * s.w);
}

// fp32 pair -> three packed bf16 pairs (low half = a, high half = b) with a = hi + mid + lo
// exactly: each conversion rounds to nearest, each residual is exact in fp32.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// a - b as ONE v_sub_f32: left to the compiler, adjacent subtractions are packed into v_pk_add_f32,
// which costs far more than its issue slot beside MFMAs (MI355X_MICROARCH.md, filler prices)
__device__ __forceinline__ float sub_f32(float a, float b)
{
#ifdef MRCNN_SPLIT_PK_SUB
    return a - b;
#else
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#endif
}
__device__ __forceinline__ void split3(float a, float b, unsigned &h, unsigned &m, unsigned &l)
{
#ifdef MRCNN_DBG_NOSPLITVALU   // ablation: no conversion arithmetic (results are garbage)
    h = __float_as_uint(a); m = __float_as_uint(b); l = h ^ m;
    return;
#endif
    h = pack_bf16(a, b);
    a = sub_f32(a, __uint_as_float(h << 16));
    b = sub_f32(b, __uint_as_float(h & 0xffff0000u));
    m = pack_bf16(a, b);
    a = sub_f32(a, __uint_as_float(m << 16));
    b = sub_f32(b, __uint_as_float(m & 0xffff0000u));
    l = pack_bf16(a, b);
}

// Round-3 tile-shape experiments, both bit-checked against this kernel and removed again
// (profiles/r03_exp_tiles.txt): 256 x 128 workgroup tiles (a wave owns 128 x 64, 255 VGPRs, two
// workgroups per CU) reach 85 vs 127 TFLOP/s on res5's 512 -> 2048 and 128 x 64 tiles (a wave
// owns 64 x 32, four workgroups per CU) 116 vs 127: with fp32 MFMA the number of interleaved
// waves per SIMD matters more than the bytes staged per MFMA, in both directions from 3.
// (A "ping-pong" variant — 512-thread workgroups running two tiles in antiphase, one group
// issuing MFMAs while the other stages — was measured slower, 110 vs 122 TFLOP/s on res5 3x3:
// one wave per SIMD cannot keep the fp32 MFMA pipe as full as interleaved free-running waves.
// Removed; see the history of this file.)
// WPERM: WGRAD with position-major pixel order (GemmParams::perm_n) — a separate instantiation
// because the natural-order kernel sits exactly at its 168-register budget.
// SPLIT (the default arithmetic; mrcnn_set_tuning("split_bf16", 0) selects fp32 MFMA everywhere;
// forward form 128x128 / 64x64 and the 128x128 natural-order weight gradient): the
// operands are staged as THREE bf16 planes each — a = a_hi + a_mid + a_lo EXACTLY (8 + 8 + 8
// significand bits) — and a K step runs six v_mfma_f32_32x32x16_bf16 (hi*hi, hi*mid, mid*hi,
// mid*mid, hi*lo, lo*hi; every product of two bf16 values is exact in fp32, accumulation is fp32).
// The three dropped cross terms are <= 2^-24 of |a*b| each, i.e. of the order of the rounding
// error of one fp32 multiply-add, so results stay fp32-accurate (tests/test_gpu_split_bf16.py
// measures the error against float64 next to the fp32 MFMA kernel's), while the matrix pipe
// does 6 x 32 instead of 8 x 64 cycles per 32x32x16 block.
// G groups of (MFMAs, one global load, LDS reads): the loads are dealt out evenly over ALL NM MFMAs
// of the K slice, the ND LDS reads (next K step's fragments) over the first NMD of them.
template <int G, int NM, int NMD, int ND, int I = 0>
__device__ __forceinline__ void sgb_interleave()
{
    if constexpr (I < G) {
        constexpr int m0 = NM * I / G, m1 = NM * (I + 1) / G;
        // LDS reads that belong in front of MFMA m1 (all of them once m1 >= NMD)
        constexpr int d0 = m0 >= NMD ? ND : ND * m0 / NMD, d1 = m1 >= NMD ? ND : ND * m1 / NMD;
        if constexpr (m1 > m0) __builtin_amdgcn_sched_group_barrier(0x008, m1 - m0, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if constexpr (d1 > d0) __builtin_amdgcn_sched_group_barrier(0x100, d1 - d0, 0);
        sgb_interleave<G, NM, NMD, ND, I + 1>();
    }
}

// W8 (round 6; SPLIT forward form only): ONE 512-thread workgroup per CU on a 256 x 128 tile — eight
// waves of 64 x 64, i.e. the two co-resident 128 x 128 workgroups of the standard configuration
// stacked on top of each other so that they share the filter panel: 48 instead of 64 operand loads
// and 3/4 of the split arithmetic per CU and K slice.  Two LDS stages, ONE barrier per slice.  What
// two independent workgroups get from drifting apart — one stages while the other multiplies — the
// two halves of this workgroup get from running the barrier interval in OPPOSITE ORDER: waves 0-3
// multiply slice k and then split / store slice k + 1, waves 4-7 store first and multiply afterwards
// (both orders read stage k & 1 and write the other one, so the interval needs no further
// synchronisation); a SIMD hosts one wave of each half (MI355X_MICROARCH.md, wave placement).
//
// PW (forward form): the launch is a 1x1 / stride 1 / pad 0 gather in natural row order that writes
// plain rows (pw_plain() on the host) — the general gather (taps, image / y / x decomposition,
// position-major rows) and the strided / deconvolution output maps are compiled out, which frees
// the scalar registers the general instantiation spills (W8: 87 -> 0 spilled SGPRs, 250 -> 182
// VGPRs, +3 % on the head's 1x1 layers).  W8 implies it.  K3: the same for the 3x3 / stride 1 / pad 1
// launches in natural row order (k3_plain(): the backbone's 3x3 layers and their transposed-filter data
// gradients) — filter size, stride and padding are constants of the gather.  Same arithmetic, same bits.
template <int TM, int TN, int MODE, bool MASKED, bool WPERM = false, bool SPLIT = false, bool W8 = false,
          bool PW = false, bool K3 = false>
__global__ void __launch_bounds__(W8 ? 512 : 256,
                                  SPLIT ? (TM == 2 ? 2 : 4) : min_blocks(TM, MODE, MASKED))
conv_gemm_kernel(const GemmParams p)
{
    static_assert(!W8 || (SPLIT && MODE == FWD && TM == 2 && TN == 2 && !MASKED),
                  "W8: the split-operand forward form on 64x64 wave tiles, unmasked");
    static_assert(!WPERM || (MODE == WGRAD && !MASKED), "WPERM is a WGRAD-only variant");
    static_assert(!PW || MODE == FWD, "PW: forward form only");
    static_assert(!K3 || (MODE == FWD && !W8 && !PW), "K3: forward form only");
    constexpr bool PWC = W8 || PW;
    // the launch geometry the gather and the output map use: compile-time facts in the PW / K3
    // instantiations (natural row order, no stem packing, plain output rows), the launch's fields otherwise
    constexpr bool GEO = PWC || K3;
    const int gR = PWC ? 1 : K3 ? 3 : p.R, gS = PWC ? 1 : K3 ? 3 : p.S;
    const int gStride = GEO ? 1 : p.stride, gPad = PWC ? 0 : K3 ? 1 : p.pad;
    const int gPermN = GEO ? 0 : p.perm_n;
    const bool gStem = !GEO && p.stem;
    constexpr bool ILV = MRCNN_SPLIT_ILV != 0 && SPLIT && MODE == FWD && !MASKED;
    static_assert(!SPLIT || (BK == 32 && TM == TN &&
                             ((MODE == FWD && (TM == 1 || TM == 2)) || (MODE == WGRAD && !WPERM && TM == 2))),
                  "SPLIT: forward form (128x128, 64x64) / weight gradient (128x128) only");
    constexpr bool SINGLEBUF = !W8 && (SPLIT || single_buffered(TM, MODE, MASKED));
    using C_ = Cfg<TM, TN, MODE>;
    constexpr int NT = W8 ? 512 : 256;                  // threads of the workgroup
    constexpr int BM = (W8 ? 2 : 1) * C_::BM, BN = C_::BN;
    constexpr int AV = BM * BK / 4 / NT, BV = BN * BK / 4 / NT;   // float4 per thread per slice
    constexpr bool HAS_MASK = MASKED;
    constexpr bool FWDLIKE = is_fwd(MODE);
    // SPLIT stage: 3 planes per operand of [row][32 bf16 + 8 pad] (80-byte rows: conflict-free b128)
    // forward form: 64-byte rows, 16-byte slots XOR-swizzled by (row >> 2) & 3 — conflict-free for
    // the b128 fragment reads (lane groups {0-3,12-15,20-27}, ...) AND for the b64 plane writes (a
    // 16-lane group writes two whole rows = one 128-byte bank row).  Weight gradient: 80-byte rows
    // (its transposing writes hit rows 4 apart; see swz).
    constexpr int SROW = MODE == WGRAD ? BK + 8 : BK;     // ushorts per plane row
    constexpr int PLA = BM * SROW, PLB = BN * SROW;       // ushorts per plane
    constexpr int STAGE_FLOATS = SPLIT ? 3 * (PLA + PLB) / 2 : C_::A_FLOATS + C_::B_FLOATS;
    __shared__ __attribute__((aligned(16))) float smem_all[1][SINGLEBUF ? 1 : 2][STAGE_FLOATS];

    float (*smem)[STAGE_FLOATS] = smem_all[0];
    // SPLIT WGRAD (80-byte rows, no swizzle): plane row 32 c + l holds channel 4 l + c of the tile
    // (WROWPERM).  The transposing writes of one instruction cover the channels 4 l + c of 32
    // lanes l: with the channel as the row they would sit 320 bytes apart — two bank positions
    // for the whole wave — and an XOR of the chunk index that spreads them breaks the b128
    // fragment reads instead (a read group spans rows r .. r + 27; round-4 PMC: half of the
    // LDS-active cycles of the former layout were bank conflicts).  With the permuted rows the
    // lanes of a write hit CONSECUTIVE rows (80 l mod 128: two-way, hidden by the store's own
    // register transfer) and a fragment read 16 rows with 16 different bank slots; the
    // permutation is undone where the tile is written (gw row 4 rr + .., column 4 li + ..).
    constexpr bool WROWPERM = SPLIT && MODE == WGRAD;
    static_assert(!(WROWPERM && MRCNN_WGRAD_INKERNEL_REDUCE),
                  "the WROWPERM epilogue returns before the in-kernel slab reduction: gw would stay unwritten");
    auto swz = [](int row) { return !SPLIT || MODE == WGRAD ? 0 : ((row >> 2) & 3) << 1; };
    const int tid = threadIdx.x;
#ifdef MRCNN_GEMM_CLOCKPROBE
    const unsigned long long probe_c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long probe_r0 = __builtin_amdgcn_s_memrealtime();
    struct ProbeEnd {
        unsigned long long c0, r0; int tid;
        __device__ ~ProbeEnd() {
            if (tid != 0) return;
            const unsigned slot = blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y;
            if (slot >= (unsigned)kProbeSlots) return;
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long *q = g_probe + (size_t)slot * 5;
            q[0] = c0; q[1] = r0;
            q[2] = __builtin_amdgcn_s_memtime(); q[3] = __builtin_amdgcn_s_memrealtime();
            q[4] = ((unsigned long long)xcc << 32) | hw;
        }
    } probe_end = {probe_c0, probe_r0, tid};
#endif
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int64_t zb = blockIdx.z;                      // batched launches: problem index
    const __amdgpu_buffer_rsrc_t rA = make_rsrc(p.A + zb * p.batch_a, p.a_bytes);
    const __amdgpu_buffer_rsrc_t rB = make_rsrc(p.B + zb * p.batch_b, p.b_bytes);
    const __amdgpu_buffer_rsrc_t rMask = make_rsrc(p.mask_y, p.a_bytes);
    const bool use_mask = HAS_MASK && p.mask_y != nullptr;

    // tile -> (m0, n0).  Workgroup b runs on XCD b % 8 (observed dispatch order); remap so
    // that each XCD owns a contiguous run of tile ids — N fastest — and neighbouring tiles,
    // which share the gathered A rows and the filter panel, hit the same private L2.
    // WGRAD (grid = tiles x splits): the remap runs over the whole 2-D grid with the split as
    // the slow index, so the tiles of one split — which all stream the same pixel range of gy
    // and x — land on one or two XCDs instead of all eight.
    const int ntn = (p.N + BN - 1) / BN;
    int tile = blockIdx.x + blockIdx.y * gridDim.x;
    int split;
    int split_len = p.split_len;
    const bool tail = MODE != WGRAD && p.tail_splits > 0 && tile >= p.tail_first;   // uniform
    if (tail) {
        const int rem = tile - p.tail_first;
        split = rem % p.tail_splits;
        tile = p.tail_first + rem / p.tail_splits;
        split_len = p.tail_split_len;
    } else {
        const int nwg = (MODE != WGRAD && p.tail_splits > 0) ? p.tail_first : (int)(gridDim.x * gridDim.y);
        const int q = nwg >> 3, r = nwg & 7, xcd = tile & 7, idx = tile >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        split = tile / (int)gridDim.x;
        tile -= split * (int)gridDim.x;
    }
    // (Walking the tiles N-panel by N-panel so that an XCD's resident workgroups sweep a filter panel
    // that fits its L2 was measured in round 5 on res5's shapes, panels of 1 .. 8 N-tiles: every
    // layer within +-2 % of the plain N-fastest walk, profiles/r05c_npanel.txt; removed.)
    const int m0 = p.m_lo + (tile / ntn) * BM;
    const int n0 = (tile % ntn) * BN;

    // ---------------- per-thread gather state -----------------------------------
    constexpr int KC_C4 = BK / 4, KC_RPP = NT / KC_C4;    // float4 per row, rows per pass
    const int kc_row = tid / KC_C4, kc_c4 = tid % KC_C4;
    constexpr int NA = AV;                              // A rows a thread addresses
    auto a_row = [&](int i) { return kc_row + KC_RPP * i; };
    int a_n[NA], a_y[NA], a_x[NA];     // FWD/DGRAD: pixel coords of each A row
    // 1x1 / stride 1 / pad 0 forward-form launches (two thirds of the RoI head's GEMMs): GEMM row
    // m IS pixel m of the gathered tensor — no (image, y, x) decomposition, i.e. none of the
    // eight integer divisions of the general set-up
    // (PW / W8 launches are pointwise by the host's rule: the general gather is compiled out)
    const bool pointwise = PWC || (FWDLIKE && gR == 1 && gS == 1 && gStride == 1 && gPad == 0 &&
                                  gPermN == 0 && !gStem && p.gp == p.sh && p.gq == p.sw);   // uniform
    if (MODE != WGRAD && pointwise) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m0 + a_row(i);
            a_n[i] = m < p.M ? m : 0;          // (pixel index; folded into a_base below)
#ifdef MRCNN_DBG_AMOD      // experiment: every tile reads the same few A rows (L2-resident operand)
            a_n[i] &= MRCNN_DBG_AMOD - 1;
#endif
            a_x[i] = 0;
            a_y[i] = m < p.M ? 0 : -(1 << 28);
        }
    } else if (MODE != WGRAD) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m0 + a_row(i);
            const bool ok = m < p.M;
            const int mm = ok ? m : 0;
            int n, rem;
            bool img_ok = true;
            if (FWDLIKE && gPermN > 0) {
                const PermRow pr = perm_row(mm, p.gp * p.gq);
                n = pr.n; rem = pr.pos;
                img_ok = n < gPermN;           // padding rows of the last image block
            } else {
                n = mm / (p.gp * p.gq); rem = mm - n * (p.gp * p.gq);
            }
            const int gy = rem / p.gq, gx = rem - gy * p.gq;
            a_n[i] = n;
            if (FWDLIKE) { a_y[i] = gy * gStride - gPad; a_x[i] = gx * gStride - gPad; }
            else { a_y[i] = gy + gPad; a_x[i] = gx + gPad; }
            if (!ok || !img_ok) a_y[i] = -(1 << 28);   // no such row: every tap out of range
        }
    }
    constexpr int A_TPR = BM / 4, B_TPR = BN / 4;            // threads per k row (K-strided tiles)
    constexpr int A_RPP = 256 / A_TPR, B_RPP = 256 / B_TPR;  // k rows per pass
    const int wa_k = tid / A_TPR, wa_c4 = tid % A_TPR;
    const int wb_k = tid / B_TPR, wb_c4 = tid % B_TPR;
    // k row (pixel of the slice) of a thread's i-th load of a K-strided tile.  SPLIT: AV
    // CONSECUTIVE pixels per thread, so that the 4 channels x AV pixels it holds go to LDS as
    // [channel][pixel] rows (the layout the bf16 MFMA fragments are read from) with 8-byte writes.
    auto a_krow = [&](int i) { return SPLIT ? wa_k * AV + i : wa_k + A_RPP * i; };
    auto b_krow = [&](int i) { return SPLIT ? wb_k * BV + i : wb_k + B_RPP * i; };
    int wr = 0, ws_ = 0, wc = 0;
    int wy_lo = 0, wx_lo = 0, wnvx = 1;   // WGRAD position-major: valid-position rectangle
    int wr_u = 0, ws_u = 0, wnv = 1;      // ... the tile's tap and its position count, uniform
    bool wcol_ok = false;
    int pn[BV], py[BV], px[BV];
    int k_begin = 0, k_end = 0;
    if (MODE == WGRAD) {
        const int j = n0 + wb_c4 * 4;
        wcol_ok = j < p.N;
        const int jj = wcol_ok ? j : 0;
        const int rs = jj / p.cin;
        wc = jj - rs * p.cin;
        wr = rs / gS;
        ws_ = rs - wr * gS;
        k_begin = split * p.split_len;
        if (WPERM) {
            // Pixel order for the reduction: (block of BK images, position, image in block),
            // restricted to the positions whose tap (wr, ws_) lies inside the map — one K
            // slice = one position of BK consecutive images, and a block's positions follow
            // each other so its pixels stay in L2.  Every column of the tile belongs to the
            // same tap (cin % BN == 0, checked by the host): tile-uniform values live in SGPRs.
            wr_u = __builtin_amdgcn_readfirstlane(wr);
            ws_u = __builtin_amdgcn_readfirstlane(ws_);
            wy_lo = max(0, gPad - wr_u);
            wx_lo = max(0, gPad - ws_u);
            const int nvy = min(p.gp - 1, p.sh - 1 + gPad - wr_u) - wy_lo + 1;
            wnvx = max(min(p.gq - 1, p.sw - 1 + gPad - ws_u) - wx_lo + 1, 0);
            wnv = max(nvy, 0) * wnvx;
            const int nblk = (gPermN + BK - 1) / BK;
            k_end = min(nblk * wnv * BK, k_begin + p.split_len);
        } else {
            k_end = min(p.Kc, k_begin + p.split_len);
#pragma unroll
            for (int i = 0; i < BV; ++i) {
                const int m = k_begin + b_krow(i);
                const int n = m / (p.gp * p.gq);
                const int rem = m - n * (p.gp * p.gq);
                pn[i] = n;
                py[i] = rem / p.gq;
                px[i] = rem - py[i] * p.gq;
            }
        }
    }

    const int adv_x = BK % p.gq, adv_y = (BK / p.gq) % p.gp, adv_n = BK / (p.gp * p.gq);
    const int cprs = (MODE == WGRAD) ? 1 : (p.Kc + BK - 1) / BK;  // K slices per (r,s)
    // FWD/DGRAD split-K (leftover rows of a small-M problem, see launch()): this workgroup
    // runs slices [kt0, kt0 + nslices) and writes raw partial sums into its slab
    // position-major rows: the taps that are inside the map for at least one position of this
    // tile, as 4-bit indices packed into a word (wave-uniform)
    int ntaps = gR * gS;
    unsigned long long tap_list = 0;
    if (FWDLIKE && !GEO && gPermN > 0) {
        const int q_lo = m0 / kPermBlock;
        const int q_hi = min(p.M - 1, m0 + BM - 1) / kPermBlock;
        const int pq = p.gp * p.gq;
        ntaps = 0;
        for (int rs = 0; rs < gR * gS; ++rs) {
            const int r = rs / gS, s = rs - r * gS;
            bool hit = q_hi - q_lo > 3;              // many positions in the tile: keep all
            for (int qq = q_lo; qq <= q_hi && !hit; ++qq) {
                const int q = qq % pq;
                const int y = q / p.gq, x = q - y * p.gq;
                hit = (unsigned)(y * gStride - gPad + r) < (unsigned)p.sh &&
                      (unsigned)(x * gStride - gPad + s) < (unsigned)p.sw;
            }
            if (hit) {
                tap_list |= (unsigned long long)rs << (4 * ntaps);
                ++ntaps;
            }
        }
    }
    int kt0 = 0;
    int nslices = (MODE == WGRAD) ? (k_end - k_begin + BK - 1) / BK : ntaps * cprs;
    if (MODE != WGRAD && split_len > 0) {
        kt0 = split * split_len;
        nslices = max(0, min(nslices - kt0, split_len));
    }

    if (p.stagger_cycles > 0) {
        const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned slot = lin >> 8;
        if (slot > 0 && slot < (unsigned)p.stagger_slots) {          // workgroup-uniform
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            const unsigned long long wait = (unsigned long long)slot * (unsigned)p.stagger_cycles;
            while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(8);
        }
    }

    float4 ra[AV], rb[BV];
    float4 rm[HAS_MASK ? AV : 1];
    float4 rscale = make_float4(1.f, 1.f, 1.f, 1.f);
    const bool use_scale = HAS_MASK && p.in_scale != nullptr;
    if (MODE == WGRAD && use_scale && m0 + wa_c4 * 4 < p.M)
        rscale = *reinterpret_cast<const float4 *>(p.in_scale + m0 + wa_c4 * 4);

    // Loop-invariant parts of every load address (element offsets; an invalid row carries the
    // sentinel 0x20000000 so that 4 * offset lands beyond num_records and reads as zero).
    constexpr unsigned kBad = 0x20000000u;
    constexpr int NB = BV;
    unsigned a_base[NA], b_base[NB];
    if (MODE != WGRAD) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const unsigned pix = pointwise ? (unsigned)a_n[i]
                                           : (unsigned)((a_n[i] * p.sh + a_y[i]) * p.sw + a_x[i]);
            a_base[i] = pix * (unsigned)p.lda + (unsigned)(kc_c4 * 4);
        }
        if (FWDLIKE) {
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int n = n0 + kc_row + KC_RPP * i;
                b_base[i] = n < p.N ? (unsigned)(n * p.ldb + kc_c4 * 4) : kBad;
            }
        } else {
#pragma unroll
            for (int i = 0; i < BV; ++i) {
                const int n = n0 + wb_c4 * 4;
                b_base[i] = n < p.N ? (unsigned)((wb_k + B_RPP * i) * p.ldb + n) : kBad;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int k = m0 + wa_c4 * 4;
            a_base[i] = k < p.M ? (unsigned)(a_krow(i) * p.ldg + k) : kBad;
        }
        if (!WPERM) {
#pragma unroll
            for (int i = 0; i < BV; ++i)
                b_base[i] = wcol_ok ? (unsigned)(b_krow(i) * p.lda + wc) : kBad;
        }
    }
    const bool wpoint = MODE == WGRAD && !WPERM && gR == 1 && gS == 1 && gStride == 1 && gPad == 0 &&
                        p.gp == p.sh && p.gq == p.sw;      // uniform
    const int RS = gR * gS;

    // ILV (SPLIT forward form without mask staging): load_slice only computes the byte offsets of
    // the slice's loads; the loads themselves are issued INSIDE the following compute(), spread
    // between its MFMAs (issue_loads + the sched_group_barrier sequence there).  Issued as one
    // burst ahead of the MFMAs, the 8 .. 12 16-byte loads of every wave of the CU queue behind
    // each other in the texture path and the wave sits in their issue for 1000 - 1700 cycles
    // (s_memtime stamps) before its first MFMA.
    unsigned oa[ILV ? NA : 1], ob[ILV ? NB : 1];
    auto ldA = [&](int i, unsigned off) {
#ifdef MRCNN_DBG_NOLOAD_A     // ablation: the A operand's loads take the out-of-range path
        off = kOOB;
#endif
        if constexpr (ILV) {
            oa[i] = off;
        } else {
            ra[i] = bload4(rA, off);
            if (HAS_MASK && use_mask) rm[i] = bload4(rMask, off);
        }
    };
    auto ldB = [&](int i, unsigned off) {
#ifdef MRCNN_DBG_NOLOAD_B
        off = kOOB;
#endif
        if constexpr (ILV) ob[i] = off;
        else rb[i] = bload4(rB, off);
    };
    auto issue_loads = [&]() {
        if constexpr (ILV) {
#pragma unroll
            for (int i = 0; i < NA; ++i) ra[i] = bload4(rA, oa[i]);
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = bload4(rB, ob[i]);
        }
    };
    // issue the global loads of slice kt (nothing here consumes a loaded value).
    // FWD/DGRAD K order: channel chunk outer, filter tap (r,s) inner — consecutive slices re-read
    // the same 32-channel slab of neighbouring pixels, which stays in the CU's L1.
    auto load_slice = [&](int kt) {
        if (FWDLIKE && pointwise) {
            // 1x1 / stride 1: slice kt is channel chunk kt of the row's own pixel — none of the
            // tap / chunk divisions and bounds checks of the general gather (a quarter of the
            // instructions a wave issues per slice, all of them competing with the MFMAs of the
            // co-resident waves for issue slots)
            const int c0 = (kt + kt0) * BK;
            const int cc = c0 + kc_c4 * 4;
            const bool c_ok = cc < p.Kc;
#pragma unroll
            for (int i = 0; i < AV; ++i)
                ldA(i, (c_ok && a_y[i] == 0) ? 4u * (a_base[i] + (unsigned)c0) : kOOB);
            if (HAS_MASK && use_scale)
                rscale = c_ok ? *reinterpret_cast<const float4 *>(p.in_scale + cc)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < BV; ++i) ldB(i, c_ok ? 4u * (b_base[i] + (unsigned)c0) : kOOB);
            return;
        }
        if (FWDLIKE || MODE == DGRAD) {
            kt += kt0;
            int chunk, rs;
            if (FWDLIKE && gPermN > 0) {
                chunk = kt / ntaps;
                rs = (int)((tap_list >> (4 * (kt - chunk * ntaps))) & 15ull);
            } else {
                chunk = kt / RS;
                rs = kt - chunk * RS;
            }
            const int c0 = chunk * BK;
            const int r = rs / gS, s = rs - r * gS;
            const int cc = c0 + kc_c4 * 4;
            const bool c_ok = gStem || cc < p.Kc;
            // wave-uniform part of the A address for this slice
            const int tap = (FWDLIKE ? (r * p.sw + s) : -(r * p.sw + s)) * p.lda + c0;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int iy, ix;
                if (FWDLIKE) { iy = a_y[i] + r; ix = a_x[i] + s + (gStem ? (cc >> 2) : 0); }
                else { iy = a_y[i] - r; ix = a_x[i] - s; }
                const bool ok = c_ok && (unsigned)iy < (unsigned)p.sh && (unsigned)ix < (unsigned)p.sw;
                ldA(i, ok ? 4u * (a_base[i] + (unsigned)tap) : kOOB);
            }
            if (HAS_MASK && use_scale)
                rscale = cc < p.Kc ? *reinterpret_cast<const float4 *>(p.in_scale + cc)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
            if (FWDLIKE) {
                const unsigned wofs = (unsigned)(rs * p.Kc + c0);
#pragma unroll
                for (int i = 0; i < BV; ++i) ldB(i, cc < p.Kc ? 4u * (b_base[i] + wofs) : kOOB);
            } else {
                const unsigned wofs = (unsigned)(c0 * p.ldb + rs * p.cin);
#pragma unroll
                for (int i = 0; i < BV; ++i) {
                    const int k = c0 + wb_k + B_RPP * i;
                    rb[i] = bload4(rB, k < p.Kc ? 4u * (b_base[i] + wofs) : kOOB);
                }
            }
        } else {
            const int kb = k_begin + kt * BK;
            if (WPERM) {
                // A and B rows of a thread are the same k rows (BM == BN); the slice's image
                // block and position are wave-uniform
                const int sidx = kb / BK;
                const int blk = sidx / max(wnv, 1), pos = sidx - blk * max(wnv, 1);
                const int yv = pos / max(wnvx, 1);
                const int y = wy_lo + yv, x = wx_lo + pos - yv * wnvx;
                const int a_col = m0 + wa_c4 * 4;
                const int pix_a = y * p.gq + x;
                const int pix_b = (y + wr_u - gPad) * p.sw + x + ws_u - gPad;
#pragma unroll
                for (int i = 0; i < BV; ++i) {
                    const int r = wb_k + B_RPP * i;
                    const int n = blk * BK + r;
                    const bool ok = kb + r < k_end && n < gPermN;
                    const unsigned offa = (unsigned)((n * (p.gp * p.gq) + pix_a) * p.ldg + a_col);
                    ra[i] = bload4(rA, ok && a_col < p.M ? 4u * offa : kOOB);
                    const unsigned offb = (unsigned)((n * (p.sh * p.sw) + pix_b) * p.lda + wc);
                    rb[i] = bload4(rB, ok && wcol_ok ? 4u * offb : kOOB);
                }
                return;
            }
            const unsigned gofs = (unsigned)(kb * p.ldg);
#pragma unroll
            for (int i = 0; i < AV; ++i) {
                const int m = kb + a_krow(i);
                const unsigned off = m < k_end ? 4u * (a_base[i] + gofs) : kOOB;
                ra[i] = bload4(rA, off);
                if (use_mask) rm[i] = bload4(rMask, off);
            }
            if (wpoint) {
                // 1x1 / stride 1: pixel m of gy is pixel m of x
                const unsigned xofs = (unsigned)(kb * p.lda);
#pragma unroll
                for (int i = 0; i < BV; ++i) {
                    const int m = kb + b_krow(i);
                    rb[i] = bload4(rB, m < k_end ? 4u * (b_base[i] + xofs) : kOOB);
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < BV; ++i) {
                const int m = kb + b_krow(i);
                const int iy = py[i] * gStride - gPad + wr;
                const int ix = px[i] * gStride - gPad + ws_;
                const bool ok = wcol_ok && m < k_end && (unsigned)iy < (unsigned)p.sh &&
                                (unsigned)ix < (unsigned)p.sw;
                rb[i] = bload4(rB, ok ? 4u * (unsigned)(((pn[i] * p.sh + iy) * p.sw + ix) * p.lda + wc)
                                      : kOOB);
                // advance this row's pixel by BK for the next slice (two carries, no loops:
                // adv_x = BK % gq, adv_y = (BK / gq) % gp, adv_n = BK / (gp*gq))
                px[i] += adv_x;
                const int cx = px[i] >= p.gq;
                px[i] -= cx ? p.gq : 0;
                py[i] += adv_y + cx;
                const int cy = py[i] >= p.gp;
                py[i] -= cy ? p.gp : 0;
                pn[i] += adv_n + cy;
            }
        }
    };

    // registers -> LDS; the fused epilogue-backward (ReLU mask, affine scale) is applied here
    auto store_slice = [&](int buf) {
        if constexpr (SPLIT) {
            unsigned short *pa = reinterpret_cast<unsigned short *>(smem[buf]);
            unsigned short *pb = pa + 3 * PLA;
            auto put = [&](unsigned short *plane0, int plane_len, int row, float4 v) {
                unsigned h0, m0_, l0, h1, m1, l1;
                split3(v.x, v.y, h0, m0_, l0);
                split3(v.z, v.w, h1, m1, l1);
                unsigned short *q = plane0 + row * SROW + ((kc_c4 ^ swz(row)) << 2);
                *reinterpret_cast<uint2 *>(q) = make_uint2(h0, h1);
                *reinterpret_cast<uint2 *>(q + plane_len) = make_uint2(m0_, m1);
                *reinterpret_cast<uint2 *>(q + 2 * plane_len) = make_uint2(l0, l1);
            };
            if constexpr (MODE == WGRAD) {
                // a thread holds 4 channels x 4 consecutive pixels: transposed in registers, one
                // 8-byte write per channel and plane at [channel][pixel chunk ^ swz(channel)]
                static_assert(AV == 4 && BV == 4, "SPLIT WGRAD: 128x128 tiles");
                auto put_t = [&](unsigned short *plane0, int plane_len, int row0, int kchunk,
                                 const float (&e)[4][4]) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        unsigned h0, m0_, l0, h1, m1, l1;
                        split3(e[0][c], e[1][c], h0, m0_, l0);
                        split3(e[2][c], e[3][c], h1, m1, l1);
                        const int row = 32 * c + (row0 >> 2);      // channel row0 + c (WROWPERM)
                        unsigned short *q = plane0 + row * SROW + ((kchunk ^ swz(row)) << 2);
                        *reinterpret_cast<uint2 *>(q) = make_uint2(h0, h1);
                        *reinterpret_cast<uint2 *>(q + plane_len) = make_uint2(m0_, m1);
                        *reinterpret_cast<uint2 *>(q + 2 * plane_len) = make_uint2(l0, l1);
                    }
                };
                float ea[4][4], eb[4][4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float4 v = ra[i];
                    if (HAS_MASK && use_mask) v = relu_mask(v, rm[i]);
                    if (HAS_MASK && use_scale) v = mul4(v, rscale);
                    ea[i][0] = v.x; ea[i][1] = v.y; ea[i][2] = v.z; ea[i][3] = v.w;
                    eb[i][0] = rb[i].x; eb[i][1] = rb[i].y; eb[i][2] = rb[i].z; eb[i][3] = rb[i].w;
                }
                put_t(pa, PLA, wa_c4 * 4, wa_k, ea);
                put_t(pb, PLB, wb_c4 * 4, wb_k, eb);
                return;
            }
#pragma unroll
            for (int i = 0; i < AV; ++i) {
                float4 v = ra[i];
                if (HAS_MASK && use_mask) v = relu_mask(v, rm[i]);
                if (HAS_MASK && use_scale) v = mul4(v, rscale);
                put(pa, PLA, kc_row + KC_RPP * i, v);
            }
#pragma unroll
            for (int i = 0; i < BV; ++i) {
#ifdef MRCNN_DBG_NOSPLIT_B    // ablation: the B operand costs no conversion (results are garbage)
                const int row = kc_row + KC_RPP * i;
                unsigned short *q = pb + row * SROW + ((kc_c4 ^ swz(row)) << 2);
                const uint2 u = make_uint2(__float_as_uint(rb[i].x), __float_as_uint(rb[i].y));
                *reinterpret_cast<uint2 *>(q) = u;
                *reinterpret_cast<uint2 *>(q + PLB) = make_uint2(__float_as_uint(rb[i].z), __float_as_uint(rb[i].w));
                *reinterpret_cast<uint2 *>(q + 2 * PLB) = u;
#else
                put(pb, PLB, kc_row + KC_RPP * i, rb[i]);
#endif
            }
            return;
        }
        float *sa = smem[buf];
        float *sb = smem[buf] + C_::A_FLOATS;
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            float4 v = ra[i];
            if (HAS_MASK && use_mask) v = relu_mask(v, rm[i]);
            if (HAS_MASK && use_scale) v = mul4(v, rscale);
            if (C_::A_KC)
                *reinterpret_cast<float4 *>(sa + (kc_row + KC_RPP * i) * (BK + KPAD) + kc_c4 * 4) = v;
            else
                *reinterpret_cast<float4 *>(sa + (wa_k + A_RPP * i) * BM + wa_c4 * 4) = v;
        }
        if (C_::B_KC) {
#pragma unroll
            for (int i = 0; i < BV; ++i)
                *reinterpret_cast<float4 *>(sb + (kc_row + KC_RPP * i) * (BK + KPAD) + kc_c4 * 4) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < BV; ++i)
                *reinterpret_cast<float4 *>(sb + (wb_k + B_RPP * i) * BN + wb_c4 * 4) = rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int li = lane & 31, lk = lane >> 5;

    // LDS -> MFMA fragments for the 8-deep K block kb of buffer `buf`
    auto load_frag = [&](const float *sa, const float *sb, int kb, float (&af)[TM][4],
                         float (&bf)[TN][4]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * (32 * TM) + i * 32 + li;
            if (C_::A_KC) {
                const float4 v = *reinterpret_cast<const float4 *>(
                    sa + row * (BK + KPAD) + kb * 8 + lk * 4);
                af[i][0] = v.x; af[i][1] = v.y; af[i][2] = v.z; af[i][3] = v.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) af[i][t] = sa[(kb * 8 + lk * 4 + t) * BM + row];
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * (32 * TN) + j * 32 + li;
            if (C_::B_KC) {
                const float4 v = *reinterpret_cast<const float4 *>(
                    sb + col * (BK + KPAD) + kb * 8 + lk * 4);
                bf[j][0] = v.x; bf[j][1] = v.y; bf[j][2] = v.z; bf[j][3] = v.w;
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) bf[j][t] = sb[(kb * 8 + lk * 4 + t) * BN + col];
            }
        }
    };

    // one K slice: fragments of block kb+1 are fetched while block kb's MFMAs issue
    auto compute_ = [&](int buf, auto issue_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;   // issue the next slice's loads here
        if constexpr (SPLIT) {
            const unsigned short *pa = reinterpret_cast<const unsigned short *>(smem[buf]);
            const unsigned short *pb = pa + 3 * PLA;
            bf16x8 fa[2][TM][3], fb[2][TN][3];
            auto frag = [&](int ks, bf16x8 (&a)[TM][3], bf16x8 (&b)[TN][3]) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        a[i][q] = *reinterpret_cast<const bf16x8 *>(
                            pa + q * PLA + (wm * (32 * TM) + i * 32 + li) * SROW +
                            (((ks * 4 + lk * 2) ^ swz(wm * (32 * TM) + i * 32 + li)) << 2));
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        b[j][q] = *reinterpret_cast<const bf16x8 *>(
                            pb + q * PLB + (wn * (32 * TN) + j * 32 + li) * SROW +
                            (((ks * 4 + lk * 2) ^ swz(wn * (32 * TN) + j * 32 + li)) << 2));
            };
            frag(0, fa[0], fb[0]);
#ifndef MRCNN_DBG_NOGLOBAL
            if constexpr (ISSUE) issue_loads();
#endif
#pragma unroll
            for (int ks = 0; ks < BK / 16; ++ks) {
                if (ks + 1 < BK / 16) frag(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
                // smallest terms first: (hi,lo) (lo,hi) (mid,mid) (mid,hi) (hi,mid) (hi,hi)
                constexpr int QA[6] = {0, 2, 1, 1, 0, 0}, QB[6] = {2, 0, 1, 0, 1, 0};
#pragma unroll
                for (int c = 0; c < 6; ++c)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                fa[ks & 1][i][QA[c]], fb[ks & 1][j][QB[c]], acc[i][j], 0, 0, 0);
            }
            if constexpr (ILV && !ISSUE) {
                // (W8, waves that multiply first: their next slice's loads are issued after the store)
                // first fragments, then one fragment read of the second K step per two MFMAs
                constexpr int NFR = 3 * (TM + TN), NMH = 6 * TM * TN;
                __builtin_amdgcn_sched_group_barrier(0x100, NFR, 0);
#pragma unroll
                for (int g = 0; g < NFR; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NMH * (BK / 16) - 2 * NFR, 0);
            } else if constexpr (ILV) {
                // issue order: first fragments, then the first K step's MFMAs with one global load
                // and the second K step's fragment reads dealt out between them, then the rest
                constexpr int NFR = 3 * (TM + TN), NMH = 6 * TM * TN;
                __builtin_amdgcn_sched_group_barrier(0x100, NFR, 0);
#if MRCNN_SPLIT_ILV == 2      // (A/B: loads bunched into the first K step's MFMAs)
                sgb_interleave<NA + NB, NMH, NMH, NFR>();
                __builtin_amdgcn_sched_group_barrier(0x008, NMH * (BK / 16 - 1), 0);
#else
                // the CU's texture path takes ~64 cycles per 16-byte-per-lane wave load whoever
                // issues it (tools/exp/planes_probe.py ablations: kernel time = compute + 64 cycles
                // x loads): a wave whose load waits for its slot issues no MFMA either, so the
                // loads sit as far apart as the slice allows — one per NM / (NA + NB) MFMAs
                sgb_interleave<NA + NB, NMH * (BK / 16), NMH * (BK / 16 - 1) - 2 * TM * TN, NFR>();
#endif
            }
            return;
        }
        const float *sa = smem[buf];
        const float *sb = smem[buf] + C_::A_FLOATS;
        float af[2][TM][4], bf[2][TN][4];
        load_frag(sa, sb, 0, af[0], bf[0]);
        if (MRCNN_GEMM_SETPRIO == 1) __builtin_amdgcn_s_setprio(1);
        if (MRCNN_GEMM_SETPRIO == 2) __builtin_amdgcn_s_setprio(0);   // staging phases run at 2
        // DS read instructions per K block (b32 pairs are merged into ds_read2_b32)
        constexpr int NR = (C_::A_KC ? TM : 2 * TM) + (C_::B_KC ? TN : 2 * TN);
        constexpr int NMFMA = 4 * TM * TN;
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
#pragma unroll
        for (int kb = 0; kb < BK / 8; ++kb) {
            if (kb + 1 < BK / 8) load_frag(sa, sb, kb + 1, af[(kb + 1) & 1], bf[(kb + 1) & 1]);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                            af[kb & 1][i][t], bf[kb & 1][j][t], acc[i][j], 0, 0, 0);
            // pin the issue order: the next block's LDS reads ride behind this block's first
            // MFMAs instead of being sunk in front of their consumers (which exposes the
            // LDS latency once per 4 MFMAs)
            if (kb + 1 < BK / 8) {
#define MRCNN_SGB_STEP(q)                                         \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            \
    __builtin_amdgcn_sched_group_barrier(0x100, (NR + 3 - (q)) / 4, 0);
                MRCNN_SGB_STEP(0) MRCNN_SGB_STEP(1) MRCNN_SGB_STEP(2) MRCNN_SGB_STEP(3)
#undef MRCNN_SGB_STEP
                __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - 4, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, NMFMA, 0);
            }
        }
        if (MRCNN_GEMM_SETPRIO == 1) __builtin_amdgcn_s_setprio(0);
        if (MRCNN_GEMM_SETPRIO == 2) __builtin_amdgcn_s_setprio(2);
    };
    auto compute = [&](int buf) { compute_(buf, std::true_type()); };

#ifdef MRCNN_GEMM_CLOCKPROBE
    const unsigned probe_slot = blockIdx.x + blockIdx.y * gridDim.x + blockIdx.z * gridDim.x * gridDim.y;
    if (tid == 0 && probe_slot < (unsigned)kProbeSlots) g_probe2[probe_slot * 8] = __builtin_amdgcn_s_memrealtime();
#endif
    if (MRCNN_GEMM_SETPRIO == 2) __builtin_amdgcn_s_setprio(2);
    if (nslices > 0) {
        load_slice(0);
        issue_loads();
        store_slice(0);
    }
    if constexpr (W8) {
        // (see the W8 note above the kernel) stage k & 1 holds slice k; slice k + 1 is in registers
        // (or in flight) while slice k is multiplied
        auto no_loads = [&]() {
#pragma unroll
            for (int i = 0; i < NA; ++i) oa[i] = kOOB;
#pragma unroll
            for (int i = 0; i < NB; ++i) ob[i] = kOOB;
        };
        if (nslices > 1) {
            load_slice(1);
            issue_loads();
        }
        __syncthreads();
        const bool stage_first = wave >= 4;               // wave-uniform
        for (int kt = 0; kt < nslices; ++kt) {
            const int cur = kt & 1;
            const bool more = kt + 1 < nslices;
            if (stage_first) {
                if (more) store_slice(cur ^ 1);
                if (kt + 2 < nslices) load_slice(kt + 2); else no_loads();
                compute_(cur, std::true_type());          // (issues the loads of slice k + 2)
            } else {
                compute_(cur, std::false_type());
                if (more) store_slice(cur ^ 1);
                if (kt + 2 < nslices) {
                    load_slice(kt + 2);
                    issue_loads();
                }
            }
            __syncthreads();
        }
    } else if (SINGLEBUF) {
        // one LDS stage (37 KB -> three workgroups per CU, three waves per SIMD): a wave spends
        // ~40 % of a slice issuing MFMAs and ~60 % staging, so three interleaved waves are
        // needed to keep the pipe full; two barriers per slice instead of one.
#ifdef MRCNN_GEMM_TRACE
        // developer instrumentation: per-phase s_memtime stamps of every wave of 64 workgroups
        const unsigned tr_lin = blockIdx.x + blockIdx.y * gridDim.x;
        unsigned long long *tr1 = g_trace + ((size_t)(tr_lin & 63) * 4 + wave) * 64 * 5;
        const bool tr1_on = tr_lin >= 256 && tr_lin < 256 + 64 && lane == 0 && blockIdx.z == 0;
#define TRACE1(slot)                                                             \
    if (tr1_on && kt >= 4 && kt < 68) {                                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       \
        tr1[(kt - 4) * 5 + slot] = __builtin_amdgcn_s_memtime();                 \
    }
#else
#define TRACE1(slot)
#endif
        for (int kt = 0; kt < nslices; ++kt) {
            TRACE1(0)
#ifndef MRCNN_DBG_NOSTAGE     // ablation: MFMA + fragment reads only (results are garbage)
            if (kt > 0) {
                __syncthreads();          // every wave is done reading the stage
#ifdef MRCNN_GEMM_TRACE
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                TRACE1(1)
                store_slice(0);
            }
            __syncthreads();
#endif
            TRACE1(2)
#ifndef MRCNN_DBG_NOGLOBAL    // ablation: no global loads in the loop (results are garbage)
            if (kt + 1 < nslices) {
                load_slice(kt + 1);
            } else if constexpr (ILV) {      // (compute() issues the loads: none left)
#pragma unroll
                for (int i = 0; i < NA; ++i) oa[i] = kOOB;
#pragma unroll
                for (int i = 0; i < NB; ++i) ob[i] = kOOB;
            }
#endif
            TRACE1(3)
            compute(0);
            TRACE1(4)
        }
#undef TRACE1
    } else {
        __syncthreads();
#ifdef MRCNN_GEMM_TRACE
        // developer instrumentation: per-phase s_memtime stamps of one wave per block
        unsigned long long *tr = g_trace + ((size_t)blockIdx.x * 4 + wave) * 64 * 5;
        const bool tr_on = blockIdx.x < 64 && lane == 0 && blockIdx.y == 0;
#define TRACE_STAMP(slot)                                                        \
    if (tr_on && kt >= 8 && kt < 72) {                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       \
        tr[(kt - 8) * 5 + slot] = __builtin_amdgcn_s_memtime();                  \
    }
#else
#define TRACE_STAMP(slot)
#endif
        for (int kt = 0; kt < nslices; ++kt) {
            const bool more = kt + 1 < nslices;
            TRACE_STAMP(0)
            if (more) load_slice(kt + 1);
            TRACE_STAMP(1)
            compute(kt & 1);
            TRACE_STAMP(2)
#ifdef MRCNN_GEMM_TRACE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            TRACE_STAMP(3)
            if (more) store_slice((kt + 1) & 1);
            __syncthreads();
            TRACE_STAMP(4)
        }
#undef TRACE_STAMP
    }

#ifdef MRCNN_GEMM_CLOCKPROBE
    if (tid == 0 && probe_slot < (unsigned)kProbeSlots) g_probe2[probe_slot * 8 + 1] = __builtin_amdgcn_s_memrealtime();
#endif
    // ---------------- epilogue ------------------------------------------------------
    // Per 32x32 MFMA tile: compute the 16 element offsets, issue every auxiliary load
    // (residual / accumulate / shortcut gradient) back to back, then combine and store.
    const float *out_base = tail ? p.tail_ws + (int64_t)split * p.tail_stride
                                 : p.C + zb * p.batch_c + (int64_t)split * p.split_stride;
    const __amdgpu_buffer_rsrc_t rC = make_rsrc(out_base, tail ? p.tail_bytes : p.c_bytes);
    const int e_flags = tail ? 0 : p.flags;              // tail pieces store raw partial sums
    const int e_row0 = tail ? p.tail_row0 : p.out_row0;
    const int e_ldc = tail ? p.N : p.ldc;
    const bool slab_rows = MODE != WGRAD && split_len > 0;   // slabs are indexed by GEMM row
    const __amdgpu_buffer_rsrc_t rRes = make_rsrc(p.residual, p.c_bytes);
    const __amdgpu_buffer_rsrc_t rResG = make_rsrc(p.res_g, p.c_bytes);
    const __amdgpu_buffer_rsrc_t rResY = make_rsrc(p.res_y, p.c_bytes);
    const __amdgpu_buffer_rsrc_t rOutM = make_rsrc(p.out_mask_y, p.c_bytes);
#ifdef MRCNN_DBG_PLAIN_EPI     // ablation: no fused epilogue arithmetic at all (plain stores)
    constexpr bool f_bias = false, f_aff = false, f_res = false, f_relu = false, f_acc = false;
    constexpr bool f_resg = false, f_resy = false, f_outm = false;
#else
    const bool f_bias = (e_flags & MRCNN_EPI_BIAS) != 0, f_aff = (e_flags & MRCNN_EPI_AFFINE) != 0;
    const bool f_res = (e_flags & MRCNN_EPI_RESIDUAL) != 0, f_relu = (e_flags & MRCNN_EPI_RELU) != 0;
    const bool f_acc = (e_flags & MRCNN_EPI_ACCUM) != 0;
    const bool f_resg = MODE != WGRAD && !tail && p.res_g != nullptr;
    const bool f_resy = f_resg && p.res_y != nullptr;
    const bool f_outm = MODE != WGRAD && !tail && p.out_mask_y != nullptr;
#endif
    constexpr int EG = 8;       // accumulator rows handled per batch of auxiliary loads

    // (PW / W8 launches write plain rows in natural order by the host's rule)
    if (MODE == FWD && TM >= 2 && MRCNN_GEMM_WIDE_EPILOGUE != 0 && (GEO || p.out_mode == OUT_PLAIN)) {   // (uniform)
      if constexpr (MODE == FWD && TM >= 2 && MRCNN_GEMM_WIDE_EPILOGUE != 0) {
        // Forward-form launches: the accumulators (one column x 16 rows per lane) are turned
        // into row-major float4s through the wave's corner of the LDS stages, so the residual
        // / mask reads and the output stores are 16 B per lane — a quarter of the memory
        // instructions of the per-element path below.  K-short layers (a bottleneck's conv3:
        // 16 K slices, 411 MB written + 411 MB residual read) spend a quarter of their time here.
        constexpr int CW = 32 * TN;                 // columns of the wave's quadrant
        constexpr int LDW = CW + 4;                 // padded LDS row
        constexpr int F4 = CW / 4, RPI = 64 / F4;   // float4 per row, rows per pass of the wave
        constexpr int NK = 32 / RPI;                // passes per 32-row half
        constexpr int QG = TM >= 2 ? 4 : 2;         // passes whose loads are in flight together
        __syncthreads();                            // every wave is done with the K loop's LDS
#ifdef MRCNN_GEMM_CLOCKPROBE
#define PROBE2(k) if (tid == 0 && probe_slot < (unsigned)kProbeSlots) g_probe2[probe_slot * 8 + (k)] = __builtin_amdgcn_s_memrealtime();
#else
#define PROBE2(k)
#endif
        PROBE2(2)
        float *ep = &smem_all[0][0][0] + wave * (32 * LDW);
        const int c4 = lane % F4, r_in = lane / F4;
        const int col = n0 + wn * CW + c4 * 4;
        const bool col_ok = col < p.N;
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f), scale4 = make_float4(1.f, 1.f, 1.f, 1.f);
        float4 shift4 = bias4;
        if (col_ok) {
            if (f_bias) bias4 = *reinterpret_cast<const float4 *>(p.bias + col);
            if (f_aff) {
                scale4 = *reinterpret_cast<const float4 *>(p.scale + col);
                if (p.shift) shift4 = *reinterpret_cast<const float4 *>(p.shift + col);
            }
        }
        // The combine step is specialised at compile time for the flag combinations the model
        // launches (F >= 0: bit k of F = flag k below), selected by ONE workgroup-uniform switch:
        // with run-time flags every element went through a chain of eight add / select pairs
        // (about 120 VALU instructions per 16-byte store, a quarter of a 16-slice tile's
        // lifetime); F = -1 keeps that generic path for any other combination.  The operation
        // order per element is the same in every instantiation, so results are bit-identical.
        enum { C_BIAS = 1, C_AFF = 2, C_RES = 4, C_RELU = 8, C_ACC = 16, C_RESG = 32, C_RESY = 64,
               C_OUTM = 128 };
        auto run = [&](auto tag) {
            constexpr int F = decltype(tag)::value;
            const bool c_bias = F >= 0 ? (F & C_BIAS) != 0 : f_bias;
            const bool c_aff = F >= 0 ? (F & C_AFF) != 0 : f_aff;
            const bool c_res = F >= 0 ? (F & C_RES) != 0 : f_res;
            const bool c_relu = F >= 0 ? (F & C_RELU) != 0 : f_relu;
            const bool c_acc = F >= 0 ? (F & C_ACC) != 0 : f_acc;
            const bool c_resg = F >= 0 ? (F & C_RESG) != 0 : f_resg;
            const bool c_resy = F >= 0 ? (F & C_RESY) != 0 : f_resy;
            const bool c_outm = F >= 0 ? (F & C_OUTM) != 0 : f_outm;
            const bool natural = F >= 0 || GEO || !(gPermN > 0 && !slab_rows);   // F >= 0: natural row order
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        ep[((e & 3) + 8 * (e >> 2) + 4 * lk) * LDW + j * 32 + li] = acc[i][j][e];
                // same wave writes and reads: LDS operations of a wave execute in order
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (i == 0) { PROBE2(3) } else { PROBE2(5) }
#pragma unroll
                for (int kg = 0; kg < NK; kg += QG) {
                    unsigned off[QG];
                    float4 v[QG], a0[QG], a1[QG], a2[QG], a3[QG];
#pragma unroll
                    for (int q = 0; q < QG; ++q) {
                        const int r = (kg + q) * RPI + r_in;
                        v[q] = *reinterpret_cast<const float4 *>(ep + r * LDW + c4 * 4);
                        const int row = m0 + wm * (32 * TM) + i * 32 + r;
                        int orow;
                        if (!natural) {
                            const PermRow pr = perm_row(row < p.M ? row : 0, p.gp * p.gq);
                            orow = pr.n < gPermN ? pr.n * (p.gp * p.gq) + pr.pos : -1;
                        } else {
                            orow = row - e_row0;
                        }
                        const bool ok = col_ok && row < p.M && orow >= 0;
                        off[q] = ok ? 4u * (unsigned)(orow * e_ldc + col) : kOOB;
                    }
                    if (c_res) {
#pragma unroll
                        for (int q = 0; q < QG; ++q) a0[q] = bload4(rRes, off[q]);
                    }
                    if (c_acc) {
#pragma unroll
                        for (int q = 0; q < QG; ++q) a1[q] = bload4(rC, off[q]);
                    }
                    if (c_resg) {
#pragma unroll
                        for (int q = 0; q < QG; ++q) a0[q] = bload4(rResG, off[q]);
                    }
                    if (c_resy) {
#pragma unroll
                        for (int q = 0; q < QG; ++q) a2[q] = bload4(rResY, off[q]);
                    }
                    if (c_outm) {
#pragma unroll
                        for (int q = 0; q < QG; ++q) a3[q] = bload4(rOutM, off[q]);
                    }
#pragma unroll
                    for (int q = 0; q < QG; ++q) {
                        float x[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
                        const float b[4] = {bias4.x, bias4.y, bias4.z, bias4.w};
                        const float sc[4] = {scale4.x, scale4.y, scale4.z, scale4.w};
                        const float sh[4] = {shift4.x, shift4.y, shift4.z, shift4.w};
                        const float r0[4] = {a0[q].x, a0[q].y, a0[q].z, a0[q].w};
                        const float r1[4] = {a1[q].x, a1[q].y, a1[q].z, a1[q].w};
                        const float r2[4] = {a2[q].x, a2[q].y, a2[q].z, a2[q].w};
                        const float r3[4] = {a3[q].x, a3[q].y, a3[q].z, a3[q].w};
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            float y = x[t];
                            if (c_bias) y += b[t];
                            if (c_aff) y = y * sc[t] + sh[t];
                            if (c_res) y += r0[t];
                            if (c_acc) y += r1[t];
                            if (c_resy) y += r2[t] > 0.f ? r0[t] : 0.f;
                            else if (c_resg) y += r0[t];
                            if (c_relu) y = fmaxf(y, 0.f);
                            if (c_outm) y = r3[t] > 0.f ? y : 0.f;
                            x[t] = y;
                        }
                        bstore4(rC, off[q], make_float4(x[0], x[1], x[2], x[3]));
                    }
                }
                if (i == 0) { PROBE2(4) } else { PROBE2(6) }
            }
        };
        const int combo = (f_bias ? C_BIAS : 0) | (f_aff ? C_AFF : 0) | (f_res ? C_RES : 0) |
                          (f_relu ? C_RELU : 0) | (f_acc ? C_ACC : 0) | (f_resg ? C_RESG : 0) |
                          (f_resy ? C_RESY : 0) | (f_outm ? C_OUTM : 0);
        const bool permuted = !GEO && gPermN > 0 && !slab_rows;
#define MRCNN_EPI_CASE(F) case (F): run(std::integral_constant<int, (F)>()); break;
        switch (permuted ? -1 : combo) {       // workgroup-uniform
            MRCNN_EPI_CASE(0)
            MRCNN_EPI_CASE(C_AFF)
            MRCNN_EPI_CASE(C_AFF | C_RELU)
            MRCNN_EPI_CASE(C_AFF | C_RES | C_RELU)
            MRCNN_EPI_CASE(C_BIAS)
            MRCNN_EPI_CASE(C_BIAS | C_RELU)
            MRCNN_EPI_CASE(C_AFF | C_OUTM)
            MRCNN_EPI_CASE(C_RESG | C_OUTM)
            MRCNN_EPI_CASE(C_RESG)
            MRCNN_EPI_CASE(C_ACC | C_OUTM)
            MRCNN_EPI_CASE(C_ACC)
        default: run(std::integral_constant<int, -1>()); break;
        }
#undef MRCNN_EPI_CASE
        return;
      }
    }
    if constexpr (WROWPERM) {
        // tile rows / columns are in plane-row order: fragment row 32 c + l = channel 4 l + c.  A
        // lane's two column tiles (j = 0, 1) are the ADJACENT columns 4 li + 2 wn + j: one 8-byte
        // store per accumulator row.
        static_assert(TM == 2 && TN == 2, "SPLIT WGRAD: 128x128 tiles");
        const int col = n0 + 4 * li + 2 * wn;
        const bool col_ok = col < p.N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int g = 0; g < 16 / EG; ++g) {
                unsigned off[EG];
                float row_scale[EG];
#pragma unroll
                for (int q = 0; q < EG; ++q) {
                    const int e = g * EG + q;
                    const int row = m0 + 4 * ((e & 3) + 8 * (e >> 2) + 4 * lk) + 2 * wm + i;
                    off[q] = (col_ok && row < p.M) ? 4u * (unsigned)(row * e_ldc + col) : kOOB;
                    row_scale[q] = (p.scale && row < p.M) ? p.scale[row] : 1.f;
                }
#pragma unroll
                for (int q = 0; q < EG; ++q) {
                    const float v0 = acc[i][0][g * EG + q] * row_scale[q];
                    const float v1 = acc[i][1][g * EG + q] * row_scale[q];
                    bstore8(rC, off[q], __float_as_uint(v0), __float_as_uint(v1));
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * (32 * TN) + j * 32 + li;
        const bool col_ok = col < p.N;
        const int colc = col_ok ? col : 0;
        float bias = 0.f, scale = 1.f, shift = 0.f;
        if (MODE != WGRAD) {
            if (f_bias) bias = p.bias[!GEO && p.out_mode == OUT_DECONV ? colc % p.ko : colc];
            if (f_aff) { scale = p.scale[colc]; shift = p.shift ? p.shift[colc] : 0.f; }
        }
        int col_off = colc;
        if (MODE != WGRAD && !GEO && p.out_mode == OUT_DECONV) {
            const int ab = colc / p.ko, o = colc - ab * p.ko;
            col_off = ((ab >> 1) * (2 * p.gq) + (ab & 1)) * p.ko + o;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int g = 0; g < 16 / EG; ++g) {  // groups of EG accumulator rows
                unsigned off[EG];
#pragma unroll
                for (int q = 0; q < EG; ++q) {
                    const int e = g * EG + q;
                    const int row = m0 + wm * (32 * TM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                    int o;
                    if (MODE != WGRAD && !GEO && p.out_mode != OUT_PLAIN) {
                        const int rr = row < p.M ? row : 0;
                        const int n = rr / (p.gp * p.gq);
                        const int rem = rr - n * (p.gp * p.gq);
                        const int gy = rem / p.gq, gx = rem - gy * p.gq;
                        if (p.out_mode == OUT_STRIDED)
                            o = ((n * p.oh + gy * (p.ostride ? p.ostride : gStride)) * p.ow +
                                 gx * (p.ostride ? p.ostride : gStride)) * p.ldc + col_off;
                        else
                            o = ((n * (2 * p.gp) + 2 * gy) * (2 * p.gq) + 2 * gx) * p.ko + col_off;
                    } else if (FWDLIKE && !GEO && gPermN > 0 && !slab_rows) {
                        // position-major GEMM row -> (image, position) row of the NHWC tensor
                        // (split-K slabs stay indexed by GEMM row; the slab-sum kernel maps)
                        const PermRow pr = perm_row(row < p.M ? row : 0, p.gp * p.gq);
                        o = pr.n < gPermN ? (pr.n * (p.gp * p.gq) + pr.pos) * p.ldc + col_off : -1;
                    } else {
                        o = (row - e_row0) * e_ldc + col_off;
                    }
                    off[q] = (col_ok && row < p.M && o >= 0) ? 4u * (unsigned)o : kOOB;
                }
                float aux0[EG], aux1[EG], aux2[EG], aux3[EG];
                if (MODE != WGRAD) {
                    if (f_res) {
#pragma unroll
                        for (int q = 0; q < EG; ++q) aux0[q] = bload1(rRes, off[q]);
                    }
                    if (f_acc) {
#pragma unroll
                        for (int q = 0; q < EG; ++q) aux1[q] = bload1(rC, off[q]);
                    }
                    if (f_resg) {
#pragma unroll
                        for (int q = 0; q < EG; ++q) aux0[q] = bload1(rResG, off[q]);
                    }
                    if (f_resy) {
#pragma unroll
                        for (int q = 0; q < EG; ++q) aux2[q] = bload1(rResY, off[q]);
                    }
                    if (f_outm) {
#pragma unroll
                        for (int q = 0; q < EG; ++q) aux3[q] = bload1(rOutM, off[q]);
                    }
                }
                float row_scale[EG];
                if (MODE == WGRAD && p.scale) {
#pragma unroll
                    for (int q = 0; q < EG; ++q) {
                        const int e = g * EG + q;
                        const int row = m0 + wm * (32 * TM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                        row_scale[q] = row < p.M ? p.scale[row] : 0.f;
                    }
                }
#pragma unroll
                for (int q = 0; q < EG; ++q) {
                    float v = acc[i][j][g * EG + q];
                    if (MODE != WGRAD) {
                        if (f_bias) v += bias;
                        if (f_aff) v = v * scale + shift;
                        if (f_res) v += aux0[q];
                        if (f_acc) v += aux1[q];
                        if (f_resy) v += aux2[q] > 0.f ? aux0[q] : 0.f;
                        else if (f_resg) v += aux0[q];
                        if (f_relu) v = fmaxf(v, 0.f);
                        if (f_outm) v = aux3[q] > 0.f ? v : 0.f;
                    } else if (p.scale) {
                        v *= row_scale[q];
                    }
                    bstore1(rC, off[q], v);
                }
            }
        }
    }

#if MRCNN_WGRAD_INKERNEL_REDUCE
    if (MODE == WGRAD && p.tile_counters != nullptr) {
        // Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): plain stores ->
        // barrier -> lane 0: agent-scope release, explicit vmcnt(0), relaxed agent atomic on
        // the tile's counter; the workgroup that observes splits-1 earlier arrivals performs
        // ONE agent-scope acquire, then everybody reads the slabs with plain loads.
        const int nsplits = (int)gridDim.y;
        __syncthreads();                 // every wave is done with the LDS stages: reuse a word
        int &s_last = *reinterpret_cast<int *>(&smem_all[0][0][0]);
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int old = __hip_atomic_fetch_add(&p.tile_counters[tile], 1, __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == nsplits - 1;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                p.tile_counters[tile] = 0;       // ready for the next launch
            }
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
        const __amdgpu_buffer_rsrc_t rOut = make_rsrc(p.reduce_out, p.c_bytes);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (32 * TN) + j * 32 + li;
            const bool col_ok = col < p.N;
            const int colc = col_ok ? col : 0;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    unsigned off[8];
                    float sum[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int e = g * 8 + q;
                        const int row = m0 + wm * (32 * TM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lk;
                        off[q] = (col_ok && row < p.M) ? 4u * (unsigned)(row * p.ldc + colc) : kOOB;
                        sum[q] = 0.f;
                    }
                    for (int sp = 0; sp < nsplits; ++sp) {       // fixed order: deterministic
                        const __amdgpu_buffer_rsrc_t rS =
                            make_rsrc(p.C + (int64_t)sp * p.split_stride, p.c_bytes);
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = bload1(rS, off[q]);
#pragma unroll
                        for (int q = 0; q < 8; ++q) sum[q] += v[q];
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) bstore1(rOut, off[q], sum[q]);
                }
            }
        }
    }
#endif
}

__global__ void splitk_reduce_kernel(const float *__restrict__ ws, int splits, int64_t n,
                                     int64_t stride, float *__restrict__ out)
{
    const int64_t nv = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv;
         i += (int64_t)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<const float4 *>(ws)[i];
        int s = 1;
        // four slab loads in flight per step; the additions keep the slab order
        for (; s + 4 <= splits; s += 4) {
            const float4 v0 = reinterpret_cast<const float4 *>(ws + (s + 0) * stride)[i];
            const float4 v1 = reinterpret_cast<const float4 *>(ws + (s + 1) * stride)[i];
            const float4 v2 = reinterpret_cast<const float4 *>(ws + (s + 2) * stride)[i];
            const float4 v3 = reinterpret_cast<const float4 *>(ws + (s + 3) * stride)[i];
            a.x += v0.x; a.y += v0.y; a.z += v0.z; a.w += v0.w;
            a.x += v1.x; a.y += v1.y; a.z += v1.z; a.w += v1.w;
            a.x += v2.x; a.y += v2.y; a.z += v2.z; a.w += v2.w;
            a.x += v3.x; a.y += v3.y; a.z += v3.z; a.w += v3.w;
        }
        for (; s < splits; ++s) {
            const float4 v = reinterpret_cast<const float4 *>(ws + s * stride)[i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        reinterpret_cast<float4 *>(out)[i] = a;
    }
}

// Workgroups resident at once: 128x128 tiles run 2 per CU (73 KB LDS), 64x64 tiles 4 per CU.
constexpr int64_t kSlotsBig = 512, kSlotsSmall = 1024;

int g_extra_lds = 0;   // developer knob: dynamic LDS bytes added to every GEMM launch (lowers
                       // the resident workgroups per CU for co-residency experiments)

int g_split_bf16 = 3;  // mrcnn_set_tuning("split_bf16"): bit 0 = 128x128 kernels, bit 1 = 64x64 forward form on the
                       // split-operand arithmetic (see SPLIT; the default since round 4), 0 = fp32 MFMA everywhere
int g_stagger = 0;    // mrcnn_set_tuning("stagger", percent of the nominal start-up stagger; 0 = off)
int g_big_split_k = -1; // mrcnn_set_tuning("big_split_k"): small-M problems as 128x128 tiles cut along K.
                      // -1 (default since round 6) = the one-round rule in launch(), 0 = off (64x64 tiles),
                      // k > 0 = aim at k workgroups
int g_stagger_min_rounds = 2;
int g_big_split_min_slices = 16;   // mrcnn_set_tuning("big_split_min_slices"): fewest K slices per slab of the one-round rule
int g_w8_min_k = 256; // mrcnn_set_tuning("w8_min_k"): shallowest K (input channels) a W8 launch takes
int g_w8 = 1;         // mrcnn_set_tuning("w8", 0/1): 256x128 tiles on 512-thread workgroups (W8) for the large
                      // pointwise forward-form launches of the split-operand arithmetic

int g_pw = 3;         // mrcnn_set_tuning("pw"): bit 0 = the PW instantiations for pointwise forward-form launches,
                      // bit 1 = the K3 instantiations for the 3x3 / stride 1 / pad 1 ones

// PW's launch rule (see the note above conv_gemm_kernel)
inline bool pw_plain(const GemmParams &p)
{
    return p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0 && p.perm_n == 0 && !p.stem && p.gp == p.sh &&
           p.gq == p.sw && p.out_mode == OUT_PLAIN;
}

// K3's launch rule
inline bool k3_plain(const GemmParams &p)
{
    return p.R == 3 && p.S == 3 && p.stride == 1 && p.pad == 1 && p.perm_n == 0 && !p.stem &&
           p.out_mode == OUT_PLAIN;
}

template <int TM, int TN, int MODE, bool MASKED>
void launch_kernel_m(const GemmParams &p0, int64_t tiles, int splits, hipStream_t s, int batch = 1)
{
    GemmParams p = p0;
    {
        // resident workgroups per CU of this instantiation (registers / LDS, see min_blocks)
        const int slots = TM == 4 ? 2 : single_buffered(TM, MODE, MASKED) ? (TM == 2 ? 3 : 6) : (TM == 2 ? 2 : 4);
        const int64_t wgs = tiles * splits * batch;
        // K slices one workgroup walks and the matrix-pipe cycles of one of its waves per slice
        const int64_t slices = MODE == WGRAD ? mrcnn::ceil_div(p.split_len, BK)
                               : (p.split_len > 0 ? p.split_len
                                                  : (int64_t)p.R * p.S * mrcnn::ceil_div(p.stem ? 32 : p.Kc, BK));
        const int64_t slice_cycles = 4 * TM * TN * (BK / 8) * 64;
        p.stagger_slots = 0; p.stagger_cycles = 0;
        if (g_stagger > 0 && wgs >= 256ll * slots) {
            int64_t cyc;
            if (wgs >= (int64_t)g_stagger_min_rounds * 256 * slots)
                cyc = slices * slice_cycles;       // a third (1 / slots) of a tile's lifetime
            else
                cyc = slice_cycles + 512;          // one round only: interleave the slice phases
            cyc = cyc * g_stagger / 100;
            p.stagger_slots = slots;
            p.stagger_cycles = (int)std::min<int64_t>(cyc, 1 << 20);
        }
    }
    hipEvent_t ev0, ev1;          // kernel-only timing (ProfKernelScope of the caller), usually null
    mrcnn::prof_take(&ev0, &ev1);
    if constexpr (MODE == WGRAD && TM == 2 && TN == 2) {
        if ((g_split_bf16 & 1) && p.perm_n == 0) {
            hipExtLaunchKernelGGL((conv_gemm_kernel<TM, TN, MODE, MASKED, false, true>),
                               dim3((unsigned)tiles, splits, batch), dim3(256), g_extra_lds, s, ev0, ev1, 0,
                                  p);
            return;
        }
    }
    if constexpr (MODE == WGRAD && !MASKED) {
        if (p.perm_n > 0) {
            hipExtLaunchKernelGGL((conv_gemm_kernel<TM, TN, MODE, MASKED, true>),
                               dim3((unsigned)tiles, splits, batch), dim3(256), g_extra_lds, s, ev0, ev1, 0,
                                  p);
            return;
        }
    }
    if constexpr (MODE == FWD && TM == TN && (TM == 1 || TM == 2)) {
        if (g_split_bf16 & (TM == 2 ? 1 : 2)) {
            if ((g_pw & 1) && pw_plain(p))
                hipExtLaunchKernelGGL((conv_gemm_kernel<TM, TN, MODE, MASKED, false, true, false, true>),
                                      dim3((unsigned)tiles, splits, batch), dim3(256), g_extra_lds, s, ev0, ev1,
                                      0, p);
            else if ((g_pw & 2) && k3_plain(p))
                hipExtLaunchKernelGGL((conv_gemm_kernel<TM, TN, MODE, MASKED, false, true, false, false, true>),
                                      dim3((unsigned)tiles, splits, batch), dim3(256), g_extra_lds, s, ev0, ev1,
                                      0, p);
            else
                hipExtLaunchKernelGGL((conv_gemm_kernel<TM, TN, MODE, MASKED, false, true>),
                                      dim3((unsigned)tiles, splits, batch), dim3(256), g_extra_lds, s, ev0, ev1,
                                      0, p);
            return;
        }
    }
    hipExtLaunchKernelGGL((conv_gemm_kernel<TM, TN, MODE, MASKED>),
                       dim3((unsigned)tiles, splits, batch), dim3(256), g_extra_lds, s, ev0, ev1, 0,
                                  p);
}

inline bool is_masked(const GemmParams &p) { return p.mask_y != nullptr || p.in_scale != nullptr; }

// ---- W8 launches (see the note above conv_gemm_kernel) -------------------------------------------
constexpr int kW8BM = 256, kW8BN = 128;

inline bool w8_pointwise(const GemmParams &p)
{
    return p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0 && p.perm_n == 0 && !p.stem &&
           p.gp == p.sh && p.gq == p.sw;
}

// a forward-form launch qualifies: split-operand arithmetic, no mask staging, plain output rows,
// 1x1 / stride 1 gather, K >= 8 slices, at least three rounds of 256 tiles, rows that fill their tiles
inline bool w8_ok(const GemmParams &p, int batch = 1)
{
    if (!g_w8 || !(g_split_bf16 & 1) || is_masked(p) || p.out_mode != OUT_PLAIN || !w8_pointwise(p) ||
        p.split_len != 0 || p.m_lo != 0)
        return false;
    const int64_t tm = mrcnn::ceil_div(p.M, kW8BM), tn = mrcnn::ceil_div(p.N, kW8BN);
    // (measured, profiles/r06b_w8_shapes.txt: the RoI head's 1x1 layers and their transposed-filter data
    // gradients gain 4 - 7 %; the batch-2 backbone's 270 - 530-tile, 2 - 8-slice launches lose 5 - 20 %)
    return p.N >= kW8BN && p.Kc >= g_w8_min_k && tm * tn * batch >= 768 && (p.M % kW8BM == 0 || p.M >= 16 * kW8BM);
}

void launch_w8_kernel(const GemmParams &p, int64_t wgs, int batch, hipStream_t s)
{
    hipEvent_t ev0, ev1;
    mrcnn::prof_take(&ev0, &ev1);
    hipExtLaunchKernelGGL((conv_gemm_kernel<2, 2, FWD, false, false, true, true>),
                          dim3((unsigned)wgs, 1, batch), dim3(512), g_extra_lds, s, ev0, ev1, 0, p);
}

template <int TM, int TN, int MODE>
void launch_kernel(const GemmParams &p, int64_t tiles, int splits, hipStream_t s, int batch = 1)
{
    if (is_masked(p)) launch_kernel_m<TM, TN, MODE, true>(p, tiles, splits, s, batch);
    else launch_kernel_m<TM, TN, MODE, false>(p, tiles, splits, s, batch);
}

template <int TM, int TN, int MODE>
void launch_tiles(GemmParams p, int m_lo, int m_hi, int splits, hipStream_t s)
{
    constexpr int BM = 64 * TM, BN = 64 * TN;
    p.m_lo = m_lo;
    p.M = m_hi;
    const int64_t blocks = mrcnn::ceil_div(m_hi - m_lo, BM) * mrcnn::ceil_div(p.N, BN);
    if (blocks <= 0) return;
    const int64_t rows = m_hi - m_lo;
    const double kdepth = (double)p.R * p.S * (p.stem ? 21.0 : (double)p.Kc);
    const double flops = 2.0 * rows * p.N * kdepth;
    const double bytes = 4.0 * ((double)rows * p.N + (double)rows * p.Kc + (double)p.N * kdepth);
    // profiler buckets follow the kernel SYMBOL (what rocprofv3 reports): the forward-form
    // instantiation runs forward convolutions and the transposed-filter dgrads alike
    {
        mrcnn::ProfKernelScope prof((MODE == FWD ? mrcnn::PROF_CONV_FWD_128 : mrcnn::PROF_CONV_DGRAD_128) +
                                        (TM >= 2 ? 0 : 1),
                                    flops, bytes);
        launch_kernel<TM, TN, MODE>(p, blocks, splits, s);
    }
}

// ---- split-K for the leftover rows of a small-M forward / dgrad ----------------------------
// 64x64 tiles are all resident at once, so a launch of T tiles costs ceil(T / 256) tile-times
// on the busiest CU (T = 536 -> 3, although the average CU has 2.09).  The rows of the whole
// multiples of 256 tiles run as usual; the leftover rows (< 154 tiles) are cut along K into
// `splits` short workgroups that spread over all CUs, write raw partial sums into slabs, and
// a small kernel sums the slabs in order and applies the epilogue (deterministic).
constexpr int64_t kSplitWsBytes = 192ll << 20;   // >= 154 leftover tiles x 16 slabs; whole small-M outputs x splits

struct FixParams {
    const float *ws;
    float *C;
    const float *bias, *scale, *shift, *residual, *res_g, *res_y, *out_mask_y;
    int splits, rows, N, row0, ldc, flags;
    int perm_n, pq;      // position-major GEMM rows (see GemmParams::perm_n), positions per image
    int64_t stride;
};

__global__ void __launch_bounds__(256) splitk_epilogue_kernel(const FixParams f)
{
    const int n4 = f.N / 4;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)f.rows * n4) return;
    const int r = (int)(i / n4), c = (int)(i - (int64_t)r * n4) * 4;
    float4 a = *reinterpret_cast<const float4 *>(f.ws + (int64_t)r * f.N + c);
    for (int s = 1; s < f.splits; ++s) {
        const float4 v = *reinterpret_cast<const float4 *>(f.ws + s * f.stride + (int64_t)r * f.N + c);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
    int orow = f.row0 + r;
    if (f.perm_n > 0) {
        const PermRow pr = perm_row(orow, f.pq);
        if (pr.n >= f.perm_n) return;            // padding row of the last image block
        orow = pr.n * f.pq + pr.pos;
    }
    const int64_t off = (int64_t)orow * f.ldc + c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float x = v[k];
        if (f.flags & MRCNN_EPI_BIAS) x += f.bias[c + k];
        if (f.flags & MRCNN_EPI_AFFINE) x = x * f.scale[c + k] + (f.shift ? f.shift[c + k] : 0.f);
        if (f.flags & MRCNN_EPI_RESIDUAL) x += f.residual[off + k];
        if (f.flags & MRCNN_EPI_ACCUM) x += f.C[off + k];
        if (f.res_g) x += (!f.res_y || f.res_y[off + k] > 0.f) ? f.res_g[off + k] : 0.f;
        if (f.flags & MRCNN_EPI_RELU) x = fmaxf(x, 0.f);
        if (f.out_mask_y) x = f.out_mask_y[off + k] > 0.f ? x : 0.f;
        v[k] = x;
    }
    *reinterpret_cast<float4 *>(f.C + off) = make_float4(v[0], v[1], v[2], v[3]);
}

template <int MODE>
bool can_split_rows(const GemmParams &p)
{
    return p.split_ws && MODE != WGRAD && p.out_mode == OUT_PLAIN && !p.stem && p.N % 4 == 0 &&
           p.ldc == p.N;
}

// rows [rows_lo, p.M) as 64 TM x 64 TM tiles cut along K into `splits` slabs + the ordered slab sum
template <int MODE, int TM = 1>
void launch_split_rows(const GemmParams &p, int rows_lo, int splits, int total_slices, hipStream_t s)
{
    const int rows_left = p.M - rows_lo;
    GemmParams q = p;
    q.C = p.split_ws;
    q.flags = 0;
    q.bias = q.scale = q.shift = q.residual = q.res_g = q.res_y = q.out_mask_y = nullptr;
    q.split_len = (int)mrcnn::ceil_div(total_slices, splits);
    splits = (int)mrcnn::ceil_div(total_slices, q.split_len);
    q.split_stride = (int64_t)rows_left * p.N;
    q.out_row0 = rows_lo;
    q.c_bytes = (unsigned)(q.split_stride * 4);
    launch_tiles<TM, TM, MODE>(q, rows_lo, p.M, splits, s);
    FixParams f = {};
    f.ws = p.split_ws; f.C = p.C;
    f.bias = p.bias; f.scale = p.scale; f.shift = p.shift; f.residual = p.residual;
    f.res_g = p.res_g; f.res_y = p.res_y; f.out_mask_y = p.out_mask_y;
    f.splits = splits; f.rows = rows_left; f.N = p.N; f.row0 = rows_lo; f.ldc = p.ldc;
    f.flags = p.flags; f.stride = q.split_stride;
    f.perm_n = MODE == FWD ? p.perm_n : 0; f.pq = p.gp * p.gq;
    const int64_t n = (int64_t)rows_left * (p.N / 4);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)mrcnn::ceil_div(n, 256)), dim3(256), 0,
                       s, f);
}

int g_fused_tail = 512;            // mrcnn_set_tuning("fused_tail", 0 = off, else target number of tail pieces)

// Rows [0, rows_main) as whole TMxTN tiles and rows [rows_main, p.M) as K-split pieces of the
// same tile shape in ONE launch (GemmParams::tail_*), then the ordered slab sum.  Returns false
// (nothing launched) when the problem does not qualify.
template <int TM, int TN, int MODE>
bool launch_fused_tail(GemmParams p, int rows_main, hipStream_t s)
{
    constexpr int BM = 64 * TM, BN = 64 * TN;
    if (!g_fused_tail || !can_split_rows<MODE>(p) || p.split_len != 0 || rows_main <= 0 ||
        rows_main % BM != 0 || rows_main >= p.M)
        return false;
    const int64_t ntn = mrcnn::ceil_div(p.N, BN);
    const int64_t main_tiles = (rows_main / BM) * ntn;
    const int rows_left = p.M - rows_main;
    const int64_t tail_tiles = mrcnn::ceil_div(rows_left, BM) * ntn;
    const int total_slices = p.R * p.S * (int)mrcnn::ceil_div(p.Kc, BK);
    int64_t splits = std::min<int64_t>(std::min<int64_t>(16, total_slices / 4),
                                       mrcnn::ceil_div(g_fused_tail, tail_tiles));
    while (splits > 1 && (int64_t)rows_left * p.N * splits * 4 > kSplitWsBytes) --splits;
    if (splits < 2) return false;
    p.tail_split_len = (int)mrcnn::ceil_div(total_slices, splits);
    splits = mrcnn::ceil_div(total_slices, p.tail_split_len);
    p.tail_first = (int)main_tiles;
    p.tail_splits = (int)splits;
    p.tail_row0 = rows_main;
    p.tail_ws = p.split_ws;
    p.tail_stride = (int64_t)rows_left * p.N;
    p.tail_bytes = (unsigned)(p.tail_stride * 4);
    p.m_lo = 0;
    {
        const double kdepth = (double)p.R * p.S * (double)p.Kc;
        const double flops = 2.0 * p.M * p.N * kdepth;
        const double bytes = 4.0 * ((double)p.M * p.N + (double)p.M * p.Kc + (double)p.N * kdepth);
        mrcnn::ProfKernelScope prof((MODE == FWD ? mrcnn::PROF_CONV_FWD_128 : mrcnn::PROF_CONV_DGRAD_128) +
                                        (TM >= 2 ? 0 : 1),
                                    flops, bytes);
        launch_kernel<TM, TN, MODE>(p, main_tiles + tail_tiles * splits, 1, s);
    }
    FixParams f = {};
    f.ws = p.split_ws; f.C = p.C;
    f.bias = p.bias; f.scale = p.scale; f.shift = p.shift; f.residual = p.residual;
    f.res_g = p.res_g; f.res_y = p.res_y; f.out_mask_y = p.out_mask_y;
    f.splits = (int)splits; f.rows = rows_left; f.N = p.N; f.row0 = rows_main; f.ldc = p.ldc;
    f.flags = p.flags; f.stride = p.tail_stride;
    f.perm_n = MODE == FWD ? p.perm_n : 0; f.pq = p.gp * p.gq;
    const int64_t n = (int64_t)rows_left * (p.N / 4);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)mrcnn::ceil_div(n, 256)), dim3(256), 0,
                       s, f);
    return true;
}

// The 64x64-tile remainder of a 128x128-tile launch (rows [rows_lo, M)) runs alone on the GPU
// after the main launch: few workgroups, each walking the whole K.  Cut along K so that about
// two workgroups per CU share the work.
template <int MODE>
void launch_remainder(const GemmParams &p, int rows_lo, hipStream_t s)
{
    const int64_t left_tiles = mrcnn::ceil_div(p.M - rows_lo, 64) * mrcnn::ceil_div(p.N, 64);
    const int total_slices = p.R * p.S * (int)mrcnn::ceil_div(p.Kc, BK);
    int64_t splits = std::min<int64_t>(std::min<int64_t>(16, total_slices / 8),
                                       mrcnn::ceil_div(512, left_tiles));
    while (splits > 1 && (int64_t)(p.M - rows_lo) * p.N * splits * 4 > kSplitWsBytes) --splits;
    if (!can_split_rows<MODE>(p) || splits < 2) {
        launch_tiles<1, 1, MODE>(p, rows_lo, p.M, 1, s);
        return;
    }
    launch_split_rows<MODE>(p, rows_lo, (int)splits, total_slices, s);
}

// One W8 launch for the whole problem: whole rounds of 256 tiles (one 512-thread workgroup per CU)
// and, when the last round would be less than ~60 % full, its rows as K-split pieces appended to
// the same grid + the ordered slab sum (the fused-tail scheme of launch_fused_tail).
void launch_w8(GemmParams p, hipStream_t s)
{
    const int64_t tm = mrcnn::ceil_div(p.M, kW8BM), tn = mrcnn::ceil_div(p.N, kW8BN);
    const int64_t T = tm * tn, full = T / 256, rem = T - full * 256;
    const double kdepth = (double)p.Kc;
    const double flops = 2.0 * p.M * p.N * kdepth;
    const double bytes = 4.0 * ((double)p.M * p.N + (double)p.M * p.Kc + (double)p.N * kdepth);
    const int total_slices = (int)mrcnn::ceil_div(p.Kc, BK);
    int64_t main_rows_tiles = tm;
    if (rem > 0 && rem < 154 && full >= 1 && g_fused_tail && p.split_ws && p.N % 4 == 0 && p.ldc == p.N)
        main_rows_tiles = (full * 256) / tn;
    const int rows_main = (int)std::min<int64_t>(p.M, main_rows_tiles * kW8BM);
    int64_t splits = 1, tail_tiles = 0;
    if (rows_main < p.M) {
        const int rows_left = p.M - rows_main;
        tail_tiles = mrcnn::ceil_div(rows_left, kW8BM) * tn;
        splits = std::min<int64_t>(std::min<int64_t>(16, total_slices / 4), mrcnn::ceil_div(256, tail_tiles));
        while (splits > 1 && (int64_t)rows_left * p.N * splits * 4 > kSplitWsBytes) --splits;
    }
    mrcnn::ProfKernelScope prof(mrcnn::PROF_CONV_FWD_W8, flops, bytes);
    if (splits < 2) {
        p.m_lo = 0;
        launch_w8_kernel(p, T, 1, s);
        return;
    }
    const int rows_left = p.M - rows_main;
    p.tail_split_len = (int)mrcnn::ceil_div(total_slices, splits);
    splits = mrcnn::ceil_div(total_slices, p.tail_split_len);
    p.tail_first = (int)(main_rows_tiles * tn);
    p.tail_splits = (int)splits;
    p.tail_row0 = rows_main;
    p.tail_ws = p.split_ws;
    p.tail_stride = (int64_t)rows_left * p.N;
    p.tail_bytes = (unsigned)(p.tail_stride * 4);
    p.m_lo = 0;
    launch_w8_kernel(p, p.tail_first + tail_tiles * splits, 1, s);
    FixParams f = {};
    f.ws = p.split_ws; f.C = p.C;
    f.bias = p.bias; f.scale = p.scale; f.shift = p.shift; f.residual = p.residual;
    f.res_g = p.res_g; f.res_y = p.res_y; f.out_mask_y = p.out_mask_y;
    f.splits = (int)splits; f.rows = rows_left; f.N = p.N; f.row0 = rows_main; f.ldc = p.ldc;
    f.flags = p.flags; f.stride = p.tail_stride;
    f.perm_n = 0; f.pq = p.gp * p.gq;
    const int64_t n = (int64_t)rows_left * (p.N / 4);
    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)mrcnn::ceil_div(n, 256)), dim3(256), 0, s, f);
}

extern float g_wino_ambiguity;
int g_small_whole_max = 1024, g_small_rem_max = 154;   // developer knobs (A/B)
int g_big_min_tiles = 384;        // mrcnn_set_tuning("big_min_tiles"): fewest 128x128 tiles that get 128x128 tiles
int g_small_m_split = 0;          // mrcnn_set_tuning("small_m_split", target workgroups per CU)
int g_tiny_split = 1;             // mrcnn_set_tuning("tiny_split", 0/1): K-split of launches with <= 128 tiles and >= 32 slices

template <int MODE>
void launch_small(const GemmParams &p, hipStream_t s)
{
    const int64_t tm = mrcnn::ceil_div(p.M, 64), tn = mrcnn::ceil_div(p.N, 64);
    const int64_t T = tm * tn, whole = (T / 256) * 256, rem = T - whole;
    const int total_slices = p.R * p.S * (int)mrcnn::ceil_div(p.Kc, BK);
    // Occupancy-driven split-K: a wave of the 64x64 kernel spends only about a quarter of a K
    // slice issuing MFMAs, so a SIMD needs ~4 co-resident waves to keep its matrix pipe busy.
    // A small-M problem has only T / 256 workgroups per CU; cutting every tile along K into
    // `splits` slabs multiplies the resident waves (each at least 8 slices deep) at the price
    // of the ordered slab sum.
    // Tiny launches that are K-deep (the head's fused cls_loc / score layer: 1024 x 408 x 2048 = 112
    // tiles of 64 slices each on 112 of 256 CUs, 80 us of pure K-loop latency): cut along K so that
    // ~512 workgroups share the walk, ordered slab sum as for the leftover rows.
    if (g_tiny_split && can_split_rows<MODE>(p) && T <= 128 && total_slices >= 32) {
        int64_t splits = std::min<int64_t>(std::min<int64_t>(16, total_slices / 8), mrcnn::ceil_div(512, T));
        while (splits > 1 && (int64_t)p.M * p.N * splits * 4 > kSplitWsBytes) --splits;
        if (splits >= 2) {
            launch_split_rows<MODE>(p, 0, (int)splits, total_slices, s);
            return;
        }
    }
    if (g_small_m_split > 0 && can_split_rows<MODE>(p) && T < 256ll * g_small_m_split &&
        total_slices >= 16) {
        int64_t splits = std::min<int64_t>(std::min<int64_t>(16, total_slices / 8),
                                           mrcnn::ceil_div(256ll * g_small_m_split, T));
        while (splits > 1 && (int64_t)p.M * p.N * splits * 4 > kSplitWsBytes) --splits;
        if (splits >= 2) {
            launch_split_rows<MODE>(p, 0, (int)splits, total_slices, s);
            return;
        }
    }
    const bool can_split = can_split_rows<MODE>(p) && whole > 0 && whole <= g_small_whole_max && rem > 0 &&
                           rem < g_small_rem_max && total_slices >= 8;   // beyond 4 tile-times per CU the
                                                             // two extra launches cost more than
                                                             // the imbalance
    const int rows_main = can_split ? (int)std::min<int64_t>(p.M, (whole / tn) * 64) : p.M;
    const int64_t left_tiles = mrcnn::ceil_div(p.M - rows_main, 64) * tn;
    int splits = 1;
    if (can_split && rows_main < p.M)
        splits = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(16, total_slices / 4),
                                                               mrcnn::ceil_div(256, left_tiles)));
    if (splits < 2) {
        launch_tiles<1, 1, MODE>(p, 0, p.M, 1, s);
        return;
    }
    if (launch_fused_tail<1, 1, MODE>(p, rows_main, s)) return;
    launch_tiles<1, 1, MODE>(p, 0, rows_main, 1, s);
    launch_split_rows<MODE>(p, rows_main, splits, total_slices, s);
}

// FWD / DGRAD launch policy.  With T 128x128 tiles and 512 resident workgroups a launch
// takes ceil(T/512) "rounds"; when the last round would be mostly empty (e.g. T = 536 or
// 1568) the rows of the full rounds run as 128x128 tiles and the leftover rows as a second,
// short launch of 64x64 tiles, instead of one nearly idle round of big tiles.
template <int MODE>
int launch(const GemmParams &p0, int splits, hipStream_t s)
{
    GemmParams p = p0;
    const int64_t tm = mrcnn::ceil_div(p.M, 128), tn = mrcnn::ceil_div(p.N, 128);
    const int64_t T = tm * tn;
    const bool big_ok = p.N > 64 && p.M > 64;
    // Small-M, K-deep problems (the batch-2 backbone's 3x3 layers on res4: M = 8568, N = 256,
    // 72 K slices): 64x64 tiles load twice the operand bytes per MFMA of 128x128 tiles and reach
    // ~100 TFLOP/s whatever their occupancy (profiles/r05f_small_m.txt: K splits of the 64x64 tiles are
    // flat), while the 134 tiles of 128x128 fill half the CUs once.  When a K split puts those tiles
    // into ONE round of the 512 resident workgroups at >= 75 % fill with >= 16 slices each, the
    // 128x128 kernel plus the ordered slab sum is faster in isolation (98 -> 83 us forward, 97 -> 80 us
    // data gradient, profiles/r05d_big_split_k.txt); other counts land in a nearly empty second round
    // and lose.  In the R-50 train step the rule is worth 0.05 - 0.1 ms (12 launches), in the R-101
    // step (46 launches: 23 res4 blocks) 0.39 ms of 32.1 (same-box A/B, profiles/r06e_ab_big_split_k.txt):
    // the default since round 6 ("big_split_k" = 0 restores the 64x64 tiles).
    const int total_slices_ = p.R * p.S * (int)mrcnn::ceil_div(p.Kc, BK);
    int64_t ksplits = 1;
    if (g_big_split_k != 0 && big_ok && T < g_big_min_tiles && splits == 1 && can_split_rows<MODE>(p)) {
        if (g_big_split_k > 0) {
            ksplits = std::min<int64_t>(std::min<int64_t>(8, total_slices_ / 8), mrcnn::ceil_div(g_big_split_k, T));
        } else {
            const int64_t fit = kSlotsBig / T;                 // splits that still make one round
            if (fit >= 2 && fit <= 8 && T * fit >= kSlotsBig * 3 / 4 && total_slices_ / fit >= g_big_split_min_slices) ksplits = fit;
        }
        while (ksplits > 1 && (int64_t)p.M * p.N * ksplits * 4 > kSplitWsBytes) --ksplits;
    }
    if (MODE == FWD && splits == 1 && w8_ok(p)) {
        launch_w8(p, s);
    } else if (ksplits >= 2) {
        launch_split_rows<MODE, 2>(p, 0, (int)ksplits, total_slices_, s);
    } else if (!big_ok || T < g_big_min_tiles) {
        launch_small<MODE>(p, s);
    } else {
        // whole "rounds" of k workgroups per CU (k = 3, 2, 1): pick the round size that leaves
        // the smallest leftover, run the leftover rows as 64x64 tiles
        // resident 128x128 workgroups per CU: 3 single-buffered fp32, 2 otherwise (and split-operand)
        const bool split_fwd = MODE == FWD && (g_split_bf16 & 1);
        const int64_t max_per_cu = !split_fwd && single_buffered(2, MODE, is_masked(p)) ? 3 : 2;
        int64_t main_tiles_m = tm, best_rem = T;
        for (int64_t k = max_per_cu; k >= 1; --k) {
            const int64_t slots = 256 * k, full = T / slots, rem = T - full * slots;
            if (full >= 1 && rem < best_rem) {
                best_rem = rem;
                main_tiles_m = rem > 0 && rem < 154 ? (full * slots) / tn : tm;
            }
        }
        const int rows_main = (int)std::min<int64_t>(p.M, main_tiles_m * 128);
        if (!(splits == 1 && launch_fused_tail<2, 2, MODE>(p, rows_main, s))) {
            launch_tiles<2, 2, MODE>(p, 0, rows_main, splits, s);
            if (rows_main < p.M) launch_remainder<MODE>(p, rows_main, s);
        }
    }
    return mrcnn::check_launch("conv_gemm");
}

int check_desc(const mrcnn_conv_desc *d)
{
    MRCNN_REQUIRE(d, "conv: null descriptor");
    MRCNN_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0 &&
                      d->stride > 0 && d->pad >= 0,
                  "conv: bad descriptor");
    MRCNN_REQUIRE(d->P == (d->H + 2 * d->pad - d->R) / d->stride + 1 &&
                      d->Q == (d->W + 2 * d->pad - d->S) / d->stride + 1,
                  "conv: output size (%d,%d) inconsistent with input (%d,%d) k=%d s=%d p=%d", d->P,
                  d->Q, d->H, d->W, d->R, d->stride, d->pad);
    MRCNN_REQUIRE(d->C % 4 == 0 && d->K % 4 == 0,
                  "conv: channel counts must be multiples of 4 (C=%d K=%d)", d->C, d->K);
    MRCNN_REQUIRE((int64_t)d->N * d->H * d->W * d->C < (int64_t)INT32_MAX &&
                      (int64_t)d->N * d->P * d->Q * d->K < (int64_t)INT32_MAX,
                  "conv: tensor exceeds 2^31 elements");
    return 0;
}

inline bool aligned16(const void *p) { return ((uintptr_t)p % 16) == 0; }

int g_position_major_rows = 1;    // mrcnn_set_tuning("position_major_rows", 0/1)

// Position-major row order (GemmParams::perm_n) pays when many small maps are convolved with
// a padded filter: n_img maps of gp x gq output positions, pad_eff = padding of the gathered
// tensor as the forward-form kernel sees it.
inline int choose_perm(int n_img, int gp, int gq, int R, int S, int stride, int pad_eff)
{
    if (!g_position_major_rows || R * S <= 1 || R * S > 16 || stride != 1 || pad_eff <= 0) return 0;
    if (gp * gq > 256 || n_img < 64) return 0;
    return n_img;
}

// buffer extents in floats -> bytes; 32-bit buffer offsets need every tensor < 2 GiB
int set_extents(GemmParams &p, int64_t a_floats, int64_t b_floats, int64_t c_floats)
{
    const int64_t lim = (int64_t)1 << 29;
    MRCNN_REQUIRE(a_floats < lim && b_floats < lim && c_floats < lim,
                  "conv: a tensor exceeds 2 GiB (%lld / %lld / %lld floats); split the batch",
                  (long long)a_floats, (long long)b_floats, (long long)c_floats);
    p.a_bytes = (unsigned)(a_floats * 4);
    p.b_bytes = (unsigned)(b_floats * 4);
    p.c_bytes = (unsigned)(c_floats * 4);
    return 0;
}

int wgrad_splits(int64_t tiles, int64_t pixels, int64_t slots)
{
    // Pick the split count whose tiles*splits workgroups fill whole rounds of `slots`
    // resident workgroups best, each split at least 8 K slices deep, at most ~3 rounds.
    const int64_t maxs = std::min<int64_t>(64, std::max<int64_t>(1, pixels / (8 * BK)));
    int best = 1;
    double best_u = -1.;
    for (int64_t sp = 1; sp <= maxs; ++sp) {
        const int64_t blocks = tiles * sp;
        if (blocks > 3 * slots && sp > 1) break;
        const double u = (double)blocks / (double)(mrcnn::ceil_div(blocks, slots) * slots);
        if (u > best_u + 1e-9) { best_u = u; best = (int)sp; }
    }
    return best;
}

}  // namespace

#ifdef MRCNN_GEMM_CLOCKPROBE
extern "C" int mrcnn_gemm_probe_read(unsigned long long *host, int n)
{
    MRCNN_HIP_TRY(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_probe), sizeof(unsigned long long) * n));
    void *dptr = nullptr;
    MRCNN_HIP_TRY(hipGetSymbolAddress(&dptr, HIP_SYMBOL(g_probe)));
    MRCNN_HIP_TRY(hipMemset(dptr, 0, sizeof(g_probe)));      // ready for the next pass
    return 0;
}
extern "C" int mrcnn_gemm_probe2_read(unsigned long long *host, int n)
{
    MRCNN_HIP_TRY(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_probe2), sizeof(unsigned long long) * n));
    return 0;
}
#endif

#ifdef MRCNN_GEMM_TRACE
extern "C" int mrcnn_gemm_trace_read(unsigned long long *host, int n)
{
    MRCNN_HIP_TRY(hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * n));
    return 0;
}
#endif

extern "C" int64_t mrcnn_conv2d_split_workspace_bytes(void) { return kSplitWsBytes; }

extern "C" int mrcnn_set_tuning(const char *name, int value)
{
    MRCNN_REQUIRE(name != nullptr, "set_tuning: null name");
    if (strcmp(name, "position_major_rows") == 0) {
        g_position_major_rows = value != 0;
        return 0;
    }
    if (strcmp(name, "gemm_extra_lds") == 0) {
        g_extra_lds = value;
        return 0;
    }
    if (strcmp(name, "fused_tail") == 0) {
        g_fused_tail = value == 1 ? 512 : value;
        return 0;
    }
    if (strcmp(name, "wino_ambiguity_ppb") == 0) {
        g_wino_ambiguity = 1e-9f * (float)value;
        return 0;
    }
    if (strcmp(name, "big_min_tiles") == 0) {
        g_big_min_tiles = value;
        return 0;
    }
    if (strcmp(name, "small_whole_max") == 0) {
        g_small_whole_max = value;
        return 0;
    }
    if (strcmp(name, "small_rem_max") == 0) {
        g_small_rem_max = value;
        return 0;
    }
    if (strcmp(name, "small_m_split") == 0) {
        g_small_m_split = value;
        return 0;
    }
    if (strcmp(name, "split_bf16") == 0) {
        g_split_bf16 = value;          // bit 0: 128x128 kernels, bit 1: 64x64 forward form
        return 0;
    }
    if (strcmp(name, "stagger") == 0) {
        g_stagger = value;
        return 0;
    }
    if (strcmp(name, "big_split_k") == 0) {
        g_big_split_k = value;
        return 0;
    }
    if (strcmp(name, "big_split_min_slices") == 0) {
        g_big_split_min_slices = value;
        return 0;
    }
    if (strcmp(name, "w8") == 0) {
        g_w8 = value;
        return 0;
    }
    if (strcmp(name, "w8_min_k") == 0) {
        g_w8_min_k = value;
        return 0;
    }
    if (strcmp(name, "pw") == 0) {
        g_pw = value;
        return 0;
    }
    if (strcmp(name, "tiny_split") == 0) {
        g_tiny_split = value;
        return 0;
    }
    if (strcmp(name, "stagger_min_rounds") == 0) {
        g_stagger_min_rounds = value;
        return 0;
    }
    if (mrcnn::roi_align_set_tuning(name, value) == 0) return 0;     // roi_fwd_lanes / roi_bwd_lanes
    MRCNN_REQUIRE(false, "set_tuning: unknown option '%s'", name);
    return 1;
}

extern "C" int mrcnn_conv2d_fwd(const mrcnn_conv_desc *d, const float *x, const float *w,
                                const float *bias, const float *scale, const float *shift,
                                const float *residual, float *y, int epi_flags, void *split_ws,
                                void *stream)
{
    if (int rc = check_desc(d)) return rc;
    MRCNN_REQUIRE(x && w && y, "conv2d_fwd: null pointer");
    MRCNN_REQUIRE(aligned16(x) && aligned16(w), "conv2d_fwd: x/w must be 16-byte aligned");
    MRCNN_REQUIRE(!(epi_flags & MRCNN_EPI_BIAS) || bias, "conv2d_fwd: bias flag without bias");
    MRCNN_REQUIRE(!(epi_flags & MRCNN_EPI_AFFINE) || (scale && shift), "conv2d_fwd: affine flag without scale/shift");
    MRCNN_REQUIRE(!(epi_flags & MRCNN_EPI_RESIDUAL) || residual, "conv2d_fwd: residual flag without residual");
    GemmParams p = {};
    p.A = x; p.B = w; p.C = y;
    p.bias = bias; p.scale = scale; p.shift = shift; p.residual = residual;
    p.M = d->N * d->P * d->Q; p.N = d->K; p.Kc = d->C;
    p.gp = d->P; p.gq = d->Q; p.sh = d->H; p.sw = d->W;
    p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad;
    p.lda = d->C; p.ldb = d->R * d->S * d->C; p.ldc = d->K;
    p.flags = epi_flags; p.out_mode = OUT_PLAIN;
    p.split_ws = (float *)split_ws;
    p.perm_n = choose_perm(d->N, d->P, d->Q, d->R, d->S, d->stride, d->pad);
    if (p.perm_n) p.M = (int)(mrcnn::ceil_div(d->N, kPermBlock) * kPermBlock) * d->P * d->Q;
    if (int rc = set_extents(p, (int64_t)d->N * d->H * d->W * d->C, (int64_t)d->K * d->R * d->S * d->C,
                             (int64_t)d->N * d->P * d->Q * d->K))
        return rc;
#ifdef MRCNN_DBG_PITCH      // experiment: rows of x and w MRCNN_DBG_PITCH floats apart (caller over-allocates)
    if (d->R == 1 && d->stride == 1) {
        p.lda = d->C + MRCNN_DBG_PITCH; p.ldb = d->C + MRCNN_DBG_PITCH;
        p.a_bytes = (unsigned)((int64_t)d->N * d->H * d->W * p.lda * 4);
        p.b_bytes = (unsigned)((int64_t)d->K * p.ldb * 4);
    }
#endif
    return launch<FWD>(p, 1, mrcnn::as_stream(stream));
}

// Stem: conv1 7x7/2 pad 3 of chainer ResNet50Layers on an input
// padded to 4 channels; filter given as (K, 7, 8, 4) with zeros at s=7 / c=3.
extern "C" int mrcnn_conv_stem_fwd(const float *x4, const float *w784, const float *bias,
                                   const float *scale, const float *shift, float *y, int N, int H,
                                   int W, int K, int epi_flags, void *stream)
{
    MRCNN_REQUIRE(x4 && w784 && y, "conv_stem: null pointer");
    MRCNN_REQUIRE(N > 0 && H > 0 && W > 0 && K > 0 && K % 4 == 0, "conv_stem: bad shape");
    MRCNN_REQUIRE(aligned16(x4) && aligned16(w784), "conv_stem: pointers must be 16-byte aligned");
    const int P = (H + 6 - 7) / 2 + 1, Q = (W + 6 - 7) / 2 + 1;
    GemmParams p = {};
    p.A = x4; p.B = w784; p.C = y;
    p.bias = bias; p.scale = scale; p.shift = shift;
    p.M = N * P * Q; p.N = K; p.Kc = 32;
    p.gp = P; p.gq = Q; p.sh = H; p.sw = W;
    p.R = 7; p.S = 1; p.stride = 2; p.pad = 3;
    p.lda = 4; p.ldb = 7 * 32; p.ldc = K;
    p.flags = epi_flags; p.out_mode = OUT_PLAIN; p.stem = 1;
    if (int rc = set_extents(p, (int64_t)N * H * W * 4, (int64_t)K * 7 * 32, (int64_t)N * P * Q * K))
        return rc;
    return launch<FWD>(p, 1, mrcnn::as_stream(stream));
}

extern "C" int mrcnn_conv2d_dgrad_ex(const mrcnn_conv_desc *d, const float *gy, const float *w,
                                     float *gx, int epi_flags, const float *mask_y,
                                     const float *in_scale, const float *res_g,
                                     const float *res_y, const float *out_mask_y,
                                     const float *out_scale, void *split_ws, void *stream);

extern "C" int mrcnn_conv2d_dgrad(const mrcnn_conv_desc *d, const float *gy, const float *w,
                                  float *gx, int epi_flags, void *stream)
{
    return mrcnn_conv2d_dgrad_ex(d, gy, w, gx, epi_flags, nullptr, nullptr, nullptr, nullptr,
                                 nullptr, nullptr, nullptr, stream);
}

extern "C" int mrcnn_conv2d_dgrad_ex(const mrcnn_conv_desc *d, const float *gy, const float *w,
                                     float *gx, int epi_flags, const float *mask_y,
                                     const float *in_scale, const float *res_g,
                                     const float *res_y, const float *out_mask_y,
                                     const float *out_scale, void *split_ws, void *stream)
{
    if (int rc = check_desc(d)) return rc;
    MRCNN_REQUIRE(!res_y || res_g, "conv2d_dgrad: res_y without res_g");
    MRCNN_REQUIRE((!res_g && !out_mask_y) || d->stride == 1,
                  "conv2d_dgrad: residual gradient / output mask need stride 1");
    MRCNN_REQUIRE(gy && w && gx, "conv2d_dgrad: null pointer");
    MRCNN_REQUIRE(aligned16(gy) && aligned16(w), "conv2d_dgrad: gy/w must be 16-byte aligned");
    MRCNN_REQUIRE((epi_flags & ~MRCNN_EPI_ACCUM) == 0, "conv2d_dgrad: only MRCNN_EPI_ACCUM is valid");
    hipStream_t s = mrcnn::as_stream(stream);
    GemmParams p = {};
    p.A = gy; p.B = w; p.C = gx;
    p.mask_y = mask_y; p.in_scale = in_scale; p.res_g = res_g; p.res_y = res_y;
    p.out_mask_y = out_mask_y; p.scale = out_scale;
    p.N = d->C; p.Kc = d->K; p.cin = d->C;
    p.R = d->R; p.S = d->S; p.pad = d->pad;
    p.lda = d->K; p.ldb = d->R * d->S * d->C; p.ldc = d->C;
    p.flags = epi_flags | (out_scale ? MRCNN_EPI_AFFINE : 0);
   
    p.split_ws = (float *)split_ws;
    if (int rc = set_extents(p, (int64_t)d->N * d->P * d->Q * d->K, (int64_t)d->K * d->R * d->S * d->C,
                             (int64_t)d->N * d->H * d->W * d->C))
        return rc;
    p.sh = d->P; p.sw = d->Q;
    if (d->stride == 1) {
        p.M = d->N * d->H * d->W; p.gp = d->H; p.gq = d->W; p.stride = 1;
        p.out_mode = OUT_PLAIN;
    } else {
        MRCNN_REQUIRE(d->R == 1 && d->S == 1 && d->pad == 0,
                      "conv2d_dgrad: stride>1 is implemented for 1x1/pad0 (the only strided "
                      "trainable convs of ResNet-C4) and the 2x2/2 adjoint (deconv entry points)");
        p.M = d->N * d->P * d->Q; p.gp = d->P; p.gq = d->Q; p.stride = d->stride;
        p.out_mode = OUT_STRIDED; p.oh = d->H; p.ow = d->W;
        if (!(epi_flags & MRCNN_EPI_ACCUM))
            MRCNN_HIP_TRY(hipMemsetAsync(gx, 0, sizeof(float) * (size_t)d->N * d->H * d->W * d->C, s));
    }
    return launch<DGRAD>(p, 1, s);
}

namespace {
// wT[c][R-1-r][S-1-s][k] = w[k][r][s][c] * row_scale[k]
__global__ void filter_flip_transpose_kernel(const float *__restrict__ w, float *__restrict__ wT,
                                             int K, int RS, int C,
                                             const float *__restrict__ row_scale)
{
    __shared__ float tile[32][33];
    const int rs = blockIdx.z;
    const int k0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int k = k0 + j, c = c0 + tx;
        tile[j][tx] = (k < K && c < C)
                          ? w[((int64_t)k * RS + rs) * C + c] * (row_scale ? row_scale[k] : 1.f)
                          : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, k = k0 + tx;
        if (c < C && k < K) wT[((int64_t)c * RS + (RS - 1 - rs)) * K + k] = tile[tx][j];
    }
}

// up to 32 filters per launch: a ResNet stage needs ~14 of these five-microsecond transposes,
// launch latency dominates them
constexpr int kFlipBatch = 32;
struct FlipBatch {
    const float *w[kFlipBatch];
    float *wT[kFlipBatch];
    const float *scale[kFlipBatch];
    int K[kFlipBatch], RS[kFlipBatch], C[kFlipBatch];
    int first_block[kFlipBatch + 1];
    int n;
};

__global__ void filter_flip_transpose_batched_kernel(const FlipBatch b)
{
    __shared__ float tile[32][33];
    int l = 0;
    while (l + 1 < b.n && (int)blockIdx.x >= b.first_block[l + 1]) ++l;
    const int K = b.K[l], RS = b.RS[l], C = b.C[l];
    int blk = blockIdx.x - b.first_block[l];
    const int ncb = (C + 31) / 32, nkb = (K + 31) / 32;
    const int cb = blk % ncb;
    blk /= ncb;
    const int kb = blk % nkb, rs = blk / nkb;
    const float *__restrict__ w = b.w[l];
    float *__restrict__ wT = b.wT[l];
    const float *__restrict__ row_scale = b.scale[l];
    const int k0 = kb * 32, c0 = cb * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int k = k0 + j, c = c0 + tx;
        tile[j][tx] = (k < K && c < C)
                          ? w[((int64_t)k * RS + rs) * C + c] * (row_scale ? row_scale[k] : 1.f)
                          : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, k = k0 + tx;
        if (c < C && k < K) wT[((int64_t)c * RS + (RS - 1 - rs)) * K + k] = tile[tx][j];
    }
}
}  // namespace

extern "C" int mrcnn_filter_flip_transpose_batched(int n, const void *const *w, void *const *wT,
                                                   const int *K, const int *R, const int *S,
                                                   const int *C, const void *const *row_scale,
                                                   void *stream)
{
    MRCNN_REQUIRE(n >= 0 && (n == 0 || (w && wT && K && R && S && C)),
                  "filter_flip_transpose_batched: bad args");
    for (int i0 = 0; i0 < n; i0 += kFlipBatch) {
        FlipBatch b = {};
        b.n = std::min(kFlipBatch, n - i0);
        int blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            const int j = i0 + i;
            MRCNN_REQUIRE(w[j] && wT[j] && K[j] > 0 && R[j] > 0 && S[j] > 0 && C[j] > 0,
                          "filter_flip_transpose_batched: bad entry %d", j);
            b.w[i] = (const float *)w[j];
            b.wT[i] = (float *)wT[j];
            b.scale[i] = row_scale ? (const float *)row_scale[j] : nullptr;
            b.K[i] = K[j]; b.RS[i] = R[j] * S[j]; b.C[i] = C[j];
            b.first_block[i] = blocks;
            blocks += ((C[j] + 31) / 32) * ((K[j] + 31) / 32) * R[j] * S[j];
        }
        b.first_block[b.n] = blocks;
        hipLaunchKernelGGL(filter_flip_transpose_batched_kernel, dim3(blocks), dim3(256), 0,
                           mrcnn::as_stream(stream), b);
    }
    return mrcnn::check_launch("filter_flip_transpose_batched");
}

extern "C" int mrcnn_filter_flip_transpose(const float *w, float *wT, int K, int R, int S, int C,
                                           const float *row_scale, void *stream)
{
    MRCNN_REQUIRE(w && wT && K > 0 && R > 0 && S > 0 && C > 0, "filter_flip_transpose: bad args");
    hipLaunchKernelGGL(filter_flip_transpose_kernel,
                       dim3((C + 31) / 32, (K + 31) / 32, R * S), dim3(256), 0,
                       mrcnn::as_stream(stream), w, wT, K, R * S, C, row_scale);
    return mrcnn::check_launch("filter_flip_transpose");
}

// Stride-1 dgrad as a forward-form convolution of gy with wT = flip-transpose(w) (C,R,S,K):
// both operands are K-contiguous (ds_read_b128 fragments, coalesced filter rows).
extern "C" int mrcnn_conv2d_dgrad_wt(const mrcnn_conv_desc *d, const float *gy, const float *wT,
                                     float *gx, int epi_flags, const float *mask_y,
                                     const float *in_scale, const float *res_g,
                                     const float *res_y, const float *out_mask_y,
                                     const float *out_scale, void *split_ws, void *stream)
{
    if (int rc = check_desc(d)) return rc;
    if (d->stride > 1) {
        // 1x1 / pad 0 strided convolution: every output pixel of the convolution sends its gradient
        // to ONE input pixel — the same forward-form GEMM on the transposed filter, its rows
        // scattered to the strided positions of a zero-filled gx (OUT_STRIDED in the per-element epilogue)
        MRCNN_REQUIRE(d->R == 1 && d->S == 1 && d->pad == 0,
                      "conv2d_dgrad_wt: stride > 1 is implemented for 1x1 / pad 0 filters");
        MRCNN_REQUIRE(gy && wT && gx, "conv2d_dgrad_wt: null pointer");
        MRCNN_REQUIRE(aligned16(gy) && aligned16(wT), "conv2d_dgrad_wt: gy/wT must be 16-byte aligned");
        MRCNN_REQUIRE((epi_flags & ~MRCNN_EPI_ACCUM) == 0, "conv2d_dgrad_wt: only MRCNN_EPI_ACCUM is valid");
        MRCNN_REQUIRE(!res_g && !res_y && !out_mask_y, "conv2d_dgrad_wt: residual gradient / output mask need stride 1");
        hipStream_t s = mrcnn::as_stream(stream);
        GemmParams p = {};
        p.A = gy; p.B = wT; p.C = gx;
        p.mask_y = mask_y; p.in_scale = in_scale; p.scale = out_scale;
        p.M = d->N * d->P * d->Q; p.N = d->C; p.Kc = d->K;
        p.gp = d->P; p.gq = d->Q; p.sh = d->P; p.sw = d->Q;
        p.R = 1; p.S = 1; p.stride = 1; p.pad = 0; p.ostride = d->stride;
        p.lda = d->K; p.ldb = d->K; p.ldc = d->C;
        p.flags = epi_flags | (out_scale ? MRCNN_EPI_AFFINE : 0);
        p.out_mode = OUT_STRIDED; p.oh = d->H; p.ow = d->W;
        if (int rc = set_extents(p, (int64_t)d->N * d->P * d->Q * d->K, (int64_t)d->K * d->C,
                                 (int64_t)d->N * d->H * d->W * d->C))
            return rc;
        if (!(epi_flags & MRCNN_EPI_ACCUM))
            MRCNN_HIP_TRY(hipMemsetAsync(gx, 0, sizeof(float) * (size_t)d->N * d->H * d->W * d->C, s));
        return launch<FWD>(p, 1, s);
    }
    MRCNN_REQUIRE(gy && wT && gx, "conv2d_dgrad_wt: null pointer");
    MRCNN_REQUIRE(aligned16(gy) && aligned16(wT), "conv2d_dgrad_wt: gy/wT must be 16-byte aligned");
    MRCNN_REQUIRE((epi_flags & ~MRCNN_EPI_ACCUM) == 0, "conv2d_dgrad_wt: only MRCNN_EPI_ACCUM is valid");
    MRCNN_REQUIRE(!res_y || res_g, "conv2d_dgrad_wt: res_y without res_g");
    GemmParams p = {};
    p.A = gy; p.B = wT; p.C = gx;
    p.mask_y = mask_y; p.in_scale = in_scale; p.res_g = res_g; p.res_y = res_y;
    p.out_mask_y = out_mask_y; p.scale = out_scale;
   
    p.split_ws = (float *)split_ws;
    p.M = d->N * d->H * d->W; p.N = d->C; p.Kc = d->K;
    p.gp = d->H; p.gq = d->W; p.sh = d->P; p.sw = d->Q;
    p.R = d->R; p.S = d->S; p.stride = 1; p.pad = d->R - 1 - d->pad;
    p.lda = d->K; p.ldb = d->R * d->S * d->K; p.ldc = d->C;
    p.flags = epi_flags | (out_scale ? MRCNN_EPI_AFFINE : 0); p.out_mode = OUT_PLAIN;
    p.perm_n = choose_perm(d->N, d->H, d->W, d->R, d->S, 1, p.pad);
    if (p.perm_n) p.M = (int)(mrcnn::ceil_div(d->N, kPermBlock) * kPermBlock) * d->H * d->W;
    MRCNN_REQUIRE(d->S - 1 - d->pad == p.pad, "conv2d_dgrad_wt: square filters / symmetric padding only");
    if (int rc = set_extents(p, (int64_t)d->N * d->P * d->Q * d->K, (int64_t)d->K * d->R * d->S * d->C,
                             (int64_t)d->N * d->H * d->W * d->C))
        return rc;
    return launch<FWD>(p, 1, mrcnn::as_stream(stream));
}

extern "C" int64_t mrcnn_conv2d_wgrad_workspace_bytes(const mrcnn_conv_desc *d)
{
    if (!d) return 0;
    const int64_t gwsz = (int64_t)d->K * d->R * d->S * d->C;
    return 64 * gwsz * 4 + kWgradCounterBytes;  // 64 split slabs + per-tile arrival counters
}

static int wgrad_impl(const float *gy, int ldg, const float *x, float *gw, int Kout, int64_t pixels,
                      int N_, int H, int W, int C, int P, int Q, int R, int S, int stride, int pad,
                      void *ws, hipStream_t s, const float *mask_y = nullptr,
                      const float *in_scale = nullptr, const float *out_row_scale = nullptr)
{
    GemmParams p = {};
    p.A = gy; p.B = x;
    p.mask_y = mask_y; p.in_scale = in_scale; p.scale = out_row_scale;
    p.M = Kout; p.N = R * S * C; p.Kc = (int)pixels;
    p.gp = P; p.gq = Q; p.sh = H; p.sw = W;
    p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.lda = C; p.ldg = ldg; p.cin = C; p.ldc = R * S * C;
    const int64_t gwsz = (int64_t)Kout * R * S * C;
    const int64_t big = mrcnn::ceil_div(p.M, 128) * mrcnn::ceil_div(p.N, 128);
    // The pixel (K) dimension supplies the parallelism through split-K, so 128x128 tiles are
    // used whenever tiles x achievable splits fills at least half of the resident slots and
    // the problem is at least one tile wide; otherwise 64x64 tiles.
    const int64_t small = mrcnn::ceil_div(p.M, 64) * mrcnn::ceil_div(p.N, 64);
    const int64_t max_splits = std::min<int64_t>(64, std::max<int64_t>(1, pixels / (8 * BK)));
    const bool use_big = p.N > 64 && p.M > 64 && (big * max_splits * 2 >= kSlotsBig || g_big_min_tiles <= 1);
    const int64_t tiles = use_big ? big : small;
    // block-position-major pixel order (WPERM kernel): border taps skip the positions where
    // they fall into the padding; the reduction then runs over whole blocks of BK images
    if (C % (use_big ? 128 : 64) == 0 && N_ >= BK && !is_masked(p))
        p.perm_n = choose_perm(N_, P, Q, R, S, stride, pad);
    const int64_t k_extent = p.perm_n ? mrcnn::ceil_div(N_, BK) * BK * P * Q : pixels;
    // resident 128x128 workgroups: 3 per CU single-buffered, 2 otherwise (and for the split-operand kernel)
    const bool split_kernel = (g_split_bf16 & 1) && p.perm_n == 0;
    const int64_t slots_big = !split_kernel && single_buffered(2, WGRAD, is_masked(p)) ? 768 : kSlotsBig;
    int splits = wgrad_splits(tiles, pixels, use_big ? slots_big : kSlotsSmall);
    if (!ws) splits = 1;
    p.split_len = (int)(mrcnn::ceil_div(mrcnn::ceil_div(k_extent, splits), BK) * BK);
    splits = (int)mrcnn::ceil_div(k_extent, p.split_len);
    p.split_stride = gwsz;
    if (int rc = set_extents(p, pixels * ldg, (int64_t)N_ * H * W * C, gwsz)) return rc;
    p.C = splits > 1 ? (float *)ws : gw;
    const bool in_kernel = MRCNN_WGRAD_INKERNEL_REDUCE != 0 && splits > 1;
    if (in_kernel) {
        // counters live behind the 64 slabs of the workspace
        p.tile_counters = (int *)((char *)ws + 64 * gwsz * 4);
        p.reduce_out = gw;
        MRCNN_HIP_TRY(hipMemsetAsync(p.tile_counters, 0, sizeof(int) * (size_t)tiles, s));
    }
    {
        mrcnn::ProfKernelScope prof(use_big ? mrcnn::PROF_CONV_WGRAD_128 : mrcnn::PROF_CONV_WGRAD_64,
                                    2.0 * p.M * p.N * (double)pixels,
                                    4.0 * ((double)p.M * p.N + (double)pixels * (p.M + (double)C)));
        if (use_big)
            launch_kernel<2, 2, WGRAD>(p, big, splits, s);
        else
            launch_kernel<1, 1, WGRAD>(p, tiles, splits, s);
    }
    if (splits > 1 && !in_kernel) {
        const int64_t blocks = mrcnn::ceil_div(gwsz / 4, 256);   // one float4 per thread
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                           (const float *)ws, splits, gwsz, gwsz, gw);
    }
    return mrcnn::check_launch("conv_wgrad");
}

extern "C" int mrcnn_conv2d_wgrad_ex(const mrcnn_conv_desc *d, const float *x, const float *gy,
                                     float *gw, void *ws, const float *mask_y,
                                     const float *in_scale, const float *out_row_scale,
                                     void *stream)
{
    if (int rc = check_desc(d)) return rc;
    MRCNN_REQUIRE(x && gy && gw, "conv2d_wgrad: null pointer");
    MRCNN_REQUIRE(aligned16(x) && aligned16(gy) && aligned16(gw) && (!ws || aligned16(ws)),
                  "conv2d_wgrad: pointers must be 16-byte aligned");
    return wgrad_impl(gy, d->K, x, gw, d->K, (int64_t)d->N * d->P * d->Q, d->N, d->H, d->W, d->C,
                      d->P, d->Q, d->R, d->S, d->stride, d->pad, ws, mrcnn::as_stream(stream),
                      mask_y, in_scale, out_row_scale);
}

extern "C" int mrcnn_conv2d_wgrad(const mrcnn_conv_desc *d, const float *x, const float *gy,
                                  float *gw, void *ws, void *stream)
{
    return mrcnn_conv2d_wgrad_ex(d, x, gy, gw, ws, nullptr, nullptr, nullptr, stream);
}

// ---- Deconvolution 2x2 stride 2 (= adjoint of a 2x2/2 convolution g: (N,2H,2W,K) -> (N,H,W,C)
//      with KRSC filter w (C,2,2,K)) -----------------------------------------------------------
extern "C" int mrcnn_deconv2x2s2_fwd(const float *x, const float *w, const float *bias, float *y,
                                     int N, int H, int W, int C, int K, int epi_flags,
                                     void *stream)
{
    MRCNN_REQUIRE(x && w && y, "deconv_fwd: null pointer");
    MRCNN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && C % 4 == 0 && K % 4 == 0,
                  "deconv_fwd: bad shape");
    MRCNN_REQUIRE(aligned16(x) && aligned16(w), "deconv_fwd: x/w must be 16-byte aligned");
    MRCNN_REQUIRE((epi_flags & ~(MRCNN_EPI_BIAS | MRCNN_EPI_RELU)) == 0, "deconv_fwd: bad flags");
    MRCNN_REQUIRE(!(epi_flags & MRCNN_EPI_BIAS) || bias, "deconv_fwd: bias flag without bias");
    GemmParams p = {};
    p.A = x; p.B = w; p.C = y; p.bias = bias;
    p.M = N * H * W; p.N = 4 * K; p.Kc = C; p.cin = 4 * K;
    p.gp = H; p.gq = W; p.sh = H; p.sw = W;
    p.R = 1; p.S = 1; p.stride = 1; p.pad = 0;
    p.lda = C; p.ldb = 4 * K; p.ldc = K;
    p.flags = epi_flags; p.out_mode = OUT_DECONV; p.ko = K;
    if (int rc = set_extents(p, (int64_t)N * H * W * C, (int64_t)C * 4 * K, (int64_t)N * 4 * H * W * K))
        return rc;
    return launch<DGRAD>(p, 1, mrcnn::as_stream(stream));
}

// The same deconvolution in FORWARD form on the transposed filter wT (4K, C) = (a, b, o; c) — both
// operands K-contiguous, i.e. the split-operand kernels (the K-strided DGRAD form above runs on fp32
// MFMA): y[n, 2y + a, 2x + b, o] = sum_c x[n, y, x, c] wT[(a, b, o), c], a 1x1 convolution whose
// output columns are scattered by the pixel-shuffle map of the epilogue.
extern "C" int mrcnn_deconv2x2s2_fwd_wt(const float *x, const float *wT, const float *bias, float *y,
                                        int N, int H, int W, int C, int K, int epi_flags,
                                        void *stream)
{
    MRCNN_REQUIRE(x && wT && y, "deconv_fwd_wt: null pointer");
    MRCNN_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && C % 4 == 0 && K % 4 == 0,
                  "deconv_fwd_wt: bad shape");
    MRCNN_REQUIRE(aligned16(x) && aligned16(wT), "deconv_fwd_wt: x/wT must be 16-byte aligned");
    MRCNN_REQUIRE((epi_flags & ~(MRCNN_EPI_BIAS | MRCNN_EPI_RELU)) == 0, "deconv_fwd_wt: bad flags");
    MRCNN_REQUIRE(!(epi_flags & MRCNN_EPI_BIAS) || bias, "deconv_fwd_wt: bias flag without bias");
    GemmParams p = {};
    p.A = x; p.B = wT; p.C = y; p.bias = bias;
    p.M = N * H * W; p.N = 4 * K; p.Kc = C;
    p.gp = H; p.gq = W; p.sh = H; p.sw = W;
    p.R = 1; p.S = 1; p.stride = 1; p.pad = 0;
    p.lda = C; p.ldb = C; p.ldc = K;
    p.flags = epi_flags; p.out_mode = OUT_DECONV; p.ko = K;
    if (int rc = set_extents(p, (int64_t)N * H * W * C, (int64_t)C * 4 * K, (int64_t)N * 4 * H * W * K))
        return rc;
    return launch<FWD>(p, 1, mrcnn::as_stream(stream));
}

extern "C" int mrcnn_deconv2x2s2_dgrad(const float *gy, const float *w, float *gx, int N, int H,
                                       int W, int C, int K, void *stream)
{
    // gx = conv2x2/2(gy) with KRSC filter w (C,2,2,K)
    mrcnn_conv_desc d = {N, 2 * H, 2 * W, K, C, 2, 2, 2, 0, H, W};
    return mrcnn_conv2d_fwd(&d, gy, w, nullptr, nullptr, nullptr, nullptr, gx, 0, nullptr, stream);
}

extern "C" int64_t mrcnn_deconv2x2s2_wgrad_workspace_bytes(int N, int H, int W, int C, int K)
{
    return 64ll * C * 4 * K * 4 + kWgradCounterBytes;
}

extern "C" int mrcnn_deconv2x2s2_wgrad(const float *x, const float *gy, float *gw, int N, int H,
                                       int W, int C, int K, void *ws, void *stream)
{
    MRCNN_REQUIRE(x && gy && gw, "deconv_wgrad: null pointer");
    MRCNN_REQUIRE(aligned16(x) && aligned16(gy) && aligned16(gw) && (!ws || aligned16(ws)),
                  "deconv_wgrad: pointers must be 16-byte aligned");
    // gw[c, (a,b,o)] = sum_m x[m, c] * gy[pix(m,a,b), o] : wgrad of the adjoint conv with
    // "gy" := x and "x" := gy.
    return wgrad_impl(x, C, gy, gw, C, (int64_t)N * H * W, N, 2 * H, 2 * W, K, H, W, 2, 2, 2, 0, ws,
                      mrcnn::as_stream(stream));
}

#include "conv_winograd.h"
