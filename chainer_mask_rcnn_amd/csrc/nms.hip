// Greedy NMS on score-sorted boxes for gfx950 (wave64).
//
// Replaces chainercv's `non_maximum_suppression` (un-vendored dependency; call
// sites /root/reference/chainer_mask_rcnn/models/mask_rcnn.py:193-194 and, via
// ProposalCreator, models/region_proposal_network.py:135-138; algorithm
// SURVEY.md Appendix A.3).  The upstream GPU path builds a 64x64-tile bitmask
// on the device, copies the whole mask to the host and scans it serially in
// Python.  Here both phases stay on the device:
//
//   1. nms_mask_kernel  — one 64-lane wave per (row block, col block) tile of the
//      upper triangle; lane i owns row box i and emits one uint64 word whose bit
//      j says IoU(row_i, col_j) >= thresh.  A wave is exactly one mask word wide.
//      For the diagonal tiles lane j also emits its COLUMN word (bit i: box i < j suppresses
//      box j) — IoU is symmetric bit for bit, so it is the lower triangle of the same tile.
//   2. nms_scan_kernel  — one workgroup per NMS problem walks the 64-row chunks, eight to a
//      super-step: the in-chunk dependency is resolved by a fixpoint iteration on the column
//      words (K <- candidates not suppressed by an earlier member of K; one ballot + AND per
//      iteration, exact: the greedy keep set is its unique fixpoint and iteration t fixes
//      at least the first t candidates), the dependency between the chunks of a super-step from
//      pre-loaded words in registers, and once per super-step all threads OR the kept rows into
//      the `removed` bit-vector held in LDS (one global round trip).
//
// IoU uses the CPU path's exact fp32 operation sequence (no FMA:
// -ffp-contract=off), so keep sets are bit-identical to oracle/nms_ref.c.
#include "common.h"

namespace {

__device__ __forceinline__ bool iou_ge(const float4 a, const float area_a, const float4 b,
                                       const float area_b, const float thresh)
{
    const float tly = fmaxf(a.x, b.x), tlx = fmaxf(a.y, b.y);
    const float bry = fminf(a.z, b.z), brx = fminf(a.w, b.w);
    float inter = (bry - tly) * (brx - tlx);
    if (!(tly < bry && tlx < brx)) inter = inter * 0.f;
    const float iou = inter / (area_a + area_b - inter);
    return iou >= thresh;
}

// grid (nblk_max, nblk_max, groups), block 64.
__global__ void __launch_bounds__(64)
nms_mask_kernel(const float4 *__restrict__ bbox_all, const int32_t *__restrict__ n_dev, int n_max,
                int nblk_max, float thresh, uint64_t *__restrict__ mask_all,
                uint64_t *__restrict__ diag_t_all)
{
    const int cb = blockIdx.x, rb = blockIdx.y, g = blockIdx.z;
    if (cb < rb) return;
    const int n = min(n_dev ? n_dev[g] : n_max, n_max);
    if (rb * 64 >= n || cb * 64 >= n) return;
    const float4 *__restrict__ bbox = bbox_all + (int64_t)g * n_max;
    uint64_t *__restrict__ mask = mask_all + (int64_t)g * n_max * nblk_max;

    __shared__ float4 cbox[64];
    __shared__ float carea[64];
    const int lane = threadIdx.x;
    const int cj = cb * 64 + lane;
    if (cj < n) {
        const float4 b = bbox[cj];
        cbox[lane] = b;
        carea[lane] = (b.z - b.x) * (b.w - b.y);
    }
    __syncthreads();
    const int ri = rb * 64 + lane;
    if (ri >= n) return;
    const float4 a = bbox[ri];
    const float area_a = (a.z - a.x) * (a.w - a.y);
    const int ncol = min(64, n - cb * 64);
    uint64_t bits = 0;
    const int j0 = (cb == rb) ? lane + 1 : 0;
    for (int j = j0; j < ncol; ++j)
        if (iou_ge(a, area_a, cbox[j], carea[j], thresh)) bits |= 1ull << j;
    mask[(int64_t)ri * nblk_max + cb] = bits;
    if (cb == rb) {
        // column word of box `lane` inside its own chunk: which earlier boxes suppress it
        // (iou_ge is symmetric in its two boxes bit for bit: max / min / a + b commute)
        uint64_t col = 0;
        for (int i = 0; i < lane; ++i)
            if (iou_ge(cbox[i], carea[i], a, area_a, thresh)) col |= 1ull << i;
        diag_t_all[(int64_t)g * n_max + ri] = col;
    }
}

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int lane)
{
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, lane);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t readfirstlane64(uint64_t v)
{
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// grid (groups), block 64 * W (W = 4 or 16 waves), dynamic LDS = nblk_max * 8 bytes.
// The keep decision of a 64-row chunk depends on every earlier chunk, so the chunks are a serial
// chain; what the chain costs per link is the design.  Round 4 paid one global round trip and two
// workgroup barriers PER CHUNK (188 chunks for the RPN's 12 000 boxes: 0.46 ms).  Here S = 8 chunks
// form a super-step:
//   * wave 0 walks the S chunks alone, without a barrier and without touching global memory on
//     the way: the words that couple the chunks of one super-step — row r's mask word for the
//     LATER chunks of the same super-step, 28 per lane — do not depend on any decision, so they were
//     loaded during the previous super-step's bulk phase (as were the S diagonal column words).  Per
//     chunk: removed-word from LDS, the fixpoint iteration on the column words (exact: the greedy
//     keep set is its unique fixpoint, iteration t fixes at least the first t candidates), keep-list
//     and kept-row-list writes, and the kept lanes OR their in-super-step words into LDS;
//   * one bulk phase per super-step: all waves OR the kept rows of its S chunks into the removed
//     bit-vector for every later column — (kept row, column) pairs dealt out evenly over the
//     threads, eight independent loads in flight per thread, one global round trip.
// Chain cost: 24 super-steps x (8 chunk links in registers / LDS + 1 round trip + 2 barriers).
constexpr int kScanS = 8;

__global__ void __launch_bounds__(1024)
nms_scan_kernel(const uint64_t *__restrict__ mask_all, const uint64_t *__restrict__ diag_t_all,
                const int32_t *__restrict__ n_dev, int n_max, int nblk_max, int limit,
                int32_t *__restrict__ keep_all, int32_t *__restrict__ n_keep_all)
{
    constexpr int S = kScanS;
    extern __shared__ __attribute__((aligned(16))) uint64_t removed[];
    __shared__ int s_rows[S * 64];      // kept rows of the current super-step, in order
    __shared__ int s_nk, s_count;
    const int g = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const uint64_t *__restrict__ mask = mask_all + (int64_t)g * n_max * nblk_max;
    const uint64_t *__restrict__ diag_t = diag_t_all + (int64_t)g * n_max;
    int32_t *__restrict__ keep = keep_all + (int64_t)g * n_max;
    const int n = min(n_dev ? n_dev[g] : n_max, n_max);
    const int nblk = (n + 63) / 64;
    for (int c = tid; c < nblk; c += nthr) removed[c] = 0;
    if (tid == 0) { s_count = 0; s_nk = 0; }

    // wave 0: per chunk c of the super-step, this lane's row: its diagonal column word and its
    // mask words for the later chunks of the same super-step (dw[c][d]: column sblk + c + 1 + d)
    uint64_t col[S], dw[S][S - 1];
    auto prefetch = [&](int sblk) {
#pragma unroll
        for (int c = 0; c < S; ++c) {
            const int row = (sblk + c) * 64 + lane;
            const bool ok = row < n;
            col[c] = ok ? diag_t[row] : 0ull;
#pragma unroll
            for (int d = 0; d < S - 1 - c; ++d)
                dw[c][d] = (ok && sblk + c + 1 + d < nblk)
                               ? mask[(int64_t)row * nblk_max + sblk + c + 1 + d] : 0ull;
        }
    };
    if (wave == 0) prefetch(0);
    __syncthreads();

    int cnt = 0;                         // wave 0: boxes kept so far (wave-uniform)
    for (int sblk = 0; sblk < nblk; sblk += S) {
        if (wave == 0) {
            int nk = 0;
#pragma unroll
            for (int c = 0; c < S; ++c) {
                const int blk = sblk + c;
                if (blk < nblk && !(limit > 0 && cnt >= limit)) {          // (wave-uniform)
                    const int row = blk * 64 + lane;
                    uint64_t rem = readfirstlane64(*(volatile uint64_t *)&removed[blk]);
                    const int nrow = n - blk * 64;
                    if (nrow < 64) rem |= ~((1ull << nrow) - 1ull);
                    // fixpoint: K <- candidates whose column word meets no member of K
                    const uint64_t cand = ~rem;
                    uint64_t kept = cand;
                    for (int it = 0; it < 64; ++it) {
                        const uint64_t sup = __ballot((col[c] & kept) != 0ull);
                        const uint64_t next = cand & ~sup;
                        if (next == kept) break;
                        kept = next;
                    }
                    if (limit > 0) {
                        const int room = limit - cnt;
                        while (__popcll(kept) > room) kept &= ~(1ull << (63 - __clzll((long long)kept)));
                    }
                    const bool mine = (kept >> lane) & 1ull;
                    const int rank = __popcll(kept & ((1ull << lane) - 1ull));
                    if (mine) {
                        keep[cnt + rank] = row;
                        s_rows[nk + rank] = row;
#pragma unroll
                        for (int d = 0; d < S - 1 - c; ++d)
                            if (dw[c][d])
                                atomicOr(reinterpret_cast<unsigned long long *>(&removed[blk + 1 + d]),
                                         (unsigned long long)dw[c][d]);
                    }
                    const int k = __popcll(kept);
                    cnt += k;
                    nk += k;
                }
            }
            if (lane == 0) { s_nk = nk; s_count = cnt; }
            // the next super-step's words: in flight across the barrier and the bulk phase
            if (sblk + S < nblk) prefetch(sblk + S);
        }
        __syncthreads();
        if (limit > 0 && s_count >= limit) break;
        // bulk phase: (kept row q, column) pairs over the columns behind this super-step
        const int col0 = sblk + S, ncols = nblk - col0, nk = s_nk;
        if (ncols > 0 && nk > 0) {
            const int npairs = nk * ncols;
            for (int p0 = 0; p0 < npairs; p0 += 8 * nthr) {
                uint64_t w[8];
                int cw[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int p = p0 + u * nthr + tid;
                    const int pc = p < npairs ? p : 0;              // clamped: no branch between the loads
                    const int q = pc / ncols, ci = pc - q * ncols;
                    cw[u] = p < npairs ? col0 + ci : -1;
                    w[u] = mask[(int64_t)s_rows[q] * nblk_max + col0 + ci];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (cw[u] >= 0 && w[u])
                        atomicOr(reinterpret_cast<unsigned long long *>(&removed[cw[u]]),
                                 (unsigned long long)w[u]);
            }
        }
        __syncthreads();
    }
    if (tid == 0) n_keep_all[g] = s_count;
}

}  // namespace

extern "C" int64_t mrcnn_nms_workspace_bytes(int n_max, int groups)
{
    const int64_t nblk = (n_max + 63) / 64;
    return (int64_t)groups * n_max * (nblk + 1) * 8;     // mask rows + one column word per box
}

extern "C" int mrcnn_nms_sorted_batched(const float *bbox, const int32_t *n_dev, int groups,
                                        int n_max, float thresh, int limit, int32_t *keep,
                                        int32_t *n_keep, void *mask_ws, void *stream)
{
    MRCNN_REQUIRE(groups >= 0 && n_max >= 0, "nms: bad shape");
    hipStream_t s = mrcnn::as_stream(stream);
    if (groups == 0) return 0;
    MRCNN_REQUIRE(n_keep, "nms: null n_keep");
    if (n_max == 0) {
        MRCNN_HIP_TRY(hipMemsetAsync(n_keep, 0, 4 * (size_t)groups, s));
        return 0;
    }
    MRCNN_REQUIRE(bbox && keep && mask_ws, "nms: null pointer");
    MRCNN_REQUIRE((uintptr_t)bbox % 16 == 0, "nms: bbox must be 16-byte aligned");
    const int nblk = (n_max + 63) / 64;
    MRCNN_REQUIRE(nblk * 8 <= 64 * 1024, "nms: n_max too large (%d)", n_max);
    MRCNN_REQUIRE(nblk <= 65535 && groups <= 65535, "nms: grid too large");
    {
        mrcnn::ProfScope prof(mrcnn::PROF_NMS_MASK, 0.,
                              (double)groups * n_max * (16.0 + 4.0 * nblk), s);
        hipLaunchKernelGGL(nms_mask_kernel, dim3(nblk, nblk, groups), dim3(64), 0, s,
                           (const float4 *)bbox, n_dev, n_max, nblk, thresh, (uint64_t *)mask_ws,
                           (uint64_t *)mask_ws + (int64_t)groups * n_max * nblk);
    }
    mrcnn::ProfScope prof(mrcnn::PROF_NMS_SCAN, 0., (double)groups * n_max * 4.0 * nblk, s);
    // long problems (RPN: 12000 boxes, 2 groups) get 16 waves, short batched ones (per-class
    // NMS: <= 1000 boxes, hundreds of groups) 4
    hipLaunchKernelGGL(nms_scan_kernel, dim3(groups), dim3(nblk > 32 ? 1024 : 256), (size_t)nblk * 8, s,
                       (const uint64_t *)mask_ws,
                       (const uint64_t *)mask_ws + (int64_t)groups * n_max * nblk, n_dev, n_max, nblk,
                       limit, keep, n_keep);
    return mrcnn::check_launch("nms_sorted");
}

extern "C" int mrcnn_nms_sorted(const float *bbox, const int32_t *n_dev, int n_max, float thresh,
                                int limit, int32_t *keep, int32_t *n_keep, void *mask_ws,
                                void *stream)
{
    return mrcnn_nms_sorted_batched(bbox, n_dev, 1, n_max, thresh, limit, keep, n_keep, mask_ws,
                                    stream);
}
