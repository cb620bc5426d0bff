// binning.hip — stage 2b: (gaussian, tile) pair emission in depth order and per-tile ranges.
//
// No reference counterpart: the reference visits every gaussian's pixel rect sequentially
// (rasterize.py:440-446); here each visible gaussian is expanded into the 16x16 tiles
// (BLOCK_SIZE, rasterize.py:34) of its rect so that tiles can be composited independently.
//
//   count   one thread per depth-sorted gaussian: tiles of its rect that belong to this shard
//           (tile rows begin, begin+step, ...) -> per-workgroup sums
//   scan    one workgroup: exclusive scan of the workgroup sums; D, overflow flag, E = min(D, max_pairs)
//   emit    recount + in-workgroup scan -> pair offsets; small rects are written by their own lane,
//           rects above 32 tiles by the whole wave (the tile count is heavy-tailed: median 4, max thousands).
//           A tile the gaussian's alpha > 1/255 footprint cannot reach (footprint.h; rect corners of oblique
//           ellipses) gets KEY_INVALID and is dropped by pass 0 of the tile sort.
//   ranges  boundaries of equal tile ids in the tile-sorted pair array -> ranges[tile] = [begin, end)
// Roofline: HBM.  Bytes: 12 B per sorted gaussian (id + rect) + 8 B per pair written; ranges reads 4 B per pair.
#include <cstring>
#include <algorithm>
#include "gsr_internal.h"
#include "footprint.h"

namespace gsr {

struct Shard {
    int begin, step;
};

// rows ty in [ty0, ty1) with (ty - begin) % step == 0:  first such row and how many
__device__ __forceinline__ void shard_rows(int ty0, int ty1, Shard sh, int *first, int *rows)
{
    int f = ty0;
    if (sh.step > 1) {
        int r = (ty0 - sh.begin) % sh.step;
        if (r < 0) r += sh.step;
        f = r == 0 ? ty0 : ty0 + (sh.step - r);
    }
    *first = f;
    *rows = f < ty1 ? (ty1 - f + sh.step - 1) / sh.step : 0;
}

__device__ __forceinline__ uint32_t tiles_of(ushort4 rc, Shard sh, int *first_row)
{
    int rows;
    shard_rows(rc.y, rc.w, sh, first_row, &rows);
    return (uint32_t)rows * (uint32_t)(rc.z - rc.x);
}

__global__ __launch_bounds__(EMIT_THREADS) void pair_count_kernel(const uint32_t *__restrict__ sorted_ids, const FrameCtrl *ctrl,
                                                                  const ushort4 *__restrict__ rect, Shard sh,
                                                                  uint32_t *__restrict__ blk_sum)
{
    __shared__ uint32_t scratch[8];
    const uint32_t n = ctrl->n_visible;
    const uint32_t r = blockIdx.x * EMIT_THREADS + threadIdx.x;
    uint32_t cnt = 0;
    if (r < n) {
        int first;
        cnt = tiles_of(rect[sorted_ids[r]], sh, &first);
    }
    uint32_t total;
    block_excl_scan_256(cnt, scratch, &total);
    if (threadIdx.x == 0) blk_sum[blockIdx.x] = total;
}

// Single workgroup (1024 threads): exclusive scan of blk_sum[0..nblk) in place; totals into ctrl.
__global__ __launch_bounds__(1024) void pair_scan_kernel(uint32_t *__restrict__ blk_sum, int nblk_bound, FrameCtrl *ctrl,
                                                         uint32_t max_pairs)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = ctrl->n_visible;
    const int nblk = min(nblk_bound, (int)((n + EMIT_THREADS - 1) / EMIT_THREADS));
    if (tid == 0) s_carry = 0;
    __syncthreads();
    unsigned long long grand = 0;  // 64-bit so that a D beyond 2^32 is still caught as overflow
    for (int base = 0; base < nblk; base += 1024) {
        const int i = base + tid;
        const uint32_t v = i < nblk ? blk_sum[i] : 0u;
        const uint32_t incl = wave_incl_scan(v);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t s = wsum[w];
            if (w < wave) wbase += s;
            tot += s;
        }
        const uint32_t carry = s_carry;
        // saturating: once the running total passes max_pairs the exact value no longer matters
        const unsigned long long ex = (unsigned long long)carry + wbase + (incl - v);
        if (i < nblk) blk_sum[i] = ex > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ex;
        grand += tot;
        __syncthreads();
        if (tid == 0) {
            const unsigned long long c = (unsigned long long)carry + tot;
            s_carry = c > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)c;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned long long D = grand;
        ctrl->n_pairs_bbox = D > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)D;
        ctrl->overflow = D > (unsigned long long)max_pairs ? 1u : 0u;
        ctrl->n_slots = D > (unsigned long long)max_pairs ? max_pairs : (uint32_t)D;
    }
}

__global__ __launch_bounds__(EMIT_THREADS) void pair_emit_kernel(const uint32_t *__restrict__ sorted_ids, const FrameCtrl *ctrl,
                                                                 const ushort4 *__restrict__ rect, Shard sh, int tiles_x,
                                                                 const GaussRec *__restrict__ rec,
                                                                 const uint32_t *__restrict__ blk_off, uint32_t max_pairs,
                                                                 uint32_t *__restrict__ pkey, uint32_t *__restrict__ pval)
{
    __shared__ uint32_t scratch[8];
    const uint32_t n = ctrl->n_visible;
    const uint32_t r = blockIdx.x * EMIT_THREADS + threadIdx.x;
    if (blockIdx.x * EMIT_THREADS >= n) return;  // uniform
    uint32_t cnt = 0, g = 0;
    int first = 0;
    ushort4 rc = make_ushort4(0, 0, 0, 0);
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
    if (r < n) {
        g = sorted_ids[r];
        rc = rect[g];
        cnt = tiles_of(rc, sh, &first);
        if (cnt > 0) { q0 = rec[g].q0; q1 = rec[g].q1; }
    }
    uint32_t total;
    const uint32_t off = blk_off[blockIdx.x] + block_excl_scan_256(cnt, scratch, &total);
    const int width = rc.z - rc.x;

    constexpr uint32_t WAVE_COOP = 32;  // rects above this many tiles are written by all 64 lanes
    // -- large rects: one at a time, the whole wave strides over its tiles
    unsigned long long big = __ballot(cnt > WAVE_COOP);
    const int lane = threadIdx.x & 63;
    while (big) {
        const int src = __ffsll((long long)big) - 1;
        big &= big - 1;
        const uint32_t b_cnt = __shfl(cnt, src, 64), b_off = __shfl(off, src, 64), b_g = __shfl(g, src, 64);
        const int b_w = __shfl(width, src, 64), b_x0 = __shfl((int)rc.x, src, 64), b_first = __shfl(first, src, 64);
        const float4 b_q0 = make_float4(__shfl(q0.x, src, 64), __shfl(q0.y, src, 64), __shfl(q0.z, src, 64), __shfl(q0.w, src, 64));
        const float4 b_q1 = make_float4(__shfl(q1.x, src, 64), __shfl(q1.y, src, 64), __shfl(q1.z, src, 64), __shfl(q1.w, src, 64));
        for (uint32_t k = lane; k < b_cnt; k += 64) {
            const uint32_t row = k / (uint32_t)b_w, col = k - row * (uint32_t)b_w;
            const uint32_t o = b_off + k;
            if (o < max_pairs) {
                const int tyy = b_first + (int)row * sh.step, txx = b_x0 + (int)col;
                const bool hit = footprint_hits_rect(b_q0, b_q1, (float)(txx * 16), (float)(txx * 16 + 15), (float)(tyy * 16),
                                                     (float)(tyy * 16 + 15));
                pkey[o] = hit ? (uint32_t)tyy * (uint32_t)tiles_x + (uint32_t)txx : KEY_INVALID;
                pval[o] = b_g;
            }
        }
    }
    // -- small rects: each lane writes its own
    if (cnt > 0 && cnt <= WAVE_COOP) {
        uint32_t o = off;
        int ty = first;
        for (; ty < rc.w; ty += sh.step) {
            for (int tx = rc.x; tx < rc.z; ++tx, ++o) {
                if (o < max_pairs) {
                    const bool hit = footprint_hits_rect(q0, q1, (float)(tx * 16), (float)(tx * 16 + 15),
                                                                     (float)(ty * 16), (float)(ty * 16 + 15));
                    pkey[o] = hit ? (uint32_t)ty * (uint32_t)tiles_x + (uint32_t)tx : KEY_INVALID;
                    pval[o] = g;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint32_t *__restrict__ pkey, const FrameCtrl *ctrl,
                                                          uint2 *__restrict__ ranges, int n_tiles)
{
    const uint32_t n = ctrl->n_pairs;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t k = pkey[i];
        if (k >= (uint32_t)n_tiles) continue;  // cannot happen; keeps a corrupt key from writing out of bounds
        if (i == 0 || pkey[i - 1] != k) ranges[k].x = i;
        if (i + 1 == n || pkey[i + 1] != k) ranges[k].y = i + 1;
    }
}

__global__ __launch_bounds__(256) void max_list_kernel(const uint2 *__restrict__ ranges, int n_tiles, FrameCtrl *ctrl)
{
    uint32_t m = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_tiles; i += gridDim.x * blockDim.x) {
        const uint2 r = ranges[i];
        m = max(m, r.y - r.x);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(&ctrl->max_list_len, m);
}

int tile_sort_passes(int tiles)
{
    int bits = 1;
    while ((1 << bits) < tiles) ++bits;
    return (bits + 7) / 8;
}

int launch_binning(const GsrCamera &cam, const GsrOptions &opts, const Workspace &ws, int sorted_buf, hipStream_t s)
{
    (void)cam;
    if (ws.n <= 0) return GSR_OK;
    const Shard sh = {opts.tile_row_begin, opts.tile_row_step < 1 ? 1 : opts.tile_row_step};
    const int nblk = (int)((ws.n + EMIT_THREADS - 1) / EMIT_THREADS);
    const uint32_t *ids = ws.val[sorted_buf];
    const uint32_t cap = (uint32_t)ws.max_pairs;
    hipLaunchKernelGGL(pair_count_kernel, dim3(nblk), dim3(EMIT_THREADS), 0, s, ids, ws.ctrl, ws.rect, sh, ws.blk_sum);
    hipLaunchKernelGGL(pair_scan_kernel, dim3(1), dim3(1024), 0, s, ws.blk_sum, nblk, ws.ctrl, cap);
    hipLaunchKernelGGL(pair_emit_kernel, dim3(nblk), dim3(EMIT_THREADS), 0, s, ids, ws.ctrl, ws.rect, sh, ws.tiles_x, ws.rec,
                       ws.blk_sum, cap, ws.pkey[0], ws.pval[0]);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

int launch_tile_ranges(const Workspace &ws, int pair_buf, hipStream_t s)
{
    const int n_tiles = ws.tiles_x * ws.tiles_y;
    GSR_HIP(hipMemsetAsync(ws.ranges, 0, sizeof(uint2) * (size_t)n_tiles, s));
    if (ws.max_pairs <= 0) return GSR_OK;
    const int grid = (int)std::min<int64_t>((ws.max_pairs + 255) / 256, 4096);
    hipLaunchKernelGGL(tile_ranges_kernel, dim3(grid), dim3(256), 0, s, ws.pkey[pair_buf], ws.ctrl, ws.ranges, n_tiles);
    hipLaunchKernelGGL(max_list_kernel, dim3(std::min((n_tiles + 255) / 256, 64)), dim3(256), 0, s, ws.ranges, n_tiles, ws.ctrl);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

}  // namespace gsr
