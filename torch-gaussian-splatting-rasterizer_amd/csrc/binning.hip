// binning.hip — stage 2b: (gaussian, tile) pairs in depth order, sorted by tile, and per-tile ranges.
//
// No reference counterpart: the reference visits every gaussian's pixel rect sequentially
// (rasterize.py:440-446); here each visible gaussian is expanded into the 16x16 tiles
// (BLOCK_SIZE, rasterize.py:34) of its rect so that tiles can be composited independently.
//
// Pair key = (tile row << bits_x) | tile column; a pair that footprint culling rejects takes the row `tiles_y` (one past
// the last) and is dropped by the first pass of the tile sort.
//
//   count   one thread per depth-sorted gaussian: tiles of its rect that belong to this shard
//           (tile rows begin, begin+step, ...) -> per-workgroup sums.  The rect arrives packed in the depth
//           sort's second payload (coalesced); frames wider than 4096 px gather it by gaussian id instead.
//   scan    one workgroup: exclusive scan of the workgroup sums; D, overflow flag, slots = min(D, max_pairs)
//   emit    load-balanced expansion: a workgroup owns 256 consecutive gaussians and the contiguous slot range
//           their pairs occupy; every thread takes slots j, j+256, ..., finds the owning gaussian by binary
//           search over the workgroup's 256 offsets in LDS and writes (tile key, gaussian id) — coalesced stores,
//           no divergence however heavy-tailed the rect sizes are (median 4 tiles, max thousands).
//           Rects above CULL_MIN_TILES tiles are tested tile by tile against the gaussian's alpha > 1/255
//           footprint (footprint.h): unreachable tiles (corners of oblique ellipses) are marked culled.
//           Smaller rects skip the test (it needs a 32-B gather per gaussian; the blend culls per 8x8 quadrant anyway).
//   sort    (sort.hip) stable radix sort by tile key, 2 passes for frames up to 4096 px
//   ranges  boundaries of equal keys in the sorted pair array -> ranges[tile] (fine) / cranges[cell] (coarse) = [begin, end)
// Coarse binning (frames up to 4096 px): pairs are generated and sorted per 32x32 CELL (2x2 tiles: half as many pairs), each
// carrying in the top four bits of its value which of the cell's tiles the gaussian's tile rect covers (and this rank owns); the
// blend of a tile walks its cell's list and keeps the entries with its bit (blend.hip, TileList).
// Measured and rejected (round 2): generating the pairs INSIDE the first tile-sort pass (a histogram of rows-per-column
// per block of 768 gaussians, then generate + rank + reorder 4096 pairs per round): it removes the emit kernel, a
// histogram kernel and one write + two reads of the pair arrays, but a workgroup then walks its rounds serially with
// rounds 69 % full and 3 instead of 4 workgroups per CU — 245 us against 149 us for emit + histogram + scatter.
// Roofline: HBM.  Bytes: 8 B per sorted gaussian (id + rect) + 8 B per pair written; ranges reads 4 B per pair.
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include "gsr_internal.h"
#include "footprint.h"

namespace gsr {

#ifndef GSR_CULL_MIN_TILES
#define GSR_CULL_MIN_TILES 8
#endif
constexpr uint32_t CULL_MIN_TILES = GSR_CULL_MIN_TILES;
// most workgroups of the count / emit grids per view (the macro: tools/ A/B builds)
#ifndef GSR_EMIT_GRID_CAP
#define GSR_EMIT_GRID_CAP 8192
#endif
constexpr int EMIT_GRID_CAP = GSR_EMIT_GRID_CAP;

// The rank's rows of a rect: strip index (RowShard::rows_before) of the first of its rows in [ty0, ty1) and how many; row k of them is
// sh.row_at(first + k).  (A RowShard at tile level, or — coarse binning — at the level of the 32x32 cells.)
__device__ __forceinline__ uint32_t tiles_of(ushort4 rc, RowShard sh, int *first_index)
{
    const int k0 = sh.rows_before(rc.y), k1 = rc.w > rc.y ? sh.rows_before(rc.w) : k0;
    *first_index = k0;
    return (uint32_t)(k1 - k0) * (uint32_t)(rc.z - rc.x);
}

__device__ __forceinline__ ushort4 unpack_rect8(uint32_t r)
{
    return make_ushort4((unsigned short)(r & 255u), (unsigned short)((r >> 8) & 255u), (unsigned short)(((r >> 16) & 255u) + 1u),
                        (unsigned short)((r >> 24) + 1u));
}

constexpr int COARSE_ID_BITS = 28;  // coarse pairs: value = gaussian id | (mask of the cell's tiles the gaussian reaches) << 28

// tile rect -> rect of the 32x32 cells (2x2 tiles) it touches; an empty rect stays empty
__device__ __forceinline__ ushort4 coarse_rect(ushort4 rc)
{
    if (rc.z <= rc.x || rc.w <= rc.y) return make_ushort4(0, 0, 0, 0);
    return make_ushort4((unsigned short)(rc.x >> 1), (unsigned short)(rc.y >> 1), (unsigned short)(((rc.z - 1) >> 1) + 1),
                        (unsigned short)(((rc.w - 1) >> 1) + 1));
}

// k-th tile of a rect of width w in row-major order -> (row, col).  k < 2^24: float division is exact up to the fix-up.
__device__ __forceinline__ void row_col(uint32_t k, uint32_t w, uint32_t *row, uint32_t *col)
{
    uint32_t r = (uint32_t)((float)k / (float)w);
    if (r * w > k) --r;
    else if ((r + 1) * w <= k) ++r;
    *row = r;
    *col = k - r * w;
}

template <bool PACKED, bool COARSE>
__device__ __forceinline__ ushort4 rect_of(uint32_t r, const uint32_t *sorted_ids, const uint32_t *sorted_rect8, const ushort4 *rect)
{
    const ushort4 rc = PACKED ? unpack_rect8(sorted_rect8[r]) : rect[sorted_ids[r]];
    return COARSE ? coarse_rect(rc) : rc;
}

// One WAVE per emit block (EMIT_THREADS = 256 consecutive gaussians of the draw order): a lane takes four consecutive gaussians
// (PACKED: one 16-B load of four packed rects), the wave sums its 256 counts with shuffles — no LDS, no barrier — and a workgroup
// covers four emit blocks.  (Round 1-3: one workgroup per emit block with a block scan, 13 K workgroups of almost no work each:
// 15 us for 13 MB; this form: 4x fewer workgroups, nothing to wait for.)
constexpr int COUNT_BLOCKS_PER_WG = EMIT_THREADS / 64;  // emit blocks per count workgroup
static_assert(EMIT_THREADS == 256, "a lane of the count kernel takes EMIT_THREADS / 64 = 4 gaussians: one uint4 of packed rects");
template <bool PACKED, bool COARSE>
__global__ __launch_bounds__(EMIT_THREADS) void pair_count_kernel(const uint32_t *__restrict__ id_a, const uint32_t *__restrict__ id_b,
                                                                  const uint32_t *__restrict__ r8_a, const uint32_t *__restrict__ r8_b,
                                                                  const FrameCtrl *ctrl, const ushort4 *__restrict__ rect, RowShard sh,
                                                                  uint32_t *__restrict__ blk_sum, uint2 *__restrict__ ranges,
                                                                  int n_tiles, uint32_t draw_limit, uint2 *__restrict__ cranges, int n_ctiles,
                                                                  FrameCtrl *ctrl_w, uint32_t ent_off, int nblk_n, int grid_wgs, size_t vstride)
{
    id_a = view_slice(id_a, vstride); id_b = view_slice(id_b, vstride); r8_a = view_slice(r8_a, vstride); r8_b = view_slice(r8_b, vstride);
    ctrl = view_slice(ctrl, vstride); rect = view_slice(rect, vstride); blk_sum = view_slice(blk_sum, vstride); ranges = view_slice(ranges, vstride);
    cranges = view_slice(cranges, vstride); ctrl_w = view_slice(ctrl_w, vstride);
    const uint32_t n = ctrl->n_visible;
    const uint32_t t = blockIdx.x * EMIT_THREADS + threadIdx.x;
    // threads in the grid (not gridDim.x: that would pull in the hidden kernarg block)
    const uint32_t stride = (uint32_t)grid_wgs * EMIT_THREADS;
    if (COARSE) {  // the cell ranges are rebuilt every frame too, and the expansion totals E with atomics
        for (uint32_t c = t; c < (uint32_t)n_ctiles; c += stride) cranges[c] = make_uint2(0u, 0u);
        if (t == 0) { ctrl_w->n_pairs = 0u; ctrl_w->ent_off = ent_off; }
    }
    for (uint32_t c = t; c < (uint32_t)n_tiles; c += stride) ranges[c] = make_uint2(0u, 0u);  // tile ranges are rebuilt every frame
    const bool odd = (ctrl->sort_passes & 1u) != 0;
    const uint32_t *sorted_ids = odd ? id_b : id_a, *sorted_rect8 = odd ? r8_b : r8_a;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lim = n < draw_limit ? n : draw_limit;
    // this wave's emit blocks: the grid is capped (launch_binning) and strides over the blocks the frame HAS — n_visible is only known
    // here; a grid sized by the bound n >= V spent most of the kernel launching waves that had nothing to count (a rank of 8 shards:
    // 24 K workgroups for 2.4 K blocks)
    for (uint32_t blk = blockIdx.x * COUNT_BLOCKS_PER_WG + (uint32_t)wave; blk < (uint32_t)nblk_n && blk * EMIT_THREADS < n;  // wave-uniform
         blk += (uint32_t)grid_wgs * COUNT_BLOCKS_PER_WG) {
    const uint32_t r0 = blk * EMIT_THREADS + (uint32_t)lane * 4u;            // r = rank in the draw order
    uint32_t cnt = 0;
    if (r0 < lim) {
        int first;
        if (PACKED && r0 + 4 <= lim) {  // the packed rects ride through the depth sort: four of them in one 16-B load (r0 is a multiple of 4)
            const uint4 p4 = *reinterpret_cast<const uint4 *>(sorted_rect8 + r0);
            const uint32_t p[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const ushort4 rc = unpack_rect8(p[j]);
                cnt += tiles_of(COARSE ? coarse_rect(rc) : rc, sh, &first);
            }
        } else {
            for (uint32_t r = r0; r < r0 + 4 && r < lim; ++r) cnt += tiles_of(rect_of<PACKED, COARSE>(r, sorted_ids, sorted_rect8, rect), sh, &first);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, d, 64);
    if (lane == 0) blk_sum[blk] = cnt;
    }
}

// Single workgroup (1024 threads): exclusive scan of blk_sum[0..nblk) in place; totals into ctrl.  Sixteen
// consecutive sums per thread (four 16-B loads), so a 6 M-gaussian frame is one trip: serial scan in registers,
// wave scan, 16 wave totals through LDS.  64-bit partials so that a D beyond 2^32 is still caught as overflow.
__global__ __launch_bounds__(1024) void pair_scan_kernel(uint32_t *__restrict__ blk_sum, int nblk_bound, FrameCtrl *ctrl,
                                                         uint32_t max_pairs, size_t vstride)
{
    constexpr int PER = 16;
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long s_carry;
    blk_sum = view_slice(blk_sum, vstride); ctrl = view_slice(ctrl, vstride);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n = ctrl->n_visible;
    const int nblk = min(nblk_bound, (int)((n + EMIT_THREADS - 1) / EMIT_THREADS));
    if (tid == 0) s_carry = 0;
    __syncthreads();
    unsigned long long grand = 0;
    for (int base = 0; base < nblk; base += 1024 * PER) {
        const int i0 = base + tid * PER;
        uint32_t v[PER];
        if (i0 + PER <= nblk) {
#pragma unroll
            for (int q = 0; q < PER / 4; ++q) {
                const uint4 t = *reinterpret_cast<const uint4 *>(blk_sum + i0 + 4 * q);
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < PER; ++j) v[j] = i0 + j < nblk ? blk_sum[i0 + j] : 0u;
        }
        unsigned long long mine = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) mine += v[j];
        unsigned long long incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long t = __shfl_up(incl, d, 64);
            if (lane >= d) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned long long wbase = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const unsigned long long s = wsum[w];
            if (w < wave) wbase += s;
            tot += s;
        }
        const unsigned long long carry = s_carry;
        // saturating: once the running total passes max_pairs the exact value no longer matters
        unsigned long long ex = carry + wbase + (incl - mine);
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (i0 + j < nblk) blk_sum[i0 + j] = ex > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)ex;
            ex += v[j];
        }
        grand += tot;
        __syncthreads();
        if (tid == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (tid == 0) {
        const unsigned long long D = grand;
        const uint32_t Dc = D > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)D;
        if (D > (unsigned long long)max_pairs) { ctrl->batch_overflow |= 1u; ctrl->batch_need = max(ctrl->batch_need, Dc); }
        ctrl->overflow = ctrl->batch_overflow;  // bit 0: pairs; bit 1: the depth sort was short of passes (sort.hip)
        ctrl->n_pairs_bbox = (ctrl->batch_overflow & 1u) ? max(ctrl->batch_need, Dc) : Dc;
        ctrl->n_slots = D > (unsigned long long)max_pairs ? max_pairs : (uint32_t)D;
    }
}

// Load-balanced expansion in emission order: a workgroup owns 256 consecutive gaussians and the contiguous slot range
// their pairs occupy; every thread takes slots j, j+256, ..., finds the owning gaussian by binary search over the
// workgroup's 256 offsets in LDS and writes (tile key, gaussian id) — coalesced stores however heavy-tailed the rects are.
// (Round 4 measured the obvious refinement — gaussians of up to four cells, most of them, write their own pairs without search or
// division, only the larger rects are load-balanced: 46.6 us either way.  The kernel is 13 K workgroups of a ~7-us dependent
// chain each — counters, ids and rects, scan, stores — eight to a CU: six rounds of latency, not search or store throughput.)
template <bool PACKED, bool COARSE, typename KeyT>  // KeyT: uint16_t when the keys fit (pair_keys_16bit), else uint32_t
__global__ __launch_bounds__(EMIT_THREADS) void pair_emit_kernel(const uint32_t *__restrict__ id_a, const uint32_t *__restrict__ id_b,
                                                                 const uint32_t *__restrict__ r8_a, const uint32_t *__restrict__ r8_b,
                                                                 const FrameCtrl *ctrl, const ushort4 *__restrict__ rect, RowShard sh,
                                                                 int bits_x, int tiles_y, const GaussRec *__restrict__ rec,
                                                                 const uint32_t *blk_off, uint32_t max_pairs,
                                                                 KeyT *__restrict__ pkey, uint32_t *__restrict__ pval,
                                                                 uint32_t draw_limit, RowShard tsh, uint32_t *blk_entries /* = blk_off: no __restrict__ on either */,
                                                                 int grid_wgs, size_t vstride)
{
    id_a = view_slice(id_a, vstride); id_b = view_slice(id_b, vstride); r8_a = view_slice(r8_a, vstride); r8_b = view_slice(r8_b, vstride);
    ctrl = view_slice(ctrl, vstride); rect = view_slice(rect, vstride); rec = view_slice(rec, vstride); blk_off = view_slice(blk_off, vstride);
    pkey = view_slice(pkey, vstride); pval = view_slice(pval, vstride); blk_entries = view_slice(blk_entries, vstride);
    __shared__ uint32_t scratch[8];
    __shared__ uint32_t s_off[EMIT_THREADS];   // exclusive pair offset of each gaussian inside the workgroup
    __shared__ uint32_t s_id[EMIT_THREADS];
    __shared__ uint32_t s_geo[EMIT_THREADS];   // x0 | width << 16
    __shared__ int s_first[EMIT_THREADS];      // strip index of the rect's first row of the shard, or -1 - that when the rect is culled per tile
    __shared__ float4 s_q0[EMIT_THREADS];
    __shared__ float4 s_q1[EMIT_THREADS];
    __shared__ uint32_t s_fine[COARSE ? EMIT_THREADS : 1];  // coarse: the gaussian's packed TILE rect
    const uint32_t n = ctrl->n_visible;
    const bool odd = (ctrl->sort_passes & 1u) != 0;
    const uint32_t *sorted_ids = odd ? id_b : id_a, *sorted_rect8 = odd ? r8_b : r8_a;
    const int tid = threadIdx.x;
    // the grid is capped (launch_binning) and strides over the emit blocks the frame has: sized by the bound n >= V it launched more
    // workgroups that found nothing to do than ones that did (n_visible is only known here)
#pragma unroll 1
    for (uint32_t blk = blockIdx.x; blk * EMIT_THREADS < n; blk += (uint32_t)grid_wgs) {  // uniform
    // the block's slot base first: its load is independent of everything below and would otherwise wait behind two barriers
    // (saturated at 2^32 - 1 by the scan: then nothing below is written)
    const unsigned long long base = blk_off[blk];
    const uint32_t r = blk * EMIT_THREADS + tid;
    uint32_t cnt = 0, g = 0;
    int first = 0;
    ushort4 rc = make_ushort4(0, 0, 0, 0);
    if (r < n && r < draw_limit) {
        g = sorted_ids[r];
        rc = rect_of<PACKED, COARSE>(r, sorted_ids, sorted_rect8, rect);
        cnt = tiles_of(rc, sh, &first);
        if (COARSE) s_fine[tid] = sorted_rect8[r];
    }
    const bool test = cnt > CULL_MIN_TILES;
    // (the record gather of a large rect flies while the workgroup scans its counts: it lands in LDS after the scan's barriers)
    float4 q0v = make_float4(0.f, 0.f, 0.f, 0.f), q1v = q0v;
    if (test) { q0v = rec[g].q0; q1v = rec[g].q1; }
    uint32_t total;
    s_off[tid] = block_excl_scan_256(cnt, scratch, &total);
    if (test) { s_q0[tid] = q0v; s_q1[tid] = q1v; }
    s_id[tid] = g;
    s_geo[tid] = (uint32_t)rc.x | ((uint32_t)(rc.z - rc.x) << 16);
    s_first[tid] = test ? -1 - first : first;
    __syncthreads();

    const uint32_t culled_row = (uint32_t)tiles_y << bits_x;
    uint32_t entries = 0;  // coarse: (gaussian, 16x16 tile) entries of the pairs this thread wrote = bits of their tile masks
    for (uint32_t j = tid; j < total; j += EMIT_THREADS) {
        // owner = last gaussian whose offset is <= j (zero-count gaussians share an offset with their successor)
        int lo = 0;
#pragma unroll
        for (int step = EMIT_THREADS / 2; step >= 1; step >>= 1)
            if (s_off[lo + step] <= j) lo += step;
        const uint32_t k = j - s_off[lo];
        const uint32_t geo = s_geo[lo];
        uint32_t row, col;
        row_col(k, geo >> 16, &row, &col);
        const int fr = s_first[lo];
        const bool tested = fr < 0;
        const int ty = sh.row_at((tested ? -1 - fr : fr) + (int)row), tx = (int)(geo & 0xFFFFu) + (int)col;
        const unsigned long long o = base + j;
        if (o < (unsigned long long)max_pairs) {
            if (COARSE) {
                // which of the cell's four tiles does the gaussian's tile rect cover — and belong to this rank's tile rows (tsh)?
                // The mask rides in the top four bits of the pair value; the blend of tile q of the cell takes the entries with
                // bit q.  Large rects are tested once per cell against the footprint (per tile it costs emit 20 us to spare the
                // blend 3 % of its staging).
                const uint32_t f = s_fine[lo];
                const int x0 = (int)(f & 255u), y0 = (int)((f >> 8) & 255u), x1 = (int)((f >> 16) & 255u), y1 = (int)(f >> 24);  // inclusive
                uint32_t mask = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int fx = 2 * tx + (q & 1), fy = 2 * ty + (q >> 1);
                    const bool mine = tsh.owns(fy);
                    mask |= (fx >= x0 && fx <= x1 && fy >= y0 && fy <= y1 && mine) ? 1u << q : 0u;
                }
                const bool hit = mask != 0u && (!tested || footprint_hits_rect(s_q0[lo], s_q1[lo], (float)(tx * 32), (float)(tx * 32 + 31),
                                                                               (float)(ty * 32), (float)(ty * 32 + 31)));
                pkey[o] = (KeyT)((hit ? (uint32_t)ty << bits_x : culled_row) | (uint32_t)tx);
                pval[o] = s_id[lo] | mask << COARSE_ID_BITS;
                entries += hit ? (uint32_t)__popc(mask) : 0u;
            } else {
                const bool hit = !tested || footprint_hits_rect(s_q0[lo], s_q1[lo], (float)(tx * 16), (float)(tx * 16 + 15),
                                                                (float)(ty * 16), (float)(ty * 16 + 15));
                pkey[o] = (KeyT)((hit ? (uint32_t)ty << bits_x : culled_row) | (uint32_t)tx);
                pval[o] = s_id[lo];
            }
        }
    }
    if (COARSE) {  // E = the sum of these partials, taken by gsr_read_stats (an atomic per workgroup here would serialise on one word).
        // The block's scanned offset (blk_off[blk], read above by this workgroup alone) is no longer needed: its slot
        // carries the partial.
        __syncthreads();
        uint32_t tot;
        block_excl_scan_256(entries, scratch, &tot);
        if (tid == 0) blk_entries[blk] = tot;
    }
    __syncthreads();  // the block's lists in LDS are done with
    }
}

template <typename KeyT>
__global__ __launch_bounds__(256) void tile_ranges_kernel(const KeyT *__restrict__ pkey, const uint32_t *n_dev,
                                                          uint2 *__restrict__ ranges, int bits_x, int tiles_x, int n_tiles,
                                                          uint32_t stride, size_t vstride)
{
    pkey = view_slice(pkey, vstride); n_dev = view_slice(n_dev, vstride); ranges = view_slice(ranges, vstride);
    // PER keys per thread from one 16-B load (four 32-bit keys, eight 16-bit ones); only the two keys flanking the group are read a
    // second time.  stride = threads in the grid, passed in: gridDim / blockDim would pull in the 256-B hidden kernarg block
    constexpr int PER = 16 / (int)sizeof(KeyT);
    const uint32_t n = *n_dev;
    const uint32_t maskx = (1u << bits_x) - 1u;
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; (unsigned long long)PER * t < n; t += stride) {
        const uint32_t i = (uint32_t)PER * t;
        uint32_t k[PER + 2];  // k[0] = key before the group, k[1..PER] = the group, k[PER + 1] = key after it
        if (i + PER <= n) {
            const uint4 v = *reinterpret_cast<const uint4 *>(pkey + i);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < PER; ++j) k[1 + j] = sizeof(KeyT) == 4 ? w[j] : ((w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu);
        } else {
#pragma unroll
            for (int j = 0; j < PER; ++j) k[1 + j] = i + j < n ? (uint32_t)pkey[i + j] : KEY_INVALID;
        }
        k[0] = i > 0 ? (uint32_t)pkey[i - 1] : KEY_INVALID;            // KEY_INVALID is no pair key: always a boundary
        k[PER + 1] = i + PER < n ? (uint32_t)pkey[i + PER] : KEY_INVALID;
#pragma unroll
        for (int j = 1; j <= PER; ++j) {
            const uint32_t idx = i + (uint32_t)(j - 1);
            const uint32_t key = k[j];
            const uint32_t tx = key & maskx, tile = (key >> bits_x) * (uint32_t)tiles_x + tx;
            if (idx >= n || tx >= (uint32_t)tiles_x || tile >= (uint32_t)n_tiles) continue;  // the last two cannot fail; they keep a corrupt key in bounds
            if (k[j - 1] != key) ranges[tile].x = idx;
            if (idx + 1 == n || k[j + 1] != key) ranges[tile].y = idx + 1;
        }
    }
}

static int ceil_log2(int v)
{
    int bits = 0;
    while ((1 << bits) < v) ++bits;
    return bits;
}

// Coarse binning when the packed rect exists (frames up to 4096 px) and the gaussian ids leave four bits for the tile mask.
// GsrOptions.fine_binning = 1 forces the fine path (A/B timing, and the test that both build the same frame).
TileKeying tile_keying(const Workspace &ws, const GsrOptions &opts)
{
    TileKeying k;
    static_assert(COARSE_ID_BITS == 28, "coarse_capable() in gsr_internal.h states the same limits");
    k.coarse = coarse_capable(ws) && opts.fine_binning == 0;
    k.grid_x = k.coarse ? ws.ctiles_x : ws.tiles_x;
    k.grid_y = k.coarse ? ws.ctiles_y : ws.tiles_y;
    k.bits_x = std::max(1, ceil_log2(k.grid_x));
    k.bits_y = std::max(1, ceil_log2(k.grid_y + 1));  // one spare row value marks culled pairs
    k.drop_from = (uint32_t)k.grid_y << k.bits_x;
    return k;
}

// Coarse binning hands the blend the sorted CELL lists (values = gaussian id | tile mask << 28) and cranges[]; the blend of a tile
// keeps the entries with its bit while it stages (blend.hip).  Round 2 expanded them into per-tile lists first (pair_expand_kernel:
// 34 us, 122 MB of traffic, 16 B of workspace per pair slot); A/B in one process on the bench frame: bin + sort 0.375 -> 0.341 ms,
// blend 0.553 -> 0.565 ms, frames and counters identical.
bool blend_reads_cell_lists(const Workspace &ws, const GsrOptions &opts) { return tile_keying(ws, opts).coarse; }

const uint32_t *tile_lists(const Workspace &ws, const GsrOptions &opts)
{
    const TileKeying tk = tile_keying(ws, opts);
    if (ws.n <= 0 || ws.max_pairs <= 0) return ws.pval[0];
    return ws.pval[((tk.bits_x + tk.bits_y + 7) / 8) & 1];  // ping-pong parity of the tile sort's passes
}

int launch_binning(const GsrOptions &opts, const Workspace &ws, hipStream_t s)
{
    const int n_tiles = ws.tiles_x * ws.tiles_y;
    if (ws.n <= 0) {  // no count kernel to clear the ranges (per tile, and per cell for a blend that reads the cell lists)
        for (int v = 0; v < ws.views; ++v) {
            GSR_HIP(hipMemsetAsync(reinterpret_cast<char *>(ws.ranges) + v * ws.view_stride, 0, sizeof(uint2) * (size_t)n_tiles, s));
            GSR_HIP(hipMemsetAsync(reinterpret_cast<char *>(ws.cranges) + v * ws.view_stride, 0, sizeof(uint2) * (size_t)ws.ctiles_x * ws.ctiles_y, s));
        }
        return GSR_OK;
    }
    const unsigned nv = (unsigned)ws.views;  // gridDim.y: one workspace slice per view
    const size_t vs = ws.view_stride;
    const bool packed_rect = rect_fits_8bit(ws);
    const RowShard sh = row_shard_of(opts);
    // cells of a shard: pairs of rows (tile_row_block = 2) ARE cell rows — the rank owns cell rows begin + k step whole; with single rows
    // and an even step the rank's tile rows begin + k step fall into the cell rows (begin >> 1) + k (step >> 1), each holding exactly one
    // of them; with an odd step (or none) every cell row can hold one.  The expansion keeps only this rank's tiles either way.
    const RowShard csh = sh.step <= 1 ? RowShard{0, 1, 0}
                       : sh.bshift == 1 ? RowShard{sh.begin, sh.step, 0}
                       : sh.step % 2 == 0 ? RowShard{sh.begin >> 1, sh.step >> 1, 0} : RowShard{0, 1, 0};
    const TileKeying tk = tile_keying(ws, opts);
    const int n_ctiles = ws.ctiles_x * ws.ctiles_y;
    const uint32_t cap = (uint32_t)ws.max_pairs;
    const uint32_t limit = opts.draw_limit > 0 ? (uint32_t)opts.draw_limit : 0xFFFFFFFFu;
    const int nblk_n = (int)((ws.n + EMIT_THREADS - 1) / EMIT_THREADS);                       // emit blocks (the bound: n >= V)
    // the count kernel: one wave per emit block (it also zeroes ranges[] / cranges[], grid-stride); both grids are capped and stride
    // over the blocks the frame has — V is known on the device only
    const int nblk_count = std::min((nblk_n + COUNT_BLOCKS_PER_WG - 1) / COUNT_BLOCKS_PER_WG, EMIT_GRID_CAP / COUNT_BLOCKS_PER_WG);
    const int nblk_emit = std::min(nblk_n, EMIT_GRID_CAP);
#define GSR_COUNT(P, C) hipLaunchKernelGGL((pair_count_kernel<P, C>), dim3(nblk_count, nv), dim3(EMIT_THREADS), 0, s, ws.val[0], ws.val[1], ws.rect8[0], ws.rect8[1], \
                                           ws.ctrl, ws.rect, C ? csh : sh, ws.blk_sum, ws.ranges, n_tiles, limit, ws.cranges, n_ctiles, ws.ctrl, \
                                           C ? (uint32_t)(reinterpret_cast<const char *>(ws.blk_sum) - reinterpret_cast<const char *>(ws.ctrl)) : 0u, nblk_n, nblk_count, vs)
    const bool k16 = pair_keys_16bit(tk.bits_x + tk.bits_y);  // two-byte keys in memory when they fit (sort.hip)
#define GSR_EMIT_T(P, C, T) hipLaunchKernelGGL((pair_emit_kernel<P, C, T>), dim3(nblk_emit, nv), dim3(EMIT_THREADS), 0, s, ws.val[0], ws.val[1], ws.rect8[0], ws.rect8[1], \
                                          ws.ctrl, ws.rect, C ? csh : sh, tk.bits_x, tk.grid_y, ws.rec, ws.blk_sum, cap, reinterpret_cast<T *>(ws.pkey[0]), ws.pval[0], limit, \
                                          sh, ws.blk_sum, nblk_emit, vs)
#define GSR_EMIT(P, C) do { if (k16) GSR_EMIT_T(P, C, uint16_t); else GSR_EMIT_T(P, C, uint32_t); } while (0)
    if (tk.coarse) GSR_COUNT(true, true); else if (packed_rect) GSR_COUNT(true, false); else GSR_COUNT(false, false);
    hipLaunchKernelGGL(pair_scan_kernel, dim3(1, nv), dim3(1024), 0, s, ws.blk_sum, nblk_n, ws.ctrl, cap, vs);
    if (tk.coarse) GSR_EMIT(true, true); else if (packed_rect) GSR_EMIT(true, false); else GSR_EMIT(false, false);
#undef GSR_COUNT
#undef GSR_EMIT
#undef GSR_EMIT_T
    GSR_HIP(hipGetLastError());
    if (ws.max_pairs <= 0) return GSR_OK;
    // stable sort by cell / tile key; the first pass drops the pairs the emit kernel culled and leaves their count in ctrl
    int pbuf = 0;
    uint32_t *n_sorted = tk.coarse ? &ws.ctrl->n_cpairs : &ws.ctrl->n_pairs;
    const int rc = launch_pair_sort(ws, 0, &ws.ctrl->n_slots, 0, tk.bits_x + tk.bits_y, tk.drop_from, n_sorted, &pbuf, s);
    if (rc) return rc;
    const int grid = (int)std::min<int64_t>((ws.max_pairs + 1023) / 1024, 8192);
    if (k16) hipLaunchKernelGGL(tile_ranges_kernel<uint16_t>, dim3(grid, nv), dim3(256), 0, s, reinterpret_cast<const uint16_t *>(ws.pkey[pbuf]), n_sorted,
                                tk.coarse ? ws.cranges : ws.ranges, tk.bits_x, tk.grid_x, tk.grid_x * tk.grid_y, (uint32_t)grid * 256u, vs);
    else hipLaunchKernelGGL(tile_ranges_kernel<uint32_t>, dim3(grid, nv), dim3(256), 0, s, ws.pkey[pbuf], n_sorted, tk.coarse ? ws.cranges : ws.ranges, tk.bits_x,
                            tk.grid_x, tk.grid_x * tk.grid_y, (uint32_t)grid * 256u, vs);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

}  // namespace gsr
