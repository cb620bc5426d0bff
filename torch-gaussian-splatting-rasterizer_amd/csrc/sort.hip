// sort.hip — stable LSD radix sort of (u32 key, u32 value[, u32 value2]) records, up to 9 bits per pass.
//
// Used twice per frame (rasterize.py:424-425 is one torch.sort; tile lists have no reference counterpart):
//   1. depth order:  keys = IEEE bits of z_cam of every gaussian (KEY_INVALID for culled ones, dropped
//      by pass 0), values = gaussian id (+ its packed tile rect).  Stable from index order => depth ties
//      resolve by gaussian index.
//   2. tile lists:   keys = (tile row, tile column) of every (gaussian,tile) pair emitted IN DEPTH ORDER, values = gaussian
//      id.  A stable sort by tile therefore leaves every tile's list depth-ordered.
//
// Per pass, three launches (no inter-workgroup hand-off inside a launch, so nothing depends on dispatch
// order or XCD placement; a dependent kernel boundary costs ~1.5 us, an in-kernel grid barrier 4-10 us):
//   hist     one workgroup per tile of 256*ITEMS keys: digit histogram -> hist[digit][tile]
//   rowscan  one workgroup per digit: exclusive scan of its row in place, row total -> digit_tot[digit]
//   scatter  one workgroup per tile: per-wave ranking through LDS peer masks -> tile reordered by digit in LDS -> digit
//            runs written out contiguously (coalesced), position = digit base + scanned hist + rank in run   (radix.h)
// The element count lives in device memory (n_dev): grids are sized by the host-side bound and surplus workgroups fall
// through.  16 keys per thread (2048-key tiles measured slower, the fixed per-workgroup costs dominate); the depth sort's
// 9-bit passes run 512 threads on 8192 keys, the pair sort's <= 8-bit passes 256 threads on 4096 (radix.h says why).
//
// Depth sort in three passes instead of four.  z_cam >= 0.2, so a key's sign and high exponent bits never vary: the sort
// runs on key - bits(0.2f), and only on the bits the frame actually uses.  Pass 0 always takes 9 bits; its histogram
// kernel, which reads every key anyway, also reduces the largest valid key (one guarded atomicMax per workgroup); the
// pass-0 rowscan turns that into the plan for the remaining passes (FrameCtrl.sort_*): the B - 9 remaining bits split
// evenly over as few passes of <= 9 bits as possible.  A scene whose depths span 0.2 .. 83 (B <= 27) sorts in 9 + 9 + 9;
// the launch sequence is fixed at four passes (the host cannot know B without a sync) and the kernels of an unused pass
// return at once.  Consumers find the sorted buffer from the plan (sort_passes & 1).
// Roofline: HBM.  Per pass per element: 4 B (hist) + 8..12 B read + 8..12 B written.
#include <algorithm>
#include "gsr_internal.h"
#include "radix.h"

namespace gsr {

__device__ __forceinline__ uint32_t load_count(const uint32_t *n_dev, uint32_t n_bound)
{
    if (n_dev == nullptr) return n_bound;
    const uint32_t n = *n_dev;
    return n < n_bound ? n : n_bound;
}

// KeyT: how the keys lie in memory — uint32_t, or uint16_t for the pair sort's cell / tile keys when they fit (frames up to 4096 px
// always do with 32x32 cells: 7 + 8 bits): the sort is bound by HBM, and a pair is then 6 B instead of 8.  Registers and LDS hold
// 32-bit keys either way.
template <int THREADS, bool DROP, int ITEMS, typename KeyT = uint32_t>
__global__ __launch_bounds__(THREADS) void radix_hist_kernel(const KeyT *__restrict__ keys, const uint32_t *n_dev,
                                                             uint32_t n_bound, PassSpec ps, FrameCtrl *ctrl,
                                                             uint32_t *__restrict__ hist, int hist_blocks, size_t vstride)
{
    constexpr int TILE = THREADS * ITEMS;
    __shared__ uint32_t h[THREADS];
    __shared__ uint32_t s_max;
    keys = view_slice(keys, vstride); n_dev = view_slice(n_dev, vstride); ctrl = view_slice(ctrl, vstride); hist = view_slice(hist, vstride);
    int shift;
    uint32_t mask;
    if (!resolve_pass(ps, ctrl, &shift, &mask)) return;  // uniform: this depth-sort pass is not needed
    const uint32_t n = load_count(n_dev, n_bound);
    const uint32_t base = blockIdx.x * (uint32_t)TILE;
    if (base >= n) return;  // rowscan and scatter stop at the live tiles too
    h[threadIdx.x] = 0;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    uint32_t kmax = 0;
    if (n - base >= (uint32_t)TILE) {  // full tile: unguarded loads, all in flight together
        uint32_t k[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) k[r] = keys[base + r * THREADS + threadIdx.x];
#pragma unroll
        for (int r = 0; r < ITEMS; ++r)
            if (!DROP || k[r] < ps.drop_from) {
                atomicAdd(&h[((k[r] - ps.key_base) >> shift) & mask], 1u);
                kmax = max(kmax, k[r]);
            }
    } else {
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const uint32_t idx = base + r * THREADS + threadIdx.x;
            if (idx < n) {
                const uint32_t k = keys[idx];
                if (!DROP || k < ps.drop_from) {
                    atomicAdd(&h[((k - ps.key_base) >> shift) & mask], 1u);
                    kmax = max(kmax, k);
                }
            }
        }
    }
    if (ps.dyn_pass == 0) {  // key range of the frame: wave max -> workgroup max -> at most one device atomic, usually none
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, d, 64));
        if ((threadIdx.x & 63) == 0) atomicMax(&s_max, kmax);
    }
    __syncthreads();
    // only the digit values the pass has: a 6-bit pass of the 256-thread pair sort would otherwise write (and its scatter read back)
    // four times the rows — every entry a 64-B sector of its own in this digit-major table
    if (threadIdx.x <= mask) hist[(size_t)threadIdx.x * hist_blocks + blockIdx.x] = h[threadIdx.x];
    if (ps.dyn_pass == 0 && threadIdx.x == 0) {
        const uint32_t m = s_max;
        // the running maximum only grows: a stale (smaller) value read here costs one redundant atomic, never a wrong result
        if (m > __hip_atomic_load(&ctrl->depth_key_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&ctrl->depth_key_max, m);
    }
}

// One workgroup per digit row.  Rows are `hist_blocks` long; only the entries of the live tiles are scanned.
// The pass-0 rowscan of the depth sort also fixes the plan of the remaining passes (see the header).
template <int TILE>
__global__ __launch_bounds__(256) void radix_rowscan_kernel(uint32_t *__restrict__ hist, int hist_blocks, const uint32_t *n_dev,
                                                            uint32_t n_bound, PassSpec ps, FrameCtrl *ctrl, size_t vstride)
{
    __shared__ uint32_t scratch[8];
    int shift;
    uint32_t mask;
    hist = view_slice(hist, vstride); n_dev = view_slice(n_dev, vstride); ctrl = view_slice(ctrl, vstride);
    if (!resolve_pass(ps, ctrl, &shift, &mask)) return;
    if (ps.dyn_pass == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t kmax = ctrl->depth_key_max;
        const uint32_t span = kmax > DEPTH_KEY_BASE ? kmax - DEPTH_KEY_BASE : 0u;
        const uint32_t bits = span ? 32u - (uint32_t)__clz((int)span) : 1u;
        const uint32_t rest = bits > DEPTH_DIGIT_BITS ? bits - DEPTH_DIGIT_BITS : 0u;
        const uint32_t rest_passes = (rest + DEPTH_DIGIT_BITS - 1) / DEPTH_DIGIT_BITS;
        ctrl->sort_passes = 1u + rest_passes;
        // what gsr_read_stats reports when the bound was too small: the most any frame since the last full clear has needed (a batch's
        // counters otherwise describe its LAST view, whose own plan may fit)
        ctrl->batch_sort_passes = max(ctrl->batch_sort_passes, 1u + rest_passes);
        if (1u + rest_passes > (uint32_t)ps.enqueued) ctrl->batch_overflow |= 2u;  // the caller's bound was too small: the frame is wrong
        ctrl->sort_key_bits = bits;
        ctrl->sort_bits_rest = rest_passes ? (rest + rest_passes - 1) / rest_passes : 0u;
        ctrl->depth_key_max = 0u;  // consumed: a repeated gsr_bin_sort starts from 0 again (the frame clear zeroes it too)
    }
    if (blockIdx.x > mask) {  // a digit value this pass does not have (uniform)
        if (threadIdx.x == 0) ctrl->digit_tot[blockIdx.x] = 0u;
        return;
    }
    const uint32_t n = load_count(n_dev, n_bound);
    const int nblk = (int)(((unsigned long long)n + TILE - 1) / TILE);
    uint32_t *row = hist + (size_t)blockIdx.x * hist_blocks;
    uint32_t carry = 0;
    for (int base = 0; base < nblk; base += 1024) {
        const int i0 = base + threadIdx.x * 4;
        uint32_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (i0 + j < nblk) ? row[i0 + j] : 0u;
        const uint32_t mine = v[0] + v[1] + v[2] + v[3];
        uint32_t total;
        uint32_t ex = block_excl_scan_256(mine, scratch, &total) + carry;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i0 + j < nblk) row[i0 + j] = ex;
            ex += v[j];
        }
        carry += total;
    }
    if (threadIdx.x == 0) ctrl->digit_tot[blockIdx.x] = carry;
}

// Register budget: two 512-thread workgroups per CU (128 VGPRs; one before: 119 KB of LDS) — depth-sort scatters 54 / 32 us ->
// 49 / 28 us on the bench frame.  The 256-thread pair sort stays at four per CU: pressed into 96 VGPRs for a fifth it spills
// and takes 53 instead of 37 us.
template <int THREADS, bool DROP, int ITEMS, bool HAS_V2, bool INDEX_VALS, typename KeyT = uint32_t>
__global__ __launch_bounds__(THREADS, 4) void radix_scatter_kernel(
    const KeyT *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, const uint32_t *__restrict__ vals2_in,
    KeyT *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint32_t *__restrict__ vals2_out, const uint32_t *n_dev,
    uint32_t n_bound, PassSpec ps, const FrameCtrl *ctrl, const uint32_t *__restrict__ hist, int hist_blocks, uint32_t *n_out,
    size_t vstride)
{
    using Smem = RadixTileSmem<THREADS, ITEMS, HAS_V2>;
    constexpr int TILE = Smem::TILE;
    __shared__ Smem sm;
    __shared__ uint32_t digit_base[THREADS];  // global position of this tile's run of digit d
    keys_in = view_slice(keys_in, vstride); vals_in = view_slice(vals_in, vstride); vals2_in = view_slice(vals2_in, vstride);
    keys_out = view_slice(keys_out, vstride); vals_out = view_slice(vals_out, vstride); vals2_out = view_slice(vals2_out, vstride);
    n_dev = view_slice(n_dev, vstride); ctrl = view_slice(ctrl, vstride); hist = view_slice(hist, vstride); n_out = view_slice(n_out, vstride);

    int shift;
    uint32_t mask;
    if (!resolve_pass(ps, ctrl, &shift, &mask)) return;  // uniform
    const uint32_t n = load_count(n_dev, n_bound);
    const uint32_t base = blockIdx.x * (uint32_t)TILE;
    if (base >= n) return;  // uniform per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    radix_clear(sm);
    __syncthreads();

    // keys first; the payloads are loaded when their turn at the tile buffer comes (their loads fly while the previous array
    // goes out), so that at most four 16-register arrays are live at once
    uint32_t key[ITEMS], rank[ITEMS];
    const bool full = n - base >= (uint32_t)TILE;  // all but the last workgroup: unguarded loads, all in flight together
    auto load = [&](const auto *__restrict__ src, uint32_t (&v)[ITEMS], uint32_t fill) {
        if (full) {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) v[r] = (uint32_t)src[base + wave * (64 * ITEMS) + r * 64 + lane];
        } else {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                const uint32_t idx = base + wave * (64 * ITEMS) + r * 64 + lane;
                v[r] = idx < n ? (uint32_t)src[idx] : fill;
            }
        }
    };
    load(keys_in, key, KEY_INVALID);
    auto dig = [&](int r) -> uint32_t {
        const uint32_t idx = base + wave * (64 * ITEMS) + r * 64 + lane;
        const bool valid = (idx < n) && (!DROP || key[r] < ps.drop_from);
        return valid ? ((key[r] - ps.key_base) >> shift) & mask : RADIX_NO_DIGIT;
    };
    radix_rank(sm, dig, rank);
    __syncthreads();

    radix_tile_layout(sm);
    {  // global digit bases: exclusive scan of the digit totals + this tile's entry of the scanned histogram
        const bool has = (uint32_t)tid <= mask;  // digit values beyond the pass's width do not occur
        const uint32_t tot = has ? ctrl->digit_tot[tid] : 0u;
        uint32_t all_total;
        const uint32_t gs = block_excl_scan<THREADS>(tot, sm.scratch, &all_total);
        digit_base[tid] = gs + (has ? hist[(size_t)tid * hist_blocks + blockIdx.x] : 0u);
        if (n_out != nullptr && blockIdx.x == 0 && tid == 0) *n_out = all_total;
    }
    __syncthreads();

    radix_positions(sm, dig, rank);  // rank[] is now the position inside the reordered tile
    uint32_t val[ITEMS];
    if (INDEX_VALS) {  // the payload is the element's index: nothing to load
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) val[r] = base + wave * (64 * ITEMS) + r * 64 + lane;
    } else {
        load(vals_in, val, 0u);
    }
    // keys through the tile buffer; every output slot's global position is fixed on the way and kept for the payloads
    radix_stage(sm, rank, key);
    __syncthreads();
    const uint32_t nvalid = sm.n_valid;
    // the depth sort's LAST pass (which one that is the frame's plan says) leaves sorted ids and rects: nobody reads the sorted keys
    const bool write_keys = !(ps.dyn_pass >= 0 && (uint32_t)ps.dyn_pass + 1u == ctrl->sort_passes);
    uint32_t gpos[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const uint32_t i = (uint32_t)tid + (uint32_t)k * THREADS;
        if (i < nvalid) {
            const uint32_t kk = sm.buf[i];
            const uint32_t d = ((kk - ps.key_base) >> shift) & mask;
            gpos[k] = digit_base[d] + (i - sm.tile_start[d]);
            if (write_keys) keys_out[gpos[k]] = (KeyT)kk;
        }
    }
    __syncthreads();
    radix_stage(sm, rank, val);
    if constexpr (HAS_V2) load(vals2_in, val, 0u);  // val is staged: its registers take the second payload
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const uint32_t i = (uint32_t)tid + (uint32_t)k * THREADS;
        if (i < nvalid) vals_out[gpos[k]] = sm.buf[i];
    }
    if constexpr (HAS_V2) {
        __syncthreads();
        radix_stage(sm, rank, val);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const uint32_t i = (uint32_t)tid + (uint32_t)k * THREADS;
            if (i < nvalid) vals2_out[gpos[k]] = sm.buf[i];
        }
    }
}

// One pass = hist + rowscan + scatter.  `first` = pass 0 of a sort (may drop, may synthesise the index payload).
template <int THREADS, int ITEMS, bool HAS_V2, typename KeyT = uint32_t>
static void launch_pass(const KeyT *kin, const uint32_t *vin, const uint32_t *v2in, KeyT *kout, uint32_t *vout, uint32_t *v2out,
                        const uint32_t *cnt_dev, int64_t n_bound, const PassSpec &ps, bool drop, bool ident, uint32_t *n_out,
                        const Workspace &ws, hipStream_t s)
{
    constexpr int TILE = THREADS * ITEMS;
    const int nblk = (int)((n_bound + TILE - 1) / TILE);
    const uint32_t nb = (uint32_t)n_bound;
    const unsigned nv = (unsigned)ws.views;  // gridDim.y: one slice of the workspace per view (gsr_internal.h, view_slice)
    const size_t vs = ws.view_stride;
#define GSR_SCATTER(DROP, IDENT)                                                                                                   \
    hipLaunchKernelGGL((radix_scatter_kernel<THREADS, DROP, ITEMS, HAS_V2, IDENT, KeyT>), dim3(nblk, nv), dim3(THREADS), 0, s, kin, vin, v2in,   \
                       kout, vout, v2out, cnt_dev, nb, ps, ws.ctrl, ws.hist, ws.hist_blocks, n_out, vs)
    if (drop) hipLaunchKernelGGL((radix_hist_kernel<THREADS, true, ITEMS, KeyT>), dim3(nblk, nv), dim3(THREADS), 0, s, kin, cnt_dev, nb, ps, ws.ctrl, ws.hist, ws.hist_blocks, vs);
    else hipLaunchKernelGGL((radix_hist_kernel<THREADS, false, ITEMS, KeyT>), dim3(nblk, nv), dim3(THREADS), 0, s, kin, cnt_dev, nb, ps, ws.ctrl, ws.hist, ws.hist_blocks, vs);
    hipLaunchKernelGGL(radix_rowscan_kernel<TILE>, dim3(THREADS, nv), dim3(256), 0, s, ws.hist, ws.hist_blocks, cnt_dev, nb, ps, ws.ctrl, vs);
    if (drop) { if (ident) GSR_SCATTER(true, true); else GSR_SCATTER(true, false); }
    else      { if (ident) GSR_SCATTER(false, true); else GSR_SCATTER(false, false); }
#undef GSR_SCATTER
}

// Depth order of the gaussians (rasterize.py:424-425).  Four passes enqueued unless the caller bounds them (GsrOptions.
// depth_sort_passes), 3 run on ordinary scenes (header); a plan that needs more than were enqueued is flagged.
// Afterwards FrameCtrl.n_visible = V and the sorted ids / packed rects are in val[p] / rect8[p], p = sort_passes & 1.
template <int ITEMS>
static int depth_sort_passes(const Workspace &ws, bool packed_rect, bool compact_input, int passes, hipStream_t s)
{
    constexpr int TILE = DEPTH_SORT_THREADS * ITEMS;
    const int nblk = (int)((ws.n + TILE - 1) / TILE);
    if (nblk > ws.hist_blocks) { set_error("radix sort: %d tiles exceed the histogram stride %d", nblk, ws.hist_blocks); return GSR_ERR_WORKSPACE; }
    // compact_input (multi-GPU shard): the input is FrameCtrl.n_records (key, id[, rect]) records in id order, left in
    // key[0] / val[0] / rect8[0] by preprocess.hip, instead of one key per gaussian
    const uint32_t *cnt_dev = compact_input ? &ws.ctrl->n_records : nullptr;
    const int enq = passes >= 1 && passes <= 4 ? passes : 4;
    for (int p = 0; p < enq; ++p) {
        const PassSpec ps = {0, 0u, DEPTH_KEY_BASE, KEY_INVALID, p, enq};
        const int in = p & 1, out = in ^ 1;
        const bool first = p == 0;
        if (packed_rect)
            launch_pass<DEPTH_SORT_THREADS, ITEMS, true>(ws.key[in], ws.val[in], ws.rect8[in], ws.key[out], ws.val[out], ws.rect8[out], cnt_dev, ws.n, ps,
                                                     first, first && !compact_input, first ? &ws.ctrl->n_visible : nullptr, ws, s);
        else
            launch_pass<DEPTH_SORT_THREADS, ITEMS, false>(ws.key[in], ws.val[in], nullptr, ws.key[out], ws.val[out], nullptr, cnt_dev, ws.n, ps,
                                                      first, first && !compact_input, first ? &ws.ctrl->n_visible : nullptr, ws, s);
        cnt_dev = &ws.ctrl->n_visible;  // later passes only see the survivors
    }
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

int launch_depth_sort(const Workspace &ws, bool packed_rect, bool compact_input, int passes, hipStream_t s)
{
    if (ws.n <= 0) return GSR_OK;
    // a shard's compact records are few (its visible gaussians): smaller tiles, more workgroups, shorter serial chains
    return compact_input ? depth_sort_passes<DEPTH_SORT_ITEMS_SHARD>(ws, packed_rect, true, passes, s)
                         : depth_sort_passes<DEPTH_SORT_ITEMS>(ws, packed_rect, false, passes, s);
}

// Stable sort of the (tile key, gaussian id) pairs over key bits [first_bit, key_bits), 8 bits or fewer per pass, the bits
// split evenly (13 tile-id bits sort as 7 + 6, not 8 + 5: with 128 digit values a workgroup's runs in the first,
// far-scattering pass are 32 entries = one full 128-B line per array instead of half a line).  The first pass run here
// drops keys >= drop_from (pairs culled at emission) and leaves the survivor count in *n_out.  in_buf: which of
// pkey[]/pval[] holds the input; *result_buf: which holds the output.
int launch_pair_sort(const Workspace &ws, int in_buf, const uint32_t *n_dev, int first_bit, int key_bits, uint32_t drop_from,
                     uint32_t *n_out, int *result_buf, hipStream_t s)
{
    *result_buf = in_buf;
    if (ws.max_pairs <= 0 || key_bits <= first_bit) return GSR_OK;
    constexpr int TILE = PAIR_SORT_THREADS * PAIR_SORT_ITEMS;
    const int nblk = (int)((ws.max_pairs + TILE - 1) / TILE);
    if (nblk > ws.hist_blocks) { set_error("radix sort: %d tiles exceed the histogram stride %d", nblk, ws.hist_blocks); return GSR_ERR_WORKSPACE; }
    const int bits = key_bits - first_bit;
    const int passes = (bits + 7) / 8;
    const int bits_pp = (bits + passes - 1) / passes;
    int cur = in_buf;
    const uint32_t *cnt_dev = n_dev;
    for (int p = 0; p < passes; ++p) {
        const int shift = first_bit + bits_pp * p;
        const PassSpec ps = {shift, (1u << std::min(bits_pp, key_bits - shift)) - 1u, 0u, drop_from, -1, 0};
        const bool first = p == 0;
        if (pair_keys_16bit(key_bits))  // the keys lie in the first half of each pkey buffer, two bytes each (binning.hip writes them so)
            launch_pass<PAIR_SORT_THREADS, PAIR_SORT_ITEMS, false, uint16_t>(reinterpret_cast<const uint16_t *>(ws.pkey[cur]), ws.pval[cur], nullptr,
                                                                            reinterpret_cast<uint16_t *>(ws.pkey[cur ^ 1]), ws.pval[cur ^ 1], nullptr, cnt_dev,
                                                                            ws.max_pairs, ps, first, false, first ? n_out : nullptr, ws, s);
        else
            launch_pass<PAIR_SORT_THREADS, PAIR_SORT_ITEMS, false>(ws.pkey[cur], ws.pval[cur], nullptr, ws.pkey[cur ^ 1], ws.pval[cur ^ 1], nullptr, cnt_dev,
                                                 ws.max_pairs, ps, first, false, first ? n_out : nullptr, ws, s);
        if (first && n_out) cnt_dev = n_out;
        cur ^= 1;
    }
    GSR_HIP(hipGetLastError());
    *result_buf = cur;
    return GSR_OK;
}

// ---- scene order (gsr_scene_order): the permutation that lays the gaussians out along a Morton curve of their means ----------------
// No reference counterpart (the reference reads the .ply in file order, rasterize.py:353-358); the statement of the permutation is
// renderer.morton_order: every axis rank-quantised to 10 bits (stable ranks: equal coordinates keep index order), bits interleaved,
// stable sort by code.  Here: per axis one stable sort of (order-preserving key of the coordinate, index) — four 8-bit passes of the
// pair-sort kernels above — whose output position IS the rank; the quantised ranks are ORed into a code per gaussian; one more
// sort of (code, index).  Sixteen radix passes and seven small kernels, ~2 ms at 6 M gaussians, once per scene at upload.

// float -> uint32 whose unsigned order is the float order; -0.0 sorts as +0.0 and every NaN last, like numpy's stable argsort
__device__ __forceinline__ uint32_t order_key_of(float x)
{
    if (x != x) return 0xFFFFFFFFu;
    const uint32_t u = __float_as_uint(x + 0.0f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void order_keys_kernel(const float *__restrict__ means, int axis, uint32_t *__restrict__ keys, uint32_t n)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) keys[i] = order_key_of(means[3 * (size_t)i + axis]);
}

__device__ __forceinline__ uint32_t spread10(uint32_t v)  // 10 bits -> every third bit
{
    v &= 0x3FFu;
    v = (v | (v << 16)) & 0x30000FFu;
    v = (v | (v << 8)) & 0x300F00Fu;
    v = (v | (v << 4)) & 0x30C30C3u;
    v = (v | (v << 2)) & 0x9249249u;
    return v;
}

// sorted_ids[r] = the gaussian of rank r on this axis: its code takes the quantised rank's bits
__global__ __launch_bounds__(256) void order_rank_kernel(const uint32_t *__restrict__ sorted_ids, uint32_t *__restrict__ code, int axis, uint32_t n)
{
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n) return;
    const uint32_t q = (uint32_t)(((unsigned long long)r * 1024ull) / n);
    const uint32_t bits = spread10(q) << axis, g = sorted_ids[r];
    code[g] = axis == 0 ? bits : (code[g] | bits);  // every gaussian has exactly one rank per axis: no two threads share a word
}

size_t scene_order_bytes(int64_t n)
{
    const size_t nn = (size_t)std::max<int64_t>(n, 1);
    const size_t nblk = (nn + PAIR_SORT_THREADS * PAIR_SORT_ITEMS - 1) / (PAIR_SORT_THREADS * PAIR_SORT_ITEMS);
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    return al(sizeof(FrameCtrl)) + 5 * al(4 * nn) + al(4 * 256 * nblk);
}

int launch_scene_order(int64_t n, const float *means, uint32_t *perm_out, void *workspace, hipStream_t s)
{
    if (n <= 0) return GSR_OK;
    const size_t nn = (size_t)n;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    char *p = static_cast<char *>(workspace);
    Workspace ws = {};
    ws.ctrl = reinterpret_cast<FrameCtrl *>(p); p += al(sizeof(FrameCtrl));
    uint32_t *key[2], *val[2], *code;
    for (int b = 0; b < 2; ++b) { key[b] = reinterpret_cast<uint32_t *>(p); p += al(4 * nn); }
    for (int b = 0; b < 2; ++b) { val[b] = reinterpret_cast<uint32_t *>(p); p += al(4 * nn); }
    code = reinterpret_cast<uint32_t *>(p); p += al(4 * nn);
    ws.hist = reinterpret_cast<uint32_t *>(p);
    ws.hist_blocks = (int)((nn + PAIR_SORT_THREADS * PAIR_SORT_ITEMS - 1) / (PAIR_SORT_THREADS * PAIR_SORT_ITEMS));
    const unsigned grid = (unsigned)((nn + 255) / 256);
    // four stable 8-bit passes over 32-bit keys; pass 0 synthesises the index payload; the result ends in buffer 0
    auto sort32 = [&](int bits) {
        int cur = 0;
        for (int sh = 0; sh < bits; sh += 8) {
            const PassSpec ps = {sh, (1u << std::min(8, bits - sh)) - 1u, 0u, KEY_INVALID, -1, 0};
            launch_pass<PAIR_SORT_THREADS, PAIR_SORT_ITEMS, false>(key[cur], val[cur], nullptr, key[cur ^ 1], val[cur ^ 1], nullptr, nullptr, n, ps,
                                                                  false, sh == 0, nullptr, ws, s);
            cur ^= 1;
        }
        return cur;
    };
    for (int axis = 0; axis < 3; ++axis) {
        hipLaunchKernelGGL(order_keys_kernel, dim3(grid), dim3(256), 0, s, means, axis, key[0], (uint32_t)n);
        const int out = sort32(32);
        hipLaunchKernelGGL(order_rank_kernel, dim3(grid), dim3(256), 0, s, val[out], code, axis, (uint32_t)n);
    }
    GSR_HIP(hipMemcpyAsync(key[0], code, 4 * nn, hipMemcpyDeviceToDevice, s));
    const int out = sort32(30);
    GSR_HIP(hipMemcpyAsync(perm_out, val[out], 4 * nn, hipMemcpyDeviceToDevice, s));
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

}  // namespace gsr
