// sort.hip — stable LSD radix sort of (u32 key, u32 value[, u32 value2]) records, up to 8 bits per pass (key bits split evenly over the passes).
//
// Used twice per frame (rasterize.py:424-425 is one torch.sort; tile lists have no reference counterpart):
//   1. depth order:  keys = IEEE bits of z_cam of every gaussian (KEY_INVALID for culled ones, dropped
//      by pass 0), values = gaussian id (+ its packed tile rect).  Stable from index order => depth ties
//      resolve by gaussian index.
//   2. tile lists:   keys = tile id of every (gaussian,tile) pair emitted IN DEPTH ORDER (KEY_INVALID for pairs
//      the emit kernel culled, dropped by pass 0), values = gaussian id.  A stable sort by tile therefore leaves
//      every tile's list depth-ordered.
//
// Per pass, three launches (no inter-workgroup hand-off inside a launch, so nothing depends on dispatch
// order or XCD placement):
//   hist     one workgroup per tile of 256*ITEMS keys: 256-bin digit histogram -> hist[digit][tile]
//   rowscan  one workgroup per digit: exclusive scan of its row in place, row total -> digit_tot[digit]
//   scatter  one workgroup per tile: per-wave ranking through LDS peer masks -> tile reordered by digit in LDS -> digit
//            runs written out contiguously (coalesced), position = digit base + scanned hist + rank in run
// The element count lives in device memory (n_dev): grids are sized by the host-side bound and
// surplus workgroups fall through.  4096-key tiles (16 keys per thread): 2048-key tiles measured slower, the
// fixed per-workgroup costs (7 barriers, two 256-wide scans, 512 table loads) dominate small tiles.
// Roofline: HBM.  Per pass per element: 4 B (hist) + 8..12 B read + 8..12 B written.
#include <algorithm>
#include "gsr_internal.h"

namespace gsr {

#ifdef GSR_SORT_TRACE  // tools/sort_trace.py: per-workgroup phase stamps of the scatter kernel (100 MHz wall clock)
__device__ uint32_t g_sort_trace[16384 * 8];
#ifndef GSR_SORT_TRACE_DROP
#define GSR_SORT_TRACE_DROP 0
#endif
#define GSR_STAMP(k) do { if (DROP_INVALID == (GSR_SORT_TRACE_DROP != 0) && !HAS_V2 && threadIdx.x == 0 && blockIdx.x < 16384) g_sort_trace[blockIdx.x * 8 + (k)] = (uint32_t)wall_clock64(); } while (0)
#else
#define GSR_STAMP(k) do {} while (0)
#endif

__device__ __forceinline__ uint32_t load_count(const uint32_t *n_dev, uint32_t n_bound)
{
    if (n_dev == nullptr) return n_bound;
    const uint32_t n = *n_dev;
    return n < n_bound ? n : n_bound;
}

template <bool DROP_INVALID, int ITEMS>
__global__ __launch_bounds__(SORT_THREADS) void radix_hist_kernel(const uint32_t *__restrict__ keys, const uint32_t *n_dev,
                                                                  uint32_t n_bound, int shift, uint32_t mask,
                                                                  uint32_t *__restrict__ hist, int hist_blocks)
{
    __shared__ uint32_t h[256];
    const uint32_t n = load_count(n_dev, n_bound);
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (SORT_THREADS * ITEMS);
    if (base + SORT_THREADS * ITEMS <= n) {  // full tile: unguarded loads, all in flight together
        uint32_t k[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) k[r] = keys[base + r * SORT_THREADS + threadIdx.x];
#pragma unroll
        for (int r = 0; r < ITEMS; ++r)
            if (!DROP_INVALID || k[r] != KEY_INVALID) atomicAdd(&h[(k[r] >> shift) & mask], 1u);
    } else if (base < n) {
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const uint32_t idx = base + r * SORT_THREADS + threadIdx.x;
            if (idx < n) {
                const uint32_t k = keys[idx];
                if (!DROP_INVALID || k != KEY_INVALID) atomicAdd(&h[(k >> shift) & mask], 1u);
            }
        }
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * hist_blocks + blockIdx.x] = h[threadIdx.x];
}

// One workgroup per digit row.  Rows are `hist_blocks` long; only the first `nblk` entries are live.
__global__ __launch_bounds__(256) void radix_rowscan_kernel(uint32_t *__restrict__ hist, int hist_blocks, int nblk,
                                                            uint32_t *__restrict__ digit_tot)
{
    __shared__ uint32_t scratch[8];
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
    if (threadIdx.x == 0) digit_tot[blockIdx.x] = carry;
}

template <bool DROP_INVALID, int ITEMS, bool HAS_V2, bool INDEX_VALS>
__global__ __launch_bounds__(SORT_THREADS) void radix_scatter_kernel(
    const uint32_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, const uint32_t *__restrict__ vals2_in,
    uint32_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint32_t *__restrict__ vals2_out, const uint32_t *n_dev,
    uint32_t n_bound, int shift, uint32_t mask, const uint32_t *__restrict__ hist, int hist_blocks,
    const uint32_t *__restrict__ digit_tot, uint32_t *n_out)
{
    constexpr int TILE = SORT_THREADS * ITEMS;
    __shared__ uint32_t wave_cnt[4][256];   // per-wave digit counts, then per-wave exclusive bases
    __shared__ uint32_t digit_base[256];    // global position of this tile's run of digit d
    __shared__ uint32_t tile_start[256];    // start of digit d inside the reordered tile
    __shared__ uint32_t skey[TILE];
    __shared__ uint32_t sval[TILE];
    __shared__ uint32_t sval2[HAS_V2 ? TILE : 1];
    __shared__ uint32_t scratch[8];
    __shared__ uint32_t s_valid;

    const uint32_t n = load_count(n_dev, n_bound);
    const uint32_t base = blockIdx.x * TILE;
    if (base >= n) return;  // uniform per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    GSR_STAMP(0);

#pragma unroll
    for (int w = 0; w < 4; ++w) {
        wave_cnt[w][tid] = 0;
        reinterpret_cast<unsigned long long *>(skey)[w * 256 + tid] = 0ull;  // peer masks (see the ranking below)
    }
    __syncthreads();

    // wave w owns items [w*64*ITEMS, (w+1)*64*ITEMS) of the tile, ITEMS rounds of 64 consecutive keys:
    // tile order == (wave, round, lane) order, which is what keeps the sort stable.
    uint32_t key[ITEMS], val[ITEMS], val2[HAS_V2 ? ITEMS : 1], rank[ITEMS];
    if (base + TILE <= n) {  // full tile (all but the last workgroup): unguarded loads, all in flight together
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const uint32_t idx = base + wave * (64 * ITEMS) + r * 64 + lane;
            key[r] = keys_in[idx];
            val[r] = INDEX_VALS ? idx : vals_in[idx];  // INDEX_VALS: the payload is the element's index, nothing to load
            if (HAS_V2) val2[r] = vals2_in[idx];
        }
    } else {
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const uint32_t idx = base + wave * (64 * ITEMS) + r * 64 + lane;
            const bool in = idx < n;
            key[r] = in ? keys_in[idx] : KEY_INVALID;
            val[r] = INDEX_VALS ? idx : (in ? vals_in[idx] : 0u);
            if (HAS_V2) val2[r] = in ? vals2_in[idx] : 0u;
        }
    }
#ifdef GSR_SORT_TRACE
    { uint32_t acc = 0;
#pragma unroll
      for (int r = 0; r < ITEMS; ++r) acc += key[r] + val[r];
      asm volatile("" ::"v"(acc)); }  // wait for the loads
#endif
    GSR_STAMP(1);
    {
    // Ranking: for every key, how many EARLIER keys of this wave carry the same digit (earlier round, or same round and
    // lower lane) -- what keeps the sort stable.  The lanes of one round that share a digit find each other through LDS:
    // each ORs its lane bit into peer[digit], reads the mask back and clears it.  DS instructions of one wave execute in
    // program order, so the read sees the whole round's ORs and the next round finds zeros; no barrier, no waiting between
    // rounds.  (The textbook alternative, eight ballots per round with a per-lane 64-bit select after each, measured
    // 8 us of this kernel's 17 us per workgroup: ~50 VALU instructions per round.)  The lowest lane of each group then
    // advances the wave's running count of that digit and hands the old value to its peers.
    unsigned long long *pm = reinterpret_cast<unsigned long long *>(skey) + wave * 256;  // skey is not live until the reorder
    uint32_t *wc = wave_cnt[wave];
    const unsigned long long my_bit = 1ull << lane;
    // GROUP rounds at a time: all their LDS traffic is issued back to back (three waits per group instead of per round)
    constexpr int GROUP = 4;
    static_assert(ITEMS % GROUP == 0, "ITEMS must be a multiple of the ranking group");
#pragma unroll
    for (int r0 = 0; r0 < ITEMS; r0 += GROUP) {
        unsigned long long m[GROUP];
        uint32_t prior[GROUP];
#pragma unroll
        for (int q = 0; q < GROUP; ++q) {
            const int r = r0 + q;
            const uint32_t idx = base + wave * (64 * ITEMS) + r * 64 + lane;
            const bool valid = (idx < n) && (!DROP_INVALID || key[r] != KEY_INVALID);
            const uint32_t d = (key[r] >> shift) & mask;
            m[q] = my_bit;
            if (valid) {
                __hip_atomic_fetch_or(&pm[d], my_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                m[q] = __hip_atomic_load(&pm[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_store(&pm[d], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
#pragma unroll
        for (int q = 0; q < GROUP; ++q) {
            const int r = r0 + q;
            const uint32_t idx = base + wave * (64 * ITEMS) + r * 64 + lane;
            const bool valid = (idx < n) && (!DROP_INVALID || key[r] != KEY_INVALID);
            const uint32_t d = (key[r] >> shift) & mask;
            prior[q] = 0;
            if (valid && (m[q] & lt_mask) == 0)
                prior[q] = __hip_atomic_fetch_add(&wc[d], (uint32_t)__popcll(m[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
#pragma unroll
        for (int q = 0; q < GROUP; ++q) {
            const int r = r0 + q;
            const uint32_t idx = base + wave * (64 * ITEMS) + r * 64 + lane;
            const bool valid = (idx < n) && (!DROP_INVALID || key[r] != KEY_INVALID);
            const uint32_t p = (uint32_t)__shfl((int)prior[q], __ffsll((long long)m[q]) - 1, 64);  // from the group's lowest lane
            rank[r] = valid ? p + (uint32_t)__popcll(m[q] & lt_mask) : 0xFFFFFFFFu;
        }
    }
    }
    __syncthreads();
    GSR_STAMP(2);

    // digit d = tid: per-wave exclusive bases, tile digit starts, global digit bases
    {
        const uint32_t c0 = wave_cnt[0][tid], c1 = wave_cnt[1][tid], c2 = wave_cnt[2][tid], c3 = wave_cnt[3][tid];
        const uint32_t cnt = c0 + c1 + c2 + c3;
        uint32_t tile_total, all_total;
        const uint32_t ts = block_excl_scan_256(cnt, scratch, &tile_total);
        const uint32_t gs = block_excl_scan_256(digit_tot[tid], scratch, &all_total);
        wave_cnt[0][tid] = 0; wave_cnt[1][tid] = c0; wave_cnt[2][tid] = c0 + c1; wave_cnt[3][tid] = c0 + c1 + c2;
        tile_start[tid] = ts;
        digit_base[tid] = gs + hist[(size_t)tid * hist_blocks + blockIdx.x];
        if (tid == 0) {
            s_valid = tile_total;
            if (n_out != nullptr && blockIdx.x == 0) *n_out = all_total;
        }
    }
    __syncthreads();
    GSR_STAMP(3);

#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        if (rank[r] != 0xFFFFFFFFu) {
            const uint32_t d = (key[r] >> shift) & mask;
            const uint32_t pos = tile_start[d] + wave_cnt[wave][d] + rank[r];
            skey[pos] = key[r];
            sval[pos] = val[r];
            if (HAS_V2) sval2[pos] = val2[r];
        }
    }
    __syncthreads();

    GSR_STAMP(4);
    const uint32_t nvalid = s_valid;
    for (uint32_t i = tid; i < nvalid; i += SORT_THREADS) {
        const uint32_t k = skey[i];
        const uint32_t d = (k >> shift) & mask;
        const uint32_t gpos = digit_base[d] + (i - tile_start[d]);
        keys_out[gpos] = k;
        vals_out[gpos] = sval[i];
        if (HAS_V2) vals2_out[gpos] = sval2[i];
    }
#ifdef GSR_SORT_TRACE
    __syncthreads();
#endif
    GSR_STAMP(5);
}

template <int ITEMS, bool HAS_V2>
static int sort_passes(uint32_t *const key[2], uint32_t *const val[2], uint32_t *const val2[2], const uint32_t *n_dev,
                       int64_t n_bound, int key_bits, bool drop_invalid_first, bool index_values,
                       uint32_t *n_out, const Workspace &ws, int *result_buf, hipStream_t s)
{
    constexpr int TILE = SORT_THREADS * ITEMS;
    const int nblk = (int)((n_bound + TILE - 1) / TILE);
    if (nblk > ws.hist_blocks) { set_error("radix sort: %d tiles exceed the histogram stride %d", nblk, ws.hist_blocks); return GSR_ERR_WORKSPACE; }
    int cur = 0;
    const uint32_t *cnt_dev = n_dev;
    uint32_t *dt = ws.ctrl->digit_tot;
    // digits: as few passes as 8-bit digits allow, the key bits split evenly over them (13 tile-id bits -> 7 + 6, not 8 + 5:
    // fewer digit values per pass mean longer runs per workgroup, i.e. fuller 128-B lines in the scattered writes)
    const int passes = (key_bits + 7) / 8;
    const int bits_pp = (key_bits + passes - 1) / passes;
    for (int p = 0; p < passes; ++p) {
        const int shift = bits_pp * p;
        const uint32_t mask = (1u << std::min(bits_pp, key_bits - shift)) - 1u;
        const bool drop = drop_invalid_first && p == 0;
        const uint32_t *v2i = HAS_V2 ? val2[cur] : nullptr;
        uint32_t *v2o = HAS_V2 ? val2[cur ^ 1] : nullptr;
        const bool ident = index_values && p == 0;  // pass 0 can synthesise value = index instead of loading it
#define GSR_HIST(DROP)                                                                                                             \
    hipLaunchKernelGGL((radix_hist_kernel<DROP, ITEMS>), dim3(nblk), dim3(SORT_THREADS), 0, s, key[cur], cnt_dev, (uint32_t)n_bound, \
                       shift, mask, ws.hist, ws.hist_blocks)
#define GSR_SCATTER(DROP, IDENT)                                                                                                   \
    hipLaunchKernelGGL((radix_scatter_kernel<DROP, ITEMS, HAS_V2, IDENT>), dim3(nblk), dim3(SORT_THREADS), 0, s, key[cur], val[cur], \
                       v2i, key[cur ^ 1], val[cur ^ 1], v2o, cnt_dev, (uint32_t)n_bound, shift, mask, ws.hist, ws.hist_blocks, dt,         \
                       drop ? n_out : (uint32_t *)nullptr)
        if (drop) {
            GSR_HIST(true);
            hipLaunchKernelGGL(radix_rowscan_kernel, dim3(256), dim3(256), 0, s, ws.hist, ws.hist_blocks, nblk, dt);
            if (ident) GSR_SCATTER(true, true); else GSR_SCATTER(true, false);
            if (n_out) cnt_dev = n_out;  // later passes only see the survivors
        } else {
            GSR_HIST(false);
            hipLaunchKernelGGL(radix_rowscan_kernel, dim3(256), dim3(256), 0, s, ws.hist, ws.hist_blocks, nblk, dt);
            if (ident) GSR_SCATTER(false, true); else GSR_SCATTER(false, false);
        }
#undef GSR_SCATTER
#undef GSR_HIST
        GSR_HIP(hipGetLastError());
        cur ^= 1;
    }
    *result_buf = cur;
    return GSR_OK;
}

int launch_radix_sort(uint32_t *const key[2], uint32_t *const val[2], uint32_t *const val2[2], const uint32_t *n_dev,
                      int64_t n_bound, int key_bits, bool drop_invalid_first, bool index_values,
                      uint32_t *n_out, int items_per_thread, const Workspace &ws, int *result_buf, hipStream_t s)
{
    *result_buf = 0;
    if (n_bound <= 0 || key_bits <= 0) return GSR_OK;
    if (items_per_thread != 16) { set_error("radix sort: unsupported items_per_thread %d", items_per_thread); return GSR_ERR_BAD_ARG; }
    return val2 ? sort_passes<16, true>(key, val, val2, n_dev, n_bound, key_bits, drop_invalid_first, index_values, n_out, ws, result_buf, s)
                : sort_passes<16, false>(key, val, val2, n_dev, n_bound, key_bits, drop_invalid_first, index_values, n_out, ws, result_buf, s);
}

}  // namespace gsr

#ifdef GSR_SORT_TRACE
extern "C" int gsr_debug_sort_trace(void *dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gsr::g_sort_trace), bytes < sizeof(gsr::g_sort_trace) ? bytes : sizeof(gsr::g_sort_trace));
}
#endif
