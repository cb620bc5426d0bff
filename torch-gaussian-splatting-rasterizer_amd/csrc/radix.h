// radix.h — workgroup-level building blocks of the stable LSD radix passes (sort.hip).  No reference counterpart: the
// reference orders gaussians with one torch.sort (rasterize.py:424-425) and has no tile lists.
//
// A pass handles a tile of THREADS * ITEMS records per workgroup; wave w owns items [w*64*ITEMS, (w+1)*64*ITEMS) in ITEMS
// rounds of 64 consecutive records, so tile order == (wave, round, lane) order — what keeps every pass stable.
//   radix_clear        zero the per-wave digit counters and the peer masks
//   radix_rank         rank[r] = number of EARLIER records of this wave with the same digit (through LDS peer masks)
//   radix_tile_layout  per-wave exclusive bases + start of every digit inside the reordered tile
//   radix_positions    where every record lands in the reordered tile; the caller then moves ONE array at a time through the
//                      tile buffer (records -> LDS in digit order -> digit runs written out contiguously)
#pragma once
#include "gsr_internal.h"

namespace gsr {

constexpr uint32_t RADIX_NO_DIGIT = 0xFFFFFFFFu;  // a record that takes no part in the pass (out of range / dropped)

// How a pass reads its digit: d = ((key - key_base) >> shift) & mask.  dyn_pass >= 0 marks a pass of the DEPTH sort,
// whose digit geometry is decided on the device from the frame's key range (see sort.hip); -1 = as given here.
struct PassSpec {
    int shift;
    uint32_t mask;
    uint32_t key_base;
    uint32_t drop_from;  // DROP passes discard records with key >= drop_from
    int dyn_pass;
    int enqueued;        // depth sort: passes the host enqueued (the plan may ask for more: flagged by the pass-0 rowscan)
};

// The depth sort's keys are the IEEE bits of z_cam >= 0.2 (rasterize.py:377): everything below bits(0.2f) is constant.
constexpr uint32_t DEPTH_KEY_BASE = 0x3E4CCCCDu;  // __float_as_uint(GSR_CULL_Z)
constexpr int DEPTH_DIGIT_BITS = 9;               // widest digit (512 values) of a depth-sort pass

// Resolves the digit geometry of a pass.  Returns false when a dynamic pass is not needed this frame (uniform).
__device__ __forceinline__ bool resolve_pass(const PassSpec &a, const FrameCtrl *ctrl, int *shift, uint32_t *mask)
{
    if (a.dyn_pass < 0) { *shift = a.shift; *mask = a.mask; return true; }
    if (a.dyn_pass == 0) { *shift = 0; *mask = (1u << DEPTH_DIGIT_BITS) - 1u; return true; }
    const uint32_t passes = ctrl->sort_passes, bits = ctrl->sort_key_bits, per = ctrl->sort_bits_rest;
    if ((uint32_t)a.dyn_pass >= passes) return false;
    const uint32_t sh = DEPTH_DIGIT_BITS + (uint32_t)(a.dyn_pass - 1) * per;
    const uint32_t nb = min(per, bits - sh);
    *shift = (int)sh;
    *mask = (1u << nb) - 1u;
    return true;
}

// One workgroup of THREADS threads sorts a tile of THREADS * ITEMS records on a digit of up to log2(THREADS) bits: thread t
// owns digit t in the per-digit steps.  256 threads / 4096 records for the pair sort (8-bit digits and fewer), 512 threads /
// 8192 records for the depth sort's 9-bit digits: with twice the digit values a 4096-record tile's runs halve (8 records =
// 32 B per array, measured +35 % per pass); doubling the tile keeps the runs at 16 records and the table reads per record equal.
// LDS budget: the arrays of a record (key, value, second value) pass through ONE tile-sized buffer one after the other instead of
// each having its own (three barriers more per tile): 52 KB instead of 116 KB for the depth sort's 512-thread tiles, so that three
// workgroups share a CU instead of one — the passes are bound by the latency of a tile's dependent phases (load, rank, layout,
// reorder, store), not by any pipe, and with one workgroup per CU a 410-tile pass ran as two rounds of 256 + 154.
template <int THREADS, int ITEMS, bool HAS_V2>
struct RadixTileSmem {
    static constexpr int TILE = THREADS * ITEMS, WAVES = THREADS / 64, DIGITS = THREADS;
    static_assert(THREADS == 256 || THREADS == 512, "digit d is owned by thread d");
    static constexpr int BUF_WORDS = TILE > WAVES * DIGITS * 2 ? TILE : WAVES * DIGITS * 2;  // the peer masks (8 B per wave and digit) live here too
    uint32_t wave_cnt[WAVES][DIGITS];  // per-wave digit counts, then per-wave exclusive bases
    uint32_t tile_start[DIGITS];       // start of digit d inside the reordered tile
    uint32_t buf[BUF_WORDS];           // peer masks while ranking, then one array of the tile at a time in digit order
    uint32_t scratch[2 * WAVES];
    uint32_t n_valid;
};

template <int THREADS, int ITEMS, bool HAS_V2>
__device__ __forceinline__ void radix_clear(RadixTileSmem<THREADS, ITEMS, HAS_V2> &sm)
{
    constexpr int WAVES = THREADS / 64;
    uint32_t *wc = &sm.wave_cnt[0][0];
    unsigned long long *pm = reinterpret_cast<unsigned long long *>(sm.buf);
#pragma unroll
    for (int i = 0; i < WAVES; ++i) {
        wc[i * THREADS + threadIdx.x] = 0u;
        pm[i * THREADS + threadIdx.x] = 0ull;
    }
}

// Ranking: for every record, how many EARLIER records of this wave carry the same digit (earlier round, or same round and
// lower lane).  The lanes of one round that share a digit find each other through LDS: each ORs its lane bit into
// peer[digit], reads the mask back and clears it.  DS instructions of one wave execute in program order, so the read sees
// the whole round's ORs and the next round finds zeros; no barrier, no waiting between rounds.  (The textbook alternative,
// eight ballots per round with a per-lane 64-bit select after each, measured 8 us of the scatter kernel's 17 us per
// workgroup: ~50 VALU instructions per round.)  The lowest lane of each group then advances the wave's running count of
// that digit and hands the old value to its peers.  dig(r) returns the digit of this thread's r-th record or RADIX_NO_DIGIT.
template <int THREADS, int ITEMS, bool HAS_V2, typename DigitOf>
__device__ __forceinline__ void radix_rank(RadixTileSmem<THREADS, ITEMS, HAS_V2> &sm, DigitOf dig, uint32_t (&rank)[ITEMS])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long *pm = reinterpret_cast<unsigned long long *>(sm.buf) + wave * THREADS;  // the tile buffer is not live until the reorder
    uint32_t *wc = sm.wave_cnt[wave];
    const unsigned long long my_bit = 1ull << lane, lt_mask = my_bit - 1ull;
    // GROUP rounds at a time: all their LDS traffic is issued back to back (three waits per group instead of per round)
    constexpr int GROUP = 4;
    static_assert(ITEMS % GROUP == 0, "ITEMS must be a multiple of the ranking group");
#pragma unroll
    for (int r0 = 0; r0 < ITEMS; r0 += GROUP) {
        unsigned long long m[GROUP];
        uint32_t prior[GROUP];
#pragma unroll
        for (int q = 0; q < GROUP; ++q) {
            const uint32_t d = dig(r0 + q);
            m[q] = my_bit;
            if (d != RADIX_NO_DIGIT) {
                __hip_atomic_fetch_or(&pm[d], my_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                m[q] = __hip_atomic_load(&pm[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                __hip_atomic_store(&pm[d], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
        }
#pragma unroll
        for (int q = 0; q < GROUP; ++q) {
            const uint32_t d = dig(r0 + q);
            prior[q] = 0;
            if (d != RADIX_NO_DIGIT && (m[q] & lt_mask) == 0)
                prior[q] = __hip_atomic_fetch_add(&wc[d], (uint32_t)__popcll(m[q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
#pragma unroll
        for (int q = 0; q < GROUP; ++q) {
            const uint32_t d = dig(r0 + q);
            const uint32_t p = (uint32_t)__shfl((int)prior[q], __ffsll((long long)m[q]) - 1, 64);  // from the group's lowest lane
            rank[r0 + q] = d != RADIX_NO_DIGIT ? p + (uint32_t)__popcll(m[q] & lt_mask) : RADIX_NO_DIGIT;
        }
    }
}

// After radix_rank + a barrier.  Thread t owns digit t: turns the per-wave counts into per-wave exclusive bases, fills
// tile_start[], leaves the tile's record count in sm.n_valid.  Contains two barriers; the caller adds one before using
// the tables.
template <int THREADS, int ITEMS, bool HAS_V2>
__device__ __forceinline__ void radix_tile_layout(RadixTileSmem<THREADS, ITEMS, HAS_V2> &sm)
{
    constexpr int WAVES = THREADS / 64;
    const int d = threadIdx.x;
    uint32_t c[WAVES], cnt = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { c[w] = sm.wave_cnt[w][d]; cnt += c[w]; }
    uint32_t total;
    sm.tile_start[d] = block_excl_scan<THREADS>(cnt, sm.scratch, &total);
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { sm.wave_cnt[w][d] = run; run += c[w]; }
    if (threadIdx.x == 0) sm.n_valid = total;
}

// After radix_tile_layout + a barrier: rank[r] (rank among the wave's earlier records of the same digit) -> the record's position
// in the reordered tile (stable), RADIX_NO_DIGIT stays.  In place.
template <int THREADS, int ITEMS, bool HAS_V2, typename DigitOf>
__device__ __forceinline__ void radix_positions(const RadixTileSmem<THREADS, ITEMS, HAS_V2> &sm, DigitOf dig, uint32_t (&rank)[ITEMS])
{
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        if (rank[r] != RADIX_NO_DIGIT) {
            const uint32_t d = dig(r);
            rank[r] = sm.tile_start[d] + sm.wave_cnt[wave][d] + rank[r];
        }
    }
}

// One array of the tile -> the tile buffer, in digit order.  The caller brackets it with barriers.
template <int THREADS, int ITEMS, bool HAS_V2>
__device__ __forceinline__ void radix_stage(RadixTileSmem<THREADS, ITEMS, HAS_V2> &sm, const uint32_t (&pos)[ITEMS], const uint32_t (&v)[ITEMS])
{
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
        if (pos[r] != RADIX_NO_DIGIT) sm.buf[pos[r]] = v[r];
}

}  // namespace gsr
