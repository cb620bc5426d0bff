// blend.hip — stage 3: front-to-back alpha compositing of every 16x16 tile.
//
// Replaces the reference's hot loop: the driver at rasterize.py:436-446 and rasterize_gaussian
// (:255-305), which together visit ONE gaussian at a time from the Python interpreter.  Here one
// 256-thread workgroup owns one tile; wave w owns the 8x8 quadrant (w&1, w>>1) and one lane one pixel.
//
// Per tile list (depth-ordered by the two radix sorts):
//   - 256 entries at a time are staged in LDS: each thread gathers one 48-B record (3 x 16-B loads)
//     addressed by the pair value, so HBM/L2 sees 16-B vector loads and the list itself is read coalesced;
//   - footprint_hits_rect (footprint.h) decides, 64 entries per instruction, whether an entry can touch this
//     wave's quadrant; the wave walks only the surviving bits of the ballot (scalar loop, no divergence),
//     two survivors per trip so the second one's LDS broadcast reads overlap the first one's arithmetic;
//   - per pixel the reference's arithmetic: power (log2 domain, coefficients pre-scaled in preprocess),
//     alpha = min(opacity * 2^power, 0.99), contribute iff alpha > 1/255 and power <= 0 (:285-291),
//     C += alpha * T * rgb, T *= 1 - alpha (:295-303);
//   - saturation early-out by wave ballot: a wave stops once T <= opts.early_out_T holds for all its 64 pixels,
//     the workgroup stops fetching once all four waves have.  The default threshold 0 is EXACT, not an
//     approximation of the reference's "blend every gaussian" (Q5): transmittance only ever shrinks, and once it
//     has underflowed to 0.0f (a few dozen near-opaque layers) alpha*T*rgb = 0 and T stays 0 — the remaining
//     entries cannot change a bit.  early_out_T > 0 (INRIA uses 1e-4) is the usual bounded approximation.
//
// Launch order: list lengths are heavy-tailed (longest ~3.5x the mean) and a frame is only ~4 rounds of
// resident workgroups, so tiles are launched longest-first (tile_order_kernel: per XCD group, bucketed by
// length).  XCD k (workgroups b with b % 8 == k) takes the tile rows k, k+8, ...: x-neighbours, which share
// most of their gaussians, hit the same L2, and heavy image regions are spread over all XCDs.
//
// Roofline: algorithmic bytes = 40 per consumed entry (4 id + 36 record) + 12 per pixel
// + 8 per tile range.  The kernel is bound on-chip, not by HBM: ~21 VALU issues per evaluated (quadrant, entry) and three
// wave-wide LDS broadcast reads of its record (10 LDS cycles; the CU's one LDS pipe is 72 % busy).  Removing VALU work
// alone does not move it (DESIGN.md §5 lists the variants); bench.py reports the HBM fraction and the VALU issue fraction.
#include "gsr_internal.h"
#include "blend_args.h"
#include "footprint.h"

namespace gsr {


// one (pixel, entry) evaluation; g = {mean_x, mean_y}, c = {A, B, C, -}, o = {log2(opacity), r, g, b}.
// 17 VALU issues (tools/valu_microbench.hip prices them):
//   - log2(opacity) rides in the quadratic's constant term: alpha = 2^p with p = power + L, and the reference's
//     `power <= 0` becomes p <= L;
//   - T*(1-alpha) is evaluated as T - alpha*T, reusing the product the colour update needs.
__device__ __forceinline__ void blend_one(const float2 g, const float4 c, const float4 o, float fpx, float fpy, float &T,
                                          float &Cr, float &Cg, float &Cb)
{
    const float dx = g.x - fpx, dy = g.y - fpy;
    const float p = fmaf(dx, fmaf(c.y, dy, c.x * dx), fmaf(c.z * dy, dy, o.x));  // log2 domain, opacity folded in
    float alpha = fminf(__builtin_amdgcn_exp2f(p), GSR_MAX_ALPHA);
    const bool valid = (alpha > GSR_MIN_ALPHA) & (p <= o.x);
    alpha = valid ? alpha : 0.0f;
    const float w = alpha * T;
    Cr = fmaf(w, o.y, Cr);
    Cg = fmaf(w, o.z, Cg);
    Cb = fmaf(w, o.w, Cb);
    T = T - w;
}

// The survivor walk of one 64-entry chunk as ONE asm statement (m != 0 on entry): for every set bit of the ballot `m`, in
// order, read the record from LDS (wave-uniform address: three broadcast reads) and blend it into the lane's pixel — the
// arithmetic of blend_one above, instruction for instruction as the compiler emits it (w = T alpha, T = fma(-T, alpha, T)),
// so frames are bit-identical to the plain kernel (blend_impl = 1; tools/blend_ab.py checks it).  What differs:
//   - the validity test goes straight into EXEC: two v_cmpx, the five update instructions run under it, one s_mov restores
//     EXEC.  tools/valu_microbench.hip: v_cmp to an SGPR pair 1.76 ns and the v_cndmask that consumes it 1.83 ns per
//     wave-instruction per SIMD, v_cmpx 1.1 ns like any plain VALU.  Lanes that fail keep C and T untouched, which is what
//     adding w = 0 did.
//   - v_cmpx_le sits between v_exp_f32 and the v_min_f32 that consumes it: on gfx940+ a VALU reading a transcendental's result
//     needs one wait state; the compiler inserts it for its own code, never inside inline asm (without it the first four lanes
//     of every eight read a stale value — found by tools/cmpx_test.hip).
//   - the walk itself costs 5 scalar instructions per survivor (s_ff1 + s_bitset0 pop, loop test, EXEC restore) instead of the
//     compiler's 10.8 (64-bit m & (m - 1) as add/addc/and, address shifts, selects), and the accumulators never change
//     registers (the compiler's one-or-two-per-trip loop copied T and the colour sums on every trip: 2 VALU per entry).
//   - survivors flagged `fast` by the culling lane (footprint_classify: neither the 0.99 clamp nor the `p <= L` test can fire
//     anywhere on this wave's quadrant) take an evaluation without v_min and the second v_cmpx: 15 instead of 17 VALU;
//   - records roll through two register sets: while survivor k is evaluated the reads of survivor k+1 are in flight (counted
//     waits: LDS returns in order; the lgkmcnt(0) up front retires anything older, scalar loads included, which do not).
// Registers v40-v63 are named explicitly (an asm statement takes at most 30 operands): v40/v41 addresses then temporaries,
// v42-v51 and v52-v61 the two records, v62/v63 temporaries; the kernel stays at 64 VGPRs = 8 waves per SIMD.
// Measured on the garden frame (tools/blend_ab.py, interleaved A/B in one process): 0.606 ms against 0.722 ms for the plain
// kernel (0.642 ms with every survivor on the guarded path).  Variants that did not pay: two quadrants per
// wave (16x8 half-tiles, one LDS read for both: +4 %, the LDS pipe is not the limit); skipping the pass of an empty 32-lane
// EXEC half (the hardware does not: 8x4 culling is pointless); the same walk without the rolling prefetch (equal: 8 waves per
// SIMD already hide the LDS latency).  The text below is generated by tools/gen_blend_walk.py.
// lds_chunk = LDS byte address of the chunk's first record (plane 0), identical in every lane.
__device__ __forceinline__ void blend_walk_asm(unsigned long long m, unsigned long long fast, unsigned lds_chunk, float fpx, float fpy,
                                               float &T, float &Cr, float &Cg, float &Cb)
{
    int ia, ib;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_ff1_i32_b64 %[ia], %[m]\n\t"
        "s_bitset0_b64 %[m], %[ia]\n\t"
        "v_lshl_add_u32 v40, %[ia], 4, %[base]\n\t"
        "ds_read_b64 v[42:43], v40\n\t"
        "ds_read_b128 v[44:47], v40 offset:4096\n\t"
        "ds_read_b128 v[48:51], v40 offset:8192\n\t"
        "1:\n\t"
        "s_cmp_eq_u64 %[m], 0\n\t"
        "s_cbranch_scc1 3f\n\t"
        "s_ff1_i32_b64 %[ib], %[m]\n\t"
        "s_bitset0_b64 %[m], %[ib]\n\t"
        "v_lshl_add_u32 v41, %[ib], 4, %[base]\n\t"
        "ds_read_b64 v[52:53], v41\n\t"
        "ds_read_b128 v[54:57], v41 offset:4096\n\t"
        "ds_read_b128 v[58:61], v41 offset:8192\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_sub_f32 v40, v42, %[fpx]\n\t"
        "v_sub_f32 v41, v43, %[fpy]\n\t"
        "v_mul_f32 v62, v44, v40\n\t"
        "v_fma_f32 v62, v45, v41, v62\n\t"
        "v_mul_f32 v63, v46, v41\n\t"
        "v_fma_f32 v63, v63, v41, v48\n\t"
        "v_fma_f32 v62, v40, v62, v63\n\t"
        "v_exp_f32 v63, v62\n\t"
        "s_bitcmp1_b64 %[fast], %[ia]\n\t"
        "s_cbranch_scc1 10f\n\t"
        "v_cmpx_le_f32 vcc, v62, v48\n\t"
        "v_min_f32 v63, 0x3f7d70a4, v63\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[T], v63\n\t"
        "v_fma_f32 %[Cr], v40, v49, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v40, v50, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v40, v51, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v63, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 11f\n\t"
        "10:\n\t"
        "s_nop 0\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[T], v63\n\t"
        "v_fma_f32 %[Cr], v40, v49, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v40, v50, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v40, v51, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v63, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "11:\n\t"
        "s_cmp_eq_u64 %[m], 0\n\t"
        "s_cbranch_scc1 4f\n\t"
        "s_ff1_i32_b64 %[ia], %[m]\n\t"
        "s_bitset0_b64 %[m], %[ia]\n\t"
        "v_lshl_add_u32 v40, %[ia], 4, %[base]\n\t"
        "ds_read_b64 v[42:43], v40\n\t"
        "ds_read_b128 v[44:47], v40 offset:4096\n\t"
        "ds_read_b128 v[48:51], v40 offset:8192\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_sub_f32 v40, v52, %[fpx]\n\t"
        "v_sub_f32 v41, v53, %[fpy]\n\t"
        "v_mul_f32 v62, v54, v40\n\t"
        "v_fma_f32 v62, v55, v41, v62\n\t"
        "v_mul_f32 v63, v56, v41\n\t"
        "v_fma_f32 v63, v63, v41, v58\n\t"
        "v_fma_f32 v62, v40, v62, v63\n\t"
        "v_exp_f32 v63, v62\n\t"
        "s_bitcmp1_b64 %[fast], %[ib]\n\t"
        "s_cbranch_scc1 12f\n\t"
        "v_cmpx_le_f32 vcc, v62, v58\n\t"
        "v_min_f32 v63, 0x3f7d70a4, v63\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[T], v63\n\t"
        "v_fma_f32 %[Cr], v40, v59, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v40, v60, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v40, v61, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v63, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 13f\n\t"
        "12:\n\t"
        "s_nop 0\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[T], v63\n\t"
        "v_fma_f32 %[Cr], v40, v59, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v40, v60, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v40, v61, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v63, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "13:\n\t"
        "s_branch 1b\n\t"
        "3:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_sub_f32 v40, v42, %[fpx]\n\t"
        "v_sub_f32 v41, v43, %[fpy]\n\t"
        "v_mul_f32 v62, v44, v40\n\t"
        "v_fma_f32 v62, v45, v41, v62\n\t"
        "v_mul_f32 v63, v46, v41\n\t"
        "v_fma_f32 v63, v63, v41, v48\n\t"
        "v_fma_f32 v62, v40, v62, v63\n\t"
        "v_exp_f32 v63, v62\n\t"
        "s_bitcmp1_b64 %[fast], %[ia]\n\t"
        "s_cbranch_scc1 14f\n\t"
        "v_cmpx_le_f32 vcc, v62, v48\n\t"
        "v_min_f32 v63, 0x3f7d70a4, v63\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[T], v63\n\t"
        "v_fma_f32 %[Cr], v40, v49, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v40, v50, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v40, v51, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v63, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 15f\n\t"
        "14:\n\t"
        "s_nop 0\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[T], v63\n\t"
        "v_fma_f32 %[Cr], v40, v49, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v40, v50, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v40, v51, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v63, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "15:\n\t"
        "s_branch 5f\n\t"
        "4:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_sub_f32 v40, v52, %[fpx]\n\t"
        "v_sub_f32 v41, v53, %[fpy]\n\t"
        "v_mul_f32 v62, v54, v40\n\t"
        "v_fma_f32 v62, v55, v41, v62\n\t"
        "v_mul_f32 v63, v56, v41\n\t"
        "v_fma_f32 v63, v63, v41, v58\n\t"
        "v_fma_f32 v62, v40, v62, v63\n\t"
        "v_exp_f32 v63, v62\n\t"
        "s_bitcmp1_b64 %[fast], %[ib]\n\t"
        "s_cbranch_scc1 16f\n\t"
        "v_cmpx_le_f32 vcc, v62, v58\n\t"
        "v_min_f32 v63, 0x3f7d70a4, v63\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[T], v63\n\t"
        "v_fma_f32 %[Cr], v40, v59, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v40, v60, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v40, v61, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v63, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 17f\n\t"
        "16:\n\t"
        "s_nop 0\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[T], v63\n\t"
        "v_fma_f32 %[Cr], v40, v59, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v40, v60, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v40, v61, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v63, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "17:\n\t"
        "5:\n\t"
        : [T] "+v"(T), [Cr] "+v"(Cr), [Cg] "+v"(Cg), [Cb] "+v"(Cb), [m] "+s"(m), [ia] "=&s"(ia), [ib] "=&s"(ib)
        : [base] "v"(lds_chunk), [fpx] "v"(fpx), [fpy] "v"(fpy), [fast] "s"(fast)
        : "vcc", "scc", "memory", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",
          "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
}

// Tile launch order.  Group g = tile rows g, g+8, ... of the shard (one XCD's share).  One workgroup per
// group bucket-sorts its tiles by list length, longest first (buckets = exponent + 3 mantissa bits of the
// length, i.e. within 12.5 %): order[8*j + g] = j-th tile of group g.  Slots past the end of a group hold -1.
// Which tile lands where inside a bucket is not deterministic; nothing observable depends on it.
// Also leaves the longest list length in ctrl (stats).
__global__ __launch_bounds__(256) void tile_order_kernel(const uint2 *__restrict__ ranges, FrameCtrl *ctrl, int tiles_x,
                                                         int row_begin, int row_step, int rows, int slots_per_group,
                                                         int *__restrict__ order, uint32_t stats_off)
{
    constexpr int NB = 256;
    __shared__ uint32_t bucket_cnt[NB];
    __shared__ uint32_t bucket_start[NB];
    __shared__ uint32_t scratch[8];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int rows_g = g < rows ? (rows - g + 7) / 8 : 0;
    const int n = rows_g * tiles_x;
    bucket_cnt[tid] = 0;
    for (int j = n + tid; j < slots_per_group; j += 256) order[8 * j + g] = -1;
    __syncthreads();
    auto tile_of = [&](int j) { return (row_begin + (g + 8 * (j / tiles_x)) * row_step) * tiles_x + (j % tiles_x); };
    auto len_of = [&](int tile) { const uint2 r = ranges[tile]; return r.y - r.x; };
    // lengths < 2^24: float conversion is exact; bits >> 20 = exponent (8 bits) and 3 mantissa bits, monotone in len
    auto bucket_of = [&](uint32_t len) { return len == 0 ? (uint32_t)(NB - 1) : min((uint32_t)(NB - 2), (151u << 3) - (__float_as_uint((float)len) >> 20)); };
    uint32_t longest = 0;
    for (int j = tid; j < n; j += 256) {
        const uint32_t len = len_of(tile_of(j));
        longest = max(longest, len);
        atomicAdd(&bucket_cnt[bucket_of(len)], 1u);
    }
    __syncthreads();
    uint32_t total;
    bucket_start[tid] = block_excl_scan_256(bucket_cnt[tid], scratch, &total);
    __syncthreads();
    for (int j = tid; j < n; j += 256) {
        const int tile = tile_of(j);
        const uint32_t slot = atomicAdd(&bucket_start[bucket_of(len_of(tile))], 1u);
        order[8 * (int)slot + g] = tile;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, d, 64));
    if ((tid & 63) == 0 && longest > 0) atomicMax(&ctrl->max_list_len, longest);
    if (g == 0 && tid == 0) {  // where gsr_read_stats finds the counters of the blend launched next
        ctrl->stats_off = stats_off;
        ctrl->stats_slots = 8u * (uint32_t)slots_per_group;
    }
}

template <bool ASM_WALK>
__global__ __launch_bounds__(256) void blend_kernel(BlendArgs a)
{
    __shared__ float4 srec[3][256];  // staged records, one plane per 16-B part: q0 at +0, q1 at +4096, q2 at +8192 bytes
    float4 *const s0 = srec[0], *const s1 = srec[1], *const s2 = srec[2];
    const unsigned lds_rec = (unsigned)(size_t)&srec[0][0];  // LDS byte address: the low half of the flat pointer
    __shared__ int s_done;

    const int tile = a.order[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *stat = a.stats + (size_t)blockIdx.x * BLEND_STAT_WORDS;
    if (tile < 0) {  // uniform: empty launch slot
        if (tid < 5) stat[tid] = 0;
        return;
    }
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;

    const int qx = tx * 16 + (wave & 1) * 8, qy = ty * 16 + (wave >> 1) * 8;
    const int px = qx + (lane & 7), py = qy + (lane >> 3);
    const float fpx = (float)px, fpy = (float)py;
    const float qx0 = (float)qx, qx1 = (float)(qx + 7), qy0 = (float)qy, qy1 = (float)(qy + 7);

    const uint2 range = a.ranges[tile];
    float T = 1.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f;
    bool wave_done = false;
    uint32_t evaluated = 0;  // wave-uniform
    uint32_t fetched = 0;    // workgroup-uniform
    if (tid == 0) s_done = 0;

    for (uint32_t batch = range.x; batch < range.y; batch += 256) {
        __syncthreads();  // previous batch fully consumed (and s_done initialised)
        if (s_done == 4) break;  // uniform: every wave saturated
        const uint32_t i = batch + tid;
        fetched += min(256u, range.y - batch);
        if (i < range.y) {
            const GaussRec *r = a.rec + a.pval[i];
            s0[tid] = r->q0;
            s1[tid] = r->q1;
            s2[tid] = r->q2;
        }
        __syncthreads();
        if (wave_done) continue;
        const int nb = min(256u, range.y - batch);
        for (int chunk = 0; chunk < nb; chunk += 64) {
            const int e = chunk + lane;
            unsigned long long m, fast = 0;
            if (ASM_WALK) {
                FootprintClass fc = {false, false};
                if (e < nb) fc = footprint_classify(s0[e], s1[e], s2[e].x, qx0, qx1, qy0, qy1);
                m = __ballot(fc.hit);
                fast = __ballot(fc.fast);
            } else {
                const bool hit = e < nb && footprint_hits_rect(s0[e], s1[e], qx0, qx1, qy0, qy1);
                m = __ballot(hit);
            }
            evaluated += (uint32_t)__popcll(m);
            if (ASM_WALK) {
                if (m) blend_walk_asm(m, fast, lds_rec + (unsigned)chunk * 16u, fpx, fpy, T, Cr, Cg, Cb);
            } else {
                // plain form: two survivors per trip so that the second one's LDS reads overlap the first one's arithmetic
                while (m) {
                    const int k0 = chunk + (__ffsll((long long)m) - 1);
                    m &= m - 1;
                    const float2 ga = *reinterpret_cast<const float2 *>(&s0[k0]);  // wave-uniform address: LDS broadcast
                    const float4 ca = s1[k0];
                    const float4 oa = s2[k0];
                    asm volatile("" ::"v"(ca.w));  // keep the read a ds_read_b128 (4 LDS cycles); a b96 costs 8
                    if (m) {
                        const int k1 = chunk + (__ffsll((long long)m) - 1);
                        m &= m - 1;
                        const float2 gb = *reinterpret_cast<const float2 *>(&s0[k1]);
                        const float4 cb = s1[k1];
                        const float4 ob = s2[k1];
                        asm volatile("" ::"v"(cb.w));
                        blend_one(ga, ca, oa, fpx, fpy, T, Cr, Cg, Cb);
                        blend_one(gb, cb, ob, fpx, fpy, T, Cr, Cg, Cb);
                    } else {
                        blend_one(ga, ca, oa, fpx, fpy, T, Cr, Cg, Cb);
                    }
                }
            }
            if (__all(T <= a.early_T)) {
                wave_done = true;
                if (lane == 0) atomicAdd(&s_done, 1);
                break;
            }
        }
    }

    if (lane == 0) stat[wave] = evaluated;
    if (tid == 0) stat[4] = fetched;
    if (px < a.W && py < a.H) {
        const bool drawn = px < a.xlim && py < a.ylim;  // Q1: last column / row stay black, T stays 1
        const float r = drawn ? Cr : 0.0f, g = drawn ? Cg : 0.0f, b = drawn ? Cb : 0.0f;
        size_t o;
        if (a.layout == 0) o = ((size_t)py * a.W + px) * 3;                                           // image [H,W,3]
        else if (a.layout == 1) o = ((size_t)px * a.H + py) * 3;                                      // screen [W,H,3]
        else o = ((size_t)(((ty - a.row_begin) / a.row_step) * 16 + (py - ty * 16)) * a.W + px) * 3;  // strip
        store_rgb(a, o, r, g, b);
        if (a.out_T) {
            const size_t ot = a.layout == 1 ? (size_t)px * a.H + py
                            : a.layout == 0 ? (size_t)py * a.W + px
                                            : (size_t)(((ty - a.row_begin) / a.row_step) * 16 + (py - ty * 16)) * a.W + px;
            a.out_T[ot] = drawn ? T : 1.0f;
        }
    }
}

int launch_blend(const GsrCamera &cam, const GsrOptions &opts, const Workspace &ws, const uint32_t *lists, void *out_image,
                 float *out_T, hipStream_t s)
{
    BlendArgs a;
    a.ranges = ws.ranges;
    a.pval = lists;
    a.rec = ws.rec;
    a.out = out_image;
    a.out_T = out_T;
    a.stats = ws.blend_stats;
    a.W = cam.width; a.H = cam.height;
    a.xlim = opts.reference_compat ? cam.width - 1 : cam.width;
    a.ylim = opts.reference_compat ? cam.height - 1 : cam.height;
    a.tiles_x = ws.tiles_x;
    a.row_step = opts.tile_row_step < 1 ? 1 : opts.tile_row_step;
    a.row_begin = opts.tile_row_begin;
    a.rows = a.row_begin < ws.tiles_y ? (ws.tiles_y - a.row_begin + a.row_step - 1) / a.row_step : 0;
    a.layout = opts.output_layout;
    a.out_bf16 = opts.output_dtype == 1;
    a.early_T = opts.early_out_T;
    if (a.rows <= 0 || a.tiles_x <= 0) return GSR_OK;
    const int rows_per_xcd = (a.rows + 7) / 8;
    const int slots_per_group = rows_per_xcd * a.tiles_x;
    a.order = ws.tile_order;
    hipLaunchKernelGGL(tile_order_kernel, dim3(8), dim3(256), 0, s, ws.ranges, ws.ctrl, a.tiles_x, a.row_begin, a.row_step, a.rows,
                       slots_per_group, ws.tile_order,
                       (uint32_t)(reinterpret_cast<const char *>(ws.blend_stats) - reinterpret_cast<const char *>(ws.ctrl)));
    if (opts.blend_impl == 2) return launch_blend_mfma(a, 8u * (unsigned)slots_per_group, s);
    if (opts.blend_impl == 1) hipLaunchKernelGGL(blend_kernel<false>, dim3(8u * (unsigned)slots_per_group), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(blend_kernel<true>, dim3(8u * (unsigned)slots_per_group), dim3(256), 0, s, a);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

// gsr_read_stats: total the per-workgroup counters of the last blend into FrameCtrl.  One workgroup, no atomics;
// the layout comes from FrameCtrl itself because gsr_read_stats is handed nothing but the workspace.
__global__ __launch_bounds__(1024) void blend_stats_kernel(FrameCtrl *ctrl, size_t workspace_bytes)
{
    __shared__ unsigned long long part[2][16];
    const uint32_t off = ctrl->stats_off, slots = ctrl->stats_slots;
    unsigned long long ev = 0, fe = 0;
    if (off >= sizeof(FrameCtrl) && (size_t)off + (size_t)slots * BLEND_STAT_WORDS * 4 <= workspace_bytes) {
        const uint32_t *st = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(ctrl) + off);
        for (uint32_t i = threadIdx.x; i < slots; i += 1024) {
            const uint4 v = *reinterpret_cast<const uint4 *>(st + (size_t)i * BLEND_STAT_WORDS);
            ev += (unsigned long long)v.x + v.y + v.z + v.w;
            fe += st[(size_t)i * BLEND_STAT_WORDS + 4];
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        ev += __shfl_xor(ev, d, 64);
        fe += __shfl_xor(fe, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = ev; part[1][threadIdx.x >> 6] = fe; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ev = fe = 0;
        for (int w = 0; w < 16; ++w) { ev += part[0][w]; fe += part[1][w]; }
        ctrl->wave_entries = ev;
        ctrl->fetched_entries = fe;
    }
}

int launch_blend_stats(FrameCtrl *ctrl, size_t workspace_bytes, hipStream_t s)
{
    hipLaunchKernelGGL(blend_stats_kernel, dim3(1), dim3(1024), 0, s, ctrl, workspace_bytes);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

}  // namespace gsr
