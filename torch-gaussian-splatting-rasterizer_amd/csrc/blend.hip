// blend.hip — stage 3: front-to-back alpha compositing of every 16x16 tile.
//
// Replaces the reference's hot loop: the driver at rasterize.py:436-446 and rasterize_gaussian
// (:255-305), which together visit ONE gaussian at a time from the Python interpreter.  Here one
// workgroup owns one tile, a wave one or two of its 8x8 quadrants, a lane one pixel of each.
//
// Per tile list (depth-ordered by the two radix sorts):
//   - 128 (256) entries at a time are staged in LDS: each thread gathers one 48-B record (3 x 16-B loads)
//     addressed by the list entry, so HBM/L2 sees 16-B vector loads and the list itself is read coalesced; a record whose colour
//     nobody has evaluated yet gets it here (staged_q2: sh_to_rgb, spherical_harmonics.py:27-73, when a tile first stages the
//     gaussian — most gaussians of a dense scene are never staged by anyone);
//   - footprint.h decides, 64 entries per instruction, whether an entry can touch the wave's quadrant(s); the wave
//     walks only the surviving bits of the ballot (scalar loop, no divergence);
//   - per pixel the reference's arithmetic: power (log2 domain, coefficients pre-scaled in preprocess),
//     alpha = min(opacity * 2^power, 0.99), contribute iff alpha > 1/255 and power <= 0 (:285-291),
//     C += alpha * T * rgb, T *= 1 - alpha (:295-303);
//   - saturation early-out by wave ballot: a quadrant stops once every one of its 64 pixels is finished (blend_args.h,
//     pixel_finished), the workgroup stops fetching once every wave has.  The default is EXACT, not an approximation of the
//     reference's "blend every gaussian" (Q5): a pixel is finished when no later entry can change a bit of its colour — T has
//     fallen below half an ulp of each colour sum (T <= 2^-25 min C: the single-rounding fma returns C unchanged from then on), or,
//     GsrOptions.saturation_rule = 1 and whenever the final T is an output, T has underflowed to 0.0f.  early_out_T > 0 (INRIA uses
//     1e-4) is the usual bounded approximation.  Bench frame: 26.1 M evaluated (quadrant, entry) pairs under the T == 0 rule.
// Two kernels share this: blend_kernel, the plain-C statement (blend_impl = 1), and blend_walk_kernel, the product, whose
// inner walk is one hand-scheduled asm statement — bit-identical frames (its comment has the measurements).
//
// Launch order: a tile's work is heavy-tailed and a frame is only ~2 rounds of resident workgroups, so tiles are launched
// heaviest-first (tile_order_kernel: per XCD group, bucketed by what the tile's blend staged in the last frame on this workspace,
// by list length where that is unknown).  XCD k (workgroups b with b % 8 == k) takes the tile rows k, k+8, ...: x-neighbours, which share
// most of their gaussians, hit the same L2, and heavy image regions are spread over all XCDs.
//
// Roofline: algorithmic bytes = 40 per consumed entry (4 id + 36 record) + 12 per pixel
// + 8 per tile range + 216 per colour evaluated here (the 192-B SH row, the mean, the write-back).  The kernel is bound on-chip, not
// by HBM: per evaluated (quadrant, entry) ~15 VALU issues incl. one quarter-rate v_exp_f32, and the record's three wave-wide LDS
// broadcast reads (10 LDS cycles); since the colour-saturation rule also by its heaviest tiles.  bench.py reports the HBM fraction
// (the contract figure) and, from the committed PMC passes, the VALU issue and LDS busy fractions.
#include "gsr_internal.h"
#include "blend_args.h"
#include "footprint.h"
#include "gauss_math.h"

namespace gsr {

// A load through a pointer that came out of memory: say that it points to global memory, or the access is a flat_load.
template <typename T>
__device__ __forceinline__ T ldg(const T *p, size_t i)
{
    return ((const __attribute__((address_space(1))) T *)p)[i];
}

// q2 = {log2(opacity), r, g, b} of gaussian `id` for the staging thread.  With GsrOptions.colour_stage = 0 the preprocess leaves
// the colour unevaluated (rgb negative; a colour is clamped to [0, 1], Q7): the first tile that STAGES the gaussian reads its mean
// and its SH row, evaluates sh_to_rgb (spherical_harmonics.py:27-73; gauss_math.h sh_eval, the same code the preprocess runs, same
// operation order, contraction off) and writes the result back into the record for the tiles that stage it later.  Gaussians no
// tile reaches before it is saturated never read their 192-B row — on the bench frame the blend stages 3.0 M of 15.2 M list
// entries.  Races are benign by construction: every writer stores the same three values, and a reader accepts the record's
// colour only when all three channels are non-negative — whatever mix of old and new words it sees (another XCD's L2 may still
// hold the line from before), it either uses the final values or evaluates them itself.
// The record's q2 is read and written with ONE 16-B access each, written as asm volatile: the race between the tiles that stage one
// gaussian is deliberate (see above), and an opaque access is how the compiler is kept out of it — it can neither cache the words nor
// merge or tear the access.  The argument above needs nothing from the hardware beyond "a 32-bit word is read and written whole": a
// reader that sees a stale or half-new record re-evaluates.  Measured alternatives (bench frame, blend kernel): the same as four /
// three relaxed atomics at wavefront scope (plain dword loads and stores: -DGSR_Q2_ATOMICS) 0.235 against 0.228 ms; at agent scope
// (`sc1` on every access: each staged entry's colour bypasses the caches on its way) 0.542 ms.
#ifndef GSR_Q2_SCOPE
#define GSR_Q2_SCOPE __HIP_MEMORY_SCOPE_WAVEFRONT
#endif
__device__ __forceinline__ float q2_load(const float *w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, GSR_Q2_SCOPE); }
__device__ __forceinline__ void q2_store(float *w, float v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, GSR_Q2_SCOPE); }

__device__ __forceinline__ float4 staged_q2(const BlendArgs &a, uint32_t id, uint32_t &evals)
{
    float *w = reinterpret_cast<float *>(&a.rec[id].q2);
#ifndef GSR_Q2_ATOMICS
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v q2v;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(q2v) : "v"(w) : "memory");
    float4 q2 = make_float4(q2v.x, q2v.y, q2v.z, q2v.w);
#else
    float4 q2 = make_float4(q2_load(w), q2_load(w + 1), q2_load(w + 2), q2_load(w + 3));
#endif
    if (!(q2.y >= 0.0f && q2.z >= 0.0f && q2.w >= 0.0f)) {
        const FrameCtrl *c = a.ctrl;
        const bool own = a.col_means != nullptr;  // uniform: gsr_blend was handed the scene
        const float *means = own ? a.col_means : c->col_means;
        const void *sh = own ? a.col_sh : c->col_sh;
        const float cc[3] = {own ? a.col_cc[0] : c->col_cc[0], own ? a.col_cc[1] : c->col_cc[1], own ? a.col_cc[2] : c->col_cc[2]};
        const float p[3] = {ldg(means, 3 * (size_t)id), ldg(means, 3 * (size_t)id + 1), ldg(means, 3 * (size_t)id + 2)};
        float rgb[3];
        // The evaluation consumes the row in memory order (sh_eval_with) and each element is loaded where it is consumed (the compiler
        // merges them into 16-B loads, one in flight at a time): this runs inside a 64-VGPR kernel whose accumulators stay live, and
        // what it costs is the bytes, not the round trips.  Measured on the bench frame (blend kernel, same box): 0.179 ms with the
        // colours already in the records; this 0.234 ms for 1.80 M evaluations; groups of three 16-B pieces, two groups in flight
        // (172 B of scratch per thread) 0.238; the whole row requested at once (spills) 0.327; without the write-back 0.281 ms for
        // 3.00 M evaluations — ~31 ns per 1000 evaluations in every variant, the rate at which the preprocess read the rows it
        // needed (650 MB in 0.11 ms): the saving is the rows never read.  The degree is a constant per branch (the reference
        // always evaluates 3, rasterize.py:368).
        const int degree = own ? a.col_degree : c->col_degree;
        if (own ? a.col_sh16 : c->col_sh16) {
            const unsigned *row = reinterpret_cast<const unsigned *>(static_cast<const char *>(sh) + 96 * (size_t)id);
            auto get = [row](int e) {
                const unsigned w = ldg(row, (size_t)(e >> 1));
                return __half2float(__ushort_as_half((unsigned short)((e & 1) ? w >> 16 : w & 0xFFFFu)));
            };
            if (degree == 3) sh_eval_with(p, get, cc, 3, rgb);
            else sh_eval_with(p, get, cc, degree, rgb);
        } else {
            const float *row = static_cast<const float *>(sh) + 48 * (size_t)id;
            auto get = [row](int e) { return ldg(row, e); };
            if (degree == 3) sh_eval_with(p, get, cc, 3, rgb);
            else sh_eval_with(p, get, cc, degree, rgb);
        }
        q2.y = rgb[0]; q2.z = rgb[1]; q2.w = rgb[2];
#ifndef GSR_Q2_ATOMICS
        q2v.y = q2.y; q2v.z = q2.z; q2v.w = q2.w;
        asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(w), "v"(q2v) : "memory");
#else
        q2_store(w + 1, q2.y); q2_store(w + 2, q2.z); q2_store(w + 3, q2.w);  // (w[0], log2 opacity, is the preprocess's and final)
#endif
        ++evals;
    }
    return q2;
}


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

// The survivor walk of one 64-entry chunk as ONE asm statement (m != 0 on entry), for a wave that owns TWO 8x8 quadrants side
// by side (a 16x8 half-tile), a lane one pixel in each: for every set bit of `m`, in order, read the record from LDS
// (wave-uniform address: three broadcast reads, once for both quadrants) and blend it into the lane's pixel of quadrant A if
// bit `ma` is set, of quadrant B if `mb` is — the arithmetic of blend_one above, instruction for instruction as the compiler
// emits it (w = T alpha, T = fma(-T, alpha, T)), with dy, C dy and t1 = fma(C dy, dy, L) computed once per record: the same
// operations on the same values as two separate evaluations, so frames are bit-identical to the plain kernel (blend_impl = 1;
// tests + tools/blend_ab.py check it).  What differs from what the compiler makes of the plain loop:
//   - the validity test goes straight into EXEC: two v_cmpx, the five update instructions run under it, one s_mov restores
//     EXEC.  tools/valu_microbench.hip: v_cmp to an SGPR pair 1.76 ns and the v_cndmask that consumes it 1.83 ns per
//     wave-instruction per SIMD, v_cmpx 1.1 ns like any plain VALU.  Lanes that fail keep C and T untouched, which is what
//     adding w = 0 did.
//   - something always sits between v_exp_f32 and its consumer (the guarded path's v_cmpx_le, the unguarded path's scalar bit
//     test + branch): on gfx940+ a VALU reading a transcendental's result needs one wait state; the compiler inserts it for
//     its own code, never inside inline asm (without it the first four lanes of every eight read a stale value — found by
//     tools/cmpx_test.hip).
//   - records flagged `fa` / `fb` by the culling lane (footprint_classify: neither the 0.99 clamp nor the `p <= L` test can
//     fire anywhere on that quadrant) take an evaluation without v_min and the second v_cmpx; that is the common case (~80 %)
//     and the straight-line one, the guarded evaluations sit out of line behind the loop (tools/walk_latency.hip: a wave
//     alone 127 -> 111 ns per record with both quadrants hit, at 8 waves per SIMD 152 -> 132 ns);
//   - the walk itself costs ~6 scalar instructions per record (s_ff1 + s_bitset0 pop, bit tests, EXEC restore) instead of the
//     compiler's 10.8 per evaluation, and the accumulators never change registers (the compiler's loop copied them every trip);
//   - records roll through two register sets: while record k is evaluated the reads of record k+1 are in flight (counted
//     waits: LDS returns in order; the lgkmcnt(0) up front retires anything older, scalar loads included, which do not).
// Registers v39-v63 are named explicitly (an asm statement takes at most 30 operands): v40/v41 addresses then temporaries,
// v42-v51 and v52-v61 the two records, v39/v62/v63 temporaries; the kernel stays at 64 VGPRs = 8 waves per SIMD.
// Measured on the bicycle frame (tools/blend_ab.py, interleaved A/B in one process): plain kernel 0.738 ms; EXEC-masked
// update + hand-written walk, one quadrant per wave 0.66 ms; + unguarded fast path 0.613 ms; two quadrants per wave (this)
// 0.570 ms.  PMC per launch, plain (round 1) -> one quadrant -> this: VALU 549 M -> 441 M -> 402 M wave-instructions,
// LDS-array cycles 267 M -> 277 M -> 177 M (82 % -> 52 % of the kernel's cycles), SALU 236 M -> 172 M -> 173 M.
// Did not pay: skipping the pass of an empty 32-lane EXEC half (the hardware does not: 8x4 culling is pointless); the walk
// without the rolling prefetch (equal: 8 waves per SIMD already hide the LDS latency); two quadrants per wave with the
// COMPILER's walk (+4 %: its register copies and selects ate the saved reads).
// The text below is generated by tools/gen_blend_walk.py.  lds_chunk = LDS byte address of the chunk's first record
// (plane 0; the planes are 2048 B apart), identical in every lane.
template <int PLANE>  // bytes between the LDS planes of the staged records = 16 * entries per batch
__device__ __forceinline__ void blend_walk2_asm(unsigned long long m, unsigned long long ma, unsigned long long mb, unsigned long long fa,
                                                unsigned long long fb, unsigned lds_chunk, float fpxa, float fpxb, float fpy, float &Ta,
                                                float &Cra, float &Cga, float &Cba, float &Tb, float &Crb, float &Cgb, float &Cbb)
{
    int ia, ib;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_ff1_i32_b64 %[ia], %[m]\n\t"
        "s_bitset0_b64 %[m], %[ia]\n\t"
        "v_lshl_add_u32 v40, %[ia], 4, %[base]\n\t"
        "ds_read_b64 v[42:43], v40\n\t"
        "ds_read_b128 v[44:47], v40 offset:%[p1]\n\t"
        "ds_read_b128 v[48:51], v40 offset:%[p2]\n\t"
        "1:\n\t"
        "s_cmp_eq_u64 %[m], 0\n\t"
        "s_cbranch_scc1 3f\n\t"
        "s_ff1_i32_b64 %[ib], %[m]\n\t"
        "s_bitset0_b64 %[m], %[ib]\n\t"
        "v_lshl_add_u32 v41, %[ib], 4, %[base]\n\t"
        "ds_read_b64 v[52:53], v41\n\t"
        "ds_read_b128 v[54:57], v41 offset:%[p1]\n\t"
        "ds_read_b128 v[58:61], v41 offset:%[p2]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_sub_f32 v41, v43, %[fpy]\n\t"
        "v_mul_f32 v39, v46, v41\n\t"
        "v_fma_f32 v39, v39, v41, v48\n\t"
        "s_bitcmp1_b64 %[ma], %[ia]\n\t"
        "s_cbranch_scc0 10f\n\t"
        "v_sub_f32 v40, v42, %[fpxa]\n\t"
        "v_mul_f32 v62, v44, v40\n\t"
        "v_fma_f32 v62, v45, v41, v62\n\t"
        "v_fma_f32 v62, v40, v62, v39\n\t"
        "v_exp_f32 v63, v62\n\t"
        "s_bitcmp1_b64 %[fa], %[ia]\n\t"
        "s_cbranch_scc0 11f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[Ta], v63\n\t"
        "v_fma_f32 %[Cra], v40, v49, %[Cra]\n\t"
        "v_fma_f32 %[Cga], v40, v50, %[Cga]\n\t"
        "v_fma_f32 %[Cba], v40, v51, %[Cba]\n\t"
        "v_fma_f32 %[Ta], -%[Ta], v63, %[Ta]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "12:\n\t"
        "10:\n\t"
        "s_bitcmp1_b64 %[mb], %[ia]\n\t"
        "s_cbranch_scc0 13f\n\t"
        "v_sub_f32 v40, v42, %[fpxb]\n\t"
        "v_mul_f32 v62, v44, v40\n\t"
        "v_fma_f32 v62, v45, v41, v62\n\t"
        "v_fma_f32 v62, v40, v62, v39\n\t"
        "v_exp_f32 v63, v62\n\t"
        "s_bitcmp1_b64 %[fb], %[ia]\n\t"
        "s_cbranch_scc0 14f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[Tb], v63\n\t"
        "v_fma_f32 %[Crb], v40, v49, %[Crb]\n\t"
        "v_fma_f32 %[Cgb], v40, v50, %[Cgb]\n\t"
        "v_fma_f32 %[Cbb], v40, v51, %[Cbb]\n\t"
        "v_fma_f32 %[Tb], -%[Tb], v63, %[Tb]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "15:\n\t"
        "13:\n\t"
        "s_cmp_eq_u64 %[m], 0\n\t"
        "s_cbranch_scc1 4f\n\t"
        "s_ff1_i32_b64 %[ia], %[m]\n\t"
        "s_bitset0_b64 %[m], %[ia]\n\t"
        "v_lshl_add_u32 v40, %[ia], 4, %[base]\n\t"
        "ds_read_b64 v[42:43], v40\n\t"
        "ds_read_b128 v[44:47], v40 offset:%[p1]\n\t"
        "ds_read_b128 v[48:51], v40 offset:%[p2]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_sub_f32 v41, v53, %[fpy]\n\t"
        "v_mul_f32 v39, v56, v41\n\t"
        "v_fma_f32 v39, v39, v41, v58\n\t"
        "s_bitcmp1_b64 %[ma], %[ib]\n\t"
        "s_cbranch_scc0 16f\n\t"
        "v_sub_f32 v40, v52, %[fpxa]\n\t"
        "v_mul_f32 v62, v54, v40\n\t"
        "v_fma_f32 v62, v55, v41, v62\n\t"
        "v_fma_f32 v62, v40, v62, v39\n\t"
        "v_exp_f32 v63, v62\n\t"
        "s_bitcmp1_b64 %[fa], %[ib]\n\t"
        "s_cbranch_scc0 17f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[Ta], v63\n\t"
        "v_fma_f32 %[Cra], v40, v59, %[Cra]\n\t"
        "v_fma_f32 %[Cga], v40, v60, %[Cga]\n\t"
        "v_fma_f32 %[Cba], v40, v61, %[Cba]\n\t"
        "v_fma_f32 %[Ta], -%[Ta], v63, %[Ta]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "18:\n\t"
        "16:\n\t"
        "s_bitcmp1_b64 %[mb], %[ib]\n\t"
        "s_cbranch_scc0 19f\n\t"
        "v_sub_f32 v40, v52, %[fpxb]\n\t"
        "v_mul_f32 v62, v54, v40\n\t"
        "v_fma_f32 v62, v55, v41, v62\n\t"
        "v_fma_f32 v62, v40, v62, v39\n\t"
        "v_exp_f32 v63, v62\n\t"
        "s_bitcmp1_b64 %[fb], %[ib]\n\t"
        "s_cbranch_scc0 20f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[Tb], v63\n\t"
        "v_fma_f32 %[Crb], v40, v59, %[Crb]\n\t"
        "v_fma_f32 %[Cgb], v40, v60, %[Cgb]\n\t"
        "v_fma_f32 %[Cbb], v40, v61, %[Cbb]\n\t"
        "v_fma_f32 %[Tb], -%[Tb], v63, %[Tb]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "21:\n\t"
        "19:\n\t"
        "s_branch 1b\n\t"
        "3:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_sub_f32 v41, v43, %[fpy]\n\t"
        "v_mul_f32 v39, v46, v41\n\t"
        "v_fma_f32 v39, v39, v41, v48\n\t"
        "s_bitcmp1_b64 %[ma], %[ia]\n\t"
        "s_cbranch_scc0 22f\n\t"
        "v_sub_f32 v40, v42, %[fpxa]\n\t"
        "v_mul_f32 v62, v44, v40\n\t"
        "v_fma_f32 v62, v45, v41, v62\n\t"
        "v_fma_f32 v62, v40, v62, v39\n\t"
        "v_exp_f32 v63, v62\n\t"
        "s_bitcmp1_b64 %[fa], %[ia]\n\t"
        "s_cbranch_scc0 23f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[Ta], v63\n\t"
        "v_fma_f32 %[Cra], v40, v49, %[Cra]\n\t"
        "v_fma_f32 %[Cga], v40, v50, %[Cga]\n\t"
        "v_fma_f32 %[Cba], v40, v51, %[Cba]\n\t"
        "v_fma_f32 %[Ta], -%[Ta], v63, %[Ta]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "24:\n\t"
        "22:\n\t"
        "s_bitcmp1_b64 %[mb], %[ia]\n\t"
        "s_cbranch_scc0 25f\n\t"
        "v_sub_f32 v40, v42, %[fpxb]\n\t"
        "v_mul_f32 v62, v44, v40\n\t"
        "v_fma_f32 v62, v45, v41, v62\n\t"
        "v_fma_f32 v62, v40, v62, v39\n\t"
        "v_exp_f32 v63, v62\n\t"
        "s_bitcmp1_b64 %[fb], %[ia]\n\t"
        "s_cbranch_scc0 26f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[Tb], v63\n\t"
        "v_fma_f32 %[Crb], v40, v49, %[Crb]\n\t"
        "v_fma_f32 %[Cgb], v40, v50, %[Cgb]\n\t"
        "v_fma_f32 %[Cbb], v40, v51, %[Cbb]\n\t"
        "v_fma_f32 %[Tb], -%[Tb], v63, %[Tb]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "27:\n\t"
        "25:\n\t"
        "s_branch 5f\n\t"
        "4:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_sub_f32 v41, v53, %[fpy]\n\t"
        "v_mul_f32 v39, v56, v41\n\t"
        "v_fma_f32 v39, v39, v41, v58\n\t"
        "s_bitcmp1_b64 %[ma], %[ib]\n\t"
        "s_cbranch_scc0 28f\n\t"
        "v_sub_f32 v40, v52, %[fpxa]\n\t"
        "v_mul_f32 v62, v54, v40\n\t"
        "v_fma_f32 v62, v55, v41, v62\n\t"
        "v_fma_f32 v62, v40, v62, v39\n\t"
        "v_exp_f32 v63, v62\n\t"
        "s_bitcmp1_b64 %[fa], %[ib]\n\t"
        "s_cbranch_scc0 29f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[Ta], v63\n\t"
        "v_fma_f32 %[Cra], v40, v59, %[Cra]\n\t"
        "v_fma_f32 %[Cga], v40, v60, %[Cga]\n\t"
        "v_fma_f32 %[Cba], v40, v61, %[Cba]\n\t"
        "v_fma_f32 %[Ta], -%[Ta], v63, %[Ta]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "30:\n\t"
        "28:\n\t"
        "s_bitcmp1_b64 %[mb], %[ib]\n\t"
        "s_cbranch_scc0 31f\n\t"
        "v_sub_f32 v40, v52, %[fpxb]\n\t"
        "v_mul_f32 v62, v54, v40\n\t"
        "v_fma_f32 v62, v55, v41, v62\n\t"
        "v_fma_f32 v62, v40, v62, v39\n\t"
        "v_exp_f32 v63, v62\n\t"
        "s_bitcmp1_b64 %[fb], %[ib]\n\t"
        "s_cbranch_scc0 32f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[Tb], v63\n\t"
        "v_fma_f32 %[Crb], v40, v59, %[Crb]\n\t"
        "v_fma_f32 %[Cgb], v40, v60, %[Cgb]\n\t"
        "v_fma_f32 %[Cbb], v40, v61, %[Cbb]\n\t"
        "v_fma_f32 %[Tb], -%[Tb], v63, %[Tb]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "33:\n\t"
        "31:\n\t"
        "s_branch 5f\n\t"
        "11:\n\t"
        "v_cmpx_le_f32 vcc, v62, v48\n\t"
        "v_min_f32 v63, 0x3f7d70a4, v63\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[Ta], v63\n\t"
        "v_fma_f32 %[Cra], v40, v49, %[Cra]\n\t"
        "v_fma_f32 %[Cga], v40, v50, %[Cga]\n\t"
        "v_fma_f32 %[Cba], v40, v51, %[Cba]\n\t"
        "v_fma_f32 %[Ta], -%[Ta], v63, %[Ta]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 12b\n\t"
        "14:\n\t"
        "v_cmpx_le_f32 vcc, v62, v48\n\t"
        "v_min_f32 v63, 0x3f7d70a4, v63\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[Tb], v63\n\t"
        "v_fma_f32 %[Crb], v40, v49, %[Crb]\n\t"
        "v_fma_f32 %[Cgb], v40, v50, %[Cgb]\n\t"
        "v_fma_f32 %[Cbb], v40, v51, %[Cbb]\n\t"
        "v_fma_f32 %[Tb], -%[Tb], v63, %[Tb]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 15b\n\t"
        "17:\n\t"
        "v_cmpx_le_f32 vcc, v62, v58\n\t"
        "v_min_f32 v63, 0x3f7d70a4, v63\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[Ta], v63\n\t"
        "v_fma_f32 %[Cra], v40, v59, %[Cra]\n\t"
        "v_fma_f32 %[Cga], v40, v60, %[Cga]\n\t"
        "v_fma_f32 %[Cba], v40, v61, %[Cba]\n\t"
        "v_fma_f32 %[Ta], -%[Ta], v63, %[Ta]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 18b\n\t"
        "20:\n\t"
        "v_cmpx_le_f32 vcc, v62, v58\n\t"
        "v_min_f32 v63, 0x3f7d70a4, v63\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[Tb], v63\n\t"
        "v_fma_f32 %[Crb], v40, v59, %[Crb]\n\t"
        "v_fma_f32 %[Cgb], v40, v60, %[Cgb]\n\t"
        "v_fma_f32 %[Cbb], v40, v61, %[Cbb]\n\t"
        "v_fma_f32 %[Tb], -%[Tb], v63, %[Tb]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 21b\n\t"
        "23:\n\t"
        "v_cmpx_le_f32 vcc, v62, v48\n\t"
        "v_min_f32 v63, 0x3f7d70a4, v63\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[Ta], v63\n\t"
        "v_fma_f32 %[Cra], v40, v49, %[Cra]\n\t"
        "v_fma_f32 %[Cga], v40, v50, %[Cga]\n\t"
        "v_fma_f32 %[Cba], v40, v51, %[Cba]\n\t"
        "v_fma_f32 %[Ta], -%[Ta], v63, %[Ta]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 24b\n\t"
        "26:\n\t"
        "v_cmpx_le_f32 vcc, v62, v48\n\t"
        "v_min_f32 v63, 0x3f7d70a4, v63\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[Tb], v63\n\t"
        "v_fma_f32 %[Crb], v40, v49, %[Crb]\n\t"
        "v_fma_f32 %[Cgb], v40, v50, %[Cgb]\n\t"
        "v_fma_f32 %[Cbb], v40, v51, %[Cbb]\n\t"
        "v_fma_f32 %[Tb], -%[Tb], v63, %[Tb]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 27b\n\t"
        "29:\n\t"
        "v_cmpx_le_f32 vcc, v62, v58\n\t"
        "v_min_f32 v63, 0x3f7d70a4, v63\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[Ta], v63\n\t"
        "v_fma_f32 %[Cra], v40, v59, %[Cra]\n\t"
        "v_fma_f32 %[Cga], v40, v60, %[Cga]\n\t"
        "v_fma_f32 %[Cba], v40, v61, %[Cba]\n\t"
        "v_fma_f32 %[Ta], -%[Ta], v63, %[Ta]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 30b\n\t"
        "32:\n\t"
        "v_cmpx_le_f32 vcc, v62, v58\n\t"
        "v_min_f32 v63, 0x3f7d70a4, v63\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v63\n\t"
        "v_mul_f32 v40, %[Tb], v63\n\t"
        "v_fma_f32 %[Crb], v40, v59, %[Crb]\n\t"
        "v_fma_f32 %[Cgb], v40, v60, %[Cgb]\n\t"
        "v_fma_f32 %[Cbb], v40, v61, %[Cbb]\n\t"
        "v_fma_f32 %[Tb], -%[Tb], v63, %[Tb]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 33b\n\t"
        "5:\n\t"
        : [Ta] "+v"(Ta), [Cra] "+v"(Cra), [Cga] "+v"(Cga), [Cba] "+v"(Cba), [Tb] "+v"(Tb), [Crb] "+v"(Crb), [Cgb] "+v"(Cgb),
          [Cbb] "+v"(Cbb), [m] "+s"(m), [ia] "=&s"(ia), [ib] "=&s"(ib)
        : [base] "v"(lds_chunk), [fpxa] "v"(fpxa), [fpxb] "v"(fpxb), [fpy] "v"(fpy), [ma] "s"(ma), [mb] "s"(mb), [fa] "s"(fa), [fb] "s"(fb),
          [p1] "i"(PLANE), [p2] "i"(2 * PLANE)
        : "vcc", "scc", "memory", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51",
          "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63");
}

// The PIPELINED one-quadrant walk, for launches that cannot fill the machine (a multi-GPU rank's shard: ~4 waves per SIMD, where a
// wave's walk runs at its own latency — tools/walk_latency.hip: 77 ns per record alone on a SIMD, ~7 cycles per instruction of one
// serial stream).  The stream is software-pipelined by one record: alpha of record k+1 (8 instructions in two independent
// sub-chains, v_exp_f32 last) is computed BEFORE record k is applied (v_cmpx + 5 masked updates), so the transcendental's latency
// and the quadratic's dependent chain sit under the previous record's update.  Three register sets of 12 (record 10, p, alpha)
// rotate: record k is applied while k+1 is evaluated and k+2's LDS reads are in flight.  The same operations on the same values in
// the same per-pixel order: bit-identical to blend_kernel (tests).  Needs 96 VGPRs (v54-v95 named here): the kernel variant that
// uses it is launched only where 5 waves per SIMD hold the whole grid.  Generated by tools/gen_blend_walk.py pipelined.
template <int PLANE>
__device__ __forceinline__ void blend_walk1p_asm(unsigned long long m, unsigned long long fa, unsigned lds_chunk, float fpx, float fpy,
                                                 float &T, float &Cr, float &Cg, float &Cb)
{
    int ia, ib, ic;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "s_ff1_i32_b64 %[ia], %[m]\n\t"
        "s_bitset0_b64 %[m], %[ia]\n\t"
        "v_lshl_add_u32 v90, %[ia], 4, %[base]\n\t"
        "ds_read_b64 v[54:55], v90\n\t"
        "ds_read_b128 v[56:59], v90 offset:%[p1]\n\t"
        "ds_read_b128 v[60:63], v90 offset:%[p2]\n\t"
        "s_cmp_eq_u64 %[m], 0\n\t"
        "s_cbranch_scc1 30f\n\t"
        "s_ff1_i32_b64 %[ib], %[m]\n\t"
        "s_bitset0_b64 %[m], %[ib]\n\t"
        "v_lshl_add_u32 v90, %[ib], 4, %[base]\n\t"
        "ds_read_b64 v[66:67], v90\n\t"
        "ds_read_b128 v[68:71], v90 offset:%[p1]\n\t"
        "ds_read_b128 v[72:75], v90 offset:%[p2]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_sub_f32 v91, v55, %[fpy]\n\t"
        "v_sub_f32 v93, v54, %[fpx]\n\t"
        "v_mul_f32 v92, v58, v91\n\t"
        "v_mul_f32 v94, v56, v93\n\t"
        "v_fma_f32 v92, v92, v91, v60\n\t"
        "v_fma_f32 v94, v57, v91, v94\n\t"
        "v_fma_f32 v64, v93, v94, v92\n\t"
        "v_exp_f32 v65, v64\n\t"
        "10:\n\t"
        "s_cmp_eq_u64 %[m], 0\n\t"
        "s_cbranch_scc1 20f\n\t"
        "s_ff1_i32_b64 %[ic], %[m]\n\t"
        "s_bitset0_b64 %[m], %[ic]\n\t"
        "v_lshl_add_u32 v90, %[ic], 4, %[base]\n\t"
        "ds_read_b64 v[78:79], v90\n\t"
        "ds_read_b128 v[80:83], v90 offset:%[p1]\n\t"
        "ds_read_b128 v[84:87], v90 offset:%[p2]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_sub_f32 v91, v67, %[fpy]\n\t"
        "v_sub_f32 v93, v66, %[fpx]\n\t"
        "v_mul_f32 v92, v70, v91\n\t"
        "v_mul_f32 v94, v68, v93\n\t"
        "v_fma_f32 v92, v92, v91, v72\n\t"
        "v_fma_f32 v94, v69, v91, v94\n\t"
        "v_fma_f32 v76, v93, v94, v92\n\t"
        "v_exp_f32 v77, v76\n\t"
        "s_bitcmp1_b64 %[fa], %[ia]\n\t"
        "s_cbranch_scc0 40f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v65\n\t"
        "v_mul_f32 v95, %[T], v65\n\t"
        "v_fma_f32 %[Cr], v95, v61, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v62, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v63, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v65, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "41:\n\t"
        "11:\n\t"
        "s_cmp_eq_u64 %[m], 0\n\t"
        "s_cbranch_scc1 21f\n\t"
        "s_ff1_i32_b64 %[ia], %[m]\n\t"
        "s_bitset0_b64 %[m], %[ia]\n\t"
        "v_lshl_add_u32 v90, %[ia], 4, %[base]\n\t"
        "ds_read_b64 v[54:55], v90\n\t"
        "ds_read_b128 v[56:59], v90 offset:%[p1]\n\t"
        "ds_read_b128 v[60:63], v90 offset:%[p2]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_sub_f32 v91, v79, %[fpy]\n\t"
        "v_sub_f32 v93, v78, %[fpx]\n\t"
        "v_mul_f32 v92, v82, v91\n\t"
        "v_mul_f32 v94, v80, v93\n\t"
        "v_fma_f32 v92, v92, v91, v84\n\t"
        "v_fma_f32 v94, v81, v91, v94\n\t"
        "v_fma_f32 v88, v93, v94, v92\n\t"
        "v_exp_f32 v89, v88\n\t"
        "s_bitcmp1_b64 %[fa], %[ib]\n\t"
        "s_cbranch_scc0 42f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v77\n\t"
        "v_mul_f32 v95, %[T], v77\n\t"
        "v_fma_f32 %[Cr], v95, v73, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v74, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v75, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v77, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "43:\n\t"
        "12:\n\t"
        "s_cmp_eq_u64 %[m], 0\n\t"
        "s_cbranch_scc1 22f\n\t"
        "s_ff1_i32_b64 %[ib], %[m]\n\t"
        "s_bitset0_b64 %[m], %[ib]\n\t"
        "v_lshl_add_u32 v90, %[ib], 4, %[base]\n\t"
        "ds_read_b64 v[66:67], v90\n\t"
        "ds_read_b128 v[68:71], v90 offset:%[p1]\n\t"
        "ds_read_b128 v[72:75], v90 offset:%[p2]\n\t"
        "s_waitcnt lgkmcnt(3)\n\t"
        "v_sub_f32 v91, v55, %[fpy]\n\t"
        "v_sub_f32 v93, v54, %[fpx]\n\t"
        "v_mul_f32 v92, v58, v91\n\t"
        "v_mul_f32 v94, v56, v93\n\t"
        "v_fma_f32 v92, v92, v91, v60\n\t"
        "v_fma_f32 v94, v57, v91, v94\n\t"
        "v_fma_f32 v64, v93, v94, v92\n\t"
        "v_exp_f32 v65, v64\n\t"
        "s_bitcmp1_b64 %[fa], %[ic]\n\t"
        "s_cbranch_scc0 44f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v89\n\t"
        "v_mul_f32 v95, %[T], v89\n\t"
        "v_fma_f32 %[Cr], v95, v85, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v86, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v87, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v89, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "45:\n\t"
        "s_branch 10b\n\t"
        "20:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_sub_f32 v91, v67, %[fpy]\n\t"
        "v_sub_f32 v93, v66, %[fpx]\n\t"
        "v_mul_f32 v92, v70, v91\n\t"
        "v_mul_f32 v94, v68, v93\n\t"
        "v_fma_f32 v92, v92, v91, v72\n\t"
        "v_fma_f32 v94, v69, v91, v94\n\t"
        "v_fma_f32 v76, v93, v94, v92\n\t"
        "v_exp_f32 v77, v76\n\t"
        "s_bitcmp1_b64 %[fa], %[ia]\n\t"
        "s_cbranch_scc0 46f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v65\n\t"
        "v_mul_f32 v95, %[T], v65\n\t"
        "v_fma_f32 %[Cr], v95, v61, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v62, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v63, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v65, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "47:\n\t"
        "s_bitcmp1_b64 %[fa], %[ib]\n\t"
        "s_cbranch_scc0 48f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v77\n\t"
        "v_mul_f32 v95, %[T], v77\n\t"
        "v_fma_f32 %[Cr], v95, v73, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v74, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v75, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v77, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "49:\n\t"
        "s_branch 39f\n\t"
        "21:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_sub_f32 v91, v79, %[fpy]\n\t"
        "v_sub_f32 v93, v78, %[fpx]\n\t"
        "v_mul_f32 v92, v82, v91\n\t"
        "v_mul_f32 v94, v80, v93\n\t"
        "v_fma_f32 v92, v92, v91, v84\n\t"
        "v_fma_f32 v94, v81, v91, v94\n\t"
        "v_fma_f32 v88, v93, v94, v92\n\t"
        "v_exp_f32 v89, v88\n\t"
        "s_bitcmp1_b64 %[fa], %[ib]\n\t"
        "s_cbranch_scc0 50f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v77\n\t"
        "v_mul_f32 v95, %[T], v77\n\t"
        "v_fma_f32 %[Cr], v95, v73, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v74, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v75, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v77, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "51:\n\t"
        "s_bitcmp1_b64 %[fa], %[ic]\n\t"
        "s_cbranch_scc0 52f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v89\n\t"
        "v_mul_f32 v95, %[T], v89\n\t"
        "v_fma_f32 %[Cr], v95, v85, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v86, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v87, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v89, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "53:\n\t"
        "s_branch 39f\n\t"
        "22:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_sub_f32 v91, v55, %[fpy]\n\t"
        "v_sub_f32 v93, v54, %[fpx]\n\t"
        "v_mul_f32 v92, v58, v91\n\t"
        "v_mul_f32 v94, v56, v93\n\t"
        "v_fma_f32 v92, v92, v91, v60\n\t"
        "v_fma_f32 v94, v57, v91, v94\n\t"
        "v_fma_f32 v64, v93, v94, v92\n\t"
        "v_exp_f32 v65, v64\n\t"
        "s_bitcmp1_b64 %[fa], %[ic]\n\t"
        "s_cbranch_scc0 54f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v89\n\t"
        "v_mul_f32 v95, %[T], v89\n\t"
        "v_fma_f32 %[Cr], v95, v85, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v86, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v87, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v89, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "55:\n\t"
        "s_bitcmp1_b64 %[fa], %[ia]\n\t"
        "s_cbranch_scc0 56f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v65\n\t"
        "v_mul_f32 v95, %[T], v65\n\t"
        "v_fma_f32 %[Cr], v95, v61, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v62, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v63, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v65, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "57:\n\t"
        "s_branch 39f\n\t"
        "30:\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_sub_f32 v91, v55, %[fpy]\n\t"
        "v_sub_f32 v93, v54, %[fpx]\n\t"
        "v_mul_f32 v92, v58, v91\n\t"
        "v_mul_f32 v94, v56, v93\n\t"
        "v_fma_f32 v92, v92, v91, v60\n\t"
        "v_fma_f32 v94, v57, v91, v94\n\t"
        "v_fma_f32 v64, v93, v94, v92\n\t"
        "v_exp_f32 v65, v64\n\t"
        "s_nop 0\n\t"
        "s_bitcmp1_b64 %[fa], %[ia]\n\t"
        "s_cbranch_scc0 58f\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v65\n\t"
        "v_mul_f32 v95, %[T], v65\n\t"
        "v_fma_f32 %[Cr], v95, v61, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v62, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v63, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v65, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "59:\n\t"
        "s_branch 39f\n\t"
        "40:\n\t"
        "v_cmpx_le_f32 vcc, v64, v60\n\t"
        "v_min_f32 v65, 0x3f7d70a4, v65\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v65\n\t"
        "v_mul_f32 v95, %[T], v65\n\t"
        "v_fma_f32 %[Cr], v95, v61, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v62, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v63, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v65, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 41b\n\t"
        "42:\n\t"
        "v_cmpx_le_f32 vcc, v76, v72\n\t"
        "v_min_f32 v77, 0x3f7d70a4, v77\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v77\n\t"
        "v_mul_f32 v95, %[T], v77\n\t"
        "v_fma_f32 %[Cr], v95, v73, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v74, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v75, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v77, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 43b\n\t"
        "44:\n\t"
        "v_cmpx_le_f32 vcc, v88, v84\n\t"
        "v_min_f32 v89, 0x3f7d70a4, v89\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v89\n\t"
        "v_mul_f32 v95, %[T], v89\n\t"
        "v_fma_f32 %[Cr], v95, v85, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v86, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v87, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v89, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 45b\n\t"
        "46:\n\t"
        "v_cmpx_le_f32 vcc, v64, v60\n\t"
        "v_min_f32 v65, 0x3f7d70a4, v65\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v65\n\t"
        "v_mul_f32 v95, %[T], v65\n\t"
        "v_fma_f32 %[Cr], v95, v61, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v62, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v63, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v65, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 47b\n\t"
        "48:\n\t"
        "v_cmpx_le_f32 vcc, v76, v72\n\t"
        "v_min_f32 v77, 0x3f7d70a4, v77\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v77\n\t"
        "v_mul_f32 v95, %[T], v77\n\t"
        "v_fma_f32 %[Cr], v95, v73, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v74, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v75, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v77, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 49b\n\t"
        "50:\n\t"
        "v_cmpx_le_f32 vcc, v76, v72\n\t"
        "v_min_f32 v77, 0x3f7d70a4, v77\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v77\n\t"
        "v_mul_f32 v95, %[T], v77\n\t"
        "v_fma_f32 %[Cr], v95, v73, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v74, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v75, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v77, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 51b\n\t"
        "52:\n\t"
        "v_cmpx_le_f32 vcc, v88, v84\n\t"
        "v_min_f32 v89, 0x3f7d70a4, v89\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v89\n\t"
        "v_mul_f32 v95, %[T], v89\n\t"
        "v_fma_f32 %[Cr], v95, v85, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v86, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v87, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v89, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 53b\n\t"
        "54:\n\t"
        "v_cmpx_le_f32 vcc, v88, v84\n\t"
        "v_min_f32 v89, 0x3f7d70a4, v89\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v89\n\t"
        "v_mul_f32 v95, %[T], v89\n\t"
        "v_fma_f32 %[Cr], v95, v85, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v86, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v87, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v89, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 55b\n\t"
        "56:\n\t"
        "v_cmpx_le_f32 vcc, v64, v60\n\t"
        "v_min_f32 v65, 0x3f7d70a4, v65\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v65\n\t"
        "v_mul_f32 v95, %[T], v65\n\t"
        "v_fma_f32 %[Cr], v95, v61, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v62, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v63, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v65, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 57b\n\t"
        "58:\n\t"
        "v_cmpx_le_f32 vcc, v64, v60\n\t"
        "v_min_f32 v65, 0x3f7d70a4, v65\n\t"
        "v_cmpx_lt_f32 vcc, 0x3b808081, v65\n\t"
        "v_mul_f32 v95, %[T], v65\n\t"
        "v_fma_f32 %[Cr], v95, v61, %[Cr]\n\t"
        "v_fma_f32 %[Cg], v95, v62, %[Cg]\n\t"
        "v_fma_f32 %[Cb], v95, v63, %[Cb]\n\t"
        "v_fma_f32 %[T], -%[T], v65, %[T]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_branch 59b\n\t"
        "39:\n\t"
        : [T] "+v"(T), [Cr] "+v"(Cr), [Cg] "+v"(Cg), [Cb] "+v"(Cb), [m] "+s"(m), [ia] "=&s"(ia), [ib] "=&s"(ib), [ic] "=&s"(ic)
        : [base] "v"(lds_chunk), [fpx] "v"(fpx), [fpy] "v"(fpy), [fa] "s"(fa), [p1] "i"(PLANE), [p2] "i"(2 * PLANE)
        : "vcc", "scc", "memory", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68",
          "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86",
          "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95");
}

// ---- staging -----------------------------------------------------------------------------------------------------------------
// A tile's depth-ordered list is either a range of per-tile entries (fine binning) or, with coarse binning, the list of its 32x32
// cell filtered by the tile's bit of the mask each entry carries in its top four bits (binning.hip): the workgroup reads the
// cell list 2 * THREADS entries at a time, keeps — in order, by ballot + wave counts — the ids with its bit in a small ring in
// LDS, and stages THREADS of them per batch.  (Round 2 expanded the cell lists into tile lists in a kernel of its own: 34 us, a
// write and a read of 61 MB per frame, 16 B of workspace per pair slot.)
constexpr uint32_t LIST_ID_MASK = (1u << 28) - 1u;

template <int THREADS>
struct TileList {
    static constexpr int WAVES = THREADS / 64, RING = 4 * THREADS;  // ring: < THREADS left over + 2 * THREADS read
    uint32_t pos, end;    // cursor into the list / its end               } workgroup-uniform
    uint32_t head, qlen;  // ring: first unread slot, entries in it       }
    int bit;              // cell lists: 28 + the tile's index in its cell; -1: plain per-tile list
};

template <int THREADS>
__device__ __forceinline__ TileList<THREADS> tile_list_of(const BlendArgs &a, int tile, int tx, int ty)
{
    TileList<THREADS> t;
    uint2 r;
    if (a.cell_lists) {
        r = a.cranges[(ty >> 1) * a.ctiles_x + (tx >> 1)];
        t.bit = 28 + (ty & 1) * 2 + (tx & 1);
    } else {
        r = a.ranges[tile];
        t.bit = -1;
    }
    t.pos = r.x; t.end = r.y; t.head = 0; t.qlen = 0;
    return t;
}

// One step of the staging loop, called by every thread after the loop's top barrier.  Returns -1: the ring was refilled, go round
// again (the top barrier publishes it); 0: the list is exhausted; nb > 0: thread tid < nb takes the batch's tid-th gaussian, *id.
template <int THREADS>
__device__ __forceinline__ int tile_list_next(const BlendArgs &a, TileList<THREADS> &t, uint32_t *s_ring, uint32_t *s_wc, uint32_t *id)
{
    constexpr int WAVES = THREADS / 64, RING = TileList<THREADS>::RING;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (t.bit < 0) {
        if (t.pos >= t.end) return 0;
        const int nb = (int)min((uint32_t)THREADS, t.end - t.pos);
        if (tid < nb) *id = a.pval[t.pos + tid];
        t.pos += nb;
        return nb;
    }
    if (t.qlen < (uint32_t)THREADS && t.pos < t.end) {  // refill: the next 2 * THREADS entries of the cell list, filtered in order
        const uint32_t i0 = t.pos + tid, i1 = i0 + THREADS;
        const uint32_t v0 = i0 < t.end ? a.pval[i0] : 0u, v1 = i1 < t.end ? a.pval[i1] : 0u;  // mask 0: nobody's
        const bool f0 = (v0 >> t.bit) & 1u, f1 = (v1 >> t.bit) & 1u;
        const unsigned long long b0 = __ballot(f0), b1 = __ballot(f1);
        if (lane == 0) { s_wc[wave] = (uint32_t)__popcll(b0); s_wc[WAVES + wave] = (uint32_t)__popcll(b1); }
        __syncthreads();
        uint32_t o0 = 0, tot0 = 0, o1 = 0, tot1 = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const uint32_t c0 = s_wc[w], c1 = s_wc[WAVES + w];
            if (w < wave) { o0 += c0; o1 += c1; }
            tot0 += c0; tot1 += c1;
        }
        const unsigned long long lt = (1ull << lane) - 1ull;
        const uint32_t tail = t.head + t.qlen;
        if (f0) s_ring[(tail + o0 + (uint32_t)__popcll(b0 & lt)) & (RING - 1)] = v0 & LIST_ID_MASK;
        if (f1) s_ring[(tail + tot0 + o1 + (uint32_t)__popcll(b1 & lt)) & (RING - 1)] = v1 & LIST_ID_MASK;
        t.qlen += tot0 + tot1;
        t.pos += 2 * THREADS;
        return -1;
    }
    if (t.qlen == 0) return 0;
    const int nb = (int)min((uint32_t)THREADS, t.qlen);
    if (tid < nb) *id = s_ring[(t.head + tid) & (RING - 1)];
    t.head += nb;
    t.qlen -= nb;
    return nb;
}

// stat[5] = the workgroup's deferred-colour evaluations: wave totals into LDS, one store.  Called once, at the end, by every thread.
__device__ __forceinline__ void stat_colour_evals(uint32_t mine, uint32_t *s_col, uint32_t *stat)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += (uint32_t)__shfl_xor((int)mine, d, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(s_col, mine);
    __syncthreads();
    if (threadIdx.x == 0) stat[5] = *s_col;
}

// Tile launch order.  Group g = tile rows g, g+8, ... of the shard (one XCD's share).  One workgroup per
// group bucket-sorts its tiles by list length, longest first (buckets = exponent + 3 mantissa bits of the
// length, i.e. within 12.5 %): order[8*j + g] = j-th tile of group g.  Slots past the end of a group hold -1.
// Which tile lands where inside a bucket is not deterministic; nothing observable depends on it.
// Also leaves the longest list length in ctrl (stats).
__global__ __launch_bounds__(256) void tile_order_kernel(const uint2 *__restrict__ ranges, FrameCtrl *ctrl, int tiles_x,
                                                         RowShard rs, int rows, int slots_per_group,
                                                         int *__restrict__ order, uint32_t stats_off, const uint2 *__restrict__ cranges,
                                                         int ctiles_x, const uint32_t *__restrict__ tile_work, size_t vstride)
{
    constexpr int NB = 256;
    ranges = view_slice(ranges, vstride); ctrl = view_slice(ctrl, vstride); order = view_slice(order, vstride);
    cranges = view_slice(cranges, vstride); tile_work = view_slice(tile_work, vstride);
    __shared__ uint32_t bucket_cnt[NB];
    __shared__ uint32_t bucket_start[NB];
    __shared__ uint32_t scratch[8];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int rows_g = g < rows ? (rows - g + 7) / 8 : 0;
    const int n = rows_g * tiles_x;
    bucket_cnt[tid] = 0;
    for (int j = n + tid; j < slots_per_group; j += 256) order[8 * j + g] = -1;
    __syncthreads();
    auto tile_of = [&](int j) { return rs.row_at(g + 8 * (j / tiles_x)) * tiles_x + (j % tiles_x); };  // the group's rows: strip rows g, g + 8, ...
    // cell lists (cranges != nullptr): the length of the tile's CELL list, an upper bound of what the tile will keep of it
    // The work of a tile is what its blend STAGES before it saturates, which its list length only bounds.  Where the last frame
    // rendered on this workspace left that count for the tile (1 + entries; a fresh workspace holds anything), it is the better
    // estimate — consecutive frames of a camera path look alike — as long as it is possible at all (<= the list): the order is a
    // schedule, every permutation renders the same frame.
    auto list_len = [&](int tile) {
        const uint2 r = cranges ? cranges[(tile / tiles_x >> 1) * ctiles_x + (tile % tiles_x >> 1)] : ranges[tile];
        return r.y - r.x;
    };
    auto len_of = [&](int tile) {
        const uint32_t len = list_len(tile), w = tile_work ? tile_work[tile] : 0u;
        return (w != 0u && w - 1u <= len) ? w - 1u : len;
    };
    // lengths < 2^24: float conversion is exact; bits >> 20 = exponent (8 bits) and 3 mantissa bits, monotone in len
    auto bucket_of = [&](uint32_t len) { return len == 0 ? (uint32_t)(NB - 1) : min((uint32_t)(NB - 2), (151u << 3) - (__float_as_uint((float)len) >> 20)); };
    uint32_t longest = 0;
    for (int j = tid; j < n; j += 256) {
        const uint32_t len = len_of(tile_of(j));
        longest = max(longest, list_len(tile_of(j)));  // GsrStats.max_list_len is the list, not the estimate
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

// The plain-C statement of the blend (GsrOptions.blend_impl = 1): one 256-thread workgroup per tile, wave = 8x8 quadrant, lane =
// pixel.  The reference for the hand-scheduled kernel below, and what round 1 shipped.
template <bool BF16ACC>  // GsrOptions.accum_dtype = 1: T and the colour sums live in bfloat16 (rounded to nearest even after every gaussian)
__global__ __launch_bounds__(256) void blend_kernel(BlendArgs args)
{
    const BlendArgs a = blend_args_of_view(args);
    auto acc_round = [](float &T, float &Cr, float &Cg, float &Cb) {
        if (BF16ACC) { T = bf16_round(T); Cr = bf16_round(Cr); Cg = bf16_round(Cg); Cb = bf16_round(Cb); }
    };
    __shared__ float4 srec[3][256];  // staged records, one plane per 16-B part: q0 at +0, q1 at +4096, q2 at +8192 bytes
    float4 *const s0 = srec[0], *const s1 = srec[1], *const s2 = srec[2];
    __shared__ int s_done;
    __shared__ uint32_t s_col;
    __shared__ uint32_t s_ring[TileList<256>::RING], s_wc[2 * TileList<256>::WAVES];

    const int tile = a.order[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *stat = a.stats + (size_t)blockIdx.x * BLEND_STAT_WORDS;
    if (tile < 0) {  // uniform: empty launch slot
        if (tid < BLEND_STAT_WORDS) stat[tid] = 0;
        return;
    }
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;

    const int qx = tx * 16 + (wave & 1) * 8, qy = ty * 16 + (wave >> 1) * 8;
    const int px = qx + (lane & 7), py = qy + (lane >> 3);
    const float fpx = (float)px, fpy = (float)py;
    const float qx0 = (float)qx, qx1 = (float)(qx + 7), qy0 = (float)qy, qy1 = (float)(qy + 7);

    TileList<256> list = tile_list_of<256>(a, tile, tx, ty);
    float T = 1.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f;
    const bool undrawn = a.sat_scale != 0.0f && !(px < a.xlim && py < a.ylim);  // never stored: finished from the start
    bool wave_done = false;
    uint32_t evaluated = 0;  // wave-uniform
    uint32_t fetched = 0;    // workgroup-uniform
    uint32_t col_evals = 0;  // per thread: deferred colours this thread evaluated while staging
    if (tid == 0) { s_done = 0; s_col = 0; }

    for (;;) {
        __syncthreads();  // previous batch fully consumed (and s_done initialised); a refilled ring published
        if (s_done == 4) break;  // uniform: every wave saturated
        uint32_t id = 0;
        const int nb = tile_list_next<256>(a, list, s_ring, s_wc, &id);
        if (nb < 0) continue;
        if (nb == 0) break;
        fetched += (uint32_t)nb;
        if (tid < nb) {
            const GaussRec *r = a.rec + id;
            s0[tid] = r->q0;
            s1[tid] = r->q1;
            s2[tid] = staged_q2(a, id, col_evals);
        }
        __syncthreads();
        if (wave_done) continue;
        for (int chunk = 0; chunk < nb; chunk += 64) {
            const int e = chunk + lane;
            const bool hit = e < nb && footprint_hits_rect(s0[e], s1[e], qx0, qx1, qy0, qy1);
            unsigned long long m = __ballot(hit);
            evaluated += (uint32_t)__popcll(m);
            // two survivors per trip so that the second one's LDS reads overlap the first one's arithmetic
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
                    acc_round(T, Cr, Cg, Cb);
                    blend_one(gb, cb, ob, fpx, fpy, T, Cr, Cg, Cb);
                    acc_round(T, Cr, Cg, Cb);
                } else {
                    blend_one(ga, ca, oa, fpx, fpy, T, Cr, Cg, Cb);
                    acc_round(T, Cr, Cg, Cb);
                }
            }
            if (__all(pixel_finished(a, T, Cr, Cg, Cb, undrawn))) {
                wave_done = true;
                if (lane == 0) atomicAdd(&s_done, 1);
                break;
            }
        }
    }

    if (lane == 0) stat[wave] = evaluated;
    if (tid == 0) { stat[4] = fetched; a.tile_work[tile] = fetched + 1u; }
    stat_colour_evals(col_evals, &s_col, stat);
    if (px < a.W && py < a.H) {
        const bool drawn = px < a.xlim && py < a.ylim;  // Q1: last column / row stay black, T stays 1
        const float r = drawn ? Cr : 0.0f, g = drawn ? Cg : 0.0f, b = drawn ? Cb : 0.0f;
        size_t o;
        if (a.layout == 0) o = ((size_t)py * a.W + px) * 3;                                           // image [H,W,3]
        else if (a.layout == 1) o = ((size_t)px * a.H + py) * 3;                                      // screen [W,H,3]
        else o = ((size_t)(a.rs.index_of(ty) * 16 + (py - ty * 16)) * a.W + px) * 3;  // strip
        store_rgb(a, o, r, g, b);
        if (a.out_T) {
            const size_t ot = a.layout == 1 ? (size_t)px * a.H + py
                            : a.layout == 0 ? (size_t)py * a.W + px
                                            : (size_t)(a.rs.index_of(ty) * 16 + (py - ty * 16)) * a.W + px;
            a.out_T[ot] = drawn ? T : 1.0f;
        }
    }
}

// The product kernel (GsrOptions.blend_impl = 0).  QPW = 2: one 128-thread workgroup per 16x16 tile, wave w = the 16x8 half
// (pixel rows 8w .. 8w+7), a lane = TWO pixels 8 columns apart, one in each 8x8 quadrant of the half; 128 entries staged per
// batch.  QPW = 1: 256 threads, wave = one quadrant, 256 entries per batch — the same walk with quadrant B switched off.
// Two quadrants per wave share LDS reads and part of the quadratic (7-8 % faster on a whole frame), but with few tiles per CU
// (a multi-GPU rank's shard) four waves per tile fill the SIMDs better and halve the per-tile critical path (G = 8 shard:
// 0.52 vs 0.65 ms), so launch_blend picks by tile count.  Same lists, same per-quadrant classification and saturation tests,
// same arithmetic as blend_kernel.
template <int QPW, bool PIPE>  // PIPE (QPW = 1 only): the pipelined walk, 96 VGPRs, for grids that 5 waves per SIMD hold
__global__ __launch_bounds__(256 / QPW, PIPE ? 5 : 8) void blend_walk_kernel(BlendArgs args)
{
    const BlendArgs a = blend_args_of_view(args);
    static_assert(!PIPE || QPW == 1, "the pipelined walk evaluates one quadrant per wave");
    constexpr int THREADS = 256 / QPW, BATCH = THREADS, WAVES = THREADS / 64;
    __shared__ float4 srec[3][BATCH];  // planes BATCH * 16 B apart
    float4 *const s0 = srec[0], *const s1 = srec[1], *const s2 = srec[2];
    const unsigned lds_rec = (unsigned)(size_t)&srec[0][0];  // LDS byte address: the low half of the flat pointer
    __shared__ int s_done;
    __shared__ uint32_t s_col;
    __shared__ uint32_t s_ring[TileList<THREADS>::RING], s_wc[2 * WAVES];

    const int tile = a.order[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *stat = a.stats + (size_t)blockIdx.x * BLEND_STAT_WORDS;
#ifdef GSR_BLEND_TIMESTAMPS  // tools/blend_wg_times.py: when does every workgroup start and end (100 MHz clock), and where
    const unsigned long long ts0 = wall_clock64();
#endif
    if (tile < 0) {  // uniform: empty launch slot
        if (tid < BLEND_STAT_WORDS) stat[tid] = 0;
        return;
    }
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int qx = tx * 16 + (QPW == 1 ? (wave & 1) * 8 : 0), qy = ty * 16 + (QPW == 1 ? wave >> 1 : wave) * 8;
    const int px = qx + (lane & 7), py = qy + (lane >> 3);  // pixel A; QPW = 2: pixel B = (px + 8, py)
    const float fpxA = (float)px, fpxB = (float)(px + 8), fpy = (float)py;
    const float xa0 = (float)qx, xa1 = (float)(qx + 7), xb0 = (float)(qx + 8), xb1 = (float)(qx + 15);
    const float qy0 = (float)qy, qy1 = (float)(qy + 7);

    TileList<THREADS> list = tile_list_of<THREADS>(a, tile, tx, ty);
    float TA = 1.0f, CrA = 0.0f, CgA = 0.0f, CbA = 0.0f;
    float TB = 1.0f, CrB = 0.0f, CgB = 0.0f, CbB = 0.0f;
    bool doneA = false, doneB = QPW == 1;  // wave-uniform: quadrant saturated (B does not exist when QPW = 1)
    // pixels whose colour is never stored (outside the frame / Q1) are finished from the start
    const bool undrawnA = a.sat_scale != 0.0f && !(px < a.xlim && py < a.ylim);
    const bool undrawnB = a.sat_scale != 0.0f && !(px + 8 < a.xlim && py < a.ylim);
    uint32_t evaluated = 0;                // wave-uniform
    uint32_t fetched = 0;                  // workgroup-uniform
    uint32_t col_evals = 0;                // per thread: deferred colours this thread evaluated while staging
    if (tid == 0) { s_done = 0; s_col = 0; }

    for (;;) {
        __syncthreads();  // previous batch fully consumed (and s_done initialised); a refilled ring published
        if (s_done == WAVES) break;  // uniform: every wave saturated
        uint32_t id = 0;
        const int nb = tile_list_next<THREADS>(a, list, s_ring, s_wc, &id);
        if (nb < 0) continue;
        if (nb == 0) break;
        fetched += (uint32_t)nb;
        if (tid < nb) {
            const GaussRec *r = a.rec + id;
            s0[tid] = r->q0;
            s1[tid] = r->q1;
            s2[tid] = staged_q2(a, id, col_evals);
        }
        __syncthreads();
        if (doneA && doneB) continue;
        for (int chunk = 0; chunk < nb; chunk += 64) {
            const int e = chunk + lane;
            FootprintClass fa = {false, false}, fb = {false, false};
            if (e < nb) {
                const float4 q0 = s0[e], q1 = s1[e];
                const float L = s2[e].x;
                fa = footprint_classify(q0, q1, L, xa0, xa1, qy0, qy1);
                if (QPW == 2) fb = footprint_classify(q0, q1, L, xb0, xb1, qy0, qy1);
            }
            const unsigned long long mA = doneA ? 0ull : __ballot(fa.hit), mB = (QPW == 1 || doneB) ? 0ull : __ballot(fb.hit);
            const unsigned long long fA = __ballot(fa.fast), fB = QPW == 1 ? 0ull : __ballot(fb.fast);
            evaluated += (uint32_t)__popcll(mA) + (uint32_t)__popcll(mB);
#ifndef GSR_BLEND_NO_WALK  // analysis build: everything but the walks (staging, culling, barriers, tail)
            if constexpr (PIPE) {
                if (mA) blend_walk1p_asm<BATCH * 16>(mA, fA, lds_rec + (unsigned)chunk * 16u, fpxA, fpy, TA, CrA, CgA, CbA);
            } else if (mA | mB)
                blend_walk2_asm<BATCH * 16>(mA | mB, mA, mB, fA, fB, lds_rec + (unsigned)chunk * 16u, fpxA, fpxB, fpy, TA, CrA, CgA, CbA, TB,
                                            CrB, CgB, CbB);
#else
            TA += __builtin_popcountll(fA) * 1e-9f; TB += __builtin_popcountll(fB) * 1e-9f;
#endif
            if (!doneA && __all(pixel_finished(a, TA, CrA, CgA, CbA, undrawnA))) doneA = true;
            if (QPW == 2 && !doneB && __all(pixel_finished(a, TB, CrB, CgB, CbB, undrawnB))) doneB = true;
            if (doneA && doneB) {
                if (lane == 0) atomicAdd(&s_done, 1);
                break;
            }
        }
    }

    if (lane == 0) {
        stat[wave] = evaluated;
        if (QPW == 2) stat[2 + wave] = 0;
    }
    if (tid == 0) { stat[4] = fetched; a.tile_work[tile] = fetched + 1u; }
    stat_colour_evals(col_evals, &s_col, stat);
#ifdef GSR_BLEND_TIMESTAMPS  // (words 6 and 7: the counters gsr_read_stats totals — 0..5 — stay what they are)
    __syncthreads();
    if (tid == 0) {
        stat[6] = (uint32_t)ts0;
        stat[7] = (uint32_t)wall_clock64();
    }
#endif
#pragma unroll
    for (int h = 0; h < QPW; ++h) {
        const int x = px + 8 * h;
        const float T = h ? TB : TA, Cr = h ? CrB : CrA, Cg = h ? CgB : CgA, Cb = h ? CbB : CbA;
        if (x < a.W && py < a.H) {
            const bool drawn = x < a.xlim && py < a.ylim;  // Q1: last column / row stay black, T stays 1
            const float r = drawn ? Cr : 0.0f, g = drawn ? Cg : 0.0f, b = drawn ? Cb : 0.0f;
            const size_t pix = a.layout == 0 ? (size_t)py * a.W + x                                           // image [H,W,3]
                             : a.layout == 1 ? (size_t)x * a.H + py                                           // screen [W,H,3]
                                             : (size_t)(a.rs.index_of(ty) * 16 + (py - ty * 16)) * a.W + x;  // strip
            store_rgb(a, pix * 3, r, g, b);
            if (a.out_T) a.out_T[pix] = drawn ? T : 1.0f;
        }
    }
}

// two quadrants per wave from this many tiles per launch on (measured: 4080 tiles better with two, 2040 with one); below, one quadrant per wave (see blend_walk_kernel)
#ifndef GSR_BLEND_HALF_MIN_TILES
#define GSR_BLEND_HALF_MIN_TILES 3000
#endif
constexpr int BLEND_HALF_MIN_TILES = GSR_BLEND_HALF_MIN_TILES;  // (the macro: tools/ A/B builds)
// the pipelined one-quadrant walk (96 VGPRs: 5 waves per SIMD = 1280 four-wave workgroups resident) up to this many tiles per launch;
// GsrOptions.blend_pipe_tiles overrides it (experiments and tests: -1 switches the variant off)
constexpr int BLEND_PIPE_MAX_TILES = 1280;

int launch_blend(const GsrCamera &cam, const GsrOptions &opts, const Workspace &ws, const uint32_t *lists, void *out_image,
                 size_t out_view_stride, float *out_T, const GsrScene *scene, hipStream_t s)
{
    if (ws.views > 1 && (out_T || scene)) { set_error("final T / an explicit scene: single views only"); return GSR_ERR_BAD_ARG; }
    BlendArgs a;
    a.col_means = scene ? scene->means : nullptr;
    a.col_sh = scene ? scene->sh : nullptr;
    a.col_cc[0] = cam.cam_center[0]; a.col_cc[1] = cam.cam_center[1]; a.col_cc[2] = cam.cam_center[2];
    a.col_degree = scene ? scene->sh_degree : 0;
    a.col_sh16 = scene ? scene->sh_dtype : 0;
    a.view_stride = ws.view_stride;
    a.out_view_stride = out_view_stride;
    const unsigned nv = (unsigned)ws.views;  // gridDim.y: one workspace slice and one frame per view
    a.ranges = ws.ranges;
    a.cranges = ws.cranges;
    a.ctiles_x = ws.ctiles_x;
    a.cell_lists = blend_reads_cell_lists(ws, opts) ? 1 : 0;
    a.pval = lists;
    a.rec = ws.rec;
    a.ctrl = ws.ctrl;
    a.out = out_image;
    a.out_T = out_T;
    a.stats = ws.blend_stats;
    a.tile_work = ws.tile_work;
    a.W = cam.width; a.H = cam.height;
    a.xlim = opts.reference_compat ? cam.width - 1 : cam.width;
    a.ylim = opts.reference_compat ? cam.height - 1 : cam.height;
    a.tiles_x = ws.tiles_x;
    a.rs = row_shard_of(opts);
    a.rows = a.rs.rows_before(ws.tiles_y);
    a.layout = opts.output_layout;
    a.out_bf16 = opts.output_dtype == 1;
    a.early_T = opts.early_out_T;
    // the colour-saturation rule (blend_args.h) leaves T unfinished: when the caller asks for the final T it is an output like the colour
    a.sat_scale = (opts.saturation_rule == 0 && out_T == nullptr) ? 0x1p-25f : 0.0f;
    const int pipe_max_tiles = opts.blend_pipe_tiles == 0 ? BLEND_PIPE_MAX_TILES : opts.blend_pipe_tiles;
    if (a.rows <= 0 || a.tiles_x <= 0) return GSR_OK;
    const int rows_per_xcd = (a.rows + 7) / 8;
    const int slots_per_group = rows_per_xcd * a.tiles_x;
    a.order = ws.tile_order;
    hipLaunchKernelGGL(tile_order_kernel, dim3(8, nv), dim3(256), 0, s, ws.ranges, ws.ctrl, a.tiles_x, a.rs, a.rows,
                       slots_per_group, ws.tile_order,
                       (uint32_t)(reinterpret_cast<const char *>(ws.blend_stats) - reinterpret_cast<const char *>(ws.ctrl)),
                       a.cell_lists ? ws.cranges : nullptr, ws.ctiles_x, opts.no_order_hint ? nullptr : ws.tile_work, ws.view_stride);
    // which walk: by the tiles of the whole launch (all views: what fills the machine)
    const int launch_tiles = a.rows * a.tiles_x * ws.views;
    const dim3 grid(8u * (unsigned)slots_per_group, nv);
    if (opts.accum_dtype == 1) hipLaunchKernelGGL(blend_kernel<true>, grid, dim3(256), 0, s, a);
    else if (opts.blend_impl == 1) hipLaunchKernelGGL(blend_kernel<false>, grid, dim3(256), 0, s, a);
    else if (launch_tiles >= BLEND_HALF_MIN_TILES) hipLaunchKernelGGL((blend_walk_kernel<2, false>), grid, dim3(128), 0, s, a);
    else if (launch_tiles <= pipe_max_tiles) hipLaunchKernelGGL((blend_walk_kernel<1, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((blend_walk_kernel<1, false>), grid, dim3(256), 0, s, a);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

// gsr_read_stats: total the per-workgroup counters of the last blend into FrameCtrl.  One workgroup, no atomics;
// the layout comes from FrameCtrl itself because gsr_read_stats is handed nothing but the workspace.
__global__ __launch_bounds__(1024) void blend_stats_kernel(FrameCtrl *ctrl, size_t workspace_bytes)
{
    __shared__ unsigned long long part[3][16];
    const uint32_t off = ctrl->stats_off, slots = ctrl->stats_slots;
    unsigned long long ev = 0, fe = 0, ce = 0;
    if (off >= sizeof(FrameCtrl) && (size_t)off + (size_t)slots * BLEND_STAT_WORDS * 4 <= workspace_bytes) {
        const uint32_t *st = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(ctrl) + off);
        for (uint32_t i = threadIdx.x; i < slots; i += 1024) {
            const uint4 v = *reinterpret_cast<const uint4 *>(st + (size_t)i * BLEND_STAT_WORDS);
            ev += (unsigned long long)v.x + v.y + v.z + v.w;
            fe += st[(size_t)i * BLEND_STAT_WORDS + 4];
            ce += st[(size_t)i * BLEND_STAT_WORDS + 5];
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        ev += __shfl_xor(ev, d, 64);
        fe += __shfl_xor(fe, d, 64);
        ce += __shfl_xor(ce, d, 64);
    }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = ev; part[1][threadIdx.x >> 6] = fe; part[2][threadIdx.x >> 6] = ce; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ev = fe = ce = 0;
        for (int w = 0; w < 16; ++w) { ev += part[0][w]; fe += part[1][w]; ce += part[2][w]; }
        ctrl->wave_entries = ev;
        ctrl->fetched_entries = fe;
        ctrl->colour_evals = ce;
    }
    // E, when the blend reads the cell lists directly (binning.hip): the emit workgroups' partial counts, one per 256 depth-sorted gaussians
    const uint32_t eoff = ctrl->ent_off;
    const size_t nblk = ((size_t)ctrl->n_visible + EMIT_THREADS - 1) / EMIT_THREADS;
    if (eoff >= sizeof(FrameCtrl) && (size_t)eoff + 4 * nblk <= workspace_bytes) {
        __syncthreads();
        const uint32_t *pe = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(ctrl) + eoff);
        unsigned long long e = 0;
        for (size_t i = threadIdx.x; i < nblk; i += 1024) e += pe[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) e += __shfl_xor(e, d, 64);
        if ((threadIdx.x & 63) == 0) part[0][threadIdx.x >> 6] = e;
        __syncthreads();
        if (threadIdx.x == 0) {
            e = 0;
            for (int w = 0; w < 16; ++w) e += part[0][w];
            ctrl->n_pairs = e > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)e;
        }
    }
}

int launch_blend_stats(FrameCtrl *ctrl, size_t workspace_bytes, hipStream_t s)
{
    hipLaunchKernelGGL(blend_stats_kernel, dim3(1), dim3(1024), 0, s, ctrl, workspace_bytes);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

}  // namespace gsr
