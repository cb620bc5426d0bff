// blend_mfma.hip — stage 3, matrix-pipe variant (GsrOptions.blend_impl = 2).
//
// Same algorithm, same lists, same culling and the same per-pixel blend as blend.hip (rasterize.py:255-305,
// :436-446); what changes is WHERE the quadratic form is evaluated.  blend.hip is bound by vector-ALU issue:
// 7 of its 17 VALU instructions per (pixel, entry) evaluation compute
//     p = A dx^2 + B dx dy + C dy^2 + log2(opacity),   (dx, dy) = mean - pixel,
// while the matrix pipe idles.  Expanded around the centre c of the wave's 8x8 quadrant (X, Y = pixel - c in
// {-3.5 .. 3.5}, m' = mean - c) p is bilinear in a per-gaussian and a per-pixel 6-vector,
//     p = theta . phi,   theta = (K0, K1, K2, A, B, C),   phi = (1, X, Y, X^2, XY, Y^2)
//     K0 = A m'x^2 + B m'x m'y + C m'y^2 + L,  K1 = -(2 A m'x + B m'y),  K2 = -(B m'x + 2 C m'y),
// so 32 gaussians x 32 pixels are one [32 x 6] . [6 x 32] product: three v_mfma_f32_32x32x2_f32 (exact fp32 FMA
// chains, no reduced precision).  The price is cancellation between K0 and the linear terms: ~1e-5 absolute
// error in p in the worst case (a 0.55-px gaussian at the far corner of the quadrant), 1e-6 typically, against
// 2e-7 for the direct form — which is why blend.hip stays the default and the reference-grade path.
//
// Accumulator layout of 32x32x2 (C/D: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)): a lane holds
// ONE pixel (column) and 16 of the 32 gaussians, in runs of 4 consecutive rows that alternate between the two
// lane halves.  Front-to-back compositing is associative: a run of entries composes to (tau, kappa) =
// (product of (1 - alpha), colour accumulated from T = 1), and runs compose as C += T kappa, T *= tau.  Each
// lane blends its 4-row runs locally, v_permlane32_swap hands both halves the two run transmittances and both
// advance the same running T; the two halves' colour sums are added once, at the end of the tile.
// A wave covers its 64 pixels as two 32-pixel sets (quadrant rows 0-3 and 4-7) that are blended jointly: they see
// the same gaussians, so one colour read serves both and the wave carries two independent dependency chains;
// 5 MFMAs per 32 survivors (the k = 0,1 step is common to both sets).
//
// Survivors of the per-quadrant footprint test are appended to a per-wave ring in LDS (theta as 6 planes for
// conflict-free operand loads, plus {r, g, b, L}); whenever 32 are queued they are consumed as one group; the tail
// group is padded with K0 = -inf entries (2^-inf = 0: no-op).
//
// Status (MI355X, bench frame): bit-for-bit the same lists and counters as blend.hip, 125.1 dB vs the oracle
// (blend.hip: 125.5 dB), 33 % fewer VALU instructions (369 M vs 549 M per launch) and 4x fewer SALU/LDS
// instructions — but 0.77 ms against 0.73 ms: at 114 VGPRs + 33 KB LDS only 3.5 waves/SIMD are resident and the
// long dependent chains leave the SIMD issuing one VALU per 5 cycles (PMC).  An 8-wave one-set-per-wave variant
// (finer 8x4 culling, 6 waves/SIMD) doubled the culling/append overhead and ran at 1.05 ms.  Kept as an option and
// as the starting point for a version that software-pipelines MFMA and blending across groups.
#include "gsr_internal.h"
#include "blend_args.h"
#include "footprint.h"

namespace gsr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int RING = 128;  // per-wave survivor ring: at most 31 left over + 64 appended

struct SetState {
    float T, Cr, Cg, Cb;  // running transmittance of this lane's pixel (same value in both halves) and this half's colour sum
};

__device__ __forceinline__ void eval_row(float p, const float4 c, float &T, float &r, float &g, float &b)
{
    float alpha = fminf(__builtin_amdgcn_exp2f(p), GSR_MAX_ALPHA);
    const bool valid = (alpha > GSR_MIN_ALPHA) & (p <= c.w);  // rasterize.py:291 (p <= L <=> power <= 0)
    alpha = valid ? alpha : 0.0f;
    const float w = alpha * T;
    r = fmaf(w, c.x, r);
    g = fmaf(w, c.y, g);
    b = fmaf(w, c.z, b);
    T = T - w;
}

__device__ __forceinline__ void compose_run(float Tl, float kr, float kg, float kb, int hsel, SetState &st)
{
    // v_permlane32_swap(a = Tl, b = Tl) leaves a = half 0's value and b = half 1's value in BOTH halves, so the two
    // lanes of a pixel multiply the same numbers in the same order and st.T stays identical in them.
    const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(Tl), __float_as_uint(Tl), false, false);
    const float tau0 = __uint_as_float(sw.x), tau1 = __uint_as_float(sw.y);
    const float T1 = st.T * tau0;
    const float W = hsel ? T1 : st.T;  // transmittance in front of MY run (half 0's run comes first)
    st.Cr = fmaf(W, kr, st.Cr);
    st.Cg = fmaf(W, kg, st.Cg);
    st.Cb = fmaf(W, kb, st.Cb);
    st.T = T1 * tau1;
}

// Blend the 16 rows this lane holds of one 32-gaussian group into its two pixels (one per 32-pixel set).
// `grp` points at this half's first colour slot of the group (ring slot of row 4*hsel): a group never wraps in the
// ring (its start is a multiple of 32), so every read is base + immediate offset.
__device__ __forceinline__ void consume_rows(const f32x16 d0, const f32x16 d1, const float4 *__restrict__ grp, int hsel,
                                             SetState &st0, SetState &st1)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // run q: rows 8q + 4 hsel + {0,1,2,3}, blended from T = 1
        float T0 = 1.0f, r0 = 0.0f, g0 = 0.0f, b0 = 0.0f, T1 = 1.0f, r1 = 0.0f, g1 = 0.0f, b1 = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 c = grp[8 * q + t];  // {r, g, b, L}: two addresses per wave (one per half)
            eval_row(d0[4 * q + t], c, T0, r0, g0, b0);
            eval_row(d1[4 * q + t], c, T1, r1, g1, b1);
        }
        compose_run(T0, r0, g0, b0, hsel, st0);
        compose_run(T1, r1, g1, b1, hsel, st1);
        // keep the next run's colour reads from being hoisted above this one: unfenced the scheduler keeps every
        // float4 read of a group live at once (170 VGPRs, 2 waves/SIMD)
        __builtin_amdgcn_sched_barrier(0);
    }
}

__global__ __launch_bounds__(256, 4) void blend_mfma_kernel(BlendArgs a)
{
    __shared__ float4 s0[256];
    __shared__ float4 s1[256];
    __shared__ float4 s2[256];
    __shared__ float ring_theta[4][6][RING];  // per wave: theta planes
    __shared__ float4 ring_col[4][RING];      // per wave: {r, g, b, L + rounding bound of the expansion}
    __shared__ int s_done;

    const int tile = a.order[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tile < 0) {  // uniform: empty launch slot
        if (tid < 5) a.stats[(size_t)blockIdx.x * BLEND_STAT_WORDS + tid] = 0;
        return;
    }
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;

    const int qx = tx * 16 + (wave & 1) * 8, qy = ty * 16 + (wave >> 1) * 8;
    const int px = qx + (lane & 7), py = qy + (lane >> 3);  // the pixel this lane finally stores
    const float qx0 = (float)qx, qx1 = (float)(qx + 7), qy0 = (float)qy, qy1 = (float)(qy + 7);
    const float cx = (float)qx + 3.5f, cy = (float)qy + 3.5f;
    const int hsel = lane >> 5, j = lane & 31;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;

    // pixel basis of the two 32-pixel sets as MFMA B operands: B[k = 2s + hsel][column j], phi = (1, X, Y, X^2, XY, Y^2)
    const float X = (float)(j & 7) - 3.5f, Y0 = (float)(j >> 3) - 3.5f, Y1 = Y0 + 4.0f;
    const float b0[3] = {hsel ? X : 1.0f, hsel ? X * X : Y0, hsel ? Y0 * Y0 : X * Y0};
    const float b1[3] = {hsel ? X : 1.0f, hsel ? X * X : Y1, hsel ? Y1 * Y1 : X * Y1};

    float(*th)[RING] = ring_theta[wave];
    float4 *col = ring_col[wave];

    const uint2 range = a.ranges[tile];
    SetState st0 = {1.0f, 0.0f, 0.0f, 0.0f}, st1 = {1.0f, 0.0f, 0.0f, 0.0f};
    bool wave_done = false;
    uint32_t evaluated = 0, fetched = 0;
    int head = 0, tail = 0;  // wave-uniform ring cursors (monotone; slots are taken mod RING)
    if (tid == 0) s_done = 0;

    auto consume_group = [&]() {
        const int base = head & (RING - 1);  // multiple of 32: the group occupies [base, base + 32) without wrapping
        float av[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) av[s] = th[2 * s + hsel][base + j];  // A[row j][k = 2s + hsel]
        f32x16 d0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b0[s], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b1[s], d1, 0, 0, 0);
        }
        consume_rows(d0, d1, col + base + 4 * hsel, hsel, st0, st1);
        head += 32;
    };

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
            float4 c0, c1;
            bool hit = false;
            if (e < nb) {
                c0 = s0[e];
                c1 = s1[e];
                hit = footprint_hits_rect(c0, c1, qx0, qx1, qy0, qy1);
            }
            const unsigned long long m = __ballot(hit);
            const int cnt = __popcll(m);
            const bool last = (batch + 256 >= range.y) && (chunk + 64 >= nb);  // uniform: final chunk of the list
            if (cnt == 0 && !last) continue;
            evaluated += (uint32_t)cnt;
            if (hit) {  // append to the ring; DS operations of one wave execute in order, no barrier needed
                const int slot = (tail + __popcll(m & lt_mask)) & (RING - 1);
                const float4 c2 = s2[e];
                const float mx = c0.x - cx, my = c0.y - cy;
                const float u = c1.x * mx, v = c1.z * my;                      // A m'x, C m'y
                const float K0 = fmaf(fmaf(c1.y, my, u), mx, fmaf(v, my, c2.x));
                const float K1 = -fmaf(c1.y, my, 2.0f * u), K2 = -fmaf(c1.y, mx, 2.0f * v);
                th[0][slot] = K0;
                th[1][slot] = K1;
                th[2][slot] = K2;
                th[3][slot] = c1.x;
                th[4][slot] = c1.y;
                th[5][slot] = c1.z;
                // `power <= 0` (rasterize.py:291) is tested as p <= L.  The direct form (blend.hip) forms the offsets first, so
                // its power is sign-exact next to the mean; the expansion about the quadrant centre cancels terms of size
                // |K0| + 3.5 (|K1| + |K2|) + 12.25 (|A| + |B| + |C|), and a pixel sitting within ~0.005 px of a sharp gaussian's
                // mean could come out at power = +1e-5 and lose the gaussian at its very peak (found by tools/fuzz_parity.py:
                // two pixels of a 1.4 M-gaussian frame off by 0.04).  The test therefore allows the expansion's rounding bound.
                const float tol = 4.0e-7f * (fabsf(K0) + fabsf(c2.x) + 3.5f * (fabsf(K1) + fabsf(K2)) +
                                             12.25f * (fabsf(c1.x) + fabsf(c1.y) + fabsf(c1.z)));
#ifdef GSR_MFMA_NO_TOL  // tests only: shows that test_matrix_pipe_keeps_gaussians_at_their_peak bites
                col[slot] = make_float4(c2.y, c2.z, c2.w, c2.x);
#else
                col[slot] = make_float4(c2.y, c2.z, c2.w, c2.x + tol);
#endif
            }
            tail += cnt;
            if (last) {  // pad the tail group with no-op entries (K0 = -inf: 2^-inf = 0)
                const int pad = (32 - ((tail - head) & 31)) & 31;
                if (lane < pad) {
                    const int slot = (tail + lane) & (RING - 1);
                    th[0][slot] = -__builtin_inff();
#pragma unroll
                    for (int k = 1; k < 6; ++k) th[k][slot] = 0.0f;
                    col[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                tail += pad;
            }
            while (tail - head >= 32) consume_group();
            if (__all((st0.T <= a.early_T) & (st1.T <= a.early_T))) {
                wave_done = true;
                if (lane == 0) atomicAdd(&s_done, 1);
                break;
            }
        }
    }

    // the two halves hold partial colour sums of the same pixels: add them, then half 0 keeps set 0, half 1 set 1
    const float r0 = st0.Cr + __shfl_xor(st0.Cr, 32, 64), g0 = st0.Cg + __shfl_xor(st0.Cg, 32, 64), bl0 = st0.Cb + __shfl_xor(st0.Cb, 32, 64);
    const float r1 = st1.Cr + __shfl_xor(st1.Cr, 32, 64), g1 = st1.Cg + __shfl_xor(st1.Cg, 32, 64), bl1 = st1.Cb + __shfl_xor(st1.Cb, 32, 64);
    const float Cr = hsel ? r1 : r0, Cg = hsel ? g1 : g0, Cb = hsel ? bl1 : bl0, T = hsel ? st1.T : st0.T;

    if (lane == 0) a.stats[(size_t)blockIdx.x * BLEND_STAT_WORDS + wave] = evaluated;
    if (tid == 0) a.stats[(size_t)blockIdx.x * BLEND_STAT_WORDS + 4] = fetched;
    if (px < a.W && py < a.H) {
        const bool drawn = px < a.xlim && py < a.ylim;  // Q1: last column / row stay black, T stays 1
        const float r = drawn ? Cr : 0.0f, g = drawn ? Cg : 0.0f, b = drawn ? Cb : 0.0f;
        size_t o;
        if (a.layout == 0) o = ((size_t)py * a.W + px) * 3;
        else if (a.layout == 1) o = ((size_t)px * a.H + py) * 3;
        else o = ((size_t)(((ty - a.row_begin) / a.row_step) * 16 + (py - ty * 16)) * a.W + px) * 3;
        store_rgb(a, o, r, g, b);
        if (a.out_T) {
            const size_t ot = a.layout == 1 ? (size_t)px * a.H + py
                            : a.layout == 0 ? (size_t)py * a.W + px
                                            : (size_t)(((ty - a.row_begin) / a.row_step) * 16 + (py - ty * 16)) * a.W + px;
            a.out_T[ot] = drawn ? T : 1.0f;
        }
    }
}

int launch_blend_mfma(const BlendArgs &a, unsigned grid, hipStream_t s)
{
    hipLaunchKernelGGL(blend_mfma_kernel, dim3(grid), dim3(256), 0, s, a);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

}  // namespace gsr
