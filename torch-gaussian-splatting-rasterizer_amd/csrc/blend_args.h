// blend_args.h — kernel arguments of the blend kernels.
#pragma once
#include "gsr_internal.h"

namespace gsr {

struct BlendArgs {
    const uint2 *ranges;  // [tiles] per-tile lists (fine binning)
    const uint2 *cranges; // [cells] cell_lists: the 32x32-cell lists; the tile's entries are those with its bit in the top four
    int ctiles_x;
    int cell_lists;
    const uint32_t *pval; // the lists: gaussian ids (cell_lists: | tile mask << 28)
    GaussRec *rec;        // read; a deferred colour is written back into its record (staged_q2)
    const FrameCtrl *ctrl; // col_*: what a deferred colour is evaluated from
    void *out;            // float32, or bfloat16 when out_bf16
    float *out_T;
    uint32_t *stats;      // [launch slots][BLEND_STAT_WORDS]
    uint32_t *tile_work;  // [tiles] 1 + entries staged: next frame's launch-order hint
    const int *order;     // tile launch order (tile_order_kernel), -1 = empty slot
    int W, H;
    int xlim, ylim;       // pixels x < xlim, y < ylim are drawn (W-1/H-1 in reference_compat: Q1)
    int tiles_x;
    RowShard rs;          // tile rows of this shard (gsr_internal.h)
    int rows;             // how many: strip rows k in [0, rows) <-> tile rows rs.row_at(k)
    int layout;
    int out_bf16;
    float early_T;
    float sat_scale;      // 2^-25: a pixel is finished once T <= sat_scale * min(Cr, Cg, Cb) (GsrOptions.saturation_rule = 0); 0: only T <= early_T
    // deferred colours (staged_q2): the scene and camera centre handed to gsr_blend; col_means == nullptr: the ones gsr_preprocess left
    // in the control block (the library's own stage sequence, and callers of gsr_blend who pass no scene)
    const float *col_means;
    const void *col_sh;
    float col_cc[3];
    int col_degree, col_sh16;
    size_t view_stride;     // several views per launch (gridDim.y): bytes between the views' workspace slices ...
    size_t out_view_stride; // ... and between their frames
};

// The arguments of view blockIdx.y (the host fills in view 0's): a new value, built field by field — mutating the kernel's argument
// block in place makes the compiler keep a private copy of it in scratch.
__device__ __forceinline__ BlendArgs blend_args_of_view(const BlendArgs &a)
{
    const size_t o = (size_t)blockIdx.y * a.view_stride;
    auto at = [](auto *p, size_t off) { return view_at(p, off); };
    BlendArgs b;
    b.ranges = at(a.ranges, o);
    b.cranges = at(a.cranges, o);
    b.ctiles_x = a.ctiles_x;
    b.cell_lists = a.cell_lists;
    b.pval = at(a.pval, o);
    b.rec = at(a.rec, o);
    b.ctrl = at(a.ctrl, o);
    b.out = at(a.out, (size_t)blockIdx.y * a.out_view_stride);
    b.out_T = a.out_T;
    b.stats = at(a.stats, o);
    b.tile_work = at(a.tile_work, o);
    b.order = at(a.order, o);
    b.W = a.W; b.H = a.H; b.xlim = a.xlim; b.ylim = a.ylim; b.tiles_x = a.tiles_x;
    b.rs = a.rs; b.rows = a.rows;
    b.layout = a.layout; b.out_bf16 = a.out_bf16; b.early_T = a.early_T; b.sat_scale = a.sat_scale;
    b.col_means = a.col_means; b.col_sh = a.col_sh;
    b.col_cc[0] = a.col_cc[0]; b.col_cc[1] = a.col_cc[1]; b.col_cc[2] = a.col_cc[2];
    b.col_degree = a.col_degree; b.col_sh16 = a.col_sh16;
    b.view_stride = a.view_stride; b.out_view_stride = a.out_view_stride;
    return b;
}

// Has this pixel stopped changing?  (wave-uniformly combined with __all by the kernels.)
//   - T <= early_T: the caller's threshold; with 0 it fires once T has underflowed to 0.0f, after which alpha*T*rgb = 0 exactly;
//   - T <= 2^-25 min(Cr, Cg, Cb): the colour update is one rounding, C = fma(w, c, C), with w = fl(alpha T) <= T (alpha < 1) and
//     0 <= c <= 1 (Q7), and C in [2^e, 2^(e+1)) has ulp(C) / 2 = 2^(e-24) > 2^-25 C: so w c <= T < ulp(C) / 2 and the fma returns C
//     bit for bit — for this entry and, T never growing and C never changing, for every later one.  The threshold itself must be
//     exact for that: 2^-25 C is (a power of two times a normal number, normal again) whenever C >= 2^-100; below that — a channel
//     that is exactly 0 so far, or one that only began to collect colour after T had decayed to ~1e-30 — the product would be rounded
//     into the denormals, possibly UP (2^-25 * 1.5 * 2^-125 -> 2^-149, where w c can be exactly half an ulp of an odd C and the fma
//     rounds to even), so there the rule falls back to T == 0 (tests: the "late red" case of test_colour_saturation_rule_is_exact).
//   - `undrawn`: a pixel whose colour is never stored (outside the frame, or Q1's last column / row) has nothing left to change.
__device__ __forceinline__ bool pixel_finished(const BlendArgs &a, float T, float Cr, float Cg, float Cb, bool undrawn)
{
    // sat_scale == 0: the caller's threshold alone (a negative one never fires: "blend every entry", the tests' ground truth)
    const float cmin = fminf(fminf(Cr, Cg), Cb);
    const float thr = a.sat_scale != 0.0f ? fmaxf(a.early_T, cmin >= 0x1p-100f ? a.sat_scale * cmin : 0.0f) : a.early_T;
    return undrawn | (T <= thr);
}

// float -> the nearest bfloat16 (ties to even), as a float.  Values are finite.
__device__ __forceinline__ float bf16_round(float x)
{
    const uint32_t u = __float_as_uint(x);
    return __uint_as_float((u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u);
}

// One pixel's colour into the frame.  bfloat16 = the upper half of the float, rounded to nearest even (values are finite).
__device__ __forceinline__ void store_rgb(const BlendArgs &a, size_t o, float r, float g, float b)
{
    if (a.out_bf16) {
        auto bf = [](float x) {
            const uint32_t u = __float_as_uint(x);
            return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
        };
        unsigned short *p = static_cast<unsigned short *>(a.out) + o;
        p[0] = bf(r); p[1] = bf(g); p[2] = bf(b);
    } else {
        float *p = static_cast<float *>(a.out) + o;
        p[0] = r; p[1] = g; p[2] = b;
    }
}

}  // namespace gsr
