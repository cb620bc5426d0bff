// preprocess.hip — stage 1: everything rasterize.py:354-420 computes per gaussian.  preprocess_kernel: one thread per
// gaussian (whole frames, and ranks of up to 4 shards); shard_preprocess_kernel: the same arithmetic in three dense phases for
// ranks of 5+ shards (below).
//
// Built with -ffp-contract=off: every expression below keeps the reference's fp32 operation order
// (file:line cited per block) so the step functions downstream (ceil of the radius, floor of the
// tile rect, det == 0) see the same values the reference sees, up to libm-vs-ocml differences in
// expf (<= 1 ulp).
//
// Roofline: HBM.  Algorithmic bytes per gaussian: 44 read (xyz 12, scale 12, rot 16, opacity 4) + 52..64 written per visible
// gaussian; the 192-B SH row only with GsrOptions.colour_stage = 1 — by default the blend evaluates a
// gaussian's colour when a tile first stages it (blend.hip), and most gaussians of a dense scene are never staged.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include "gsr_internal.h"
#include "gauss_math.h"

namespace gsr {

struct Cam {
    float V[16];
    float F[16];
    float cc[3];
    float fx, fy, limx, limy;
    float w_sigma2;  // largest eigenvalue of A^T A, A = w2c[:3,:3] (1 for a unit qvec), padded: bounds |J A| in the shard kernel's phase 1
    int W, H;
};

// What the geometry half of the per-gaussian work hands on (to the colour half and to whoever writes the outputs).
struct GeoOut {
    bool visible;     // enters this rank's depth sort with a non-empty tile rect
    bool keep_empty;  // progressive render only: enters the sort with an empty rect (counts in the reference's draw order)
    uint32_t key;     // IEEE bits of z_cam
    int tx0, ty0, tx1, ty1;
    float mx, my, sx, sy, sxy, pthr, op;  // screen mean, conic, culling threshold, opacity: what record_of() needs
    float p[3];       // the mean (world), for the colour half
};

// GaussRec.q0, q1 and q2.x of a visible gaussian (q2.yzw is its colour).
__device__ __forceinline__ void record_of(const GeoOut &g, float4 *q0, float4 *q1, float *log2_op)
{
    const float LOG2E = 1.4426950408889634f;
    const float A = (-0.5f * g.sx) * LOG2E, B = (-g.sxy) * LOG2E, C = (-0.5f * g.sy) * LOG2E;
    // -B/(2C), -B/(2A): only used by the culling test, and only when the conic is positive definite (A, C < 0)
    const float rc = g.pthr < -1e37f ? 0.0f : -0.5f * B / C, ra = g.pthr < -1e37f ? 0.0f : -0.5f * B / A;
    *q0 = make_float4(g.mx, g.my, rc, ra);
    *q1 = make_float4(A, B, C, g.pthr);
    *log2_op = log2f(g.op);
}

// The camera-independent half of a gaussian's geometry: what rasterize.py:354-358 reads and derives before any camera is involved.
struct GeoIn {
    float C3[3][3];  // cov3D, :357
    float op;        // sigmoid(opacity_logit), :358
};

__device__ __forceinline__ GeoIn geometry_load(const GsrScene &sc, int64_t i)
{
    GeoIn in;
    const float ls[3] = {sc.log_scales[3 * i], sc.log_scales[3 * i + 1], sc.log_scales[3 * i + 2]};
    const float4 q = reinterpret_cast<const float4 *>(sc.quats)[i];
    const float op_logit = sc.opacity_logit[i];
    cov3d_of(ls, q, in.C3);                      // :357
    in.op = 1.0f / (1.0f + expf(-op_logit));    // sigmoid, :358
    return in;
}

// Is the gaussian at `p` behind the cull plane of this camera (rasterize.py:377)?  The same expression geometry_view evaluates.
__device__ __forceinline__ bool culled_by(const Cam &cam, const float p[3])
{
    const float *V = cam.V;
    return ((p[0] * V[2] + p[1] * V[6]) + p[2] * V[10]) + V[14] < GSR_CULL_Z;
}

// Can NO gaussian of a block be drawn in this view?  `bb` = the block's entry of GsrScene.block_bounds: {min xyz, max xyz, largest
// log-scale, -}.  Evaluated once per block and view by block_flags_kernel (the frame's first kernel when the scene has bounds); the
// preprocess kernels read the flag.  Conservative by construction, so skipping the block changes no bit of the frame:
//   - z_cam (rasterize.py:84) is linear in the mean: over the box it stays within zc +- ze.  If even the largest stays below the
//     cull plane (rasterize.py:377) — with a margin for the different roundings of this bound and of geometry_view's own sum —
//     every gaussian of the block is culled.  A box that straddles the plane is kept.
//   - in front of the plane (w = z_cam > 0 on the whole box) the map mean -> (x_clip / w, y_clip / w) is projective, so the extremes of
//     either coordinate over the box are taken at its corners: eight corners give the range of the pixel means (rasterize.py:391); the
//     radius of every gaussian is bounded by shard_preprocess_kernel's phase-1 bound (which see) evaluated at the block's largest scale
//     and nearest depth with the ray clamp at its limit: Rb.  As there, a gaussian whose columns [mx - Rb, mx + Rb + 15] or tile rows
//     [floor((my - Rb) / 16), floor((my + Rb + 15) / 16)) miss the frame — or miss this rank's tile rows — cannot be drawn.  Rb carries
//     0.1 % + 1 px + 1e-4 of the coordinates' magnitude more here for this function's own roundings.
// Anything non-finite compares false and keeps the block.
__device__ __forceinline__ bool block_dead(const float *__restrict__ bb, const Cam &cam, RowShard rs)
{
    const float4 b0 = *reinterpret_cast<const float4 *>(bb), b1 = *reinterpret_cast<const float4 *>(bb + 4);
    const float lo[3] = {b0.x, b0.y, b0.z}, hi[3] = {b0.w, b1.x, b1.y};
    const float ls_max = b1.z;
    const float *V = cam.V, *F = cam.F;
    // z_cam over the box: centre +- half range, and the magnitude its fp32 evaluation errors scale with
    float zc = V[14], ze = 0.0f, zm = fabsf(V[14]);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float c = 0.5f * (lo[j] + hi[j]), e = 0.5f * (hi[j] - lo[j]) * 1.00001f + 1.0e-6f * (fabsf(lo[j]) + fabsf(hi[j]));
        zc += c * V[4 * j + 2];
        ze += e * fabsf(V[4 * j + 2]);
        zm += (fabsf(c) + e) * fabsf(V[4 * j + 2]);
    }
    const float zpad = 1.0e-5f * zm + 1.0e-6f;
    if (zc + ze + zpad < GSR_CULL_Z) return true;        // all behind the cull plane
    const float zmin = zc - ze - zpad;
    if (!(zmin >= GSR_CULL_Z)) return false;             // straddles it (or NaN): no screen bound
    float nx_lo = 3.0e38f, nx_hi = -3.0e38f, ny_lo = 3.0e38f, ny_hi = -3.0e38f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float p0 = (k & 1) ? hi[0] : lo[0], p1 = (k & 2) ? hi[1] : lo[1], p2 = (k & 4) ? hi[2] : lo[2];
        const float x = ((p0 * F[0] + p1 * F[4]) + p2 * F[8]) + F[12];
        const float y = ((p0 * F[1] + p1 * F[5]) + p2 * F[9]) + F[13];
        const float w = ((p0 * F[3] + p1 * F[7]) + p2 * F[11]) + F[15];
        if (!(w > 0.05f)) return false;                  // w = z_cam for the reference's projection (rasterize.py:123-151); stay away from 0
        const float iw = 1.0f / w, nx = x * iw, ny = y * iw;
        // the corner's own rounding: each of x, y, w is good to ~1e-6 of its terms' magnitude
        const float ex = 1.0e-5f * (fabsf(p0 * F[0]) + fabsf(p1 * F[4]) + fabsf(p2 * F[8]) + fabsf(F[12])) * iw + 1.0e-5f * fabsf(nx);
        const float ey = 1.0e-5f * (fabsf(p0 * F[1]) + fabsf(p1 * F[5]) + fabsf(p2 * F[9]) + fabsf(F[13])) * iw + 1.0e-5f * fabsf(ny);
        nx_lo = fminf(nx_lo, nx - ex); nx_hi = fmaxf(nx_hi, nx + ex);
        ny_lo = fminf(ny_lo, ny - ey); ny_hi = fmaxf(ny_hi, ny + ey);
    }
    const float Wf = (float)cam.W, Hf = (float)cam.H;
    const float mx_lo = ((nx_lo + 1.0f) * Wf - 1.0f) * 0.5f, mx_hi = ((nx_hi + 1.0f) * Wf - 1.0f) * 0.5f;
    const float my_lo = ((ny_lo + 1.0f) * Hf - 1.0f) * 0.5f, my_hi = ((ny_hi + 1.0f) * Hf - 1.0f) * 0.5f;
    const float iz = 1.0f / zmin, jx = cam.fx * iz, jy = cam.fy * iz;
    const float smax = expf(2.0f * ls_max);
    const float trb = cam.w_sigma2 * smax * (jx * jx * (1.0f + cam.limx * cam.limx) + jy * jy * (1.0f + cam.limy * cam.limy)) + 0.6f;
    const float Rb = (3.0f * sqrtf(1.02f * trb + 0.4f) + 1.5f) * 1.001f + 1.0f + 1.0e-4f * (fabsf(mx_lo) + fabsf(mx_hi) + fabsf(my_lo) + fabsf(my_hi));
    if (!(Rb < 1.0e8f) || !(fabsf(my_lo) < 1.0e8f) || !(fabsf(my_hi) < 1.0e8f) || !(fabsf(mx_lo) < 1.0e8f) || !(fabsf(mx_hi) < 1.0e8f)) return false;
    if (mx_hi + Rb < 0.0f || mx_lo - Rb - 16.0f > Wf) return true;   // left or right of the frame
    const int tiles_y = (cam.H + GSR_TILE - 1) / GSR_TILE;
    const int row_lo = max((int)floorf((my_lo - Rb) * 0.0625f), 0), row_hi = min((int)floorf((my_hi + Rb + 15.0f) * 0.0625f) - 1, tiles_y - 1);
    if (row_lo > row_hi) return true;                                  // above or below it
    return !rs.any_in(row_lo, row_hi + 1);                             // none of this rank's tile rows in between
}

// Geometry of gaussian `i` seen from one camera: everything rasterize.py:354-420 computes per gaussian except the colour.  `in` is only
// read when the gaussian is not culled (or DEBUG): callers may leave it unloaded for a gaussian culled_by() the camera.
template <bool DEBUG>
__device__ __forceinline__ GeoOut geometry_view(const float p[3], const GeoIn &in, const Cam &cam, int compat, int no_cull, RowShard rs,
                                                int keep_ref_drawn, const GsrDebugOut &dbg, int64_t i)
{
    GeoOut g;
    g.visible = g.keep_empty = false;
    g.p[0] = p[0]; g.p[1] = p[1]; g.p[2] = p[2];
    const float *V = cam.V, *F = cam.F;
    // project_to_camera_space, rasterize.py:80-86
    float cm[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) cm[j] = ((p[0] * V[0 + j] + p[1] * V[4 + j]) + p[2] * V[8 + j]) + V[12 + j];
    const bool culled = cm[2] < GSR_CULL_Z;  // :377
    if (!DEBUG && culled) return g;

    // clip-space point, :374, cull zeroing :378, perspective divide :381-382
    float pt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pt[j] = ((p[0] * F[0 + j] + p[1] * F[4 + j]) + p[2] * F[8 + j]) + F[12 + j];
    if (culled) pt[0] = pt[1] = pt[2] = pt[3] = 0.0f;
    const float p_w = 1.0f / (pt[3] + 0.0000001f);
    const float ndc_x = pt[0] * p_w, ndc_y = pt[1] * p_w;

    // NDC -> pixel, :391
    const float Wf = (float)cam.W, Hf = (float)cam.H;
    const float mx = ((ndc_x + 1.0f) * Wf - 1.0f) / 2.0f, my = ((ndc_y + 1.0f) * Hf - 1.0f) / 2.0f;

    const float (&C3)[3][3] = in.C3;

    // compute_2d_covariance, :201-252
    float c2[4];
    ewa_cov2d(V, C3, cm, cam.fx, cam.fy, cam.limx, cam.limy, c2);
    float a = c2[0], b01 = c2[1], b10 = c2[2], c = c2[3];
    if (DEBUG && dbg.cov2d) { dbg.cov2d[4 * i] = a; dbg.cov2d[4 * i + 1] = b01; dbg.cov2d[4 * i + 2] = b10; dbg.cov2d[4 * i + 3] = c; }
    if (culled) a = b01 = b10 = c = 0.0f;  // :388

    // compute_covering_bbox, :154-198
    float tb[4], det, spread;
    covering_bbox(mx, my, a, b01, b10, c, Wf, Hf, tb, &det, &spread);
    const float tb0 = tb[0], tb1 = tb[1], tb2 = tb[2], tb3 = tb[3];

    // conic, :395-411
    const float det_inv = det == 0.0f ? 0.0f : 1.0f / det;
    const float sx = c * det_inv, sy = a * det_inv, sxy = (-b01) * det_inv;

    // pixel rect, :415-419.  reference_compat clamps to W-1/H-1 (Q1); otherwise to W/H.
    const int xlim = compat ? cam.W - 1 : cam.W, ylim = compat ? cam.H - 1 : cam.H;
    // tb* are small non-negative integers (<= W-1); tile*16 fits int32 for any sane frame
    const int x_min = clampi((int)tb0 * GSR_TILE, 0, xlim), y_min = clampi((int)tb1 * GSR_TILE, 0, ylim);
    const int x_max = clampi((int)tb2 * GSR_TILE, 0, xlim), y_max = clampi((int)tb3 * GSR_TILE, 0, ylim);

    const float op = in.op;

    if (DEBUG) {
        if (dbg.cov3d) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) dbg.cov3d[9 * i + 3 * r + cc] = C3[r][cc];
        }
        if (dbg.cam_means) { dbg.cam_means[3 * i] = cm[0]; dbg.cam_means[3 * i + 1] = cm[1]; dbg.cam_means[3 * i + 2] = cm[2]; }
        if (dbg.screen_means) { dbg.screen_means[2 * i] = mx; dbg.screen_means[2 * i + 1] = my; }
        if (dbg.tile_bboxes) { dbg.tile_bboxes[4 * i] = (int64_t)tb0; dbg.tile_bboxes[4 * i + 1] = (int64_t)tb1; dbg.tile_bboxes[4 * i + 2] = (int64_t)tb2; dbg.tile_bboxes[4 * i + 3] = (int64_t)tb3; }
        if (dbg.sigmas) { dbg.sigmas[3 * i] = sx; dbg.sigmas[3 * i + 1] = sy; dbg.sigmas[3 * i + 2] = sxy; }
        if (dbg.pixel_bboxes) { dbg.pixel_bboxes[4 * i] = x_min; dbg.pixel_bboxes[4 * i + 1] = y_min; dbg.pixel_bboxes[4 * i + 2] = x_max; dbg.pixel_bboxes[4 * i + 3] = y_max; }
        if (dbg.opacity) dbg.opacity[i] = op;
    }

    // ---- skip guard of the driver loop, rasterize.py:441 ------------------------------------------
    // bbox_area == 0, or (Q2) any conic entry == 0.  Without reference_compat only a zero determinant
    // (degenerate / culled gaussian) is skipped.
    bool visible = !culled && (x_max - x_min) > 0 && (y_max - y_min) > 0;
    visible = visible && (compat ? (sx != 0.0f && sy != 0.0f && sxy != 0.0f) : (det != 0.0f));
    // finite inputs only (the reference would write garbage rects for NaN/Inf; we drop them)
    visible = visible && (fabsf(mx) < 1e9f) && (fabsf(my) < 1e9f) && (spread < 1e9f) && (sx == sx) && (sy == sy) && (sxy == sxy);
    const bool ref_drawn = visible;  // exactly the gaussians the reference's loop rasterizes (its iteration_step counts these)
    // Exact extra cull: alpha = opacity * exp(power), power <= 0, can never exceed 1/255 if opacity <= 1/255 (:285-291)
    visible = visible && (op > GSR_MIN_ALPHA);

    // ---- footprint AABB: where can alpha > 1/255 (and power <= 0) hold? -----------------------------
    // alpha > 1/255  <=>  sx dx^2 + sy dy^2 + 2 sxy dx dy < 2 ln(255 op).  For a positive-definite conic that
    // ellipse has the axis-aligned half extents sqrt(2 tau sy / D), sqrt(2 tau sx / D), D = sx sy - sxy^2.
    // All fp32, made conservative explicitly: D suffers cancellation, its computed value is within 3 ulp of
    // max(sx sy, sxy^2) of the true one, so D is lowered by 4e-7 * sx * sy (and the test fails over to "no
    // culling" when that leaves nothing); tau carries a 1 % margin, the extents 1e-5 relative + 0.05 px.  The
    // per-pixel fp32 evaluation in the blend can therefore never put a contributing pixel outside
    // (tests/test_gpu_parity.py::test_footprint_culling_is_exact compares against culling disabled, bit for bit).
    float hx = 3.0e38f, hy = 3.0e38f, pthr = -3.0e38f;
    if (visible && !no_cull) {
        const float sxsy = sx * sy;
        const float D = (sxsy - sxy * sxy) - 4.0e-7f * sxsy;
        if (sx > 0.0f && sy > 0.0f && D > 0.0f && D > 1.0e-6f * sxsy) {  // also keep the conic's condition number sane
            const float tau = logf(255.0f * op);  // > 0 because op > 1/255
            const float tau2 = 2.02f * tau + 1.0e-5f;
            const float invD = 1.0f / D;
            hx = sqrtf(tau2 * sy * invD) * 1.00001f + 0.05f;
            hy = sqrtf(tau2 * sx * invD) * 1.00001f + 0.05f;
            pthr = -1.01f * 1.4426950408889634f * tau - 1.0e-3f;  // -log2(255 op), loosened (footprint.h)
        }
    }
    // refine the reference rect [x_min,x_max) x [y_min,y_max) by the footprint; pixel centres are integers (Q9)
    int tx0 = 0, ty0 = 0, tx1 = 0, ty1 = 0;
    if (visible) {
        const float fx0 = fmaxf(ceilf(mx - hx), (float)x_min), fx1 = fminf(floorf(mx + hx), (float)(x_max - 1));
        const float fy0 = fmaxf(ceilf(my - hy), (float)y_min), fy1 = fminf(floorf(my + hy), (float)(y_max - 1));
        if (fx0 > fx1 || fy0 > fy1) visible = false;
        else {
            tx0 = (int)fx0 >> 4; tx1 = ((int)fx1 >> 4) + 1;
            ty0 = (int)fy0 >> 4; ty1 = ((int)fy1 >> 4) + 1;
            // multi-GPU shard (RowShard: the rank's tile rows): a gaussian that touches none of this rank's rows leaves here,
            // before the 192-B SH read, and never enters this rank's sorts
            if (!rs.any_in(ty0, ty1)) visible = false;
        }
    }

    g.visible = visible;
    g.keep_empty = !visible && keep_ref_drawn && ref_drawn;
    g.key = __float_as_uint(cm[2]);  // z >= 0.2 > 0: IEEE bits are monotone in z (rasterize.py:424-425)
    g.tx0 = tx0; g.ty0 = ty0; g.tx1 = tx1; g.ty1 = ty1;
    g.mx = mx; g.my = my; g.sx = sx; g.sy = sy; g.sxy = sxy; g.pthr = pthr; g.op = op;
    return g;
}

// One gaussian, one camera: load + view.  A gaussian behind the cull plane leaves after its 12-B mean (no debug outputs wanted).
template <bool DEBUG>
__device__ __forceinline__ GeoOut geometry_one(const GsrScene &sc, const Cam &cam, int compat, int no_cull, RowShard rs,
                                               int keep_ref_drawn, const GsrDebugOut &dbg, int64_t i)
{
    const float p[3] = {sc.means[3 * i], sc.means[3 * i + 1], sc.means[3 * i + 2]};
    if (!DEBUG && culled_by(cam, p)) {
        GeoOut g;
        g.visible = g.keep_empty = false;
        return g;
    }
    const GeoIn in = geometry_load(sc, i);
    return geometry_view<DEBUG>(p, in, cam, compat, no_cull, rs, keep_ref_drawn, dbg, i);
}

// Colour of gaussian `i` seen from the camera: sh_to_rgb, spherical_harmonics.py:27-73 (rasterize.py:368).
template <bool SH16>
__device__ __forceinline__ void colour_one(const GsrScene &sc, const Cam &cam, int64_t i, const float p[3], float rgb[3])
{
    float sh[48];
    if (SH16) load_sh48_f16(sc.sh, i, sh);
    else load_sh48(reinterpret_cast<const float *>(sc.sh), i, sh);
    sh_eval(p, sh, cam.cc, sc.sh_degree, rgb);
}

// The SH rows of a whole wave (64 consecutive gaussians: 12 KB, contiguous) through LDS, for waves most of whose gaussians are
// visible — what a scene uploaded in spatial order (renderer.morton_order) makes of nearly every wave that is not culled whole.
// Per-lane loads of a 192-B row at 192-B stride touch 64 different lines per instruction; here every instruction is one fully
// coalesced 1-KB load, written straight to LDS (global_load_lds_dwordx4: no staging registers), and a lane then reads its own row
// with 12 ds_read_b128.  Two rounds of 32 rows keep it to 6 KB of LDS per wave.  Same values into the same registers: nothing
// downstream can tell.  (A timing probe in FILE order, where 55 % of a wave's lanes are visible, measured this pattern SLOWER than
// per-lane loads — it fetches every line of the array instead of 74 % of them and the kernel is bound by HBM: hence the density
// test at the call site.)
__device__ __forceinline__ void load_sh48_wave(const float *sh, int64_t i0, int lane, float4 *lds /* this wave's 384 float4 */, float out[48])
{
    const float4 *base = reinterpret_cast<const float4 *>(sh + 48 * i0) + lane;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int k = 0; k < 6; ++k)  // LDS destination: wave-uniform base + lane * 16
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (6 * h + k) * 64),
                                             (__attribute__((address_space(3))) void *)(lds + k * 64), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the rows have landed (an LDS-DMA counts on vmcnt)
        if ((lane >> 5) == h) {
            const float4 *row = lds + (lane & 31) * 12;
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const float4 v = row[j];
                out[4 * j] = v.x; out[4 * j + 1] = v.y; out[4 * j + 2] = v.z; out[4 * j + 3] = v.w;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read before the next round's DMA overwrites the buffer
    }
}

__device__ __forceinline__ uint32_t pack_rect8(const GeoOut &g)
{
    return (uint32_t)(g.tx0 & 255) | ((uint32_t)(g.ty0 & 255) << 8) | ((uint32_t)((g.tx1 - 1) & 255) << 16) | ((uint32_t)((g.ty1 - 1) & 255) << 24);
}

// Frame reset, by workgroup 0 of the frame's first kernel: nothing in that kernel reads FrameCtrl and every later kernel of the frame
// is stream-ordered behind it (a hipMemsetAsync costs two blit kernels and a dispatch bubble, ~20 us).  Then what a blend that
// evaluates colours itself needs to know (GsrOptions.colour_stage; gsr_blend is handed no scene).
template <int THREADS>
__device__ __forceinline__ void frame_reset(uint32_t *ctrl_words, int ctrl_reset_words, const GsrScene &sc, const Cam &cam)
{
    for (int w = threadIdx.x; w < ctrl_reset_words; w += THREADS) ctrl_words[w] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        FrameCtrl *c = reinterpret_cast<FrameCtrl *>(ctrl_words);
        c->col_means = sc.means;
        c->col_sh = sc.sh;
        c->col_cc[0] = cam.cc[0]; c->col_cc[1] = cam.cc[1]; c->col_cc[2] = cam.cc[2];
        c->col_degree = sc.sh_degree;
        c->col_sh16 = sc.sh_dtype;
    }
}

// Whole frame: one thread per gaussian.
#ifndef GSR_PRE_THREADS
#define GSR_PRE_THREADS 256
#endif
// threads per workgroup of the whole-frame kernel.  128- and 64-thread builds were measured for one question only — does a smaller
// workgroup find room beside another frame's resident blend workgroups? — and change nothing (881 -> 882 / 884 frames/s with four
// frames in flight): tools/overlap_probe.py has the rest of that story.
constexpr int PRE_THREADS = GSR_PRE_THREADS;
// Register budget: fp32 SH 94 VGPRs (5 waves per SIMD); fp16 storage 67 (7 waves) — asked for 8 waves it fits 64 with a few bytes
// of spill and is 3-5 % slower (0.158 vs 0.150 ms on the Morton-ordered bench scene, the latter measured at 76 VGPRs / 6 waves).
// The wave-wide row loads give fp16 nothing (0.150 ms with and without: six per-lane loads at a 96-B stride are already cheap),
// so only fp32 has them.
// COLOUR = false (GsrOptions.colour_stage = 0): the colour is left to the blend — the record's rgb holds the "not evaluated" mark
// (negative: a colour is clamped to [0, 1], Q7) and the 192-B SH row is not touched here: 44 B read per gaussian instead of up to 236.
constexpr float COLOUR_PENDING = -1.0f;
template <bool DEBUG, bool SH16, bool COLOUR>
__global__ __launch_bounds__(PRE_THREADS, !COLOUR ? 8 : (SH16 && !DEBUG) ? 6 : 4) void preprocess_kernel(GsrScene sc, Cam cam, int compat, int no_cull, RowShard rs, int keep_ref_drawn, GaussRec *__restrict__ rec,
                                                         ushort4 *__restrict__ rect, uint32_t *__restrict__ rect8, uint32_t *__restrict__ depth_key,
                                                         GsrDebugOut dbg,
                                                         uint32_t *__restrict__ ctrl_words, int ctrl_reset_words, int packed_rect, int sh_dense_min,
                                                         const unsigned char *__restrict__ blk_dead)
{
    static_assert(COLOUR || !DEBUG, "the debug outputs include rgb");
    constexpr bool WAVE_SH = COLOUR && !DEBUG && !SH16;  // fp32 rows of a dense wave go through LDS (load_sh48_wave)
    __shared__ float4 s_sh[WAVE_SH ? (PRE_THREADS / 64) * 384 : 1];
    if (blockIdx.x == 0) frame_reset<PRE_THREADS>(ctrl_words, ctrl_reset_words, sc, cam);
    const int64_t i = (int64_t)blockIdx.x * PRE_THREADS + threadIdx.x;  // the constant, not blockDim.x: that would pull in the hidden kernarg block
    if (i >= sc.n) return;
    // block-level culling (GsrScene.block_bounds; block_flags_kernel): a wave is one block; if none of its gaussians can be drawn it
    // leaves before reading any of them (the depth sort still finds a key per gaussian: the invalid one)
    static_assert(GSR_BOUNDS_BLOCK == 64, "a block of GsrScene.block_bounds is one wave of this kernel");
    if (!DEBUG && blk_dead != nullptr && blk_dead[i >> 6]) {  // wave-uniform
        depth_key[i] = KEY_INVALID;
        return;
    }
    // (the gaussian id is not written: pass 0 of the depth sort synthesises the identity payload, 8 B per gaussian less traffic)
    const GeoOut g = geometry_one<DEBUG>(sc, cam, compat, no_cull, rs, keep_ref_drawn, dbg, i);
    float rgb[3] = {COLOUR_PENDING, COLOUR_PENDING, COLOUR_PENDING};
    bool coloured = !COLOUR;
    if constexpr (WAVE_SH) {
        const int lane = threadIdx.x & 63;
        const int64_t i0 = i - lane;
        // wave-uniform: a full wave (the last one of the grid may not be) with at least sh_dense_min visible gaussians
        if (i0 + 64 <= sc.n && (int)__popcll(__ballot(g.visible)) >= sh_dense_min) {
            float sh[48];
            load_sh48_wave(reinterpret_cast<const float *>(sc.sh), i0, lane, s_sh + (threadIdx.x >> 6) * 384, sh);
            if (g.visible) sh_eval(g.p, sh, cam.cc, sc.sh_degree, rgb);
            coloured = true;
        }
    }
    if (!coloured && (g.visible || (DEBUG && dbg.rgb))) {
        colour_one<SH16>(sc, cam, i, g.p, rgb);
        if (DEBUG && dbg.rgb) { dbg.rgb[3 * i] = rgb[0]; dbg.rgb[3 * i + 1] = rgb[1]; dbg.rgb[3 * i + 2] = rgb[2]; }
    }
    if (!g.visible) {
        // progressive render (draw_limit): the depth rank must count every gaussian the reference draws, also those
        // that cannot touch a pixel here; they stay in the sort with an empty tile rect
        if (g.keep_empty) {
            depth_key[i] = g.key;
            // only the form the binning will read is written: packed bytes ride through the depth sort as its second
            // payload when the frame has at most 256 x 256 tiles, otherwise the rect is gathered by gaussian id
            if (packed_rect) rect8[i] = 1u;  // packed {x0 = 1, y0 = 0, x1 - 1 = 0, y1 - 1 = 0}: zero width, i.e. no tiles
            else rect[i] = make_ushort4(0, 0, 0, 0);
        } else {
            depth_key[i] = KEY_INVALID;
        }
        return;
    }
    depth_key[i] = g.key;
    if (packed_rect) rect8[i] = pack_rect8(g);
    else rect[i] = make_ushort4((unsigned short)g.tx0, (unsigned short)g.ty0, (unsigned short)g.tx1, (unsigned short)g.ty1);
    GaussRec r;
    float log2_op;
    record_of(g, &r.q0, &r.q1, &log2_op);
    r.q2 = make_float4(log2_op, rgb[0], rgb[1], rgb[2]);
    rec[i] = r;
}

// Several views per launch (gsr_render_batch; gridDim.y is NOT used here): a thread takes its gaussian through every camera of the
// batch — the mean is read once, scales / rotation / opacity once (when the first camera that does not cull it comes along) and the
// 3D covariance and the sigmoid are evaluated once; what depends on the camera is geometry_view, the same code on the same values as
// the one-view kernel, so every view's records, keys and rects are bit for bit those of a single-view frame.  View v writes into
// slice v of the workspace (gsr_internal.h, view_slice).  Bytes per gaussian: 44 read once + 64 written per view that sees it.
struct CamBatch {
    Cam cam[MAX_VIEWS];
};
template <typename T>
__device__ __forceinline__ T *slice_of(T *p, int v, size_t vstride)
{
    return view_at(p, (size_t)v * vstride);
}

template <bool SH16, bool COLOUR>
__global__ __launch_bounds__(PRE_THREADS, COLOUR ? 4 : 6) void preprocess_views_kernel(GsrScene sc, CamBatch cams, int views, size_t vstride, int compat, int no_cull,
                                                                                RowShard rs, int keep_ref_drawn, GaussRec *__restrict__ rec0,
                                                                                ushort4 *__restrict__ rect0, uint32_t *__restrict__ rect80,
                                                                                uint32_t *__restrict__ depth_key0, uint32_t *__restrict__ ctrl_words0,
                                                                                int ctrl_reset_words, int packed_rect, const unsigned char *__restrict__ blk_dead0)
{
    if (blockIdx.x == 0)
        for (int v = 0; v < views; ++v) {
            frame_reset<PRE_THREADS>(slice_of(ctrl_words0, v, vstride), ctrl_reset_words, sc, cams.cam[v]);
            __syncthreads();
        }
    const int64_t i = (int64_t)blockIdx.x * PRE_THREADS + threadIdx.x;
    if (i >= sc.n) return;
    // block-level culling (GsrScene.block_bounds; block_flags_kernel), per view: bit v of `dead` = no gaussian of this wave's block can
    // be drawn in view v.  A block dead in every view is never read.
    unsigned dead = 0;  // wave-uniform
    if (blk_dead0 != nullptr)
        for (int v = 0; v < views; ++v) dead |= slice_of(blk_dead0, v, vstride)[i >> 6] ? 1u << v : 0u;
    float p[3] = {0.0f, 0.0f, 0.0f};
    if (dead != (1u << views) - 1u) { p[0] = sc.means[3 * i]; p[1] = sc.means[3 * i + 1]; p[2] = sc.means[3 * i + 2]; }
    const GsrDebugOut none = {};
    GeoIn in = {};
    bool loaded = false;
#pragma unroll 1
    for (int v = 0; v < views; ++v) {
        const Cam &cam = cams.cam[v];
        if ((dead >> v) & 1u) {  // uniform
            slice_of(depth_key0, v, vstride)[i] = KEY_INVALID;
            continue;
        }
        if (!loaded && !culled_by(cam, p)) { in = geometry_load(sc, i); loaded = true; }
        const GeoOut g = geometry_view<false>(p, in, cam, compat, no_cull, rs, keep_ref_drawn, none, i);
        uint32_t *depth_key = slice_of(depth_key0, v, vstride);
        if (!g.visible) {  // as in preprocess_kernel
            if (g.keep_empty) {
                depth_key[i] = g.key;
                if (packed_rect) slice_of(rect80, v, vstride)[i] = 1u;
                else slice_of(rect0, v, vstride)[i] = make_ushort4(0, 0, 0, 0);
            } else {
                depth_key[i] = KEY_INVALID;
            }
            continue;
        }
        float rgb[3] = {COLOUR_PENDING, COLOUR_PENDING, COLOUR_PENDING};
        if constexpr (COLOUR) colour_one<SH16>(sc, cam, i, g.p, rgb);
        depth_key[i] = g.key;
        if (packed_rect) slice_of(rect80, v, vstride)[i] = pack_rect8(g);
        else slice_of(rect0, v, vstride)[i] = make_ushort4((unsigned short)g.tx0, (unsigned short)g.ty0, (unsigned short)g.tx1, (unsigned short)g.ty1);
        GaussRec r;
        float log2_op;
        record_of(g, &r.q0, &r.q1, &log2_op);
        r.q2 = make_float4(log2_op, rgb[0], rgb[1], rgb[2]);
        slice_of(rec0, v, vstride)[i] = r;
    }
}

// ---- multi-GPU shard (the tile rows of a RowShard: begin, begin + step, ... or pairs of rows) ---------------------------------------------------
// A rank of G keeps ~1/G of the gaussians, but which ones depends on the camera, so every rank has to look at all N.  Run
// through the kernel above, nearly every wave still holds a few survivors and walks the whole path with most lanes idle
// (G = 8: 41 % of the lanes pass a cheap bound, 13 % are visible), and pass 0 of the depth sort then scans N keys to drop
// 7 of 8.  shard_preprocess_kernel instead works in three phases per workgroup of SHARD_SPAN consecutive gaussians, each in
// dense waves, handing the survivors on through id-ordered lists in LDS:
//   1. bound   24 B per gaussian (mean, log-scales): which rows can the reference's rect reach at most?
//   2. geometry of the candidates (exact: the rows of the refined rect)
//   3. colour of the visible ones (the 192-B SH read), then the workgroup's (key, id, packed rect) records go out as one
//      contiguous run; shard_compact_kernel closes the gaps between the workgroups' runs, so the depth sort starts on the
//      V_rank visible records instead of N keys, and stays stable by id.  (Letting pass 0 of the sort read the runs in
//      place measured slower than this copy: 18 + 44 us against 17 + 6 + 18.)
// Every geometry array is read once and the SH rows of the visible gaussians once.
// The bound (reference radius, rasterize.py:179-181, from the largest scale alone): cov2D = T Sigma T^T + 0.3 I with
// T = J A (:224-232), so its trace is at most |A|^2 smax^2 (|J_0|^2 + |J_1|^2) + 0.6, its largest eigenvalue at most ~the
// trace (det >= 0 up to rounding; the 0.1 floor of :172 adds < 0.32), and the radius at most 3 sqrt(.) + 1 (ceil).  Padded
// by 2 % + 1.5 px against fp32 rounding.  The final tile rows are a subset of the reference rect's rows, which are a subset
// of [floor((my - Rb) / 16), floor((my + Rb + 15) / 16)) — the reference rect ends BEFORE tile row tb3 (pixel rows < 16 tb3,
// rasterize.py:271-272, :415-418) — and of the frame's [0, tiles_y); if no row of this rank lies in there the gaussian cannot
// reach it, and neither can one whose columns, bounded the same way, miss the frame.
// Anything non-finite stays a candidate.  (Property-tested against the whole frame: shards reassemble bit-exactly.)
constexpr int SHARD_PER = 4, SHARD_SPAN = 256 * SHARD_PER;  // gaussians per thread / per workgroup in phase 1

// Stable append of the flagged threads' items to an LDS list, in thread order.  Returns this thread's position (valid when
// `flag`); *count advances by the number of flagged threads.  Two barriers.
__device__ __forceinline__ uint32_t block_append_256(bool flag, uint32_t *s_wave /* [4] */, uint32_t *count /* LDS */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t pos = *count + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t c = s_wave[w];
        if (w < wave) pos += c;
        tot += c;
    }
    __syncthreads();
    if (threadIdx.x == 0) *count += tot;
    return pos;
}

// Several views per launch (gsr_render_batch): blockIdx.y is the view, its records and runs go to slice v of the workspace — four views
// are four times the workgroups, each with ONE three-phase chain (a quarter of them walking four chains one after the other measured
// 301 against 288 us; phase 1's 24 B per gaussian come from L2 / the infinity cache for three of the four).
// amdgpu_waves_per_eu: left alone the compiler keeps 106 SGPRs, and a SIMD's 800 hold seven such waves — the eighth costs 25 scalar spills.
template <bool SH16, bool COLOUR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void shard_preprocess_kernel(GsrScene sc, CamBatch cams, int views, size_t vstride, int compat, int no_cull, RowShard rs,
                                                               GaussRec *__restrict__ rec0, ushort4 *__restrict__ rect0,
                                                               uint32_t *__restrict__ run_key0, uint32_t *__restrict__ run_id0,
                                                               uint32_t *__restrict__ run_rect80, uint32_t *__restrict__ run_cnt0,
                                                               uint32_t *__restrict__ ctrl_words0, int ctrl_reset_words, int packed_rect,
                                                               const unsigned char *__restrict__ blk_dead0)
{
    __shared__ uint32_t s_cand[SHARD_SPAN];
    __shared__ uint32_t s_key[SHARD_SPAN], s_rect8[SHARD_SPAN];
    // the visible list's ids overwrite the candidate list in place: phase 2 reads s_cand[c0 + t] before the barriers of its append and
    // writes s_id[pos] after them, pos <= c0 + t — only entries every thread has already read.  16 KB + a few words of LDS per
    // workgroup instead of 20 KB + a few words, which was one workgroup per CU too many for eight (7 waves per SIMD).
    uint32_t *const s_id = s_cand;
    __shared__ float s_l2op[SHARD_SPAN];
    __shared__ uint32_t s_wave[4], s_scan[8], s_nvis;
    const int v = (int)blockIdx.y;
    if (blockIdx.x == 0) frame_reset<256>(slice_of(ctrl_words0, v, vstride), ctrl_reset_words, sc, cams.cam[v]);  // as in preprocess_kernel
    const int64_t base = (int64_t)blockIdx.x * SHARD_SPAN;
    const int tiles_y = (cams.cam[0].H + GSR_TILE - 1) / GSR_TILE;

    const Cam &cam = cams.cam[v];
    GaussRec *__restrict__ rec = slice_of(rec0, v, vstride);
    ushort4 *__restrict__ rect = slice_of(rect0, v, vstride);
    uint32_t *__restrict__ run_key = slice_of(run_key0, v, vstride), *__restrict__ run_id = slice_of(run_id0, v, vstride);
    uint32_t *__restrict__ run_rect8 = slice_of(run_rect80, v, vstride), *__restrict__ run_cnt = slice_of(run_cnt0, v, vstride);
    const float *V = cam.V, *F = cam.F;
    const float Wf = (float)cam.W, Hf = (float)cam.H;
    if (threadIdx.x == 0) s_nvis = 0;  // (read after the barriers of phase 1's scan)
    // what phase 1 reads: a thread takes FOUR CONSECUTIVE gaussians, ids base + 4 t + r — 48 contiguous bytes of means and of
    // log-scales, three 16-B loads each when the arrays are 16-B aligned (a wave: 3 KB in a row) — so that ONE workgroup scan of the
    // threads' candidate counts puts the list in id order (round 5; before: ids base + 256 r + t, four stable appends of two barriers
    // each).  The four lie in one block of GSR_BOUNDS_BLOCK: one flag per thread.
    float p[SHARD_PER][3], ls[SHARD_PER];
    const unsigned char *blk_dead = blk_dead0 != nullptr ? slice_of(blk_dead0, v, vstride) : nullptr;
    const int64_t i0 = base + (int64_t)SHARD_PER * threadIdx.x;
    const bool blk_live = i0 < sc.n && !(blk_dead != nullptr && blk_dead[i0 >> 6]);
    const bool vec_ok = ((reinterpret_cast<uintptr_t>(sc.means) | reinterpret_cast<uintptr_t>(sc.log_scales)) & 15u) == 0;
    if (blk_live && vec_ok && i0 + SHARD_PER <= sc.n) {
        const float4 *m4 = reinterpret_cast<const float4 *>(sc.means + 3 * i0), *s4 = reinterpret_cast<const float4 *>(sc.log_scales + 3 * i0);
        const float4 ma = m4[0], mb = m4[1], mc = m4[2], sa = s4[0], sb = s4[1], sc4 = s4[2];
        p[0][0] = ma.x; p[0][1] = ma.y; p[0][2] = ma.z; p[1][0] = ma.w; p[1][1] = mb.x; p[1][2] = mb.y;
        p[2][0] = mb.z; p[2][1] = mb.w; p[2][2] = mc.x; p[3][0] = mc.y; p[3][1] = mc.z; p[3][2] = mc.w;
        ls[0] = fmaxf(sa.x, fmaxf(sa.y, sa.z)); ls[1] = fmaxf(sa.w, fmaxf(sb.x, sb.y));
        ls[2] = fmaxf(sb.z, fmaxf(sb.w, sc4.x)); ls[3] = fmaxf(sc4.y, fmaxf(sc4.z, sc4.w));
    } else {
#pragma unroll
        for (int r = 0; r < SHARD_PER; ++r) {
            const int64_t i = i0 + r;
            const bool in = blk_live && i < sc.n;
#pragma unroll
            for (int j = 0; j < 3; ++j) p[r][j] = in ? sc.means[3 * i + j] : 0.0f;
            ls[r] = in ? fmaxf(sc.log_scales[3 * i], fmaxf(sc.log_scales[3 * i + 1], sc.log_scales[3 * i + 2])) : 0.0f;
        }
    }

    // ---- phase 1: the bound ----
    uint32_t cand = 0;  // bit r: gaussian i0 + r may reach one of this rank's tile rows
#pragma unroll
    for (int r = 0; r < SHARD_PER; ++r) {
        // the same expressions as geometry_view (this file is built with -ffp-contract=off)
        float cm[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) cm[j] = ((p[r][0] * V[0 + j] + p[r][1] * V[4 + j]) + p[r][2] * V[8 + j]) + V[12 + j];
        bool k = blk_live && i0 + r < sc.n && !(cm[2] < GSR_CULL_Z);
        const float pt0 = ((p[r][0] * F[0] + p[r][1] * F[4]) + p[r][2] * F[8]) + F[12];
        const float pt1 = ((p[r][0] * F[1] + p[r][1] * F[5]) + p[r][2] * F[9]) + F[13];
        const float pt3 = ((p[r][0] * F[3] + p[r][1] * F[7]) + p[r][2] * F[11]) + F[15];
        const float p_w = 1.0f / (pt3 + 0.0000001f);
        const float ndc_x = pt0 * p_w, ndc_y = pt1 * p_w;
        const float mx = ((ndc_x + 1.0f) * Wf - 1.0f) / 2.0f, my = ((ndc_y + 1.0f) * Hf - 1.0f) / 2.0f;
        const float iz = 1.0f / cm[2];
        const float u = fminf(cam.limx, fmaxf(-cam.limx, cm[0] * iz)), vv = fminf(cam.limy, fmaxf(-cam.limy, cm[1] * iz));
        const float jx = cam.fx * iz, jy = cam.fy * iz;
        const float smax = expf(2.0f * ls[r]);
        const float trb = cam.w_sigma2 * smax * (jx * jx * (1.0f + u * u) + jy * jy * (1.0f + vv * vv)) + 0.6f;
        const float Rb = 3.0f * sqrtf(1.02f * trb + 0.4f) + 1.5f;
        if (k && Rb < 1.0e8f && fabsf(my) < 1.0e8f) {
            const int lo = max((int)floorf((my - Rb) * 0.0625f), 0), hi = min((int)floorf((my + Rb + 15.0f) * 0.0625f) - 1, tiles_y - 1);
            if (!rs.any_in(lo, hi + 1)) k = false;
            // columns: a rect that lies left or right of the frame clamps to zero width (covering_bbox + the pixel clamp)
            if (fabsf(mx) < 1.0e8f && (mx + Rb < 0.0f || mx - Rb - 16.0f > Wf)) k = false;
        }
        cand |= k ? 1u << r : 0u;
    }
    uint32_t ncand;
    uint32_t cpos = block_excl_scan_256((uint32_t)__popc(cand), s_scan, &ncand);
#pragma unroll
    for (int r = 0; r < SHARD_PER; ++r)
        if ((cand >> r) & 1u) s_cand[cpos++] = (uint32_t)(i0 + r);
    __syncthreads();

    // ---- phase 2: geometry of the candidates, 256 at a time; the visible ones are appended in order ----
    const GsrDebugOut none = {};
    for (uint32_t c0 = 0; c0 < ncand; c0 += 256) {
        const uint32_t j = c0 + threadIdx.x;
        GeoOut g;
        g.visible = false;
        int64_t i = 0;
        if (j < ncand) {
            i = (int64_t)s_cand[j];
            g = geometry_one<false>(sc, cam, compat, no_cull, rs, 0, none, i);
        }
        const uint32_t pos = block_append_256(g.visible, s_wave, &s_nvis);
        if (g.visible) {
            s_id[pos] = (uint32_t)i;
            s_key[pos] = g.key;
            s_rect8[pos] = pack_rect8(g);
            float4 q0, q1;
            float log2_op;
            record_of(g, &q0, &q1, &log2_op);
            s_l2op[pos] = log2_op;
            rec[i].q0 = q0;
            rec[i].q1 = q1;
            if (!packed_rect) rect[i] = make_ushort4((unsigned short)g.tx0, (unsigned short)g.ty0, (unsigned short)g.tx1, (unsigned short)g.ty1);
        }
    }
    __syncthreads();

    // ---- phase 3: colour of the visible ones; the workgroup's run of sort records ----
    const uint32_t nvis = s_nvis;
    if (threadIdx.x == 0) run_cnt[blockIdx.x] = nvis;
    for (uint32_t j = threadIdx.x; j < nvis; j += 256) {
        const int64_t i = (int64_t)s_id[j];
        float rgb[3] = {COLOUR_PENDING, COLOUR_PENDING, COLOUR_PENDING};
        if constexpr (COLOUR) {
            const float pm[3] = {sc.means[3 * i], sc.means[3 * i + 1], sc.means[3 * i + 2]};
            colour_one<SH16>(sc, cam, i, pm, rgb);
        }
        rec[i].q2 = make_float4(s_l2op[j], rgb[0], rgb[1], rgb[2]);
        run_key[base + j] = s_key[j];
        run_id[base + j] = (uint32_t)i;
        if (packed_rect) run_rect8[base + j] = s_rect8[j];
    }
}

// Workgroup b moves the runs of shard_preprocess workgroups [8 b, 8 b + 8) to their place in the compact arrays: position = records
// of the workgroups before them (the counts are a few KB, L2-resident, summed here four at a time: cheaper than a scan kernel and
// its boundary) + rank inside.  Eight runs per workgroup (round 5; one until then: every workgroup paid the sum, the scan's two
// barriers and one short dependent copy — 44 us for four views of a rank of 8, i.e. ~6000 x 4 workgroups of ~130 records).
constexpr int COMPACT_RUNS = 8;
__global__ __launch_bounds__(256) void shard_compact_kernel(const uint32_t *__restrict__ run_key, const uint32_t *__restrict__ run_id,
                                                            const uint32_t *__restrict__ run_rect8, const uint32_t *__restrict__ run_cnt,
                                                            uint32_t *__restrict__ key, uint32_t *__restrict__ id, uint32_t *__restrict__ rect8,
                                                            uint32_t *__restrict__ n_records, int n_runs, int packed_rect, size_t vstride)
{
    __shared__ uint32_t scratch[8];
    run_key = view_slice(run_key, vstride); run_id = view_slice(run_id, vstride); run_rect8 = view_slice(run_rect8, vstride);
    run_cnt = view_slice(run_cnt, vstride); key = view_slice(key, vstride); id = view_slice(id, vstride); rect8 = view_slice(rect8, vstride);
    n_records = view_slice(n_records, vstride);
    const int b0 = (int)blockIdx.x * COMPACT_RUNS, quads = b0 >> 2;  // COMPACT_RUNS is a multiple of 4
    uint32_t mine = 0;
    for (int q = threadIdx.x; q < quads; q += 256) {
        const uint4 c = reinterpret_cast<const uint4 *>(run_cnt)[q];
        mine += (c.x + c.y) + (c.z + c.w);
    }
    uint32_t before;
    block_excl_scan_256(mine, scratch, &before);
    // the eight runs as ONE list of T records: record j lies in the run whose prefix it has passed (the eight prefixes stay in
    // registers), so that every thread's loads are independent of one another — four in flight before the first store
    uint32_t pre[COMPACT_RUNS + 1];
    pre[0] = 0;
    if (b0 + COMPACT_RUNS <= n_runs) {
        const uint4 c0 = reinterpret_cast<const uint4 *>(run_cnt)[quads], c1 = reinterpret_cast<const uint4 *>(run_cnt)[quads + 1];
        pre[1] = c0.x; pre[2] = pre[1] + c0.y; pre[3] = pre[2] + c0.z; pre[4] = pre[3] + c0.w;
        pre[5] = pre[4] + c1.x; pre[6] = pre[5] + c1.y; pre[7] = pre[6] + c1.z; pre[8] = pre[7] + c1.w;
    } else {
#pragma unroll
        for (int r = 0; r < COMPACT_RUNS; ++r) pre[r + 1] = pre[r] + (b0 + r < n_runs ? run_cnt[b0 + r] : 0u);
    }
    const uint32_t T = pre[COMPACT_RUNS];
    const size_t src0 = (size_t)b0 * SHARD_SPAN;
    for (uint32_t j0 = threadIdx.x; j0 < T; j0 += 4 * 256) {
        uint32_t k[4], d[4], q[4];
        size_t src[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t j = j0 + u * 256;
            uint32_t r = 0;
#pragma unroll
            for (int t = 1; t < COMPACT_RUNS; ++t) r += j >= pre[t] ? 1u : 0u;
            uint32_t first = 0;
#pragma unroll
            for (int t = 1; t < COMPACT_RUNS; ++t) first = r == (uint32_t)t ? pre[t] : first;
            src[u] = src0 + (size_t)r * SHARD_SPAN + (j - first);
            k[u] = d[u] = q[u] = 0;
            if (j < T) {
                k[u] = run_key[src[u]];
                d[u] = run_id[src[u]];
                if (packed_rect) q[u] = run_rect8[src[u]];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t j = j0 + u * 256;
            if (j < T) {
                key[before + j] = k[u];
                id[before + j] = d[u];
                if (packed_rect) rect8[before + j] = q[u];
            }
        }
    }
    before += T;
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_records = before;
}

static Cam make_cam(const GsrCamera &c)
{
    Cam k;
    for (int j = 0; j < 16; ++j) { k.V[j] = c.w2c[j]; k.F[j] = c.full_proj[j]; }
    for (int j = 0; j < 3; ++j) k.cc[j] = c.cam_center[j];
    k.fx = c.focal_x; k.fy = c.focal_y; k.limx = c.lim_x; k.limy = c.lim_y;
    k.W = c.width; k.H = c.height;
    // largest eigenvalue of A^T A (A = w2c[:3,:3]) by power iteration in float64; exactly 1 for a unit qvec.  Only an
    // upper bound is needed (shard early-out): pad by 1e-3 and never go below the largest column norm.
    double M[3][3];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            M[a][b] = 0.0;
            for (int r = 0; r < 3; ++r) M[a][b] += (double)c.w2c[4 * r + a] * (double)c.w2c[4 * r + b];
        }
    double v[3] = {0.6, 0.5, 0.62}, lam = 0.0;
    for (int it = 0; it < 64; ++it) {
        double w[3];
        for (int a = 0; a < 3; ++a) w[a] = M[a][0] * v[0] + M[a][1] * v[1] + M[a][2] * v[2];
        lam = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        if (!(lam > 0.0)) break;
        for (int a = 0; a < 3; ++a) v[a] = w[a] / lam;
    }
    const double tr = M[0][0] + M[1][1] + M[2][2];  // >= lambda_max >= tr / 3: if the iteration went wrong, fall back to the trace
    if (!(lam >= tr / 3.0 * 0.999) || !(lam <= tr * 1.001)) lam = tr;
    k.w_sigma2 = (float)(lam * 1.001 + 1e-12);
    return k;
}

// The frame's first kernel when the scene has block bounds: one thread per block and view -> blk_dead (1 = the preprocess skips the
// block).  ~100 K threads of a few hundred flops: a few microseconds, against reading 44 B per gaussian of the blocks it rules out.
// (Round 5 first ran the test inside the preprocess, once per workgroup: 256 lanes each redoing it took the kernel from 128 to
// 173 us.)  Also gsr_block_visibility (tests and tooling: which blocks does a view skip?).
__global__ __launch_bounds__(256) void block_flags_kernel(GsrScene sc, CamBatch cams, size_t vstride, RowShard rs,
                                                          unsigned char *__restrict__ dead0)
{
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nblk = (sc.n + GSR_BOUNDS_BLOCK - 1) / GSR_BOUNDS_BLOCK;
    if (b < nblk) view_slice(dead0, vstride)[b] = block_dead(sc.block_bounds + 8 * b, cams.cam[blockIdx.y], rs) ? 1 : 0;
}

static void launch_block_flags(const GsrScene &scene, const CamBatch &kb, int views, size_t vstride, const GsrOptions &opts, unsigned char *dead, hipStream_t s)
{
    const int64_t nblk = (scene.n + GSR_BOUNDS_BLOCK - 1) / GSR_BOUNDS_BLOCK;
    // progressive frames rank every gaussian the reference draws, whatever rows it touches: no row test then
    const RowShard rs = opts.draw_limit > 0 ? RowShard{0, 1, 0} : row_shard_of(opts);
    hipLaunchKernelGGL(block_flags_kernel, dim3((unsigned)((nblk + 255) / 256), (unsigned)views), dim3(256), 0, s, scene, kb, vstride, rs, dead);
}

int launch_block_visibility(const GsrScene &scene, const GsrCamera &cam, const GsrOptions &opts, unsigned char *dead, hipStream_t s)
{
    if (scene.n <= 0) return GSR_OK;
    CamBatch kb;
    for (int v = 0; v < MAX_VIEWS; ++v) kb.cam[v] = make_cam(cam);
    launch_block_flags(scene, kb, 1, 0, opts, dead, s);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

int launch_preprocess(const GsrScene &scene, const GsrCamera *cams, const GsrOptions &opts, const Workspace &ws,
                      const GsrDebugOut *dbg, int ctrl_reset_words, hipStream_t s)
{
    const int views = ws.views;
    if (views < 1 || views > MAX_VIEWS || (dbg && views != 1)) { set_error("bad view count %d", views); return GSR_ERR_BAD_ARG; }
    if (scene.n <= 0) {  // no kernel to carry the frame reset
        for (int v = 0; v < views; ++v)
            GSR_HIP(hipMemsetAsync(reinterpret_cast<char *>(ws.ctrl) + (size_t)v * ws.view_stride, 0, 4 * (size_t)ctrl_reset_words, s));
        return GSR_OK;
    }
    const unsigned grid = (unsigned)((scene.n + PRE_THREADS - 1) / PRE_THREADS);
    CamBatch kb;
    for (int v = 0; v < MAX_VIEWS; ++v) kb.cam[v] = make_cam(cams[v < views ? v : 0]);
    const Cam &k = kb.cam[0];
    GsrDebugOut d;
    memset(&d, 0, sizeof d);
    if (dbg) d = *dbg;
    const RowShard rs = row_shard_of(opts);
    const int packed = rect_fits_8bit(ws) ? 1 : 0;
    const bool h16 = scene.sh_dtype == 1;
    // from how many visible gaussians per wave on the wave's SH rows are fetched whole through LDS (load_sh48_wave);
    // GsrOptions.sh_dense_min overrides it for experiments (65 = never).  Swept on the bench frame (file order / Morton order): >= 56: 0.273 / 0.216 ms,
    // >= 48: 0.275 / 0.216, >= 32: 0.359 / 0.202, >= 16: 0.381 / 0.205, never: 0.288 / 0.241.
    const int sh_dense_min = opts.sh_dense_min > 0 ? opts.sh_dense_min : 48;
    const bool colour = opts.colour_stage == 1;  // 0: the blend evaluates a gaussian's colour when a tile first stages it
    const int keep_drawn = opts.draw_limit > 0 ? 1 : 0;
    // block-level culling: the flags first (not for the debug pass, whose outputs cover every gaussian; not for the three-phase shard
    // kernel either: its phase 1 already leaves a gaussian after 24 B, and measured on rank 3 of 8 of the bench frame the flags
    // kernel costs what the skipped loads return — 5.4 + 84.2 us against 82.6 us without; with rows in pairs, rank 2 of 8, four views:
    // 9.4 + 218.7 against 222)
    const unsigned char *blk_dead = nullptr;
    if (scene.block_bounds != nullptr && !dbg && !shard_compact(opts)) {
        launch_block_flags(scene, kb, views, ws.view_stride, opts, ws.blk_dead, s);
        blk_dead = ws.blk_dead;
    }
#define GSR_LAUNCH_PRE(DBG, H16, COL)                                                                                         \
    hipLaunchKernelGGL((preprocess_kernel<DBG, H16, COL>), dim3(grid), dim3(PRE_THREADS), 0, s, scene, k, opts.reference_compat,     \
                       opts.no_footprint_cull, rs, keep_drawn, ws.rec, ws.rect,     \
                       ws.rect8[0], ws.key[0], d, reinterpret_cast<uint32_t *>(ws.ctrl), ctrl_reset_words, packed, sh_dense_min, DBG ? nullptr : blk_dead)
#define GSR_LAUNCH_VIEWS(H16, COL)                                                                                            \
    hipLaunchKernelGGL((preprocess_views_kernel<H16, COL>), dim3(grid), dim3(PRE_THREADS), 0, s, scene, kb, views, ws.view_stride,     \
                       opts.reference_compat, opts.no_footprint_cull, rs, keep_drawn, ws.rec, ws.rect,   \
                       ws.rect8[0], ws.key[0], reinterpret_cast<uint32_t *>(ws.ctrl), ctrl_reset_words, packed, blk_dead)
    // debug outputs cover every gaussian: for a shard (below) that is a pass of its own, whose other outputs are then
    // overwritten
    if (dbg) { if (h16) GSR_LAUNCH_PRE(true, true, true); else GSR_LAUNCH_PRE(true, false, true); }
    if (shard_compact(opts)) {
        const unsigned sgrid = (unsigned)((scene.n + SHARD_SPAN - 1) / SHARD_SPAN);
        // the workgroups' runs go to the "out" halves of the depth sort's ping-pong buffers, idle until its pass 0 scatters
        // into them (after shard_compact_kernel has read them); their lengths to blk_sum, idle until the pair count
        uint32_t *run_cnt = ws.blk_sum;
#define GSR_LAUNCH_SHARD(H16, COL)                                                                                            \
    hipLaunchKernelGGL((shard_preprocess_kernel<H16, COL>), dim3(sgrid, (unsigned)views), dim3(256), 0, s, scene, kb, views, ws.view_stride, opts.reference_compat,             \
                       opts.no_footprint_cull, rs, ws.rec, ws.rect, ws.key[1], ws.val[1], ws.rect8[1], \
                       run_cnt, reinterpret_cast<uint32_t *>(ws.ctrl), ctrl_reset_words, packed, blk_dead)
        if (h16) { if (colour) GSR_LAUNCH_SHARD(true, true); else GSR_LAUNCH_SHARD(true, false); }
        else { if (colour) GSR_LAUNCH_SHARD(false, true); else GSR_LAUNCH_SHARD(false, false); }
#undef GSR_LAUNCH_SHARD
        hipLaunchKernelGGL(shard_compact_kernel, dim3((sgrid + COMPACT_RUNS - 1) / COMPACT_RUNS, (unsigned)views), dim3(256), 0, s, ws.key[1], ws.val[1], ws.rect8[1],
                           run_cnt, ws.key[0], ws.val[0], ws.rect8[0], &ws.ctrl->n_records, (int)sgrid, packed, ws.view_stride);
    } else if (!dbg) {
        if (views > 1) {
            if (h16) { if (colour) GSR_LAUNCH_VIEWS(true, true); else GSR_LAUNCH_VIEWS(true, false); }
            else { if (colour) GSR_LAUNCH_VIEWS(false, true); else GSR_LAUNCH_VIEWS(false, false); }
        } else {
            if (h16) { if (colour) GSR_LAUNCH_PRE(false, true, true); else GSR_LAUNCH_PRE(false, true, false); }
            else { if (colour) GSR_LAUNCH_PRE(false, false, true); else GSR_LAUNCH_PRE(false, false, false); }
        }
    }
#undef GSR_LAUNCH_PRE
#undef GSR_LAUNCH_VIEWS
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

}  // namespace gsr
