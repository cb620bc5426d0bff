// footprint.h — exact, conservative test "can this gaussian contribute to any pixel of this rectangle?"
//
// A gaussian contributes to a pixel iff alpha = opacity * 2^p > 1/255 and p <= 0 (rasterize.py:285-291), with
// p(dx,dy) = A dx^2 + B dx dy + C dy^2 in the log2 domain (A, C < 0 for a positive-definite conic).  So it can
// touch a rectangle of pixel centres only if max over the rectangle of p exceeds -log2(255 opacity).  `pthr` is
// that bound, loosened by 1 % + 1e-3 in preprocess so fp32 rounding in the per-pixel evaluation can never put a
// contributing pixel outside; pthr < -1e37 marks "no culling" (indefinite conic or culling disabled).
// The tests only ever skip work whose contribution is exactly zero: tests/test_gpu_parity.py proves the frame
// is bit-identical with culling disabled.
#pragma once
#include <hip/hip_runtime.h>

namespace gsr {

// q0 = {mean_x, mean_y, -B/(2C), -B/(2A)}, q1 = {A, B, C, pthr}; rectangle of pixel centres [x0,x1]x[y0,y1].
// p here is the quadratic WITHOUT the log2(opacity) term the blend folds in; pthr accounts for the opacity.
__device__ __forceinline__ bool footprint_hits_rect(const float4 q0, const float4 q1, float x0, float x1, float y0, float y1)
{
    if (q1.w < -1e37f) return true;
    // offset from the mean to the nearest point of the rectangle; both zero <=> the mean is inside
    const float dxn = q0.x - fminf(fmaxf(q0.x, x0), x1);
    const float dyn = q0.y - fminf(fmaxf(q0.y, y0), y1);
    if (dxn == 0.0f && dyn == 0.0f) return true;
    // p is concave with its maximum at the mean, so over the rectangle it peaks on an edge facing the mean:
    // maximise the 1-D quadratic along the (at most two) facing edges.
    float best = -3.0e38f;
    if (dxn != 0.0f) {  // vertical edge at dx = dxn, dy in [my - y1, my - y0]
        const float dy = fminf(fmaxf(q0.z * dxn, q0.y - y1), q0.y - y0);
        best = fmaxf(best, dxn * (q1.x * dxn + q1.y * dy) + q1.z * dy * dy);
    }
    if (dyn != 0.0f) {  // horizontal edge at dy = dyn, dx in [mx - x1, mx - x0]
        const float dx = fminf(fmaxf(q0.w * dyn, q0.x - x1), q0.x - x0);
        best = fmaxf(best, dx * (q1.x * dx + q1.y * dyn) + q1.z * dyn * dyn);
    }
    return best >= q1.w;
}

// The same test, plus: may the blend evaluate this entry on this rectangle WITHOUT its two guards?  (blend.hip, fast path.)
//   - min(alpha, 0.99) is the identity when log2(opacity) <= log2(0.99) - 1e-4: alpha = 2^p with p <= L, and v_exp_f32 is
//     accurate to 1 ulp, so alpha <= opacity (1 + 2^-22) < 0.99;
//   - `p <= L` (power <= 0, rasterize.py:291) holds for every pixel of the rectangle when the quadratic's maximum over it,
//     `best` (<= 0, reached on an edge facing the mean), is below -1e-3 and the conic is well conditioned, B^2 <= 0.99 * 4AC:
//     then |A| dx^2 + |B dx dy| + |C| dy^2 <= 2 q / (1 - rho) <= 400 q (q = -quadratic, rho = |B| / 2 sqrt(AC) <= 0.995), the
//     rounding error of the blend's three-FMA evaluation is below 4 * 2^-24 * (400 q + |L|) < q for q >= 1e-3, |L| < 8, and the
//     computed p stays below L.  With the mean inside the rectangle (best = 0) the guard stays on: next to its mean a
//     gaussian's computed power can round to +1e-7 and the reference then skips the pixel.
// Both guards are therefore no-ops wherever `fast` is set: dropping them cannot change a bit.
struct FootprintClass {
    bool hit, fast;
};
__device__ __forceinline__ FootprintClass footprint_classify(const float4 q0, const float4 q1, float L, float x0, float x1, float y0, float y1)
{
    FootprintClass r = {true, false};
    if (q1.w < -1e37f) return r;
    const float dxn = q0.x - fminf(fmaxf(q0.x, x0), x1);
    const float dyn = q0.y - fminf(fmaxf(q0.y, y0), y1);
    if (dxn == 0.0f && dyn == 0.0f) return r;
    float best = -3.0e38f;
    if (dxn != 0.0f) {
        const float dy = fminf(fmaxf(q0.z * dxn, q0.y - y1), q0.y - y0);
        best = fmaxf(best, dxn * (q1.x * dxn + q1.y * dy) + q1.z * dy * dy);
    }
    if (dyn != 0.0f) {
        const float dx = fminf(fmaxf(q0.w * dyn, q0.x - x1), q0.x - x0);
        best = fmaxf(best, dx * (q1.x * dx + q1.y * dyn) + q1.z * dyn * dyn);
    }
    r.hit = best >= q1.w;
    r.fast = r.hit && best <= -1.0e-3f && q1.y * q1.y <= 3.96f * (q1.x * q1.z) && L <= -0.0146f;  // log2(0.99) = -0.014500
    return r;
}

}  // namespace gsr
