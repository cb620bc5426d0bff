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

}  // namespace gsr
