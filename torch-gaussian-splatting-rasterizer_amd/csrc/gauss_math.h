// gauss_math.h — per-gaussian device math shared by the fused preprocess kernel and the stand-alone helpers.
// Every function keeps the reference's fp32 operation order (file:line cited); translation units that include
// this header are built with -ffp-contract=off (blend.hip, which is not, uses sh_eval only: it carries its own pragma).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "../../include/gsr.h"

namespace gsr {

// rasterize.py:41-56 on fp32 inputs
__device__ __forceinline__ void quat_to_rot(float w, float x, float y, float z, float R[3][3])
{
    R[0][0] = 1.0f - 2.0f * (y * y) - 2.0f * (z * z); R[0][1] = 2.0f * x * y - 2.0f * z * w;           R[0][2] = 2.0f * x * z + 2.0f * y * w;
    R[1][0] = 2.0f * x * y + 2.0f * z * w;           R[1][1] = 1.0f - 2.0f * (x * x) - 2.0f * (z * z); R[1][2] = 2.0f * y * z - 2.0f * x * w;
    R[2][0] = 2.0f * x * z - 2.0f * y * w;           R[2][1] = 2.0f * y * z + 2.0f * x * w;           R[2][2] = 1.0f - 2.0f * (x * x) - 2.0f * (y * y);
}

// get_covariance_matrix_from_mesh, rasterize.py:89-120
__device__ __forceinline__ void cov3d_of(const float ls[3], const float4 q, float cov[3][3])
{
    const float s[3] = {expf(ls[0]), expf(ls[1]), expf(ls[2])};
    float nrm = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    nrm = nrm < 1e-12f ? 1e-12f : nrm;  // F.normalize eps, :112
    float R[3][3], M[3][3];
    quat_to_rot(q.x / nrm, q.y / nrm, q.z / nrm, q.w / nrm, R);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) M[i][j] = R[i][j] * s[j];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) cov[i][j] = M[i][0] * M[j][0] + M[i][1] * M[j][1] + M[i][2] * M[j][2];
}

// sh_to_rgb, spherical_harmonics.py:27-73, for one gaussian.  `get(e)` returns element e = 3 k + c of its [16][3] coefficient row.
// The evaluation walks the row in MEMORY order (coefficient-major), each element consumed once into its channel's running sums, so a
// caller that loads the row 16 B at a time needs only a few of its 48 values in registers at once (the blend evaluates deferred
// colours inside a 64-VGPR kernel); per channel the operations and their association are the reference's:
//   col = sh0 c0;  col += ((-c1 y) sh1 + (c1 z) sh2) - (c1 x) sh3;  col += (((t4 + t5) + t6) + t7) + t8;  col += ((...(t9 + t10) ... ) + t15)
// with t_k = basis_k sh_k, basis_k written exactly as spherical_harmonics.py:45-65 writes it; + 0.5, clamp to [0, 1] (:69-71, Q7).
template <typename Get>
__device__ __forceinline__ void sh_eval_with(const float p[3], Get get, const float cc[3], int degree, float rgb[3])
{
#pragma clang fp contract(off)  // also where the including file is built with contraction on (blend.hip evaluates deferred colours)
    const float d0 = p[0] - cc[0], d1 = p[1] - cc[1], d2 = p[2] - cc[2];
    const float n = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
    const float x = d0 / n, y = d1 / n, z = d2 / n;
    const float c0 = 0.28209479177387814f, c1 = 0.4886025119029199f, nc1 = -0.4886025119029199f;
    const float k20 = 1.0925484305920792f, k21 = -1.0925484305920792f, k22 = 0.31539156525252005f,
                k23 = -1.0925484305920792f, k24 = 0.5462742152960396f;
    const float k30 = -0.5900435899266435f, k31 = 2.890611442640554f, k32 = -0.4570457994644658f,
                k33 = 0.3731763325901154f, k34 = -0.4570457994644658f, k35 = 1.445305721320277f,
                k36 = -0.5900435899266435f;
    float col[3], part[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) col[c] = get(c) * c0;
    if (degree > 0) {  // :45-46
        const float b1 = nc1 * y, b2 = c1 * z, b3 = c1 * x;
#pragma unroll
        for (int c = 0; c < 3; ++c) part[c] = b1 * get(3 + c);
#pragma unroll
        for (int c = 0; c < 3; ++c) part[c] = part[c] + b2 * get(6 + c);
#pragma unroll
        for (int c = 0; c < 3; ++c) col[c] = col[c] + (part[c] - b3 * get(9 + c));
        if (degree > 1) {  // :48-55
            const float xx = x * x, yy = y * y;
            const float b4 = (k20 * x) * y, b5 = (k21 * y) * z, b6 = k22 * (((2.0f * z) * z - xx) - yy), b7 = (k23 * x) * z,
                        b8 = k24 * (xx - yy);
#pragma unroll
            for (int c = 0; c < 3; ++c) part[c] = b4 * get(12 + c);
#pragma unroll
            for (int c = 0; c < 3; ++c) part[c] = part[c] + b5 * get(15 + c);
#pragma unroll
            for (int c = 0; c < 3; ++c) part[c] = part[c] + b6 * get(18 + c);
#pragma unroll
            for (int c = 0; c < 3; ++c) part[c] = part[c] + b7 * get(21 + c);
#pragma unroll
            for (int c = 0; c < 3; ++c) col[c] = col[c] + (part[c] + b8 * get(24 + c));
            if (degree > 2) {  // :56-65
                const float b9 = (k30 * y) * ((3.0f * x) * x - yy), b10 = ((k31 * x) * y) * z,
                            b11 = (k32 * y) * (((4.0f * z) * z - xx) - yy),
                            b12 = (k33 * z) * (((2.0f * z) * z - (3.0f * x) * x) - (3.0f * y) * y),
                            b13 = (k34 * x) * (((4.0f * z) * z - xx) - yy), b14 = (k35 * z) * (xx - yy),
                            b15 = (k36 * x) * (xx - (3.0f * y) * y);
#pragma unroll
                for (int c = 0; c < 3; ++c) part[c] = b9 * get(27 + c);
#pragma unroll
                for (int c = 0; c < 3; ++c) part[c] = part[c] + b10 * get(30 + c);
#pragma unroll
                for (int c = 0; c < 3; ++c) part[c] = part[c] + b11 * get(33 + c);
#pragma unroll
                for (int c = 0; c < 3; ++c) part[c] = part[c] + b12 * get(36 + c);
#pragma unroll
                for (int c = 0; c < 3; ++c) part[c] = part[c] + b13 * get(39 + c);
#pragma unroll
                for (int c = 0; c < 3; ++c) part[c] = part[c] + b14 * get(42 + c);
#pragma unroll
                for (int c = 0; c < 3; ++c) col[c] = col[c] + (part[c] + b15 * get(45 + c));
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = col[c] + 0.5f;                        // :69
        rgb[c] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);     // :71 (Q7)
    }
}

// sh = 48 floats [16][3] of one gaussian, already in registers
__device__ __forceinline__ void sh_eval(const float p[3], const float *sh, const float cc[3], int degree, float rgb[3])
{
    sh_eval_with(p, [sh](int e) { return sh[e]; }, cc, degree, rgb);
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// fp16 storage: 96-B rows, six 16-B loads, widened to fp32 before evaluation
__device__ __forceinline__ void load_sh48_f16(const void *sh, int64_t i, float out[48])
{
    const uint4 *p = reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(sh) + 96 * i);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const uint4 v = p[k];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            out[8 * k + 2 * j] = __half2float(__ushort_as_half((unsigned short)(w[j] & 0xFFFFu)));
            out[8 * k + 2 * j + 1] = __half2float(__ushort_as_half((unsigned short)(w[j] >> 16)));
        }
    }
}

__device__ __forceinline__ void load_sh48(const float *sh, int64_t i, float out[48])
{
    const float4 *p = reinterpret_cast<const float4 *>(sh + 48 * i);  // 192-B rows are 16-B aligned
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float4 v = p[k];
        out[4 * k] = v.x; out[4 * k + 1] = v.y; out[4 * k + 2] = v.z; out[4 * k + 3] = v.w;
    }
}


// compute_2d_covariance, rasterize.py:201-252, one gaussian.  V = w2c (row-vector convention, 16 floats),
// C3 = 3x3 covariance, cm = camera-space mean.  Returns the 2x2 block {a, b01, b10, c} incl. the 0.3 low-pass.
__device__ __forceinline__ void ewa_cov2d(const float *V, const float C3[3][3], const float cm[3], float fx, float fy, float limx,
                                          float limy, float out[4])
{
    const float tz = cm[2];
    const float txtz = cm[0] / tz, tytz = cm[1] / tz;
    const float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;  // :218-221
    const float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    float J[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};  // :224-228
    J[0][0] = fx / tz;
    J[0][2] = -(fx * tx) / (tz * tz);
    J[1][1] = fy / tz;
    J[1][2] = -(fy * ty) / (tz * tz);
    float T[3][3], TV[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) T[r][c] = (V[4 * c + 0] * J[r][0] + V[4 * c + 1] * J[r][1]) + V[4 * c + 2] * J[r][2];  // :230-232
    const float vrk[3][3] = {{C3[0][0], C3[0][1], C3[0][2]}, {C3[0][1], C3[1][1], C3[1][2]}, {C3[0][2], C3[1][2], C3[2][2]}};  // :234-243
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) TV[r][c] = (T[r][0] * vrk[0][c] + T[r][1] * vrk[1][c]) + T[r][2] * vrk[2][c];
    float PC[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) PC[r][c] = (TV[r][0] * T[c][0] + TV[r][1] * T[c][1]) + TV[r][2] * T[c][2];  // :245
    out[0] = PC[0][0] + GSR_LOWPASS;  // :249
    out[1] = PC[0][1];
    out[2] = PC[1][0];
    out[3] = PC[1][1] + GSR_LOWPASS;  // :250
}

// compute_covering_bbox, rasterize.py:154-198, one gaussian: tile-unit bbox as floats (already floored).
__device__ __forceinline__ void covering_bbox(float mx, float my, float a, float b01, float b10, float c, float Wf, float Hf,
                                              float tb[4], float *det_out, float *spread_out)
{
    const float det = a * c - b10 * b01;
    const float trace = a + c;
    const float disc = sqrtf(fmaxf((trace * trace) / 4.0f - det, GSR_EIG_FLOOR));
    const float l1 = trace / 2.0f + disc, l2 = trace / 2.0f - disc;
    const float spread = ceilf(GSR_GAUSSIAN_SPREAD * sqrtf(fmaxf(l1, l2)));
    tb[0] = floorf(clampf((mx - spread) / 16.0f, 0.0f, Wf - 1.0f));
    tb[1] = floorf(clampf((my - spread) / 16.0f, 0.0f, Hf - 1.0f));
    tb[2] = floorf(clampf((mx + (spread + 16.0f - 1.0f)) / 16.0f, 0.0f, Wf - 1.0f));
    tb[3] = floorf(clampf((my + (spread + 16.0f - 1.0f)) / 16.0f, 0.0f, Hf - 1.0f));
    *det_out = det;
    *spread_out = spread;
}

}  // namespace gsr
