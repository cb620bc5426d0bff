// helpers.hip — stand-alone kernels behind the reference's helper functions (same maths as the fused preprocess,
// gauss_math.h; built with -ffp-contract=off), plus the single-gaussian compositing step rasterize_gaussian.
// They exist for API parity with the reference's module surface and for unit parity tests;
// the render path uses the fused kernels.
#include "gsr_internal.h"
#include "gauss_math.h"

namespace gsr {

struct Mat16 {
    float m[16];
};

__global__ __launch_bounds__(256) void sh_to_rgb_kernel(int64_t n, const float *__restrict__ means, const float *__restrict__ sh,
                                                        float cx, float cy, float cz, int degree, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float p[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
    const float cc[3] = {cx, cy, cz};
    float c[48], rgb[3];
    load_sh48(sh, i, c);
    sh_eval(p, c, cc, degree, rgb);
    out[3 * i] = rgb[0]; out[3 * i + 1] = rgb[1]; out[3 * i + 2] = rgb[2];
}

__global__ __launch_bounds__(256) void cov3d_kernel(int64_t n, const float *__restrict__ log_scales, const float *__restrict__ quats,
                                                    float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float ls[3] = {log_scales[3 * i], log_scales[3 * i + 1], log_scales[3 * i + 2]};
    const float4 q = reinterpret_cast<const float4 *>(quats)[i];
    float C3[3][3];
    cov3d_of(ls, q, C3);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) out[9 * i + 3 * r + c] = C3[r][c];
}

// gsr_scene_bounds: one wave per block of GSR_BOUNDS_BLOCK = 64 consecutive gaussians -> {min xyz, max xyz, max log-scale, 0}.
// A non-finite mean or scale makes the block's box unbounded (the preprocess then never skips it).
__global__ __launch_bounds__(256) void scene_bounds_kernel(int64_t n, const float *__restrict__ means, const float *__restrict__ log_scales,
                                                           float *__restrict__ bounds)
{
    static_assert(GSR_BOUNDS_BLOCK == 64, "one wave per block");
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t blk = i >> 6;
    if (blk * 64 >= n) return;  // wave-uniform
    const float INF = __builtin_huge_valf();
    float lo[3] = {INF, INF, INF}, hi[3] = {-INF, -INF, -INF}, ls = -INF, bad = 0.0f;
    if (i < n) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float p = means[3 * i + j], s = log_scales[3 * i + j];
            lo[j] = hi[j] = p;
            ls = fmaxf(ls, s);
            if (!(fabsf(p) < INF) || !(fabsf(s) < INF)) bad = 1.0f;  // NaN or Inf
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            lo[j] = fminf(lo[j], __shfl_xor(lo[j], d, 64));
            hi[j] = fmaxf(hi[j], __shfl_xor(hi[j], d, 64));
        }
        ls = fmaxf(ls, __shfl_xor(ls, d, 64));
        bad = fmaxf(bad, __shfl_xor(bad, d, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        if (bad != 0.0f) { lo[0] = lo[1] = lo[2] = -INF; hi[0] = hi[1] = hi[2] = INF; ls = INF; }
        float4 *dst = reinterpret_cast<float4 *>(bounds + 8 * (size_t)blk);
        dst[0] = make_float4(lo[0], lo[1], lo[2], hi[0]);
        dst[1] = make_float4(hi[1], hi[2], ls, 0.0f);
    }
}

int launch_scene_bounds(int64_t n, const float *means, const float *log_scales, float *bounds, hipStream_t s)
{
    if (n <= 0) return GSR_OK;
    hipLaunchKernelGGL(scene_bounds_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, means, log_scales, bounds);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

// project_to_camera_space, rasterize.py:80-86
__global__ __launch_bounds__(256) void project_kernel(int64_t n, const float *__restrict__ means, Mat16 V, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float p[3] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
#pragma unroll
    for (int j = 0; j < 3; ++j) out[3 * i + j] = ((p[0] * V.m[0 + j] + p[1] * V.m[4 + j]) + p[2] * V.m[8 + j]) + V.m[12 + j];
}

// compute_2d_covariance, rasterize.py:201-252
__global__ __launch_bounds__(256) void cov2d_kernel(int64_t n, const float *__restrict__ cov3d, const float *__restrict__ cam_means,
                                                    Mat16 V, float fx, float fy, float limx, float limy, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float C3[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C3[r][c] = cov3d[9 * i + 3 * r + c];
    const float cm[3] = {cam_means[3 * i], cam_means[3 * i + 1], cam_means[3 * i + 2]};
    float c2[4];
    ewa_cov2d(V.m, C3, cm, fx, fy, limx, limy, c2);
    reinterpret_cast<float4 *>(out)[i] = make_float4(c2[0], c2[1], c2[2], c2[3]);
}

// compute_covering_bbox, rasterize.py:154-198
__global__ __launch_bounds__(256) void bbox_kernel(int64_t n, const float *__restrict__ screen_means, const float *__restrict__ cov2d,
                                                   float Wf, float Hf, int64_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 c = reinterpret_cast<const float4 *>(cov2d)[i];
    float tb[4], det, spread;
    covering_bbox(screen_means[2 * i], screen_means[2 * i + 1], c.x, c.y, c.z, c.w, Wf, Hf, tb, &det, &spread);
#pragma unroll
    for (int j = 0; j < 4; ++j) out[4 * i + j] = (int64_t)tb[j];
}

// rasterize_gaussian, rasterize.py:255-305: ONE gaussian blended in place into screen [W,H,3] / opacity_buffer [W,H]
// (x-major, Q9) over its pixel rect bboxes[g] = {x_min, y_min, x_max, y_max} (end-exclusive, Q1).  The rect lives in
// device memory, so a fixed grid strides over it.  Reference operation order, libm-grade expf.
__global__ __launch_bounds__(256) void rasterize_gaussian_kernel(int64_t g, const int64_t *__restrict__ bboxes, float *__restrict__ screen,
                                                                 const float *__restrict__ screen_means, const float *__restrict__ sigmas,
                                                                 const float *__restrict__ rgb, float *__restrict__ opacity_buffer,
                                                                 const float *__restrict__ opacity, int W, int H)
{
    const int64_t x0 = bboxes[4 * g], y0 = bboxes[4 * g + 1], x1 = bboxes[4 * g + 2], y1 = bboxes[4 * g + 3];
    if (x1 <= x0 || y1 <= y0 || x0 < 0 || y0 < 0 || x1 > W || y1 > H) return;
    const int64_t w = x1 - x0, h = y1 - y0, count = w * h;
    const float mx = screen_means[2 * g], my = screen_means[2 * g + 1];
    const float sx = sigmas[3 * g], sy = sigmas[3 * g + 1], sxy = sigmas[3 * g + 2];
    const float op = opacity[g], cr = rgb[3 * g], cg = rgb[3 * g + 1], cb = rgb[3 * g + 2];
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t x = x0 + k / h, y = y0 + k % h;  // meshgrid 'ij': x outer, y inner (:274)
        const float dx = mx - (float)x, dy = my - (float)y;
        const float power = -0.5f * (sx * (dx * dx) + sy * (dy * dy)) - (sxy * dx) * dy;  // :279-283
        float alpha = op * expf(power);                                                 // :285-288
        alpha = alpha > GSR_MAX_ALPHA ? GSR_MAX_ALPHA : alpha;
        if (!(alpha > GSR_MIN_ALPHA && power <= 0.0f)) continue;                        // :291
        const int64_t pix = x * H + y;
        const float T = opacity_buffer[pix];
        screen[3 * pix] = screen[3 * pix] + (alpha * cr) * T;                            // :295-297
        screen[3 * pix + 1] = screen[3 * pix + 1] + (alpha * cg) * T;
        screen[3 * pix + 2] = screen[3 * pix + 2] + (alpha * cb) * T;
        opacity_buffer[pix] = T * (1.0f - alpha);                                       // :301-303
    }
}

static unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256); }

int launch_sh_to_rgb(int64_t n, const float *means, const float *sh, const float cc[3], int degree, float *rgb, hipStream_t s)
{
    if (n <= 0) return GSR_OK;
    hipLaunchKernelGGL(sh_to_rgb_kernel, dim3(grid_for(n)), dim3(256), 0, s, n, means, sh, cc[0], cc[1], cc[2], degree, rgb);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

int launch_cov3d(int64_t n, const float *log_scales, const float *quats, float *out, hipStream_t s)
{
    if (n <= 0) return GSR_OK;
    hipLaunchKernelGGL(cov3d_kernel, dim3(grid_for(n)), dim3(256), 0, s, n, log_scales, quats, out);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

int launch_project(int64_t n, const float *means, const float w2c[16], float *out, hipStream_t s)
{
    if (n <= 0) return GSR_OK;
    Mat16 V;
    for (int j = 0; j < 16; ++j) V.m[j] = w2c[j];
    hipLaunchKernelGGL(project_kernel, dim3(grid_for(n)), dim3(256), 0, s, n, means, V, out);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

int launch_cov2d(int64_t n, const float *cov3d, const float *cam_means, const float w2c[16], float fx, float fy, float limx, float limy,
                 float *out, hipStream_t s)
{
    if (n <= 0) return GSR_OK;
    Mat16 V;
    for (int j = 0; j < 16; ++j) V.m[j] = w2c[j];
    hipLaunchKernelGGL(cov2d_kernel, dim3(grid_for(n)), dim3(256), 0, s, n, cov3d, cam_means, V, fx, fy, limx, limy, out);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

int launch_bbox(int64_t n, const float *screen_means, const float *cov2d, float W, float H, int64_t *out, hipStream_t s)
{
    if (n <= 0) return GSR_OK;
    hipLaunchKernelGGL(bbox_kernel, dim3(grid_for(n)), dim3(256), 0, s, n, screen_means, cov2d, W, H, out);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

int launch_rasterize_gaussian(int64_t g, const int64_t *bboxes, float *screen, const float *screen_means, const float *sigmas,
                              const float *rgb, float *opacity_buffer, const float *opacity, int W, int H, hipStream_t s)
{
    hipLaunchKernelGGL(rasterize_gaussian_kernel, dim3(256), dim3(256), 0, s, g, bboxes, screen, screen_means, sigmas, rgb,
                       opacity_buffer, opacity, W, H);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

}  // namespace gsr
