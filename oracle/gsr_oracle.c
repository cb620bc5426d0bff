/*
 * gsr_oracle.c — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's forward rasterization path
 * (arnaudstiegler/torch-gaussian-splatting-rasterizer: rasterize.py,
 * spherical_harmonics.py), one function per reference stage, each citing the
 * reference file:line it follows.  All arithmetic is IEEE fp32 in the reference's
 * operation order (build with -ffp-contract=off, no fast-math).
 *
 * Parity status: PINNED — checked against golden vectors produced by running the
 * unmodified reference in the build container (tools/make_golden.py ->
 * tests/golden/f1..f4, tests/test_oracle_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker.  The product path (libgsr.so) never
 * links, loads or calls it.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* rasterize.py:29-38 */
#define Z_FAR 100.0
#define Z_NEAR 0.01
#define GAUSSIAN_SPREAD 3.0f
#define BLOCK_SIZE 16
#define MAX_GAUSSIAN_DENSITY 0.99f
#define MIN_ALPHA ((float)(1.0 / 255.0))

/* Camera block shared with include/gsr.h (same field order; see GsrCamera). */
typedef struct {
    float w2c[16];       /* Mᵀ, row-vector convention: x_cam = x_w @ w2c[:3,:3] + w2c[3,:3]  (rasterize.py:361) */
    float full_proj[16]; /* w2c @ Pᵀ (rasterize.py:364) */
    float cam_center[3]; /* inverse(w2c)[3,:3] (spherical_harmonics.py:35) */
    float focal_x, focal_y;       /* focal used by the EWA Jacobian = full-res fx,fy / 2 (rasterize.py:216,336-337; Q3) */
    float lim_x, lim_y;           /* fp32(1.3 * tan(fov/2)), rasterize.py:210-211 */
    float tan_fov_x, tan_fov_y;   /* rasterize.py:344-345 (informational) */
    int32_t width, height;        /* rasterize.py:338 */
} OracleCamera;

/* ------------------------------------------------------------------------------------------------
 * Camera set-up: rasterize.py:41-77 (quat -> R, M), :123-151 (P), :342-345 (fov), :361-364 (Mᵀ, Pᵀ, Mᵀ·Pᵀ),
 * spherical_harmonics.py:35 (camera centre).  qvec/tvec are float64 as COLMAP stores them; the
 * reference evaluates the quaternion formula in float64 and casts to fp32 (`.float()` at :56, the
 * fp32 matrix at :69-75).
 * ---------------------------------------------------------------------------------------------- */
void gsr_oracle_camera(const double qvec[4], const double tvec[3], double fx_full, double fy_full,
                       int64_t cam_width, int64_t cam_height, int32_t width, int32_t height, OracleCamera *cam)
{
    const double w = qvec[0], x = qvec[1], y = qvec[2], z = qvec[3]; /* used un-normalised (Q8) */
    double R[3][3];
    R[0][0] = 1 - 2 * (y * y) - 2 * (z * z); R[0][1] = 2 * x * y - 2 * z * w;         R[0][2] = 2 * x * z + 2 * y * w;
    R[1][0] = 2 * x * y + 2 * z * w;         R[1][1] = 1 - 2 * (x * x) - 2 * (z * z); R[1][2] = 2 * y * z - 2 * x * w;
    R[2][0] = 2 * x * z - 2 * y * w;         R[2][1] = 2 * y * z + 2 * x * w;         R[2][2] = 1 - 2 * (x * x) - 2 * (y * y);
    float M[4][4];
    memset(M, 0, sizeof M);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) M[i][j] = (float)R[i][j];
        M[i][3] = (float)tvec[i]; /* +tvec (Q8) */
    }
    M[3][3] = 1.0f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) cam->w2c[4 * i + j] = M[j][i]; /* .transpose(0,1) at :361 */

    /* fov from the FULL-RES camera (rasterize.py:342-345) */
    const double fov_x = 2.0 * atan((double)cam_width / (2.0 * fx_full));
    const double fov_y = 2.0 * atan((double)cam_height / (2.0 * fy_full));
    cam->tan_fov_x = (float)tan(fov_x * 0.5);
    cam->tan_fov_y = (float)tan(fov_y * 0.5);
    /* 1.3*tan is formed in float64 and then becomes an fp32 tensor (rasterize.py:210-211) */
    cam->lim_x = (float)(1.3 * tan(fov_x * 0.5));
    cam->lim_y = (float)(1.3 * tan(fov_y * 0.5));
    /* get_projection_matrix, rasterize.py:123-151 (python float64 arithmetic, stored into an fp32 tensor) */
    const double thx = tan(fov_x / 2), thy = tan(fov_y / 2);
    const double top = thy * Z_NEAR, bottom = -top, right = thx * Z_NEAR, left = -right;
    float P[4][4];
    memset(P, 0, sizeof P);
    P[0][0] = (float)(2.0 * Z_NEAR / (right - left));
    P[1][1] = (float)(2.0 * Z_NEAR / (top - bottom));
    P[0][2] = (float)((right + left) / (right - left));
    P[1][2] = (float)((top + bottom) / (top - bottom));
    P[3][2] = 1.0f;
    P[2][2] = (float)(1.0 * Z_FAR / (Z_FAR - Z_NEAR));
    P[2][3] = (float)(-(Z_FAR * Z_NEAR) / (Z_FAR - Z_NEAR));
    /* full = w2c @ Pᵀ in fp32 (rasterize.py:364) */
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) acc = acc + cam->w2c[4 * i + k] * P[j][k];
            cam->full_proj[4 * i + j] = acc;
        }
    /* camera centre = inverse(w2c)[3,:3] = -t·R⁻¹ in row-vector form; R need not be orthonormal
     * because qvec is not normalised, so invert the 3x3 properly (float64, cast once). */
    double A[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = (double)cam->w2c[4 * i + j];
    const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                       A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    double inv[3][3];
    inv[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det; inv[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det; inv[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
    inv[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det; inv[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det; inv[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
    inv[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det; inv[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det; inv[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
    for (int j = 0; j < 3; ++j) {
        double c = 0.0;
        for (int k = 0; k < 3; ++k) c -= (double)cam->w2c[12 + k] * inv[k][j];
        cam->cam_center[j] = (float)c;
    }
    cam->focal_x = (float)(fx_full / 2.0); /* `focals / 2`, rasterize.py:216 */
    cam->focal_y = (float)(fy_full / 2.0);
    cam->width = width;
    cam->height = height;
}

/* rasterize.py:41-56 on fp32 inputs (called from :113 with the normalised ply quaternion) */
static void quat_to_rot_f32(float w, float x, float y, float z, float R[3][3])
{
    R[0][0] = 1.0f - 2.0f * (y * y) - 2.0f * (z * z); R[0][1] = 2.0f * x * y - 2.0f * z * w;           R[0][2] = 2.0f * x * z + 2.0f * y * w;
    R[1][0] = 2.0f * x * y + 2.0f * z * w;           R[1][1] = 1.0f - 2.0f * (x * x) - 2.0f * (z * z); R[1][2] = 2.0f * y * z - 2.0f * x * w;
    R[2][0] = 2.0f * x * z - 2.0f * y * w;           R[2][1] = 2.0f * y * z + 2.0f * x * w;           R[2][2] = 1.0f - 2.0f * (x * x) - 2.0f * (y * y);
}

/* get_covariance_matrix_from_mesh, rasterize.py:89-120: Σ = (R S)(R S)ᵀ, q normalised (eps 1e-12), S = diag(exp(scale)) */
static void cov3d_from_scale_rot(const float ls[3], const float q[4], float cov[3][3])
{
    float s[3] = {expf(ls[0]), expf(ls[1]), expf(ls[2])};
    float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (nrm < 1e-12f) nrm = 1e-12f;
    float R[3][3], M[3][3];
    quat_to_rot_f32(q[0] / nrm, q[1] / nrm, q[2] / nrm, q[3] / nrm, R);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i][j] = R[i][j] * s[j];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) cov[i][j] = M[i][0] * M[j][0] + M[i][1] * M[j][1] + M[i][2] * M[j][2];
}

/* spherical_harmonics.py:4-24 */
static const double SH_0 = 0.28209479177387814, SH_C1 = 0.4886025119029199;
static const double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396};
static const double SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                                -0.4570457994644658, 1.445305721320277, -0.5900435899266435};

/* sh_to_rgb, spherical_harmonics.py:27-73.  sh is [16][3] for one gaussian. */
static void sh_to_rgb_one(const float p[3], const float *sh, const float cc[3], int degree, float rgb[3])
{
    float d[3] = {p[0] - cc[0], p[1] - cc[1], p[2] - cc[2]};
    float n = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const float x = d[0] / n, y = d[1] / n, z = d[2] / n;
    const float c0 = (float)SH_0, c1 = (float)SH_C1, nc1 = (float)(-SH_C1);
    float k2[5], k3[7];
    for (int i = 0; i < 5; ++i) k2[i] = (float)SH_C2[i];
    for (int i = 0; i < 7; ++i) k3[i] = (float)SH_C3[i];
    for (int c = 0; c < 3; ++c) {
#define SHC(k) sh[(k) * 3 + c]
        float col = SHC(0) * c0;
        if (degree > 0) { /* :45-46 */
            col = col + (((nc1 * y) * SHC(1) + (c1 * z) * SHC(2)) - (c1 * x) * SHC(3));
            if (degree > 1) { /* :48-55 */
                const float t4 = ((k2[0] * x) * y) * SHC(4);
                const float t5 = ((k2[1] * y) * z) * SHC(5);
                const float t6 = (k2[2] * (((2.0f * z) * z - x * x) - y * y)) * SHC(6);
                const float t7 = ((k2[3] * x) * z) * SHC(7);
                const float t8 = (k2[4] * (x * x - y * y)) * SHC(8);
                col = col + ((((t4 + t5) + t6) + t7) + t8);
                if (degree > 2) { /* :56-65 */
                    const float t9 = ((k3[0] * y) * ((3.0f * x) * x - y * y)) * SHC(9);
                    const float t10 = (((k3[1] * x) * y) * z) * SHC(10);
                    const float t11 = ((k3[2] * y) * (((4.0f * z) * z - x * x) - y * y)) * SHC(11);
                    const float t12 = ((k3[3] * z) * (((2.0f * z) * z - (3.0f * x) * x) - (3.0f * y) * y)) * SHC(12);
                    const float t13 = ((k3[4] * x) * (((4.0f * z) * z - x * x) - y * y)) * SHC(13);
                    const float t14 = ((k3[5] * z) * (x * x - y * y)) * SHC(14);
                    const float t15 = ((k3[6] * x) * (x * x - (3.0f * y) * y)) * SHC(15);
                    col = col + ((((((t9 + t10) + t11) + t12) + t13) + t14) + t15);
                }
            }
        }
#undef SHC
        col = col + 0.5f;                     /* :69 */
        col = col < 0.0f ? 0.0f : (col > 1.0f ? 1.0f : col); /* :71, both sides (Q7) */
        rgb[c] = col;
    }
}

static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int64_t clampi(int64_t v, int64_t lo, int64_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* ------------------------------------------------------------------------------------------------
 * Per-gaussian preprocessing: everything rasterize.py:354-420 computes before the loop.
 * Inputs: means[n,3], log_scales[n,3], quats[n,4] (w,x,y,z raw), opacity_logit[n], sh[n,16,3].
 * Any output pointer may be NULL.
 *   cov3d[n,9] cam_means[n,3] cov2d[n,4] (before cull zeroing, as compute_2d_covariance returns it)
 *   screen_means[n,2] tile_bboxes[n,4]i64 sigmas[n,3] pixel_bboxes[n,4]i64 rgb[n,3] opacity[n]
 * ---------------------------------------------------------------------------------------------- */
void gsr_oracle_preprocess(int64_t n, const float *means, const float *log_scales, const float *quats,
                           const float *opacity_logit, const float *sh, int sh_degree, const OracleCamera *cam,
                           float *cov3d, float *cam_means, float *cov2d, float *screen_means, int64_t *tile_bboxes,
                           float *sigmas, int64_t *pixel_bboxes, float *rgb, float *opacity)
{
    const float *V = cam->w2c, *F = cam->full_proj;
    const float Wf = (float)cam->width, Hf = (float)cam->height;
    const float limx = cam->lim_x, limy = cam->lim_y; /* rasterize.py:210-211 */
    const float fx = cam->focal_x, fy = cam->focal_y; /* :216 (Q3) */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float *p = means + 3 * i;
        float C3[3][3];
        cov3d_from_scale_rot(log_scales + 3 * i, quats + 4 * i, C3); /* :357 */
        if (cov3d) memcpy(cov3d + 9 * i, C3, sizeof C3);
        const float op = 1.0f / (1.0f + expf(-opacity_logit[i])); /* sigmoid, :358 */
        if (opacity) opacity[i] = op;
        if (rgb) sh_to_rgb_one(p, sh + 48 * i, cam->cam_center, sh_degree, rgb + 3 * i); /* :368 */
        /* project_to_camera_space, :80-86 */
        float cm[3];
        for (int j = 0; j < 3; ++j) cm[j] = ((p[0] * V[0 + j] + p[1] * V[4 + j]) + p[2] * V[8 + j]) + V[12 + j];
        if (cam_means) memcpy(cam_means + 3 * i, cm, sizeof cm);
        /* clip-space point, :374 */
        float pt[4];
        for (int j = 0; j < 4; ++j) pt[j] = ((p[0] * F[0 + j] + p[1] * F[4 + j]) + p[2] * F[8 + j]) + F[12 + j];
        const int culled = cm[2] < 0.2f; /* :377 */
        if (culled) pt[0] = pt[1] = pt[2] = pt[3] = 0.0f; /* :378 */
        const float p_w = 1.0f / (pt[3] + 0.0000001f); /* :381 */
        const float ndc_x = pt[0] * p_w, ndc_y = pt[1] * p_w; /* :382 */
        /* compute_2d_covariance, :201-252 */
        const float tz = cm[2];
        const float txtz = cm[0] / tz, tytz = cm[1] / tz;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
        float J[3][3] = {{0}};
        J[0][0] = fx / tz;
        J[0][2] = -(fx * tx) / (tz * tz);
        J[1][1] = fy / tz;
        J[1][2] = -(fy * ty) / (tz * tz);
        /* W = w2c[:3,:3].T (:230); T = (Wᵀ Jᵀ)ᵀ = J·W (:232) */
        float T[3][3], TV[3][3], PC[3][3];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                /* element [c][r] of Wᵀ Jᵀ = sum_k Wᵀ[c][k] Jᵀ[k][r] = sum_k w2c[c][k]... with Wᵀ = w2c[:3,:3] */
                T[r][c] = (V[4 * c + 0] * J[r][0] + V[4 * c + 1] * J[r][1]) + V[4 * c + 2] * J[r][2];
            }
        float vrk[3][3] = {{C3[0][0], C3[0][1], C3[0][2]}, {C3[0][1], C3[1][1], C3[1][2]}, {C3[0][2], C3[1][2], C3[2][2]}}; /* :234-243 */
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) TV[r][c] = (T[r][0] * vrk[0][c] + T[r][1] * vrk[1][c]) + T[r][2] * vrk[2][c];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) PC[r][c] = (TV[r][0] * T[c][0] + TV[r][1] * T[c][1]) + TV[r][2] * T[c][2]; /* :245 */
        PC[0][0] += 0.3f; /* :249 */
        PC[1][1] += 0.3f; /* :250 */
        if (cov2d) { cov2d[4 * i + 0] = PC[0][0]; cov2d[4 * i + 1] = PC[0][1]; cov2d[4 * i + 2] = PC[1][0]; cov2d[4 * i + 3] = PC[1][1]; }
        float a = PC[0][0], b01 = PC[0][1], b10 = PC[1][0], c = PC[1][1];
        if (culled) a = b01 = b10 = c = 0.0f; /* :388 */
        /* NDC -> pixel, :391 */
        const float mx = ((ndc_x + 1.0f) * Wf - 1.0f) / 2.0f, my = ((ndc_y + 1.0f) * Hf - 1.0f) / 2.0f;
        if (screen_means) { screen_means[2 * i] = mx; screen_means[2 * i + 1] = my; }
        /* compute_covering_bbox, :154-198 */
        const float det = a * c - b10 * b01;
        const float trace = a + c;
        const float disc = sqrtf(fmaxf((trace * trace) / 4.0f - det, 0.1f));
        const float l1 = trace / 2.0f + disc, l2 = trace / 2.0f - disc;
        const float spread = ceilf(GAUSSIAN_SPREAD * sqrtf(fmaxf(l1, l2)));
        const float tb0 = floorf(clampf((mx - spread) / (float)BLOCK_SIZE, 0.0f, Wf - 1.0f));
        const float tb1 = floorf(clampf((my - spread) / (float)BLOCK_SIZE, 0.0f, Hf - 1.0f));
        const float tb2 = floorf(clampf((mx + (spread + (float)BLOCK_SIZE - 1.0f)) / (float)BLOCK_SIZE, 0.0f, Wf - 1.0f));
        const float tb3 = floorf(clampf((my + (spread + (float)BLOCK_SIZE - 1.0f)) / (float)BLOCK_SIZE, 0.0f, Hf - 1.0f));
        const int64_t t0 = (int64_t)tb0, t1 = (int64_t)tb1, t2 = (int64_t)tb2, t3 = (int64_t)tb3;
        if (tile_bboxes) { tile_bboxes[4 * i] = t0; tile_bboxes[4 * i + 1] = t1; tile_bboxes[4 * i + 2] = t2; tile_bboxes[4 * i + 3] = t3; }
        /* conic, :395-411 (det again, same formula) */
        const float det_inv = det == 0.0f ? 0.0f : 1.0f / det;
        if (sigmas) { sigmas[3 * i] = c * det_inv; sigmas[3 * i + 1] = a * det_inv; sigmas[3 * i + 2] = (-b01) * det_inv; }
        /* pixel rects, :415-419 */
        if (pixel_bboxes) {
            pixel_bboxes[4 * i + 0] = clampi(t0 * BLOCK_SIZE, 0, cam->width - 1);
            pixel_bboxes[4 * i + 1] = clampi(t1 * BLOCK_SIZE, 0, cam->height - 1);
            pixel_bboxes[4 * i + 2] = clampi(t2 * BLOCK_SIZE, 0, cam->width - 1);
            pixel_bboxes[4 * i + 3] = clampi(t3 * BLOCK_SIZE, 0, cam->height - 1);
        }
    }
}

/* depth order, rasterize.py:424-425: argsort(z_cam) ascending over ALL n.  The reference's sort is
 * unstable (ties unordered); the oracle (and the HIP path) break ties by gaussian index. */
typedef struct { float z; int64_t i; } DepthKey;
static int depth_cmp(const void *pa, const void *pb)
{
    const DepthKey *a = (const DepthKey *)pa, *b = (const DepthKey *)pb;
    if (a->z < b->z) return -1;
    if (a->z > b->z) return 1;
    return (a->i > b->i) - (a->i < b->i);
}
void gsr_oracle_depth_order(int64_t n, const float *cam_means, int64_t *order)
{
    DepthKey *k = (DepthKey *)malloc(sizeof(DepthKey) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) { k[i].z = cam_means[3 * i + 2]; k[i].i = i; }
    qsort(k, (size_t)n, sizeof(DepthKey), depth_cmp);
    for (int64_t i = 0; i < n; ++i) order[i] = k[i].i;
    free(k);
}

/* ------------------------------------------------------------------------------------------------
 * The per-gaussian compositing loop: rasterize.py:436-446 (driver + skip guard) and :255-305
 * (rasterize_gaussian).  screen is [W,H,3] x-major, transmittance [W,H] (Q9); caller zero/one-fills.
 * Rows of the x range [xlo,xhi) only are touched, so disjoint x bands can run on different threads
 * without changing any pixel's blend order.  Returns the number of gaussians drawn (not skipped).
 * `limit` < 0 = all; otherwise stop after `limit` drawn gaussians (progressive render, :448-450).
 * ---------------------------------------------------------------------------------------------- */
static int64_t composite_band(int64_t n, const int64_t *order, const int64_t *pixel_bboxes, const float *screen_means,
                              const float *sigmas, const float *rgb, const float *opacity, int32_t H, int64_t xlo,
                              int64_t xhi, int64_t limit, float *screen, float *transmittance)
{
    int64_t drawn = 0;
    for (int64_t k = 0; k < n; ++k) {
        const int64_t g = order[k];
        const int64_t *bb = pixel_bboxes + 4 * g;
        const int64_t area = (bb[2] - bb[0]) * (bb[3] - bb[1]);
        const float sx = sigmas[3 * g], sy = sigmas[3 * g + 1], sxy = sigmas[3 * g + 2];
        if (area == 0 || sx == 0.0f || sy == 0.0f || sxy == 0.0f) continue; /* :441 (Q2) */
        if (limit >= 0 && drawn >= limit) break;
        ++drawn;
        const int64_t x0 = bb[0] > xlo ? bb[0] : xlo, x1 = bb[2] < xhi ? bb[2] : xhi;
        const float mx = screen_means[2 * g], my = screen_means[2 * g + 1], op = opacity[g];
        const float *col = rgb + 3 * g;
        for (int64_t x = x0; x < x1; ++x) {           /* arange is end-exclusive (Q1), :271-272 */
            const float dx = mx - (float)x;
            const float ex = sx * (dx * dx);
            for (int64_t y = bb[1]; y < bb[3]; ++y) {
                const float dy = my - (float)y;
                const float power = -0.5f * (ex + sy * (dy * dy)) - (sxy * dx) * dy; /* :279-283 */
                float alpha = op * expf(power);                                     /* :285-288 */
                if (alpha > MAX_GAUSSIAN_DENSITY) alpha = MAX_GAUSSIAN_DENSITY;
                if (!(alpha > MIN_ALPHA && power <= 0.0f)) continue;                /* :291 (Q6) */
                const int64_t pix = x * (int64_t)H + y;
                const float T = transmittance[pix];
                float *s = screen + 3 * pix;
                s[0] = s[0] + (alpha * col[0]) * T;                                 /* :295-297 */
                s[1] = s[1] + (alpha * col[1]) * T;
                s[2] = s[2] + (alpha * col[2]) * T;
                transmittance[pix] = T * (1.0f - alpha);                            /* :301-303 */
            }
        }
    }
    return drawn;
}

int64_t gsr_oracle_composite(int64_t n, const int64_t *order, const int64_t *pixel_bboxes, const float *screen_means,
                             const float *sigmas, const float *rgb, const float *opacity, int32_t W, int32_t H,
                             int64_t limit, int threads, float *screen, float *transmittance)
{
    if (threads <= 1)
        return composite_band(n, order, pixel_bboxes, screen_means, sigmas, rgb, opacity, H, 0, W, limit, screen, transmittance);
    int64_t drawn = 0;
    /* narrow interleaved bands balance the load; every band walks the whole depth order */
    const int64_t band = 8;
    const int64_t nbands = (W + band - 1) / band;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int64_t b = 0; b < nbands; ++b) {
        const int64_t lo = b * band, hi = (lo + band < W) ? lo + band : W;
        const int64_t d = composite_band(n, order, pixel_bboxes, screen_means, sigmas, rgb, opacity, H, lo, hi, limit, screen, transmittance);
        if (b == 0) drawn = d;
    }
    return drawn;
}

/* Whole frame: returns the viewable image [H,W,3] = screen.transpose(1,0) (rasterize.py:471, Q9) and,
 * optionally, the final transmittance [H,W].  Returns number of gaussians drawn, <0 on allocation failure. */
int64_t gsr_oracle_render(int64_t n, const float *means, const float *log_scales, const float *quats,
                          const float *opacity_logit, const float *sh, int sh_degree, const OracleCamera *cam,
                          int threads, float *image_hw3, float *final_T_hw)
{
    const int32_t W = cam->width, H = cam->height;
    const size_t nn = (size_t)(n > 0 ? n : 1);
    float *cm = malloc(nn * 3 * sizeof(float)), *sm = malloc(nn * 2 * sizeof(float)), *sg = malloc(nn * 3 * sizeof(float));
    float *col = malloc(nn * 3 * sizeof(float)), *op = malloc(nn * sizeof(float));
    int64_t *bb = malloc(nn * 4 * sizeof(int64_t)), *ord = malloc(nn * sizeof(int64_t));
    float *screen = calloc((size_t)W * H * 3, sizeof(float)), *T = malloc((size_t)W * H * sizeof(float));
    int64_t drawn = -1;
    if (cm && sm && sg && col && op && bb && ord && screen && T) {
        for (size_t i = 0; i < (size_t)W * H; ++i) T[i] = 1.0f; /* :437-438 */
        gsr_oracle_preprocess(n, means, log_scales, quats, opacity_logit, sh, sh_degree, cam,
                              NULL, cm, NULL, sm, NULL, sg, bb, col, op);
        gsr_oracle_depth_order(n, cm, ord);
        drawn = gsr_oracle_composite(n, ord, bb, sm, sg, col, op, W, H, -1, threads, screen, T);
        for (int32_t y = 0; y < H; ++y)
            for (int32_t x = 0; x < W; ++x) {
                const size_t src = (size_t)x * H + y, dst = (size_t)y * W + x;
                image_hw3[3 * dst] = screen[3 * src]; image_hw3[3 * dst + 1] = screen[3 * src + 1]; image_hw3[3 * dst + 2] = screen[3 * src + 2];
                if (final_T_hw) final_T_hw[dst] = T[src];
            }
    }
    free(cm); free(sm); free(sg); free(col); free(op); free(bb); free(ord); free(screen); free(T);
    return drawn;
}

int gsr_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
