// api.hip — the extern "C" surface of libgsr.so (include/gsr.h) and its host-side logic.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstddef>
#include <cstring>

#include "gsr_internal.h"

namespace gsr {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return GSR_ERR_HIP;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t carve_workspace(void *base, int64_t n, int width, int height, int64_t max_pairs, Workspace *ws)
{
    char *p = static_cast<char *>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) -> void * {
        void *r = p ? p + off : nullptr;
        off = align_up(off + bytes, 256);
        return r;
    };
    const size_t nn = (size_t)std::max<int64_t>(n, 1), np = (size_t)std::max<int64_t>(max_pairs, 1);
    ws->n = n;
    ws->max_pairs = max_pairs;
    ws->tiles_x = (width + GSR_TILE - 1) / GSR_TILE;
    ws->tiles_y = (height + GSR_TILE - 1) / GSR_TILE;
    ws->ctiles_x = (ws->tiles_x + 1) / 2;
    ws->ctiles_y = (ws->tiles_y + 1) / 2;
    // row stride of the histogram table: tiles of the depth sort, tiles of the pair sort
    ws->hist_blocks = (int)std::max((nn + DEPTH_SORT_THREADS * DEPTH_SORT_ITEMS_SHARD - 1) / (DEPTH_SORT_THREADS * DEPTH_SORT_ITEMS_SHARD),
                                    (np + PAIR_SORT_THREADS * PAIR_SORT_ITEMS - 1) / (PAIR_SORT_THREADS * PAIR_SORT_ITEMS));
    ws->ctrl = static_cast<FrameCtrl *>(take(sizeof(FrameCtrl)));
    ws->rec = static_cast<GaussRec *>(take(sizeof(GaussRec) * nn));
    ws->rect = static_cast<ushort4 *>(take(sizeof(ushort4) * nn));
    for (int b = 0; b < 2; ++b) ws->key[b] = static_cast<uint32_t *>(take(4 * nn));
    for (int b = 0; b < 2; ++b) ws->val[b] = static_cast<uint32_t *>(take(4 * nn));
    for (int b = 0; b < 2; ++b) ws->rect8[b] = static_cast<uint32_t *>(take(4 * nn));
    ws->blk_dead = static_cast<unsigned char *>(take((nn + GSR_BOUNDS_BLOCK - 1) / GSR_BOUNDS_BLOCK));  // (with the per-gaussian arrays: stage 1 carves with max_pairs = 0)
    ws->blk_sum = static_cast<uint32_t *>(
        take(4 * ((std::max(nn, (size_t)ws->tiles_x * ws->tiles_y) + EMIT_THREADS - 1) / EMIT_THREADS + 1)));
    ws->hist = static_cast<uint32_t *>(take(4 * 512 * (size_t)ws->hist_blocks));
    for (int b = 0; b < 2; ++b) ws->pkey[b] = static_cast<uint32_t *>(take(4 * np));
    for (int b = 0; b < 2; ++b) ws->pval[b] = static_cast<uint32_t *>(take(4 * np));
    ws->ranges = static_cast<uint2 *>(take(sizeof(uint2) * (size_t)ws->tiles_x * ws->tiles_y));
    ws->cranges = static_cast<uint2 *>(take(sizeof(uint2) * (size_t)ws->ctiles_x * ws->ctiles_y));
    const size_t order_slots = 8 * (size_t)((ws->tiles_y + 7) / 8) * ws->tiles_x;
    ws->tile_order = static_cast<int *>(take(sizeof(int) * order_slots));
    ws->blend_stats = static_cast<uint32_t *>(take(sizeof(uint32_t) * BLEND_STAT_WORDS * order_slots));
    ws->tile_work = static_cast<uint32_t *>(take(sizeof(uint32_t) * (size_t)ws->tiles_x * ws->tiles_y));

    ws->pair_off = nullptr;
    ws->bytes = off;
    return off;
}

static int check_frame(int64_t n, const GsrCamera *cam, const GsrOptions *opts, int64_t max_pairs, void *workspace,
                       size_t workspace_bytes, Workspace *ws)
{
    if (!cam || !opts || !workspace) { set_error("null camera/options/workspace"); return GSR_ERR_BAD_ARG; }
    if (n < 0 || n > 0x7FFFFFFF) { set_error("n = %lld out of range", (long long)n); return GSR_ERR_BAD_ARG; }
    if (cam->width <= 0 || cam->height <= 0 || cam->width > 65535 * GSR_TILE || cam->height > 65535 * GSR_TILE) {
        set_error("bad frame size %dx%d", cam->width, cam->height); return GSR_ERR_BAD_ARG;
    }
    // the sort kernels index pairs with 32-bit arithmetic, one 4096-pair tile past the last pair at most
    if (max_pairs < 0 || max_pairs > GSR_MAX_PAIRS) { set_error("max_pairs = %lld out of range [0, %lld]", (long long)max_pairs, (long long)GSR_MAX_PAIRS); return GSR_ERR_BAD_ARG; }
    if (opts->tile_row_step < 0 || opts->tile_row_begin < 0 || opts->tile_row_begin >= std::max(opts->tile_row_step, 1)) {
        set_error("bad tile-row shard %d/%d", opts->tile_row_begin, opts->tile_row_step); return GSR_ERR_BAD_ARG;
    }
    if (opts->tile_row_block < 0 || opts->tile_row_block > 2) { set_error("bad tile_row_block %d (0 / 1: single rows, 2: pairs)", opts->tile_row_block); return GSR_ERR_BAD_ARG; }
    if (opts->draw_limit < 0) { set_error("bad draw_limit %d", opts->draw_limit); return GSR_ERR_BAD_ARG; }
    if (opts->output_dtype != 0 && opts->output_dtype != 1) { set_error("bad output_dtype %d", opts->output_dtype); return GSR_ERR_BAD_ARG; }
    if (opts->blend_impl < 0 || opts->blend_impl > 1) { set_error("bad blend_impl %d", opts->blend_impl); return GSR_ERR_BAD_ARG; }
    if (opts->output_layout < 0 || opts->output_layout > 2) { set_error("bad output_layout %d", opts->output_layout); return GSR_ERR_BAD_ARG; }
    if (opts->depth_sort_passes < 0 || opts->depth_sort_passes > 4) { set_error("bad depth_sort_passes %d", opts->depth_sort_passes); return GSR_ERR_BAD_ARG; }
    if (opts->accum_dtype != 0 && opts->accum_dtype != 1) { set_error("bad accum_dtype %d", opts->accum_dtype); return GSR_ERR_BAD_ARG; }
    if (opts->keep_flags != 0 && opts->keep_flags != 1) { set_error("bad keep_flags %d", opts->keep_flags); return GSR_ERR_BAD_ARG; }
    if (opts->saturation_rule != 0 && opts->saturation_rule != 1) { set_error("bad saturation_rule %d", opts->saturation_rule); return GSR_ERR_BAD_ARG; }
    if (opts->fine_binning != 0 && opts->fine_binning != 1) { set_error("bad fine_binning %d", opts->fine_binning); return GSR_ERR_BAD_ARG; }
    if (opts->shard_preprocess < 0 || opts->shard_preprocess > 2) { set_error("bad shard_preprocess %d", opts->shard_preprocess); return GSR_ERR_BAD_ARG; }
    if (opts->blend_pipe_tiles < -1) { set_error("bad blend_pipe_tiles %d", opts->blend_pipe_tiles); return GSR_ERR_BAD_ARG; }
    if (opts->no_order_hint != 0 && opts->no_order_hint != 1) { set_error("bad no_order_hint %d", opts->no_order_hint); return GSR_ERR_BAD_ARG; }
    if (opts->colour_stage != 0 && opts->colour_stage != 1) { set_error("bad colour_stage %d", opts->colour_stage); return GSR_ERR_BAD_ARG; }
    if (opts->sh_dense_min < 0 || opts->sh_dense_min > 65) { set_error("bad sh_dense_min %d", opts->sh_dense_min); return GSR_ERR_BAD_ARG; }
    if (opts->batch_views < 0 || opts->batch_views > MAX_VIEWS) { set_error("bad batch_views %d (0 .. %d)", opts->batch_views, MAX_VIEWS); return GSR_ERR_BAD_ARG; }
    if (reinterpret_cast<uintptr_t>(workspace) % 256 != 0) { set_error("workspace must be 256-byte aligned"); return GSR_ERR_BAD_ARG; }
    const size_t need = carve_workspace(workspace, n, cam->width, cam->height, max_pairs, ws);
    if (workspace_bytes < need) {
        set_error("workspace too small: %zu bytes given, %zu needed", workspace_bytes, need);
        return GSR_ERR_WORKSPACE;
    }
    return GSR_OK;
}

}  // namespace gsr

using namespace gsr;

extern "C" {

int gsr_version(void) { return GSR_VERSION; }

const char *gsr_last_error(void) { return g_err; }

void gsr_default_options(GsrOptions *o)
{
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->reference_compat = 1;
    o->early_out_T = 0.0f;
    o->tile_row_begin = 0;
    o->tile_row_step = 1;
    o->output_layout = 0;
}

int gsr_camera_setup(const double qvec[4], const double tvec[3], double fx_full, double fy_full, int64_t cam_width,
                     int64_t cam_height, int32_t width, int32_t height, GsrCamera *cam)
{
    if (!qvec || !tvec || !cam) { set_error("null argument"); return GSR_ERR_BAD_ARG; }
    if (!(fx_full > 0.0) || !(fy_full > 0.0) || cam_width <= 0 || cam_height <= 0 || width <= 0 || height <= 0) {
        set_error("bad intrinsics"); return GSR_ERR_BAD_ARG;
    }
    const double Z_FAR = 100.0, Z_NEAR = 0.01;  // rasterize.py:29-30
    // world -> camera: the quaternion formula of rasterize.py:41-56 evaluated in float64 on the COLMAP qvec
    // (un-normalised, Q8), cast to fp32 (:56), +tvec in the last column (:75), then transposed (:361).
    const double w = qvec[0], x = qvec[1], y = qvec[2], z = qvec[3];
    const double R[3][3] = {
        {1 - 2 * (y * y) - 2 * (z * z), 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w},
        {2 * x * y + 2 * z * w, 1 - 2 * (x * x) - 2 * (z * z), 2 * y * z - 2 * x * w},
        {2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * (x * x) - 2 * (y * y)}};
    float M[4][4] = {{0}};
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) M[i][j] = (float)R[i][j];
        M[i][3] = (float)tvec[i];
    }
    M[3][3] = 1.0f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) cam->w2c[4 * i + j] = M[j][i];
    // field of view from the full-resolution camera, rasterize.py:342-345
    const double fov_x = 2.0 * std::atan((double)cam_width / (2.0 * fx_full));
    const double fov_y = 2.0 * std::atan((double)cam_height / (2.0 * fy_full));
    cam->tan_fov_x = (float)std::tan(fov_x * 0.5);
    cam->tan_fov_y = (float)std::tan(fov_y * 0.5);
    cam->lim_x = (float)(1.3 * std::tan(fov_x * 0.5));  // :210
    cam->lim_y = (float)(1.3 * std::tan(fov_y * 0.5));  // :211
    cam->focal_x = (float)(fx_full / 2.0);              // :216 (Q3: always full-res / 2)
    cam->focal_y = (float)(fy_full / 2.0);
    // perspective matrix, rasterize.py:123-151 (float64 python arithmetic stored into an fp32 tensor)
    const double thx = std::tan(fov_x / 2), thy = std::tan(fov_y / 2);
    const double top = thy * Z_NEAR, bottom = -top, right = thx * Z_NEAR, left = -right;
    float P[4][4] = {{0}};
    P[0][0] = (float)(2.0 * Z_NEAR / (right - left));
    P[1][1] = (float)(2.0 * Z_NEAR / (top - bottom));
    P[0][2] = (float)((right + left) / (right - left));
    P[1][2] = (float)((top + bottom) / (top - bottom));
    P[3][2] = 1.0f;
    P[2][2] = (float)(Z_FAR / (Z_FAR - Z_NEAR));
    P[2][3] = (float)(-(Z_FAR * Z_NEAR) / (Z_FAR - Z_NEAR));
    for (int i = 0; i < 4; ++i)  // full = w2c @ P^T in fp32, :364
        for (int j = 0; j < 4; ++j) {
            float acc = 0.0f;
            for (int k = 0; k < 4; ++k) acc = acc + cam->w2c[4 * i + k] * P[j][k];
            cam->full_proj[4 * i + j] = acc;
        }
    // camera centre = inverse(w2c)[3,:3] (spherical_harmonics.py:35) = -t A^-1, A = w2c[:3,:3]; float64, one cast
    double A[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = (double)cam->w2c[4 * i + j];
    const double c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1], c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2],
                 c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    const double det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02;
    if (det == 0.0 || !std::isfinite(det)) { set_error("singular world-to-camera rotation"); return GSR_ERR_BAD_ARG; }
    const double inv[3][3] = {
        {c00 / det, (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det, (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det},
        {c01 / det, (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det, (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det},
        {c02 / det, (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det, (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det}};
    for (int j = 0; j < 3; ++j) {
        double c = 0.0;
        for (int k = 0; k < 3; ++k) c -= (double)cam->w2c[12 + k] * inv[k][j];
        cam->cam_center[j] = (float)c;
    }
    cam->width = width;
    cam->height = height;
    return GSR_OK;
}

int gsr_workspace_bytes(int64_t n, int32_t width, int32_t height, int64_t max_pairs, size_t *bytes)
{
    if (!bytes || n < 0 || width <= 0 || height <= 0 || max_pairs < 0) { set_error("bad argument"); return GSR_ERR_BAD_ARG; }
    Workspace ws;
    *bytes = carve_workspace(nullptr, n, width, height, max_pairs, &ws);
    return GSR_OK;
}

static int check_scene(const GsrScene *sc)
{
    if (!sc) { set_error("null scene"); return GSR_ERR_BAD_ARG; }
    if (sc->n > 0 && (!sc->means || !sc->log_scales || !sc->quats || !sc->opacity_logit || !sc->sh)) {
        set_error("null scene array"); return GSR_ERR_BAD_ARG;
    }
    if (sc->sh_degree < 0 || sc->sh_degree > 3) { set_error("sh_degree %d not in 0..3", sc->sh_degree); return GSR_ERR_BAD_ARG; }
    if (sc->sh_dtype != 0 && sc->sh_dtype != 1) { set_error("sh_dtype %d: 0 (float32) or 1 (float16)", sc->sh_dtype); return GSR_ERR_BAD_ARG; }
    if (reinterpret_cast<uintptr_t>(sc->sh) % 16 != 0 || reinterpret_cast<uintptr_t>(sc->quats) % 16 != 0 ||
        reinterpret_cast<uintptr_t>(sc->block_bounds) % 16 != 0) {
        set_error("sh, quats and block_bounds must be 16-byte aligned"); return GSR_ERR_BAD_ARG;
    }
    return GSR_OK;
}

// how much of a view's control block its frame clears: a fresh workspace may hold anything, so the default clears the whole block
// (a 0xFF-filled one renders correctly); keep_flags / the later views of a batch keep the sticky record at its tail
static int reset_words_of(const GsrOptions *opts, bool keep_batch_words)
{
    static_assert(sizeof(FrameCtrl) % 4 == 0 && offsetof(FrameCtrl, batch_overflow) % 4 == 0, "FrameCtrl is cleared by words");
    return (int)((keep_batch_words || opts->keep_flags ? offsetof(FrameCtrl, batch_overflow) : sizeof(FrameCtrl)) / 4);
}

int gsr_preprocess(const GsrScene *scene, const GsrCamera *cam, const GsrOptions *opts, void *workspace,
                   size_t workspace_bytes, const GsrDebugOut *debug, void *stream)
{
    int rc = check_scene(scene);
    if (rc) return rc;
    Workspace ws;
    // stage 1 does not touch the pair buffers: any max_pairs >= 0 gives the same per-gaussian layout,
    // so size-check against the smallest one.
    rc = check_frame(scene->n, cam, opts, 0, workspace, workspace_bytes, &ws);
    if (rc) return rc;
    return launch_preprocess(*scene, cam, *opts, ws, debug, reset_words_of(opts, false), static_cast<hipStream_t>(stream));
}

// depth order: pass 0 drops culled gaussians and leaves V in ctrl; 3 passes on ordinary scenes (sort.hip); then pairs in depth
// order, stably sorted by tile -> per-tile lists and their ranges; E in ctrl
static int bin_sort_impl(const GsrOptions *opts, const Workspace &ws, hipStream_t s)
{
    const int rc = launch_depth_sort(ws, rect_fits_8bit(ws), shard_compact(*opts), opts->depth_sort_passes, s);
    if (rc) return rc;
    return launch_binning(*opts, ws, s);
}

int gsr_bin_sort(int64_t n, const GsrCamera *cam, const GsrOptions *opts, int64_t max_pairs, void *workspace,
                 size_t workspace_bytes, void *stream)
{
    Workspace ws;
    int rc = check_frame(n, cam, opts, max_pairs, workspace, workspace_bytes, &ws);
    if (rc) return rc;
    return bin_sort_impl(opts, ws, static_cast<hipStream_t>(stream));
}

int gsr_blend(const GsrScene *scene, int64_t n, const GsrCamera *cam, const GsrOptions *opts, int64_t max_pairs, void *workspace,
              size_t workspace_bytes, void *out_image, float *out_final_T, void *stream)
{
    if (!out_image) { set_error("null output image"); return GSR_ERR_BAD_ARG; }
    if (scene) {
        const int rc = check_scene(scene);
        if (rc) return rc;
        if (scene->n != n) { set_error("scene->n = %lld but n = %lld", (long long)scene->n, (long long)n); return GSR_ERR_BAD_ARG; }
    }
    Workspace ws;
    int rc = check_frame(n, cam, opts, max_pairs, workspace, workspace_bytes, &ws);
    if (rc) return rc;
    return launch_blend(*cam, *opts, ws, tile_lists(ws, *opts), out_image, 0, out_final_T, scene, static_cast<hipStream_t>(stream));
}

// Stages 1-3 for `views` cameras through ONE launch sequence: view v works in slice v of the workspace (slices of
// gsr_workspace_bytes() bytes) and renders into out_images + v * out_view_stride bytes.
static int render_views(const GsrScene *scene, const GsrCamera *cams, int views, const GsrOptions *opts, int64_t max_pairs,
                        void *workspace, size_t workspace_bytes, void *out_images, size_t out_view_stride, float *out_final_T,
                        void *stream, bool keep_batch_words)
{
    int rc = check_scene(scene);
    if (rc) return rc;
    if (!out_images) { set_error("null output image"); return GSR_ERR_BAD_ARG; }
    if (!cams || views < 1 || views > MAX_VIEWS || (views > 1 && out_final_T)) { set_error("bad view count %d", views); return GSR_ERR_BAD_ARG; }
    Workspace ws;
    rc = check_frame(scene->n, &cams[0], opts, max_pairs, workspace, workspace_bytes, &ws);
    if (rc) return rc;
    if (views > 1) {
        if (workspace_bytes / ws.bytes < (size_t)views) {
            set_error("workspace too small for %d views per launch: %zu bytes given, %zu needed", views, workspace_bytes, ws.bytes * (size_t)views);
            return GSR_ERR_WORKSPACE;
        }
        ws.views = views;
        ws.view_stride = ws.bytes;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    rc = launch_preprocess(*scene, cams, *opts, ws, nullptr, reset_words_of(opts, keep_batch_words), s);
    if (rc) return rc;
    rc = bin_sort_impl(opts, ws, s);
    if (rc) return rc;
    // (the blend finds the scene through what the preprocess above left in each view's control block: same call, same arrays)
    return launch_blend(cams[0], *opts, ws, tile_lists(ws, *opts), out_images, out_view_stride, out_final_T, nullptr, s);
}

int gsr_render_forward(const GsrScene *scene, const GsrCamera *cam, const GsrOptions *opts, int64_t max_pairs,
                       void *workspace, size_t workspace_bytes, void *out_image, float *out_final_T, void *stream)
{
    if (!cam || !opts) { set_error("null camera/options"); return GSR_ERR_BAD_ARG; }
    return render_views(scene, cam, 1, opts, max_pairs, workspace, workspace_bytes, out_image, 0, out_final_T, stream, false);
}

// How many views go through one launch sequence: as many slices as the workspace holds, at most MAX_VIEWS, GsrOptions.batch_views
// (when set) and the views there are.  0: the workspace does not hold one view.
static int views_per_launch(const GsrScene *scene, const GsrCamera *cam0, const GsrOptions *opts, int64_t max_pairs, size_t workspace_bytes, int32_t n_cams)
{
    Workspace ws;
    const size_t per = carve_workspace(nullptr, scene->n, cam0->width, cam0->height, max_pairs, &ws);
    size_t k = workspace_bytes / per;
    k = std::min<size_t>(k, (size_t)MAX_VIEWS);
    if (opts->batch_views > 0) k = std::min<size_t>(k, (size_t)opts->batch_views);
    return (int)std::min<size_t>(k, (size_t)std::max<int32_t>(n_cams, 1));
}

static int check_batch(const GsrScene *scene, const GsrCamera *cams, int32_t n_cams, const GsrOptions *opts, int64_t max_pairs, const void *out_images,
                       int64_t frame_stride)
{
    if (!scene || !cams || n_cams < 0 || !out_images || !opts) { set_error("bad argument"); return GSR_ERR_BAD_ARG; }
    if (scene->n < 0 || scene->n > 0x7FFFFFFF || max_pairs < 0) { set_error("bad argument"); return GSR_ERR_BAD_ARG; }
    for (int32_t i = 0; i < n_cams; ++i) {
        if (cams[i].width <= 0 || cams[i].height <= 0) { set_error("bad frame size %dx%d", cams[i].width, cams[i].height); return GSR_ERR_BAD_ARG; }
        if (cams[i].width != cams[0].width || cams[i].height != cams[0].height) { set_error("views of a batch must share one frame size"); return GSR_ERR_BAD_ARG; }
    }
    // what one view writes: the frame, or (output_layout = 2) the strip of its shard's tile rows
    int64_t rows_px = cams[0].height;
    if (n_cams > 0 && opts->output_layout == 2) {
        const int tiles_y = (cams[0].height + GSR_TILE - 1) / GSR_TILE;
        rows_px = (int64_t)row_shard_of(*opts).rows_before(tiles_y) * GSR_TILE;
    }
    if (n_cams > 0 && frame_stride < (int64_t)cams[0].width * rows_px * 3) { set_error("frame_stride smaller than a frame"); return GSR_ERR_BAD_ARG; }
    return GSR_OK;
}

int gsr_render_batch(const GsrScene *scene, const GsrCamera *cams, int32_t n_cams, const GsrOptions *opts, int64_t max_pairs,
                     void *workspace, size_t workspace_bytes, void *out_images, int64_t frame_stride, void *stream)
{
    int rc = check_batch(scene, cams, n_cams, opts, max_pairs, out_images, frame_stride);
    if (rc || n_cams == 0) return rc;
    const int K = views_per_launch(scene, &cams[0], opts, max_pairs, workspace_bytes, n_cams);
    const size_t fbytes = (size_t)frame_stride * (opts->output_dtype == 1 ? 2 : 4);
    for (int32_t i = 0; i < n_cams; i += std::max(K, 1)) {
        // K = 0: the workspace does not hold one view — render_views reports it.  A slice's first view clears its whole control
        // block, its later views (i >= K) keep the batch-sticky overflow words
        const int k = std::max(1, std::min<int32_t>(K, n_cams - i));
        rc = render_views(scene, &cams[i], k, opts, max_pairs, workspace, workspace_bytes, static_cast<char *>(out_images) + (size_t)i * fbytes,
                          fbytes, nullptr, stream, i > 0);
        if (rc) return rc;
    }
    return GSR_OK;
}

int gsr_render_batch_slots(const GsrScene *scene, const GsrCamera *cams, int32_t n_cams, const GsrOptions *opts, int64_t max_pairs,
                           void *const *workspaces, size_t workspace_bytes, void *const *streams, int32_t n_slots, void *out_images,
                           int64_t frame_stride)
{
    if (!workspaces || !streams || n_slots < 1) { set_error("bad argument"); return GSR_ERR_BAD_ARG; }
    int rc = check_batch(scene, cams, n_cams, opts, max_pairs, out_images, frame_stride);
    if (rc) return rc;
    for (int32_t k = 0; k < n_slots; ++k) {
        if (!workspaces[k]) { set_error("null workspace in slot %d", (int)k); return GSR_ERR_BAD_ARG; }
        for (int32_t j = 0; j < k; ++j)
            if (workspaces[j] == workspaces[k]) { set_error("slots %d and %d share a workspace", (int)j, (int)k); return GSR_ERR_BAD_ARG; }
    }
    if (n_cams == 0) return GSR_OK;
    const int K = views_per_launch(scene, &cams[0], opts, max_pairs, workspace_bytes, n_cams);
    const size_t fbytes = (size_t)frame_stride * (opts->output_dtype == 1 ? 2 : 4);
    int32_t g = 0;
    for (int32_t i = 0; i < n_cams; i += std::max(K, 1), ++g) {
        const int32_t slot = g % n_slots;
        const int k = std::max(1, std::min<int32_t>(K, n_cams - i));
        // a slot's first group clears its slices' control blocks, its later groups keep the batch-sticky overflow words
        rc = render_views(scene, &cams[i], k, opts, max_pairs, workspaces[slot], workspace_bytes, static_cast<char *>(out_images) + (size_t)i * fbytes,
                          fbytes, nullptr, streams[slot], g >= n_slots);
        if (rc) return rc;
    }
    return GSR_OK;
}

int gsr_read_stats(void *workspace, size_t workspace_bytes, GsrStats *out, void *stream)
{
    if (!workspace || !out || workspace_bytes < sizeof(FrameCtrl)) { set_error("bad argument"); return GSR_ERR_BAD_ARG; }
    static_assert(sizeof(GsrStats) == 48 && offsetof(FrameCtrl, digit_tot) == sizeof(GsrStats), "GsrStats is the head of FrameCtrl");
    hipStream_t s = static_cast<hipStream_t>(stream);
    {  // total the blend's per-workgroup counters; the kernel finds them through the offsets kept in FrameCtrl
        const int rc = launch_blend_stats(static_cast<FrameCtrl *>(workspace), workspace_bytes, s);
        if (rc != GSR_OK) return rc;
    }
    // the whole control block (2 KB): GsrStats is its head, the sticky record of earlier frames its tail
    FrameCtrl host;
    GSR_HIP(hipMemcpyAsync(&host, workspace, sizeof(FrameCtrl), hipMemcpyDeviceToHost, s));
    GSR_HIP(hipStreamSynchronize(s));
    memcpy(out, &host, sizeof(GsrStats));
    if (out->overflow & 1u) {
        set_error("pair overflow: the frame needs %u (gaussian,tile) pairs", out->n_pairs_bbox);
        return GSR_ERR_PAIR_OVERFLOW;
    }
    if (out->overflow & 2u) {
        // the flag is sticky over a batch / a run of keep_flags frames while sort_passes describes the LAST frame: report what
        // the worst frame needed, or a caller that re-renders with "GsrStats.sort_passes" could be handed its own bound back
        out->sort_passes = std::max(out->sort_passes, host.batch_sort_passes);
        set_error("the depth sort needs %u passes, more than depth_sort_passes allowed", out->sort_passes);
        return GSR_ERR_SORT_PASSES;
    }
    return GSR_OK;
}

int gsr_scene_order_bytes(int64_t n, size_t *bytes)
{
    if (!bytes || n < 0 || n > 0x7FFFFFFF) { set_error("bad argument"); return GSR_ERR_BAD_ARG; }
    *bytes = scene_order_bytes(n);
    return GSR_OK;
}

int gsr_scene_order(int64_t n, const float *means, uint32_t *perm_out, void *workspace, size_t workspace_bytes, void *stream)
{
    if (n < 0 || n > 0x7FFFFFFF || (n > 0 && (!means || !perm_out || !workspace))) { set_error("bad argument"); return GSR_ERR_BAD_ARG; }
    if (n > 0 && reinterpret_cast<uintptr_t>(workspace) % 256 != 0) { set_error("workspace must be 256-byte aligned"); return GSR_ERR_BAD_ARG; }
    if (workspace_bytes < scene_order_bytes(n)) {
        set_error("workspace too small: %zu bytes given, %zu needed", workspace_bytes, scene_order_bytes(n));
        return GSR_ERR_WORKSPACE;
    }
    return launch_scene_order(n, means, perm_out, workspace, static_cast<hipStream_t>(stream));
}

int gsr_scene_bounds(int64_t n, const float *means, const float *log_scales, float *bounds_out, void *stream)
{
    if (n < 0 || n > 0x7FFFFFFF || (n > 0 && (!means || !log_scales || !bounds_out))) { set_error("bad argument"); return GSR_ERR_BAD_ARG; }
    if (reinterpret_cast<uintptr_t>(bounds_out) % 16 != 0) { set_error("bounds_out must be 16-byte aligned"); return GSR_ERR_BAD_ARG; }
    return launch_scene_bounds(n, means, log_scales, bounds_out, static_cast<hipStream_t>(stream));
}

int gsr_block_visibility(const GsrScene *scene, const GsrCamera *cam, const GsrOptions *opts, uint8_t *dead_out, void *stream)
{
    int rc = check_scene(scene);
    if (rc) return rc;
    if (!cam || !opts || !scene->block_bounds || (scene->n > 0 && !dead_out)) { set_error("null camera / options / block_bounds / output"); return GSR_ERR_BAD_ARG; }
    if (cam->width <= 0 || cam->height <= 0) { set_error("bad frame size %dx%d", cam->width, cam->height); return GSR_ERR_BAD_ARG; }
    return launch_block_visibility(*scene, *cam, *opts, dead_out, static_cast<hipStream_t>(stream));
}

int gsr_sh_to_rgb(int64_t n, const float *means, const float *sh, const float cam_center[3], int32_t degree, float *rgb_out,
                  void *stream)
{
    if (n < 0 || (n > 0 && (!means || !sh || !rgb_out)) || !cam_center || degree < 0 || degree > 3) {
        set_error("bad argument"); return GSR_ERR_BAD_ARG;
    }
    if (reinterpret_cast<uintptr_t>(sh) % 16 != 0) { set_error("sh must be 16-byte aligned"); return GSR_ERR_BAD_ARG; }
    return launch_sh_to_rgb(n, means, sh, cam_center, degree, rgb_out, static_cast<hipStream_t>(stream));
}

int gsr_cov3d(int64_t n, const float *log_scales, const float *quats, float *cov3d_out, void *stream)
{
    if (n < 0 || (n > 0 && (!log_scales || !quats || !cov3d_out))) { set_error("bad argument"); return GSR_ERR_BAD_ARG; }
    if (reinterpret_cast<uintptr_t>(quats) % 16 != 0) { set_error("quats must be 16-byte aligned"); return GSR_ERR_BAD_ARG; }
    return launch_cov3d(n, log_scales, quats, cov3d_out, static_cast<hipStream_t>(stream));
}

int gsr_project_to_camera_space(int64_t n, const float *means, const float w2c[16], float *out, void *stream)
{
    if (n < 0 || !w2c || (n > 0 && (!means || !out))) { set_error("bad argument"); return GSR_ERR_BAD_ARG; }
    return launch_project(n, means, w2c, out, static_cast<hipStream_t>(stream));
}

int gsr_compute_2d_covariance(int64_t n, const float *cov3d, const float *cam_means, double tan_fov_x, double tan_fov_y,
                              double focal_x, double focal_y, const float w2c[16], float *cov2d_out, void *stream)
{
    if (n < 0 || !w2c || (n > 0 && (!cov3d || !cam_means || !cov2d_out))) { set_error("bad argument"); return GSR_ERR_BAD_ARG; }
    if (reinterpret_cast<uintptr_t>(cov2d_out) % 16 != 0) { set_error("cov2d_out must be 16-byte aligned"); return GSR_ERR_BAD_ARG; }
    // focals / 2 (rasterize.py:216) and 1.3 * tan_fov (rasterize.py:210-211), formed in float64 like the reference
    return launch_cov2d(n, cov3d, cam_means, w2c, (float)(focal_x / 2.0), (float)(focal_y / 2.0), (float)(1.3 * tan_fov_x),
                        (float)(1.3 * tan_fov_y), cov2d_out, static_cast<hipStream_t>(stream));
}

int gsr_compute_covering_bbox(int64_t n, const float *screen_means, const float *cov2d, double width, double height,
                              int64_t *tile_bboxes_out, void *stream)
{
    if (n < 0 || (n > 0 && (!screen_means || !cov2d || !tile_bboxes_out))) { set_error("bad argument"); return GSR_ERR_BAD_ARG; }
    if (reinterpret_cast<uintptr_t>(cov2d) % 16 != 0) { set_error("cov2d must be 16-byte aligned"); return GSR_ERR_BAD_ARG; }
    return launch_bbox(n, screen_means, cov2d, (float)width, (float)height, tile_bboxes_out, static_cast<hipStream_t>(stream));
}

int gsr_rasterize_gaussian(int64_t gaussian_index, int64_t n, const int64_t *bboxes, float *screen, const float *screen_means,
                           const float *sigmas, const float *rgb, float *opacity_buffer, const float *opacity, int32_t width,
                           int32_t height, void *stream)
{
    if (!bboxes || !screen || !screen_means || !sigmas || !rgb || !opacity_buffer || !opacity || width <= 0 || height <= 0) {
        set_error("bad argument"); return GSR_ERR_BAD_ARG;
    }
    if (gaussian_index < 0 || gaussian_index >= n) { set_error("gaussian_index %lld out of range", (long long)gaussian_index); return GSR_ERR_BAD_ARG; }
    return launch_rasterize_gaussian(gaussian_index, bboxes, screen, screen_means, sigmas, rgb, opacity_buffer, opacity, width, height,
                                     static_cast<hipStream_t>(stream));
}

}  // extern "C"
