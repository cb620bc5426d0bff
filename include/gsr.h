/*
 * gsr.h — C ABI of libgsr.so, the MI355X (gfx950) forward rasterizer for 3D Gaussian Splatting.
 *
 * Boundary: the reference (arnaudstiegler/torch-gaussian-splatting-rasterizer)
 * has no FFI of its own — its "render call" is the body of rasterize.py:315-478.  This header is what a
 * binding for that path binds: every entry point names the reference lines it replaces.  Rules:
 *   - extern "C", POD structs, plain pointers and sizes; no C++/torch types; no exceptions cross.
 *   - every call returns int: GSR_OK or a negative GsrStatus; gsr_last_error() gives thread-local text.
 *   - the CALLER owns all device memory (inputs, outputs, one scratch workspace sized by
 *     gsr_workspace_bytes); the library never allocates on the hot path and holds no global state,
 *     so calls with distinct workspaces/streams may run concurrently.
 *   - all device work is enqueued on the caller's hipStream_t (passed as void*); nothing synchronises
 *     except gsr_read_stats.
 *   - fp32 throughout (the reference's arithmetic type); device pointers unless marked [host].
 *   - no environment variable is read anywhere: the A/B switches of rounds 1-3 are GsrOptions fields since 0.5.0
 *     (fine_binning, shard_preprocess, blend_pipe_tiles, sh_dense_min) — same frames whatever they say.
 */
#ifndef GSR_H
#define GSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_VERSION 600 /* 0.6.0: several views per launch sequence — gsr_render_batch / gsr_render_batch_slots put as many views through ONE
                            preprocess / sort / blend launch sequence as the workspace holds slices of gsr_workspace_bytes() (GsrOptions.batch_views
                            caps it); gsr_blend takes the scene again (NULL = what gsr_preprocess left in the workspace); GsrScene.block_bounds + gsr_scene_bounds /
                            gsr_block_visibility (block-level culling); GsrOptions.tile_row_block (tile-row shards in pairs of rows).  0.5.0: GsrOptions.saturation_rule (the exact colour-saturation early-out), the four environment switches became GsrOptions
                            fields (the library reads no environment and holds no function statics), GsrOptions.colour_stage / no_order_hint,
                            GsrStats.colour_evals, + gsr_scene_order.  0.4.0: GsrOptions.keep_flags, GsrOptions.accum_dtype; GsrStats.sort_passes reports the worst frame when the bound was exceeded; blend_impl 2
                            (matrix-pipe experiment) removed; the frame clear covers every word of the control block.  0.3.0: GsrOptions.depth_sort_passes, GsrStats.sort_passes, GSR_ERR_SORT_PASSES.  0.2.1: + gsr_render_batch_slots.
                            0.2.0: gsr_preprocess_geometry/_color removed (measured slower), gsr_read_stats takes a non-const workspace */

typedef enum GsrStatus {
    GSR_OK = 0,
    GSR_ERR_BAD_ARG = -1,       /* null pointer, non-positive size, unsupported option */
    GSR_ERR_WORKSPACE = -2,     /* workspace smaller than gsr_workspace_bytes() says */
    GSR_ERR_PAIR_OVERFLOW = -3, /* (gaussian,tile) pairs exceeded max_pairs: frame is incomplete, re-render with more */
    GSR_ERR_HIP = -4,           /* a HIP runtime call failed; see gsr_last_error() */
    GSR_ERR_SORT_PASSES = -5    /* the frame's depth keys span more bits than GsrOptions.depth_sort_passes passes cover: the frame is
                                   wrong, re-render with GsrStats.sort_passes (or 0) */
} GsrStatus;

/* Constants of rasterize.py:29-38 and the literals buried in the reference's glue. */
#define GSR_TILE 16                  /* BLOCK_SIZE, rasterize.py:34 */
#define GSR_GAUSSIAN_SPREAD 3.0f     /* rasterize.py:32 */
#define GSR_MAX_ALPHA 0.99f          /* MAX_GAUSSIAN_DENSITY, rasterize.py:36 */
#define GSR_MIN_ALPHA (1.0f / 255.0f)/* rasterize.py:38 */
#define GSR_CULL_Z 0.2f              /* rasterize.py:377 */
#define GSR_LOWPASS 0.3f             /* rasterize.py:249-250 */
#define GSR_EIG_FLOOR 0.1f           /* rasterize.py:172,175 */
#define GSR_MAX_PAIRS 0xFFFFE000ll   /* largest max_pairs: pairs are indexed in 32 bits, one 4096-pair sort tile of headroom */
#define GSR_MAX_BATCH_VIEWS 8        /* most views gsr_render_batch puts through one launch sequence */
#define GSR_BOUNDS_BLOCK 64          /* consecutive gaussians per entry of GsrScene.block_bounds */

/* Camera-independent trained gaussians, exactly the values stored in the INRIA .ply
 * (rasterize.py:98-106,354-358; utils.py:10-31).  Row-major dense arrays.
 * The ORDER of the arrays is the caller's; it decides one thing: gaussians at EXACTLY equal depth (fp32 z_cam) are drawn in array-
 * index order (the reference's torch.sort, rasterize.py:425, leaves their order undefined; in practice torch's CPU sort keeps file
 * order).  Arrays in .ply file order therefore reproduce the reference's practical frame bit for bit in its tie order as well; arrays
 * reordered for speed (gsr_scene_order: what the Python loaders do by default, `spatial_order=False` / `--scene-order file` turn it
 * off) render the same frame except where two tied gaussians overlap — measured up to ~1e-2 per pixel on scenes with many exact
 * ties (tools/fuzz_parity.py: 0.0086), ~1e5 tie pairs among the 3.4 M visible gaussians of a 6 M-gaussian scene. */
typedef struct GsrScene {
    int64_t n;                  /* number of gaussians */
    const float *means;         /* [n,3]  x,y,z */
    const float *log_scales;    /* [n,3]  scale_0..2, BEFORE exp (rasterize.py:97) */
    const float *quats;         /* [n,4]  rot_0..3 = (w,x,y,z), un-normalised (rasterize.py:99-112) */
    const float *opacity_logit; /* [n]    BEFORE sigmoid (rasterize.py:358) */
    const void *sh;             /* [n,16,3] sh[i][k][c], k=0 is f_dc (utils.py:21-31); fp32, or IEEE fp16 if sh_dtype = 1 */
    int32_t sh_degree;          /* 0..3; the reference always evaluates 3 (rasterize.py:368) */
    int32_t sh_dtype;           /* 0 = float32 (the reference's type), 1 = float16 storage (evaluated in fp32; halves the
                                   192-B/gaussian stream that dominates stage 1; ~91 dB vs fp32 coefficients) */
    const float *block_bounds;  /* optional (NULL = none): [ceil(n / GSR_BOUNDS_BLOCK)][8] from gsr_scene_bounds — per block of
                                   GSR_BOUNDS_BLOCK consecutive gaussians the box of their means and their largest log-scale.  The
                                   first kernel of a frame tests every block against the view (cull plane rasterize.py:377, frame, this
                                   rank's tile rows) and the preprocess skips, BEFORE touching their gaussians' 44 B each, the blocks none
                                   of whose gaussians can be drawn: a conservative bound, frames are bit-identical with and without.
                                   Worth having when the arrays are in a spatial order (gsr_scene_order).  16-byte aligned. */
} GsrScene;

/* One view.  Filled by gsr_camera_setup() or by hand.  All matrices are in the reference's
 * row-vector convention (transposed at rasterize.py:361-362): x_cam = x_w @ w2c[:3,:3] + w2c[3,:3]. */
typedef struct GsrCamera {
    float w2c[16];        /* get_world_to_camera_matrix(...).T, rasterize.py:59-77,:361 */
    float full_proj[16];  /* w2c @ get_projection_matrix(...).T, rasterize.py:123-151,:362-364 */
    float cam_center[3];  /* inverse(w2c)[3,:3], spherical_harmonics.py:35 */
    float focal_x, focal_y;     /* focal of the EWA Jacobian = full-res fx,fy / 2 (rasterize.py:216; quirk Q3) */
    float lim_x, lim_y;         /* fp32(1.3*tan(fov/2)), rasterize.py:210-211 */
    float tan_fov_x, tan_fov_y; /* rasterize.py:344-345 */
    int32_t width, height;      /* output size in pixels, rasterize.py:338 */
} GsrCamera;

typedef struct GsrOptions {
    int32_t reference_compat; /* 1 (default): reproduce Q1 (column W-1 / row H-1 never drawn, rasterize.py:271-272,
                                 :415-418) and Q2 (skip if ANY conic entry == 0, :441).  0: draw every pixel. */
    float early_out_T;        /* a wave of 64 pixels stops once all their transmittances are <= this.  0 (default) is
                                 exact — identical bits to blending every gaussian like the reference (Q5); see
                                 saturation_rule for what "exact" stops on.  >0 is a bounded approximation (INRIA
                                 uses 1e-4). */
    int32_t tile_row_begin;   /* multi-GPU sharding: this call bins+blends tile rows begin, begin+step, ... (blocks of rows: tile_row_block) */
    int32_t tile_row_step;    /* default 0 / 1 = all rows */
    int32_t output_layout;    /* 0 (default): image [H,W,3] (= screen.transpose(1,0), rasterize.py:471);
                                 1: reference `screen` layout [W,H,3] (rasterize.py:437);
                                 2: compact strip [rows_of_this_shard*16, W, 3] for the framebuffer gather */
    int32_t no_footprint_cull;/* 0 (default): tile rects / quadrant tests are tightened to the AABB of the region where
                                 alpha > 1/255 can hold (exact: skipped work contributes nothing).  1: bin the
                                 reference's full 3-sigma tile rect — only to prove that property in tests. */
    int32_t blend_impl;       /* 0 (default): the blend kernel with its inner walk hand-scheduled (EXEC-masked update, csrc/blend.hip).
                                 1: the same kernel with the walk in plain C — the readable statement of the arithmetic and the A/B
                                 reference; frames are bit-identical to 0. */
    int32_t draw_limit;       /* 0 (default): blend everything.  k > 0: blend only the first k gaussians of the reference's
                                 draw order (depth order restricted to those its skip guard rasterize.py:441 lets through) —
                                 the progressive frames of --generate_video (rasterize.py:448-450). */
    int32_t output_dtype;     /* 0 (default): the frame is float32.  1: the frame is stored as bfloat16 (round to nearest even),
                                 6 B per pixel (BASELINE configs[2]); T and the colour sums are ALWAYS accumulated in fp32
                                 registers — bf16 accumulators measure 41 dB, below the 50 dB bar.
                                 out_final_T stays float32. */
    int32_t depth_sort_passes;/* 0 (default): enqueue the four radix passes any depth range can need; those a frame does not need
                                 return at once, but still cost their launches (3 per pass, ~4.6 us each).  1..4: the caller's bound on
                                 what the frames need — GsrStats.sort_passes of an earlier frame of the scene (3 whenever the depths stay
                                 within 0.2 .. 13 000): only that many are enqueued.  Verified on the device like max_pairs: a frame
                                 that needs more is flagged and gsr_read_stats returns GSR_ERR_SORT_PASSES. */
    int32_t accum_dtype;      /* 0 (default): T and the colour sums are accumulated in fp32 registers.  1: they are rounded to bfloat16 after
                                 every gaussian — BASELINE configs[2] as literally worded ("bf16 blend accumulator").  A measurement
                                 option, not a fast path (it runs the plain-C blend kernel, whatever blend_impl says): 8 mantissa bits
                                 cannot carry a transmittance through hundreds of layers, the frame lands far below the fp32 one
                                 (tests/test_gpu_configs.py prints the PSNR; SURVEY.md §7.3 probed 41 dB on the CPU). */
    int32_t keep_flags;       /* 0 (default): the frame starts from a cleared control block (a fresh workspace needs no initialisation).
                                 1: the frame keeps the overflow record of the frames rendered before it on this workspace (bits of
                                 GsrStats.overflow, the largest D, the most depth-sort passes), so ONE gsr_read_stats after a run of
                                 unchecked frames — first frame 0, the rest 1 — reports whether ANY of them exceeded max_pairs or
                                 depth_sort_passes.  The views of gsr_render_batch are chained this way internally. */
    int32_t saturation_rule;  /* when may a quadrant of 64 pixels stop EXACTLY (early_out_T = 0)?
                                 0 (default): once no later gaussian can change a bit of its COLOUR: every pixel has
                                 T <= 2^-25 * min(Cr, Cg, Cb).  The update is C = fma(w, c, C) with w = fl(alpha T) <= T and c <= 1 (Q7),
                                 so w c <= T < ulp(C) / 2 for each channel and the fma returns C unchanged, now and ever after (T only
                                 shrinks).  Pixels the reference never draws (Q1) count as finished.  Frames are bit-identical to rule 1;
                                 ~4x fewer evaluated entries on the bench frame.  When out_final_T is requested rule 1 applies (T itself
                                 is then an output and keeps shrinking).
                                 1: once T has underflowed to 0.0f for all 64 pixels (rounds 1-3) — the A/B reference of rule 0. */
    int32_t fine_binning;     /* 0 (default): (gaussian, 32x32 cell) pairs whenever the frame allows (<= 4096 px, n <= 2^28), the blend filters
                                 the cell lists by tile.  1: (gaussian, 16x16 tile) pairs — the path wider frames always take; A/B timing and
                                 the test that both build the same frame (csrc/binning.hip).  Was GSR_FINE_BINNING. */
    int32_t shard_preprocess; /* tile-row shards only: 0 (default) = the three-phase shard preprocess from tile_row_step 5 on, 1 = always the
                                 whole-frame kernel, 2 = always the three-phase kernel (csrc/preprocess.hip).  Was GSR_SHARD_PREPROCESS. */
    int32_t blend_pipe_tiles; /* tile count up to which the blend runs its pipelined one-quadrant walk: 0 (default) = 1280, -1 = never
                                 (csrc/blend.hip).  Was GSR_BLEND_PIPE_TILES. */
    int32_t no_order_hint;    /* 0 (default): tiles are launched heaviest first by what each tile's blend STAGED in the last frame rendered on
                                 this workspace (consecutive frames of a camera path look alike; a tile saturates long before the end of
                                 its list), falling back to the list length where that record is missing or impossible.  1: by list length
                                 alone (rounds 1-3).  A schedule only: every order renders the same bits. */
    int32_t colour_stage;     /* where is sh_to_rgb (spherical_harmonics.py:27-73) evaluated?
                                 0 (default): in the blend, when a tile first STAGES a gaussian (its 192-B SH row is read and its colour
                                 evaluated then, and remembered in the gaussian's record for the tiles that stage it later): gaussians
                                 that no tile reaches before it is saturated — most of a dense scene — never read their SH row.  The
                                 preprocess then reads 44 B per gaussian instead of up to 236.  Same arithmetic in the same order on the
                                 same inputs: frames are bit-identical to 1.
                                 1: in gsr_preprocess, for every visible gaussian (rounds 1-3) — the A/B reference of 0.
                                 (gsr_preprocess with a GsrDebugOut always evaluates there: `rgb` is one of its outputs.)  Only
                                 gsr_preprocess reads this field; gsr_blend finds out from the records. */
    int32_t sh_dense_min;     /* visible gaussians per wave from which the preprocess fetches the wave's 64 SH rows whole through LDS:
                                 0 (default) = 48, 65 = never (csrc/preprocess.hip).  Was GSR_SH_DENSE. */
    int32_t batch_views;      /* gsr_render_batch / gsr_render_batch_slots: at most this many views per launch sequence; 0 (default) = as many
                                 as the workspace holds slices for, up to GSR_MAX_BATCH_VIEWS.  1 = one view at a time (rounds 1-4).  Same
                                 frames whatever it says. */
    int32_t tile_row_block;   /* multi-GPU sharding: how many consecutive tile rows form one unit of the interleave.  0 / 1 (default): single
                                 rows — this call owns tile rows begin, begin + step, ...  2: PAIRS of rows, i.e. the two tile rows of one
                                 32x32 binning cell — block b = rows 2b, 2b + 1 is owned when b % step == begin.  A rank then bins and sorts
                                 only the cells it owns (with single rows and an even step every cell row is shared by two ranks, and each
                                 of them emits and sorts all its pairs); the strip (output_layout = 2) holds the owned rows in ascending
                                 order either way.  Same pixels; only the partition differs. */
} GsrOptions;

/* Counters of one frame (device -> host with gsr_read_stats). */
typedef struct GsrStats {
    uint32_t n_visible;     /* V: gaussians that survive cull and have a non-empty footprint */
    uint32_t n_pairs_bbox;  /* D: pair slots this frame needs, i.e. what max_pairs must cover: (gaussian, 32x32 cell) pairs of the
                               visible gaussians' rects (frames wider than 4096 px: (gaussian, 16x16 tile) pairs of this shard) */
    uint32_t n_pairs;       /* E: entries of the per-tile lists the blend consumes (this shard; after footprint culling) */
    uint32_t overflow;      /* bit 0: D exceeded max_pairs (frame incomplete); bit 1: the depth sort needed more passes than
                               depth_sort_passes allowed (frame wrong) */
    uint32_t max_list_len;  /* longest list a blend workgroup walks: per 32x32 cell (frames wider than 4096 px: per 16x16 tile) */
    uint32_t sort_passes;   /* radix passes the depth sort of this frame needs (1..4): the bound to pass as depth_sort_passes.  When
                               overflow bit 1 is set: the most any frame since the last cleared one (keep_flags, batches) needed */
    uint64_t wave_entries;  /* (8x8 quadrant, entry) pairs the blend actually evaluated: 64 pixel evaluations each */
    uint64_t fetched_entries; /* list entries the blend staged (<= n_pairs: a saturated tile stops fetching) */
    uint64_t colour_evals;  /* sh_to_rgb evaluations (192-B SH rows read) by the blend: colour_stage = 0 evaluates a gaussian when a tile
                               first stages it; tiles racing for the same gaussian may each evaluate it (same result), so this is an
                               UPPER-BOUND estimate of the gaussians coloured and varies by a few per cent from run to run.  0 when the
                               preprocess evaluated the colours (colour_stage = 1: n_visible evaluations there). */
} GsrStats;

/* Optional intermediates of gsr_preprocess, one entry per gaussian, any pointer may be NULL.
 * They are the tensors the reference's helpers return (used by the drop-in helper functions and parity tests). */
typedef struct GsrDebugOut {
    float *cov3d;         /* [n,9]  get_covariance_matrix_from_mesh, rasterize.py:89-120 */
    float *cam_means;     /* [n,3]  project_to_camera_space, rasterize.py:80-86 */
    float *cov2d;         /* [n,4]  compute_2d_covariance, rasterize.py:201-252 (before the cull zeroing of :388) */
    float *screen_means;  /* [n,2]  rasterize.py:391 */
    int64_t *tile_bboxes; /* [n,4]  compute_covering_bbox, rasterize.py:154-198 */
    float *sigmas;        /* [n,3]  conic (sigma_x, sigma_y, sigma_xy), rasterize.py:404-411 */
    int64_t *pixel_bboxes;/* [n,4]  rasterize.py:415-419 */
    float *rgb;           /* [n,3]  sh_to_rgb, spherical_harmonics.py:27-73 */
    float *opacity;       /* [n]    sigmoid(opacity_logit), rasterize.py:358 */
} GsrDebugOut;

int gsr_version(void);
const char *gsr_last_error(void);
void gsr_default_options(GsrOptions *opts /* [host] */);

/* [host] Camera set-up from COLMAP extrinsics/intrinsics: replaces rasterize.py:336-345 (fov, focals),
 * :59-77 + :361 (world->camera), :123-151 + :362-364 (projection, full transform) and
 * spherical_harmonics.py:35 (camera centre).  fx_full/fy_full/cam_width/cam_height are the FULL-RES camera
 * of cameras.bin (cam_info[1]); width/height the size of the images_{K} frame (rasterize.py:338). */
int gsr_camera_setup(const double qvec[4], const double tvec[3], double fx_full, double fy_full, int64_t cam_width,
                     int64_t cam_height, int32_t width, int32_t height, GsrCamera *out /* [host] */);

/* Bytes of scratch one frame needs for n gaussians, a width x height target and room for max_pairs
 * pairs.  max_pairs is the caller's bound on D = the (gaussian, 32x32 cell) pairs of a frame (frames wider than 4096 px:
 * (gaussian, 16x16 tile) pairs); GsrStats.n_pairs_bbox reports what a frame needed; exceeding it is reported, never UB. */
int gsr_workspace_bytes(int64_t n, int32_t width, int32_t height, int64_t max_pairs, size_t *bytes /* [host] */);

/* Stage 1 — per-gaussian preprocessing, fused: rasterize.py:354-420 (means, cov3D, opacity, SH colour,
 * camera/clip projection, z<0.2 cull, EWA 2D covariance, NDC->pixel, radius + tile rect, conic, pixel rect). */
int gsr_preprocess(const GsrScene *scene, const GsrCamera *cam, const GsrOptions *opts, void *workspace,
                   size_t workspace_bytes, const GsrDebugOut *debug /* [host], may be NULL */, void *stream);

/* Stage 2 — depth order (rasterize.py:424-425, ties broken by gaussian index) and 16x16 tile binning:
 * radix sort of the visible gaussians by depth, pair emission in depth order, stable radix sort by tile,
 * per-tile [begin,end) ranges.  Needs gsr_preprocess on the same workspace first. */
int gsr_bin_sort(int64_t n, const GsrCamera *cam, const GsrOptions *opts, int64_t max_pairs, void *workspace,
                 size_t workspace_bytes, void *stream);

/* Stage 3 — front-to-back compositing of every tile list: rasterize.py:436-446 (driver loop + skip guard)
 * and :255-305 (rasterize_gaussian).  out_image layout per opts->output_layout; out_final_T [H,W] may be NULL.
 * n and max_pairs must be the values given to the earlier stages (the library keeps no state; they fix the
 * workspace layout).  With GsrOptions.colour_stage = 0 (default) this stage evaluates sh_to_rgb (spherical_harmonics.py:27-73) for
 * the gaussians its tiles stage, from `scene`'s means / sh arrays and cam->cam_center: pass the scene gsr_preprocess was given
 * (the same values: the arrays may have MOVED since, they may not have changed).  scene = NULL: the pointers gsr_preprocess left
 * in the workspace are used instead — then those arrays must still be alive where they were, which no argument check can see
 * (a binding should always pass the scene). */
int gsr_blend(const GsrScene *scene /* may be NULL */, int64_t n, const GsrCamera *cam, const GsrOptions *opts, int64_t max_pairs,
              void *workspace, size_t workspace_bytes, void *out_image /* float32 or bfloat16, opts->output_dtype */,
              float *out_final_T, void *stream);

/* Stages 1-3 back to back: the whole render call of rasterize.py:354-446. */
int gsr_render_forward(const GsrScene *scene, const GsrCamera *cam, const GsrOptions *opts, int64_t max_pairs,
                       void *workspace, size_t workspace_bytes, void *out_image, float *out_final_T, void *stream);

/* Several views of ONE resident scene (the reference renders one view per process, rasterize.py:315-329; BASELINE configs[3] is a
 * camera set).  cams[n_cams] [host] must share width/height; frame i goes to out_images + i * frame_stride (in elements of the
 * output dtype).
 * Views per launch sequence.  With S = gsr_workspace_bytes(n, width, height, max_pairs), a workspace of K * S bytes is K slices,
 * and K = min(workspace_bytes / S, GSR_MAX_BATCH_VIEWS, opts->batch_views if set, n_cams) views at a time go through ONE sequence
 * of launches — one preprocess that reads every gaussian's 44 B once for the K cameras, one set of depth-sort and cell-sort launches
 * over K x the keys, one blend over K x the tiles: ~22 dispatches per K frames instead of per frame, each K times better filled.
 * View i works in slice i % K (workspace + (i % K) * S) and its frame is bit for bit the frame gsr_render_forward renders.
 * Counters: gsr_read_stats on slice j (pointer workspace + j * S, size S) describes the LAST view rendered there (views j, j + K, ...);
 * an overflow in any of them is sticky in it, and n_pairs_bbox / sort_passes then report what the WORST of them needed (the values
 * to re-render with).  A caller that checks a batch reads slices 0 .. min(K, n_cams) - 1. */
int gsr_render_batch(const GsrScene *scene, const GsrCamera *cams /* [host] */, int32_t n_cams, const GsrOptions *opts,
                     int64_t max_pairs, void *workspace, size_t workspace_bytes, void *out_images, int64_t frame_stride,
                     void *stream);

/* The same with n_slots batches in flight: with K = the views per launch sequence as above (every workspace of workspace_bytes = K
 * slices), group g = views g K .. g K + K - 1 is enqueued with workspaces[g % n_slots] on streams[g % n_slots], so that the ramps
 * and tails of one group's kernels overlap another's (the library keeps no state; each slot is one more (workspace, stream) pair).
 * Frames are bit-identical to gsr_render_batch.  Each slice's counters afterwards describe the LAST view rendered there; an overflow
 * in any of them is sticky.  The caller orders the streams against whatever produced the scene and whatever consumes the frames. */
int gsr_render_batch_slots(const GsrScene *scene, const GsrCamera *cams /* [host] */, int32_t n_cams, const GsrOptions *opts,
                           int64_t max_pairs, void *const *workspaces /* [host] n_slots device pointers */, size_t workspace_bytes,
                           void *const *streams /* [host] n_slots streams */, int32_t n_slots, void *out_images, int64_t frame_stride);

/* Totals the blend's per-workgroup counters (one small kernel on `stream`: wave_entries / fetched_entries describe the
 * LAST gsr_blend of the frame, 0 if none ran; the two totals are written into the workspace's counter block, no frame
 * data changes), copies the frame counters to host memory and waits for the stream.
 * Returns GSR_ERR_PAIR_OVERFLOW if the frame overflowed max_pairs, GSR_ERR_SORT_PASSES if depth_sort_passes was too small. */
int gsr_read_stats(void *workspace, size_t workspace_bytes, GsrStats *out /* [host] */, void *stream);

/* Scene order (no reference counterpart: the reference reads the .ply in file order, rasterize.py:353-358, and its result does not
 * depend on the storage order except for gaussians at exactly equal depth, which its unstable torch.sort leaves undefined and this
 * library draws in array-index order).  perm_out[i] = index, in the caller's arrays, of the gaussian that should be stored i-th: the
 * gaussians along a Morton curve of their means (every axis rank-quantised to 10 bits, bits interleaved, stable).  Gathering all five
 * scene arrays with it before rendering makes the same frames ~10 % faster on MI355X (culled waves, L2 hits).  Once per scene:
 * sixteen radix passes; at 6.13 M gaussians the permutation plus the caller's gather of the five arrays take 3.8 ms in a warm process
 * (the first call of a process also loads this library's code objects: 19 - 71 ms seen).  workspace: gsr_scene_order_bytes(n) bytes,
 * 256-B aligned, free afterwards. */
int gsr_scene_order_bytes(int64_t n, size_t *bytes /* [host] */);
int gsr_scene_order(int64_t n, const float *means /* [n,3] */, uint32_t *perm_out /* [n] */, void *workspace, size_t workspace_bytes,
                    void *stream);

/* Camera-independent block bounds for GsrScene.block_bounds (no reference counterpart; once per scene, after any reordering):
 * bounds_out [ceil(n / GSR_BOUNDS_BLOCK)][8] = {min x, min y, min z, max x, max y, max z, max log-scale, 0} over each block of
 * GSR_BOUNDS_BLOCK consecutive gaussians (a block with a non-finite value gets an unbounded box: never skipped). */
int gsr_scene_bounds(int64_t n, const float *means /* [n,3] */, const float *log_scales /* [n,3] */, float *bounds_out, void *stream);

/* Which blocks of scene->block_bounds does the preprocess skip for this view (and, with a tile-row shard in opts, this rank)?
 * dead_out [ceil(n / GSR_BOUNDS_BLOCK)] = 1 where no gaussian of the block can be drawn.  The test the kernels apply, exposed for
 * tests (every gaussian of a skipped block must fail the reference's own skip guard, rasterize.py:441) and for tooling. */
int gsr_block_visibility(const GsrScene *scene, const GsrCamera *cam, const GsrOptions *opts, uint8_t *dead_out, void *stream);

/* Stand-alone helpers behind the reference's helper functions (same maths as inside gsr_preprocess). */
/* sh_to_rgb, spherical_harmonics.py:27-73 */
int gsr_sh_to_rgb(int64_t n, const float *means, const float *sh, const float cam_center[3] /* [host] */, int32_t degree,
                  float *rgb_out, void *stream);
/* get_covariance_matrix_from_mesh, rasterize.py:89-120: out [n,9] */
int gsr_cov3d(int64_t n, const float *log_scales, const float *quats, float *cov3d_out, void *stream);

/* project_to_camera_space, rasterize.py:80-86: out[n,3] = means @ w2c[:3,:3] + w2c[3,:3]; w2c [host] row-major 4x4 */
int gsr_project_to_camera_space(int64_t n, const float *means, const float w2c[16] /* [host] */, float *out, void *stream);
/* compute_2d_covariance, rasterize.py:201-252: cov3d [n,9], cam_means [n,3] -> cov2d_out [n,4] (2x2 row-major).
 * tan_fov_*, focal_* are the reference's arguments as given (focals are halved inside, rasterize.py:216). */
int gsr_compute_2d_covariance(int64_t n, const float *cov3d, const float *cam_means, double tan_fov_x, double tan_fov_y,
                              double focal_x, double focal_y, const float w2c[16] /* [host] */, float *cov2d_out, void *stream);
/* compute_covering_bbox, rasterize.py:154-198: screen_means [n,2], cov2d [n,4] -> tile-unit bboxes int64 [n,4] */
int gsr_compute_covering_bbox(int64_t n, const float *screen_means, const float *cov2d, double width, double height,
                              int64_t *tile_bboxes_out, void *stream);
/* rasterize_gaussian, rasterize.py:255-305: blend ONE gaussian in place into screen [W,H,3] / opacity_buffer [W,H]
 * (the reference's x-major layout).  n = number of gaussians in the per-gaussian arrays (bounds check). */
int gsr_rasterize_gaussian(int64_t gaussian_index, int64_t n, const int64_t *bboxes, float *screen, const float *screen_means,
                           const float *sigmas, const float *rgb, float *opacity_buffer, const float *opacity, int32_t width,
                           int32_t height, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSR_H */
