// gsr_internal.h — shared declarations of libgsr.so (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <type_traits>

#include "../../include/gsr.h"

namespace gsr {

// ---------------------------------------------------------------------------------------------
// Device control block at the head of the workspace.  Zeroed at the start of every frame.
// The first 48 bytes are GsrStats verbatim.
// ---------------------------------------------------------------------------------------------
struct FrameCtrl {
    uint32_t n_visible;     // V  (written by the first depth-sort scatter pass)
    uint32_t n_pairs_bbox;  // D  (total of the tile-count scan, before clamping)
    uint32_t n_pairs;       // E  (pairs that survive footprint culling; written by tile-sort pass 0)
    uint32_t overflow;
    uint32_t max_list_len;
    uint32_t sort_passes;   // depth-sort plan of this frame (sort.hip): passes it needs (1..4) — the sorted ids end up in val[sort_passes & 1]
    unsigned long long wave_entries;  // (quadrant, entry) pairs evaluated by the blend   } totals of blend_stats[], filled in
    unsigned long long fetched_entries;  // list entries staged by the blend              } by gsr_read_stats
    unsigned long long colour_evals;     // deferred colours evaluated by the blend       }
    uint32_t digit_tot[512]; // per-digit totals of the radix pass in flight
    uint32_t stats_off;      // byte offset of blend_stats[] from this struct, and the number of launch slots the last
    uint32_t stats_slots;    // blend filled (tile_order_kernel writes both; 0 = no blend since the frame was reset)
    uint32_t n_slots;        // min(D, max_pairs): pair slots written by the emit kernel
    uint32_t sort_key_bits;  // rest of the depth-sort plan: significant bits of key - bits(0.2f); digit width of the passes
    uint32_t sort_bits_rest; //   after the first.  Written by the pass-0 rowscan (with sort_passes).
    uint32_t n_cpairs;       // coarse binning: 32x32-cell pairs that survive the sort's drop (count of the coarse ranges pass)
    uint32_t n_records;      // multi-GPU shard: records entering the depth sort (= this rank's visible gaussians; preprocess.hip)
    uint32_t ent_off;        // coarse binning without expansion: byte offset (from this struct) of the emit workgroups' partial counts
                             // of tile-list entries; gsr_read_stats totals them into n_pairs.  0: n_pairs is already final.
    // deferred colour (GsrOptions.colour_stage = 0): what the blend needs to evaluate sh_to_rgb for a gaussian it stages — written by
    // the preprocess kernel's workgroup 0 after the clear, so that gsr_blend (which is handed no scene) finds it in the workspace
    const float *col_means;  // GsrScene.means
    const void *col_sh;      // GsrScene.sh
    float col_cc[3];         // GsrCamera.cam_center
    int32_t col_degree;      // GsrScene.sh_degree
    int32_t col_sh16;        // GsrScene.sh_dtype
    uint32_t _pad0;
    uint32_t depth_key_max;  // maximum of the frame's valid depth keys (pass-0 histogram).  Cleared with the frame AND by the pass-0
                             // rowscan once consumed (gsr_bin_sort may be repeated on one gsr_preprocess).
    // ---- everything below survives the per-frame clear of a frame rendered with GsrOptions.keep_flags (and of the later views
    // ---- of a batch): the record of what went wrong in ANY frame since the last full clear
    uint32_t batch_overflow;     // bit 0 pair overflow, bit 1 depth sort short of passes
    uint32_t batch_need;         // largest D seen
    uint32_t batch_sort_passes;  // most radix passes any frame's depth sort has needed
    uint32_t _pad1;
};
constexpr int BLEND_STAT_WORDS = 8;  // per launch slot: [0..3] evaluated entries of waves 0..3, [4] staged entries, [5] deferred colours evaluated

// Per-gaussian record consumed by pair emission and the blend (48 B, three 16-B loads):
//   q0 = {mean_x, mean_y, -B/(2C), -B/(2A)}   the two ratios locate the edge maxima in footprint.h
//   q1 = {A, B, C, pthr}    power2(dx,dy) = A dx^2 + B dx dy + C dy^2 (log2 domain); pthr: see footprint.h
//   q2 = {log2(opacity), red, green, blue}     alpha = 2^(power2 + log2(opacity))
struct alignas(16) GaussRec {
    float4 q0, q1, q2;
};

constexpr int DEPTH_SORT_THREADS = 512;  // depth sort: 9-bit digits on 8192-key tiles (radix.h)
constexpr int PAIR_SORT_THREADS = 256;   // pair sort: <= 8-bit digits on 4096-pair tiles
constexpr int DEPTH_SORT_ITEMS = 16;     // keys per thread per pass (2048-key tiles measured slower: fixed per-workgroup costs dominate)
#ifndef GSR_DS_SHARD_ITEMS
#define GSR_DS_SHARD_ITEMS 8
#endif
constexpr int DEPTH_SORT_ITEMS_SHARD = GSR_DS_SHARD_ITEMS;  // (the macro: tools/ A/B builds; 2048-key tiles measured on a rank of 8 with rows in pairs: 0.138 against 0.126 ms per frame)  // a multi-GPU rank's compact records (~0.8 M at G = 8): 4096-key tiles, twice the workgroups, half the serial chain each
constexpr int PAIR_SORT_ITEMS = 16;
constexpr int EMIT_THREADS = 256;                     // threads per workgroup in the binning kernels
constexpr uint32_t KEY_INVALID = 0xFFFFFFFFu;

struct Workspace {
    FrameCtrl *ctrl;
    GaussRec *rec;        // [n]
    ushort4 *rect;        // [n]   tile rect {tx0, ty0, tx1, ty1} (exclusive upper), after footprint refinement
    uint32_t *rect8[2];   // [n]   the same rect packed x0 | y0<<8 | (x1-1)<<16 | (y1-1)<<24; rides through the depth
                          //       sort as a second payload when the tile grid fits 8 bits (frames up to 4096 px)
    uint32_t *key[2];     // [n]   depth keys (ping-pong)
    uint32_t *val[2];     // [n]   gaussian ids (ping-pong)
    uint32_t *pair_off;   // [n]   exclusive pair offsets in depth order
    uint32_t *blk_sum;    // [ceil(n/EMIT_THREADS)+1]
    uint32_t *hist;       // [512 * hist_blocks] digit-major: hist[digit][block]
    uint32_t *pkey[2];    // [max_pairs] tile ids
    uint32_t *pval[2];    // [max_pairs] gaussian ids
    uint2 *ranges;        // [tiles]
    uint2 *cranges;       // [ctiles]  coarse binning: [begin, end) of every 32x32 cell's list in the sorted pair array
    int *tile_order;      // [8 * ceil(tiles_y/8) * tiles_x] blend launch order
    unsigned char *blk_dead; // [ceil(n / GSR_BOUNDS_BLOCK)] block-level culling: 1 = the preprocess skips the block (preprocess.hip, block_flags_kernel)
    uint32_t *tile_work;  // [tiles] 1 + entries the tile's blend staged in the LAST frame rendered on this workspace (0 / garbage: unknown):
                          // the launch-order hint of the next frame (blend.hip, tile_order_kernel); never cleared, never trusted
    uint32_t *blend_stats; // [tile_order slots][BLEND_STAT_WORDS] per-workgroup counters: plain stores, no atomics (40 k
                          // same-address atomics per frame put a 0.45 ms floor under the blend kernel)
    int64_t n;
    int64_t max_pairs;
    int tiles_x, tiles_y;
    int ctiles_x, ctiles_y;  // 32x32 cells = 2x2 tiles
    int hist_blocks;      // row stride of `hist`
    size_t bytes;
    // Several views through ONE launch sequence (gsr_render_batch): every kernel of the render path is launched with gridDim.y = views
    // and view v works on the v-th slice of the caller's workspace.  The pointers above are view 0's; slice v lies v * view_stride
    // bytes further (a slice is a complete one-view workspace: gsr_read_stats reads any of them).
    int views = 1;
    size_t view_stride = 0;
};
constexpr int MAX_VIEWS = 8;  // views per launch sequence (the cameras travel in the preprocess kernel's argument block)

// p + off bytes, as pointer arithmetic (a round trip through an integer would hide from the compiler that the result still points into the
// kernel argument's global buffer: flat loads instead of global / scalar ones).
template <typename T>
__device__ __forceinline__ T *view_at(T *p, size_t off)
{
    using Byte = std::conditional_t<std::is_const<T>::value, const char, char>;
    return reinterpret_cast<T *>(reinterpret_cast<Byte *>(p) + off);
}
__device__ __forceinline__ void *view_at(void *p, size_t off) { return static_cast<char *>(p) + off; }
// The v-th view's instance of a workspace pointer (v = blockIdx.y; the host passes view 0's).  nullptr stays nullptr.
template <typename T>
__device__ __forceinline__ T *view_slice(T *p, size_t view_stride)
{
    return p == nullptr ? nullptr : view_at(p, (size_t)blockIdx.y * view_stride);
}

// Carves `base` (may be nullptr to only size).  Returns total bytes.
size_t carve_workspace(void *base, int64_t n, int width, int height, int64_t max_pairs, Workspace *ws);

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);

#define GSR_HIP(call)                                      \
    do {                                                   \
        hipError_t e__ = (call);                           \
        if (e__ != hipSuccess) return gsr::hip_fail(e__, #call); \
    } while (0)

// ---- kernels' host launchers (each returns GSR_OK / GSR_ERR_HIP) --------------------------------
// cams: ws.views cameras (one frame size); ctrl_reset_words: how much of each view's control block the frame clears
int launch_preprocess(const GsrScene &scene, const GsrCamera *cams, const GsrOptions &opts, const Workspace &ws,
                      const GsrDebugOut *dbg, int ctrl_reset_words, hipStream_t s);
int launch_block_visibility(const GsrScene &scene, const GsrCamera &cam, const GsrOptions &opts, unsigned char *dead, hipStream_t s);
int launch_scene_bounds(int64_t n, const float *means, const float *log_scales, float *bounds, hipStream_t s);
int launch_sh_to_rgb(int64_t n, const float *means, const float *sh, const float cc[3], int degree, float *rgb, hipStream_t s);
int launch_cov3d(int64_t n, const float *log_scales, const float *quats, float *out, hipStream_t s);
int launch_project(int64_t n, const float *means, const float w2c[16], float *out, hipStream_t s);
int launch_cov2d(int64_t n, const float *cov3d, const float *cam_means, const float w2c[16], float fx, float fy, float limx, float limy,
                 float *out, hipStream_t s);
int launch_bbox(int64_t n, const float *screen_means, const float *cov2d, float W, float H, int64_t *out, hipStream_t s);
int launch_rasterize_gaussian(int64_t g, const int64_t *bboxes, float *screen, const float *screen_means, const float *sigmas,
                              const float *rgb, float *opacity_buffer, const float *opacity, int W, int H, hipStream_t s);

// Depth order (sort.hip): stable LSD radix sort of the depth keys; leaves V in FrameCtrl.n_visible and the sorted ids (+ packed
// rects) in val[p] / rect8[p], p = FrameCtrl.sort_passes & 1 (decided on the device from the frame's key range).  passes: how
// many to enqueue (GsrOptions.depth_sort_passes; 0 = 4).
int launch_depth_sort(const Workspace &ws, bool packed_rect, bool compact_input, int passes, hipStream_t s);
// gsr_scene_order (sort.hip): Morton-curve permutation of the gaussians, built with the radix passes of the pair sort
size_t scene_order_bytes(int64_t n);
int launch_scene_order(int64_t n, const float *means, uint32_t *perm_out, void *workspace, hipStream_t s);
// Stable radix sort of the pair arrays over key bits [first_bit, key_bits); the first pass drops keys >= drop_from and leaves the
// survivor count in *n_out.  in_buf / *result_buf: which of pkey[]/pval[] holds input / output.
int launch_pair_sort(const Workspace &ws, int in_buf, const uint32_t *n_dev, int first_bit, int key_bits, uint32_t drop_from,
                     uint32_t *n_out, int *result_buf, hipStream_t s);
// The pair sort keeps its keys as uint16_t in memory when every key (the culled row included) fits: 6 B per pair instead of 8 through
// emit, both sort passes and the range scan, all of them bound by HBM.  key_bits = TileKeying.bits_x + bits_y.
inline bool pair_keys_16bit(int key_bits) { return key_bits <= 16; }
// Pair keys: (tile row << bits_x) | tile column; culled pairs carry the row `tiles_y` and are dropped from drop_from on.
struct TileKeying {
    int bits_x, bits_y;   // of the grid the pairs are generated on: tiles, or 32x32 cells when `coarse`
    uint32_t drop_from;
    bool coarse;          // pairs are generated and sorted per 32x32 cell; the blend filters the cell lists by tile (binning.hip)
    int grid_x, grid_y;
};
TileKeying tile_keying(const Workspace &ws, const GsrOptions &opts);
inline bool rect_fits_8bit(const Workspace &ws) { return ws.tiles_x <= 256 && ws.tiles_y <= 256; }
// Can this frame bin per 32x32 cell (binning.hip)?  Needs the packed rect and pair values of 28 id bits + 4 mask bits.
inline bool coarse_capable(const Workspace &ws) { return rect_fits_8bit(ws) && ws.n <= ((int64_t)1 << 28); }
// Which tile rows a rank owns (multi-GPU sharding, GsrOptions.tile_row_begin / _step / _block): blocks of 2^bshift consecutive tile
// rows, block b is the rank's when b % step == begin.  bshift 0: the rows begin, begin + step, ...; bshift 1: pairs of rows = the
// tile rows of one 32x32 cell row.  The rank's rows in ascending order are its STRIP rows (output_layout = 2): index k <-> row_at(k).
struct RowShard {
    int begin, step, bshift;
    // how many of the rank's rows lie below tile row t (t >= 0) = strip index of its first row >= t
    __host__ __device__ __forceinline__ int rows_before(int t) const
    {
        if (step <= 1) return t;
        const int b = t >> bshift, q = b / step, r = b - q * step;
        int k = (q + (begin < r ? 1 : 0)) << bshift;         // whole blocks of the rank below block b
        if (r == begin) k += t & ((1 << bshift) - 1);        // t lies inside one of its blocks: that block's rows below t
        return k;
    }
    __host__ __device__ __forceinline__ int row_at(int k) const
    {
        if (bshift == 0) return k * step + begin;
        return ((((k >> bshift) * step) + begin) << bshift) | (k & ((1 << bshift) - 1));
    }
    __host__ __device__ __forceinline__ bool owns(int t) const { return step <= 1 || (t >> bshift) % step == begin; }
    // strip index of a row the rank owns
    __host__ __device__ __forceinline__ int index_of(int t) const { return (((t >> bshift) / step) << bshift) | (t & ((1 << bshift) - 1)); }
    // any of the rank's rows in [t0, t1)?
    __host__ __device__ __forceinline__ bool any_in(int t0, int t1) const { return t1 > t0 && (step <= 1 || rows_before(t1) > rows_before(t0)); }
};
inline RowShard row_shard_of(const GsrOptions &o)
{
    return RowShard{o.tile_row_begin, o.tile_row_step < 1 ? 1 : o.tile_row_step, o.tile_row_block == 2 ? 1 : 0};
}

// A multi-GPU shard's preprocess (preprocess.hip) hands the depth sort a compact list of (key, id, rect) records of the rank's
// visible gaussians instead of one key per gaussian.  Progressive frames (draw_limit) rank ALL gaussians the reference
// draws, so they take the whole-frame path.
inline bool shard_compact(const GsrOptions &o)
{
    if (o.tile_row_step <= 1 || o.draw_limit != 0) return false;
    if (o.shard_preprocess != 0) return o.shard_preprocess == 2;  // A/B: 1 = whole-frame kernel, 2 = three-phase kernel
    return o.tile_row_step >= 5;  // measured: 2 and 4 shards are as fast or faster through the whole-frame kernel
}
// Stage 2b: pairs of the depth-sorted gaussians -> per-tile depth-ordered lists + ranges[] (count, scan, emit, sort, ranges,
// and with coarse binning the expansion).
int launch_binning(const GsrOptions &opts, const Workspace &ws, hipStream_t s);
const uint32_t *tile_lists(const Workspace &ws, const GsrOptions &opts);  // the array ranges[] / cranges[] index after launch_binning (gaussian ids [| tile mask << 28])
bool blend_reads_cell_lists(const Workspace &ws, const GsrOptions &opts);  // coarse binning: the blend filters the 32x32-cell lists by tile bit itself
// out_image: view 0's frame, view v's lies v * out_view_stride BYTES further (out_T: single views only).  scene: where deferred
// colours are evaluated from (single views only); nullptr = what the preprocess left in the workspace's control block
int launch_blend(const GsrCamera &cam, const GsrOptions &opts, const Workspace &ws, const uint32_t *lists, void *out_image,
                 size_t out_view_stride, float *out_T, const GsrScene *scene, hipStream_t s);
int launch_blend_stats(FrameCtrl *ctrl, size_t workspace_bytes, hipStream_t s);

// ---- small device helpers -----------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Exclusive scan across a 256-thread workgroup (4 waves).  `scratch` = 8 uint32 of LDS.  Returns the
// exclusive prefix of `v`; *total receives the workgroup sum.  Contains two __syncthreads().
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *scratch, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v);
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint32_t s = scratch[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}


// The same for THREADS = 64 * WAVES threads; `scratch` = 2 * WAVES uint32 of LDS.
template <int THREADS>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *scratch, uint32_t *total)
{
    constexpr int WAVES = THREADS / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v);
    if (lane == 63) scratch[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const uint32_t s = scratch[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

}  // namespace gsr
