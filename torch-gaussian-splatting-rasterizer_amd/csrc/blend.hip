// blend.hip — stage 3: front-to-back alpha compositing of every 16x16 tile.
//
// Replaces the reference's hot loop: the driver at rasterize.py:436-446 and rasterize_gaussian
// (:255-305), which together visit ONE gaussian at a time from the Python interpreter.  Here one
// 256-thread workgroup owns one tile; wave w owns the 8x8 quadrant (w&1, w>>1) and one lane one pixel.
//
// Per tile list (depth-ordered by the two radix sorts):
//   - 256 entries at a time are staged in LDS: each thread gathers one 48-B record (3 x 16-B loads)
//     addressed by the pair value, so HBM/L2 sees 16-B vector loads and the list itself is read coalesced;
//   - each wave tests, 64 entries per instruction, whether the entry's alpha > 1/255 footprint AABB
//     touches its quadrant, and walks only the surviving bits of the ballot (scalar loop, no divergence);
//   - per pixel the reference's arithmetic: power (log2 domain, coefficients pre-scaled in preprocess),
//     alpha = min(opacity * 2^power, 0.99), contribute iff alpha > 1/255 and power <= 0 (:285-291),
//     C += alpha * T * rgb, T *= 1 - alpha (:295-303);
//   - optional early-out (opts.early_out_T > 0): a wave whose 64 pixels are all below the threshold
//     stops (wave ballot); the workgroup stops fetching once all four waves have.
// No early termination by default: the reference blends every gaussian (Q5).
//
// Roofline: algorithmic bytes = 40 per consumed entry (4 id + 36 record) + 12 per pixel
// + 8 per tile range.  At ~20 VALU issues per (pixel, entry) evaluation the kernel is VALU/exp bound,
// not HBM bound; both fractions are reported by bench.py.
#include "gsr_internal.h"

namespace gsr {

struct BlendArgs {
    const uint2 *ranges;
    const uint32_t *pval;
    const GaussRec *rec;
    float *out;
    float *out_T;
    int W, H;
    int xlim, ylim;       // pixels x < xlim, y < ylim are drawn (W-1/H-1 in reference_compat: Q1)
    int tiles_x;
    int row_begin, row_step, rows;  // tile rows of this shard: row_begin + k*row_step, k in [0, rows)
    int layout;
    float early_T;
};

__global__ __launch_bounds__(256) void blend_kernel(BlendArgs a)
{
    __shared__ float4 s0[256];
    __shared__ float4 s1[256];
    __shared__ float4 s2[256];
    __shared__ int s_done;

    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so workgroup b and b+8
    // share an L2.  Give XCD k the tile rows k, k+8, ... of the shard: x-neighbours (which share most of
    // their gaussians) hit the same L2, and heavy image regions are spread over all XCDs.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, j = bid >> 3;
    const int rows_per_xcd = (a.rows + 7) >> 3;
    const int jr = j / a.tiles_x, tx = j - jr * a.tiles_x;
    const int shard_row = xcd + 8 * jr;
    if (jr >= rows_per_xcd || shard_row >= a.rows) return;  // uniform
    const int ty = a.row_begin + shard_row * a.row_step;
    const int tile = ty * a.tiles_x + tx;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qx = tx * 16 + (wave & 1) * 8, qy = ty * 16 + (wave >> 1) * 8;
    const int px = qx + (lane & 7), py = qy + (lane >> 3);
    const float fpx = (float)px, fpy = (float)py;
    const float qx0 = (float)qx, qx1 = (float)(qx + 7), qy0 = (float)qy, qy1 = (float)(qy + 7);

    const uint2 range = a.ranges[tile];
    float T = 1.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f;
    bool wave_done = false;
    if (tid == 0) s_done = 0;

    for (uint32_t batch = range.x; batch < range.y; batch += 256) {
        __syncthreads();  // previous batch fully consumed (and s_done initialised)
        if (a.early_T > 0.0f && s_done == 4) break;  // uniform: every wave saturated
        const uint32_t i = batch + tid;
        if (i < range.y) {
            const GaussRec *r = a.rec + a.pval[i];
            s0[tid] = r->r0;
            s1[tid] = r->r1;
            s2[tid] = r->r2;
        }
        __syncthreads();
        if (wave_done) continue;
        const int nb = min(256u, range.y - batch);
        for (int chunk = 0; chunk < nb; chunk += 64) {
            const int e = chunk + lane;
            bool hit = false;
            if (e < nb) {
                const float4 g0 = s0[e];
                const float4 g2 = s2[e];
                hit = (g0.x + g2.y >= qx0) & (g0.x - g2.y <= qx1) & (g0.y + g2.z >= qy0) & (g0.y - g2.z <= qy1);
            }
            unsigned long long m = __ballot(hit);
            while (m) {
                const int k = chunk + (__ffsll((long long)m) - 1);
                m &= m - 1;
                const float4 g0 = s0[k];  // wave-uniform address: LDS broadcast
                const float4 g1 = s1[k];
                const float g2x = s2[k].x;
                const float dx = g0.x - fpx, dy = g0.y - fpy;
                const float power = dx * (g0.z * dx + g0.w * dy) + (g1.x * dy) * dy;  // log2 domain
                float alpha = g1.y * __builtin_amdgcn_exp2f(power);
                alpha = fminf(alpha, GSR_MAX_ALPHA);
                const bool valid = (alpha > GSR_MIN_ALPHA) & (power <= 0.0f);
                alpha = valid ? alpha : 0.0f;
                const float w = alpha * T;
                Cr = fmaf(w, g1.z, Cr);
                Cg = fmaf(w, g1.w, Cg);
                Cb = fmaf(w, g2x, Cb);
                T = T * (1.0f - alpha);
            }
            if (a.early_T > 0.0f && __all(T < a.early_T)) {
                wave_done = true;
                if (lane == 0) atomicAdd(&s_done, 1);
                break;
            }
        }
    }

    if (px < a.W && py < a.H) {
        const bool drawn = px < a.xlim && py < a.ylim;  // Q1: last column / row stay black, T stays 1
        const float r = drawn ? Cr : 0.0f, g = drawn ? Cg : 0.0f, b = drawn ? Cb : 0.0f;
        size_t o;
        if (a.layout == 0) o = ((size_t)py * a.W + px) * 3;                                           // image [H,W,3]
        else if (a.layout == 1) o = ((size_t)px * a.H + py) * 3;                                      // screen [W,H,3]
        else o = ((size_t)(((ty - a.row_begin) / a.row_step) * 16 + (py - ty * 16)) * a.W + px) * 3;  // strip
        a.out[o] = r; a.out[o + 1] = g; a.out[o + 2] = b;
        if (a.out_T) {
            const size_t ot = a.layout == 1 ? (size_t)px * a.H + py
                            : a.layout == 0 ? (size_t)py * a.W + px
                                            : (size_t)(((ty - a.row_begin) / a.row_step) * 16 + (py - ty * 16)) * a.W + px;
            a.out_T[ot] = drawn ? T : 1.0f;
        }
    }
}

int launch_blend(const GsrCamera &cam, const GsrOptions &opts, const Workspace &ws, int pair_buf, float *out_image,
                 float *out_T, hipStream_t s)
{
    BlendArgs a;
    a.ranges = ws.ranges;
    a.pval = ws.pval[pair_buf];
    a.rec = ws.rec;
    a.out = out_image;
    a.out_T = out_T;
    a.W = cam.width; a.H = cam.height;
    a.xlim = opts.reference_compat ? cam.width - 1 : cam.width;
    a.ylim = opts.reference_compat ? cam.height - 1 : cam.height;
    a.tiles_x = ws.tiles_x;
    a.row_step = opts.tile_row_step < 1 ? 1 : opts.tile_row_step;
    a.row_begin = opts.tile_row_begin;
    a.rows = a.row_begin < ws.tiles_y ? (ws.tiles_y - a.row_begin + a.row_step - 1) / a.row_step : 0;
    a.layout = opts.output_layout;
    a.early_T = opts.early_out_T;
    if (a.rows <= 0 || a.tiles_x <= 0) return GSR_OK;
    const int rows_per_xcd = (a.rows + 7) / 8;
    const unsigned grid = 8u * (unsigned)rows_per_xcd * (unsigned)a.tiles_x;
    hipLaunchKernelGGL(blend_kernel, dim3(grid), dim3(256), 0, s, a);
    GSR_HIP(hipGetLastError());
    return GSR_OK;
}

}  // namespace gsr
