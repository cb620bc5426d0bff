// blend_args.h — kernel arguments shared by the two blend implementations.
#pragma once
#include "gsr_internal.h"

namespace gsr {

struct BlendArgs {
    const uint2 *ranges;
    const uint32_t *pval;
    const GaussRec *rec;
    float *out;
    float *out_T;
    uint32_t *stats;      // [launch slots][BLEND_STAT_WORDS]
    const int *order;     // tile launch order (tile_order_kernel), -1 = empty slot
    int W, H;
    int xlim, ylim;       // pixels x < xlim, y < ylim are drawn (W-1/H-1 in reference_compat: Q1)
    int tiles_x;
    int row_begin, row_step, rows;  // tile rows of this shard: row_begin + k*row_step, k in [0, rows)
    int layout;
    float early_T;
};

int launch_blend_mfma(const BlendArgs &a, unsigned grid, hipStream_t s);

}  // namespace gsr
