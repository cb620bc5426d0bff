"""CPU baseline port of the reference's per-gaussian torch loop.  TEST/BENCH INFRASTRUCTURE ONLY.

What bench.py times as `cpu_baseline` ("port"): the same torch-CPU operations, per gaussian, that the
reference issues from its Python `for` loop — the skip guard of rasterize.py:441 and the body of
`rasterize_gaussian` (rasterize.py:255-305: arange/meshgrid over the rect, power, alpha, validity mask,
advanced-index read-modify-write of screen and transmittance).  The per-gaussian inputs come from the C
oracle's preprocess (oracle/gsr_oracle.c), which is pinned to the reference's own intermediates.

The product never imports this module.
"""
from __future__ import annotations

import time
from typing import Dict, Sequence, Tuple

import numpy as np
import torch

MAX_ALPHA = 0.99
MIN_ALPHA = 1 / 255


def _as_tensors(pre: Dict[str, np.ndarray]):
    return (torch.from_numpy(pre["pixel_bboxes"]), torch.from_numpy(pre["screen_means"]), torch.from_numpy(pre["sigmas"]),
            torch.from_numpy(pre["rgb"]), torch.from_numpy(pre["opacity"]))


def splat_one(g: int, rects, centres, conics, colours, opac, canvas, trans) -> None:
    """One iteration of the reference's loop body (rasterize.py:255-305) for gaussian g, in place."""
    xs = torch.arange(int(rects[g, 0]), int(rects[g, 2]))
    ys = torch.arange(int(rects[g, 1]), int(rects[g, 3]))
    gx, gy = torch.meshgrid(xs, ys, indexing="ij")
    pix = torch.stack((gx, gy), dim=-1).reshape(-1, 2)
    d = centres[g] - pix
    cx, cy, cxy = conics[g]
    power = -0.5 * (cx * d[:, 0] ** 2 + cy * d[:, 1] ** 2) - cxy * d[:, 0] * d[:, 1]
    alpha = torch.clamp_max(opac[g] * torch.exp(power), MAX_ALPHA).float()
    keep = (alpha > MIN_ALPHA) & (power <= 0)
    kp = pix[keep]
    a = alpha[keep]
    t = trans[kp[:, 0], kp[:, 1]]
    canvas[kp[:, 0], kp[:, 1], :] += a[:, None] * colours[g] * t[:, None]
    trans[kp[:, 0], kp[:, 1]] = t * (1 - a)


def render(pre: Dict[str, np.ndarray], order: Sequence[int], width: int, height: int) -> Tuple[np.ndarray, int]:
    """Full loop (small scenes only): returns (image [H,W,3], drawn)."""
    rects, centres, conics, colours, opac = _as_tensors(pre)
    canvas = torch.zeros((width, height, 3))
    trans = torch.ones((width, height))
    area = (rects[:, 2] - rects[:, 0]) * (rects[:, 3] - rects[:, 1])
    drawn = 0
    for g in order:
        g = int(g)
        if area[g] == 0 or torch.any(conics[g] == 0):  # rasterize.py:441
            continue
        splat_one(g, rects, centres, conics, colours, opac, canvas, trans)
        drawn += 1
    return canvas.transpose(1, 0).numpy(), drawn


def timed_sample(pre: Dict[str, np.ndarray], order: Sequence[int], width: int, height: int, budget_s: float = 15.0,
                 max_gaussians: int = 20_000, seed: int = 0) -> Dict[str, float]:
    """Time the loop on a uniform random subsample of the depth order under a wall budget.

    Every loop iteration is either the skip guard alone (rasterize.py:441) or guard + body; which one is known
    exactly for all gaussians from the preprocessed arrays.  A shuffled uniform subsample of each class is timed
    (the cost of an iteration does not depend on what was blended before it), and the frame time is
    n_drawn * mean(body) + n_skipped * mean(guard) — reported as an extrapolation, not a measurement."""
    rects, centres, conics, colours, opac = _as_tensors(pre)
    canvas = torch.zeros((width, height, 3))
    trans = torch.ones((width, height))
    area = (rects[:, 2] - rects[:, 0]) * (rects[:, 3] - rects[:, 1])
    skip_all = ((area == 0) | (conics == 0).any(dim=1)).numpy()
    order = np.asarray(order)
    n = len(order)
    n_skip_total = int(skip_all[order].sum())
    n_draw_total = n - n_skip_total
    rng = np.random.default_rng(seed)
    sample = rng.permutation(order)[: max_gaussians]
    t_guard = t_body = 0.0
    n_skip = n_draw = 0
    start = time.perf_counter()
    for g in sample:
        g = int(g)
        t0 = time.perf_counter()
        skip = bool(area[g] == 0 or torch.any(conics[g] == 0))
        t1 = time.perf_counter()
        if skip:
            t_guard += t1 - t0
            n_skip += 1
        else:
            splat_one(g, rects, centres, conics, colours, opac, canvas, trans)
            t_body += time.perf_counter() - t0
            n_draw += 1
        if time.perf_counter() - start > budget_s:
            break
    per_draw = t_body / max(n_draw, 1)
    per_skip = t_guard / max(n_skip, 1)
    est = n_draw_total * per_draw + n_skip_total * per_skip
    return dict(sampled=n_skip + n_draw, drawn=n_draw, seconds=time.perf_counter() - start, s_per_drawn=per_draw,
                s_per_skipped=per_skip, extrapolated_frame_s=est, total_iterations=n, total_drawn=n_draw_total)
