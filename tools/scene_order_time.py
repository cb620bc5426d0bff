import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import gsr_amd
from gsr_amd import renderer, synthetic, utils
packed = utils.pack_gaussians(synthetic.mip360_like(6_131_954, 361))
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = renderer.GaussianScene.from_packed(packed, spatial_order=False)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    s.sort_spatially()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"upload {1e3*(t1-t0):.1f} ms, sort_spatially {1e3*(t2-t1):.1f} ms (events: {s.order_ms:.1f} ms)")
    del s
