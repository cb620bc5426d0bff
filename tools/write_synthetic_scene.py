#!/usr/bin/env python3
"""Write a synthetic scene in the on-disk layout the reference (and bench.py --input_dir/--trained_model_path) reads:
<out>/scene/sparse/0/{cameras,images}.bin, <out>/scene/images_2/<name>.png, <out>/model/point_cloud/iteration_30000/point_cloud.ply
usage: tools/write_synthetic_scene.py OUT_DIR [n=100000] [seed=360] [W=1920] [H=1080]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gsr_amd
from gsr_amd import data_reader as dr, ply, synthetic
from PIL import Image

out = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 360
W = int(sys.argv[4]) if len(sys.argv) > 4 else 1920
H = int(sys.argv[5]) if len(sys.argv) > 5 else 1080
sparse = os.path.join(out, "scene", "sparse", "0")
os.makedirs(sparse, exist_ok=True)
fx = synthetic.pinhole_focal(W)
dr.write_intrinsics_binary(os.path.join(sparse, "cameras.bin"),
                           [dr.Camera(id=1, model="PINHOLE", width=2 * W, height=2 * H, params=np.array([2 * fx, 2 * fx, W, H], float))])
poses = synthetic.ring_cameras(25)
dr.write_extrinsics_binary(os.path.join(sparse, "images.bin"), poses)
os.makedirs(os.path.join(out, "scene", "images_2"), exist_ok=True)
for p in poses:
    Image.new("RGB", (W, H)).save(os.path.join(out, "scene", "images_2", p.name))
model = os.path.join(out, "model", "point_cloud", "iteration_30000")
os.makedirs(model, exist_ok=True)
ply.write_gaussians_ply(os.path.join(model, "point_cloud.ply"), synthetic.mip360_like(n, seed))
print(os.path.join(out, "scene"), os.path.join(out, "model"))
