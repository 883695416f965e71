#!/usr/bin/env python3
"""tools/pose_encoder_time.py -- k_grid_encode time per pose next to the pose's viewing direction: does the encoder's cost follow the axis the
rays run along?  (hashed entries are contiguous along x: a wave whose 64 samples spread along x shares cache lines, one that spreads along y / z does not)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import bench
from nerficg_amd import _lib

dev = torch.device('cuda', 0)
model, renderer, cam, poses = bench.build_scene(dev)
for p in poses[:2]:
    renderer.render_image_fused(cam, p)
torch.cuda.synchronize()
rows = []
for i, p in enumerate(poses[:int(sys.argv[1]) if len(sys.argv) > 1 else 24]):
    r = bench.time_dominant_kernel(renderer, cam, [p], reps=2)
    fwd = np.asarray(p, dtype=np.float64)[:3, 2]
    rows.append((i, r['enc_ms'], r['launches_per_image'], fwd))
    print(f'pose {i:3d}  encode {r["enc_ms"] * 1e3:7.1f} us/launch x {r["launches_per_image"]:4.1f}  mlp {r["mlp_ms"] * 1e3:6.1f}   forward = ({fwd[0]:+.2f}, {fwd[1]:+.2f}, {fwd[2]:+.2f})')
a = np.array([[r[1], abs(r[3][0]), abs(r[3][1]), abs(r[3][2])] for r in rows])
for k, name in ((1, '|fx|'), (2, '|fy|'), (3, '|fz|')):
    print('correlation of the encode time with', name, round(float(np.corrcoef(a[:, 0], a[:, k])[0, 1]), 3))
