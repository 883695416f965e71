#!/usr/bin/env python3
"""tools/exp_pose_shapes.py [N] -- k_grid_encode time per pose for every brick shape of a wave (nrc_ngp_set_encoder_shape) next to the pose's axes:
what a per-pose choice of the shape could buy, and which geometric quantity predicts the best shape."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import bench
from nerficg_amd import _lib

dev = torch.device('cuda', 0)
model, renderer, cam, poses = bench.build_scene(dev)
lib = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shapes = [(3, 1), (3, 0), (2, 1), (1, 2), (0, 3), (1, 3), (3, 2), (2, 2), (2, 0), (0, 2), (1, 1), (2, 3), (3, 3) if False else (1, 1)]
shapes = sorted(set(shapes))
for p in poses[:2]:
    renderer.render_image_fused(cam, p)
torch.cuda.synchronize()
table = []
for i, p in enumerate(poses[:n]):
    row = {}
    for (lx, ly) in shapes:
        _lib.check(lib.nrc_ngp_set_encoder_shape(lx, ly), 'set_encoder_shape')
        row[(lx, ly)] = bench.time_dominant_kernel(renderer, cam, [p], reps=2)['enc_ms'] * 1e3
    lib.nrc_ngp_set_encoder_shape(-1, -1)
    R = np.asarray(p, dtype=np.float64)[:3, :3]
    best = min(row, key=row.get)
    table.append((i, row, best, R))
    print(f'pose {i:2d}  ' + ' '.join(f'{lx}{ly}:{row[(lx, ly)]:5.0f}' for lx, ly in shapes) + f'   best {best[0]}{best[1]} {row[best]:.0f} (default 31: {row[(3, 1)]:.0f})'
          f'   |right.x| {abs(R[0, 0]):.2f} |down.x| {abs(R[0, 1]):.2f} |fwd.x| {abs(R[0, 2]):.2f}')
d = np.mean([t[1][(3, 1)] for t in table]); b = np.mean([t[1][t[2]] for t in table])
print(f'mean us per launch: default {d:.1f}, best shape per pose {b:.1f} ({(1 - b / d) * 100:.1f} % less)')
for s in shapes:
    print(s, 'mean', round(float(np.mean([t[1][s] for t in table])), 1))
