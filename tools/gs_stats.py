#!/usr/bin/env python3
"""tools/gs_stats.py -- developer statistics of the 3DGS bench scene: tile list lengths, blended prefix lengths, rect sizes."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
gs = bench.build_gs_scene(torch.device('cuda', 0), n)
t = {k: v.detach().requires_grad_(True) for k, v in gs['tensors'].items()}
color, radii = gs['rast'](means3D=t['means3D'], means2D=torch.zeros_like(t['means3D']), opacities=t['opacities'], shs=t['shs'], scales=t['scales'],
                          rotations=t['rotations'])
names = ['means3D', 'sh', 'col', 'sc', 'rot', 'cov', 'radii', 'points_xy', 'conic_opacity', 'rgb', 'clamped', 'cov3D', 'point_list', 'ranges', 'n_contrib', 'final_T']
sv = dict(zip(names, color.grad_fn.saved_tensors))
H, W = gs['cam']['height'], gs['cam']['width']
ranges = sv['ranges'].cpu().numpy().astype(np.int64).reshape(-1, 2)
lens = ranges[:, 1] - ranges[:, 0]
nc = sv['n_contrib'].cpu().numpy().reshape(H, W).astype(np.int64)
gy, gx = (H + 15) // 16, (W + 15) // 16
pad = np.zeros((gy * 16, gx * 16), np.int64); pad[:H, :W] = nc
tiles = pad.reshape(gy, 16, gx, 16).transpose(0, 2, 1, 3).reshape(gy * gx, 256)
tmax = tiles.max(1)
strips = tiles.reshape(gy * gx, 4, 64).max(2)
print('tiles', gx * gy, 'instances', lens.sum(), 'mean list', lens.mean(), 'max list', lens.max())
print('mean n_contrib per pixel', nc.mean(), ' mean tile max(last)', tmax.mean(), ' mean strip max(last)', strips.mean())
print('sum over tiles of max(last) =', tmax.sum(), ' (x4 waves =', 4 * tmax.sum(), ') sum strips', strips.sum(), ' pixel-steps', nc.sum())
r = sv['radii'].cpu().numpy()
print('visible', (r > 0).sum(), 'radius px: mean', r[r > 0].mean(), 'median', np.median(r[r > 0]), 'p90', np.percentile(r[r > 0], 90), 'max', r.max())
