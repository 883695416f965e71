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

# ---- activity of the backward walk: per (tile, Gaussian of the blended prefix), how many 8x8 quadrants / pixels take part
pl = sv['point_list'].long(); xy = sv['points_xy']; co = sv['conic_opacity']
ncd = sv['n_contrib'].reshape(H, W)
rng = np.random.default_rng(0)
tot_pairs = act_quads = act_pix = any_tile = 0
for t in rng.choice(gx * gy, 60, replace=False):
    ty, tx = divmod(int(t), gx)
    r0, r1 = int(ranges[t, 0]), int(ranges[t, 1])
    ys = torch.arange(ty * 16, min(ty * 16 + 16, H), device=xy.device); xs = torch.arange(tx * 16, min(tx * 16 + 16, W), device=xy.device)
    last = ncd[ys][:, xs].long()                                    # (h, w)
    n = int(last.max())
    if n == 0:
        continue
    ids = pl[r0:r0 + n]
    dx = xy[ids, 0][:, None, None] - xs[None, None, :].float(); dy = xy[ids, 1][:, None, None] - ys[None, :, None].float()
    c = co[ids]
    power = -0.5 * (c[:, 0, None, None] * dx * dx + c[:, 2, None, None] * dy * dy) - c[:, 1, None, None] * dx * dy
    alpha = torch.clamp(c[:, 3, None, None] * torch.exp(power), max=0.99)
    act = (power <= 0) & (alpha >= 1.0 / 255.0) & (torch.arange(n, device=xy.device)[:, None, None] < last[None])
    hh, ww = act.shape[1], act.shape[2]
    pad = torch.zeros(n, 16, 16, dtype=torch.bool, device=xy.device); pad[:, :hh, :ww] = act
    q = pad.reshape(n, 2, 8, 2, 8).any(dim=4).any(dim=2)             # (n, 2, 2) quadrants
    tot_pairs += n; act_quads += int(q.sum()); act_pix += int(act.sum()); any_tile += int(q.reshape(n, 4).any(dim=1).sum())
print(f'backward walk: {tot_pairs} (tile,Gaussian) pairs sampled; active quadrants per pair {act_quads / tot_pairs:.2f} of 4; '
      f'pairs with any active quadrant {any_tile / tot_pairs:.2f}; active pixels per pair {act_pix / tot_pairs:.1f} of 256')
