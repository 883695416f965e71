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

# ---- 4x4-block work lists: blocks selected by the box test of the kernels vs blocks with an active pixel (what an exact ellipse test would keep),
#      and the wave cost (a wave = 4 blocks of a quadrant, its cost per entry = 1 if any of its 4 rows takes the entry)
tot = box_blocks = act_blocks = box_wave = act_wave = slab_blocks = slab_wave = 0
for t in rng.choice(gx * gy, 60, replace=False):
    ty, tx = divmod(int(t), gx)
    r0 = int(ranges[t, 0])
    ys = torch.arange(ty * 16, ty * 16 + 16, device=xy.device); xs = torch.arange(tx * 16, tx * 16 + 16, device=xy.device)
    inside = (ys[:, None] < H) & (xs[None, :] < W)
    last = torch.zeros(16, 16, dtype=torch.long, device=xy.device)
    last[:min(16, H - ty * 16), :min(16, W - tx * 16)] = ncd[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16].long()
    n = int(last.max())
    if n == 0:
        continue
    ids = pl[r0:r0 + n]
    X, Y = xy[ids, 0], xy[ids, 1]
    c = co[ids]
    A, B, C, o = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
    dx = X[:, None, None] - xs[None, None, :].float(); dy = Y[:, None, None] - ys[None, :, None].float()
    power = -0.5 * (A[:, None, None] * dx * dx + C[:, None, None] * dy * dy) - B[:, None, None] * dx * dy
    alpha = torch.clamp(o[:, None, None] * torch.exp(power), max=0.99)
    act = (power <= 0) & (alpha >= 1.0 / 255.0) & inside[None]
    ab = act.reshape(n, 4, 4, 4, 4).any(dim=4).any(dim=2)            # (n, cy, cx)
    tau = 2 * (torch.log(255 * o) + 1e-3)
    det = A * C - B * B
    ok = (tau > 0) & (det > 0) & (A > 0) & (C > 0)
    ex = torch.where(ok, torch.sqrt(tau * C / det) * 1.001 + 0.01, torch.full_like(A, 1e9))
    ey = torch.where(ok, torch.sqrt(tau * A / det) * 1.001 + 0.01, torch.full_like(A, 1e9))
    ex = torch.where(tau > 0, ex, torch.full_like(ex, -1.0)); ey = torch.where(tau > 0, ey, torch.full_like(ey, -1.0))
    bx0 = (tx * 16 + 4 * torch.arange(4, device=xy.device)).float()
    by0 = (ty * 16 + 4 * torch.arange(4, device=xy.device)).float()
    col = (X[:, None] + ex[:, None] >= bx0[None]) & (X[:, None] - ex[:, None] <= bx0[None] + 3)
    row = (Y[:, None] + ey[:, None] >= by0[None]) & (Y[:, None] - ey[:, None] <= by0[None] + 3)
    bb = row[:, :, None] & col[:, None, :]                           # (n, cy, cx)
    # slab test: x-interval of the ellipse over the y-slab [by0, by0 + 3] (lo convex / hi concave in y: ends + the ellipse's extreme points)
    def xrange_at(yv):  # yv (n, k) absolute y -> lo, hi absolute x (nan where the row misses the ellipse)
        d = Y[:, None] - yv                                       # dy = Y - y  (same sign convention as the kernels' dx = X - x)
        disc = A[:, None] * tau[:, None] - det[:, None] * d * d
        w = torch.sqrt(disc.clamp_min(0)) / A[:, None]
        mid = X[:, None] + (B / A)[:, None] * d                     # centre of the chord in absolute x
        miss = disc < 0
        return torch.where(miss, torch.full_like(w, float('inf')), mid - w), torch.where(miss, torch.full_like(w, -float('inf')), mid + w)
    ya, yb = by0[None, :].expand(n, 4), (by0 + 3)[None, :].expand(n, 4)
    # clamp the slab to the ellipse's y-extent so that the ends are inside it
    ylo, yhi = Y[:, None] - ey[:, None], Y[:, None] + ey[:, None]
    ca, cb = torch.maximum(ya, ylo), torch.minimum(yb, yhi)
    empty = ca > cb
    loa, hia = xrange_at(ca); lob, hib = xrange_at(cb)
    lo, hi = torch.minimum(loa, lob), torch.maximum(hia, hib)
    # extreme points: leftmost at y_l, rightmost at y_r; if inside the slab the slab interval reaches X -+ ex
    yl = Y[:, None] + (B / C * torch.sqrt(tau * C / det))[:, None]; yr = Y[:, None] - (B / C * torch.sqrt(tau * C / det))[:, None]
    lo = torch.where((yl >= ca) & (yl <= cb) | (yr >= ca) & (yr <= cb), torch.minimum(lo, (X - ex)[:, None]), lo)
    hi = torch.where((yl >= ca) & (yl <= cb) | (yr >= ca) & (yr <= cb), torch.maximum(hi, (X + ex)[:, None]), hi)
    lo, hi = lo - 0.02, hi + 0.02
    sb = (~empty)[:, :, None] & (hi[:, :, None] >= bx0[None, None, :]) & (lo[:, :, None] <= bx0[None, None, :] + 3)
    sb = torch.where(ok[:, None, None], sb, bb) & bb
    assert bool((sb | ~ab).all()), 'slab test dropped an active block'
    def wave_cost(m):  # (n, cy, cx) -> number of (entry, wave) steps
        return int(m.reshape(n, 2, 2, 2, 2).any(dim=4).any(dim=2).sum())
    tot += n; box_blocks += int(bb.sum()); act_blocks += int(ab.sum()); slab_blocks += int(sb.sum())
    box_wave += wave_cost(bb); act_wave += wave_cost(ab); slab_wave += wave_cost(sb)
print(f'4x4 blocks per (tile,Gaussian) pair of the blended prefix: box test {box_blocks / tot:.2f}, slab test {slab_blocks / tot:.2f}, exact {act_blocks / tot:.2f} of 16; '
      f'wave steps per pair: box {box_wave / tot:.2f}, slab {slab_wave / tot:.2f}, exact {act_wave / tot:.2f} of 4')
