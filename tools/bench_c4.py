#!/usr/bin/env python3
"""tools/bench_c4.py -- BASELINE config 4 shape on one GPU: InstantNGP 1600x1060 image (1 696 000 rays), whole image and one 1/8 tile shard."""
import sys, time, math
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from nerficg_amd.instant_ngp import Camera
from nerficg_amd import parallel

dev = torch.device('cuda', 0)
model, renderer, cam, poses = bench.build_scene(dev)
W, H = 1600, 1060
big = Camera(width=W, height=H, focal_x=cam.focal_x * W / cam.width, focal_y=cam.focal_x * W / cam.width, center_x=W / 2, center_y=H / 2,
             near_plane=cam.near_plane, far_plane=cam.far_plane, background_color=cam.background_color)
nt = renderer.n_image_tiles(big)
b3, e3 = parallel.shard_range(nt, 3, 8)
for label, (b, n) in (('whole image', (0, nt)), ('shard 3 of 8', (b3, e3 - b3))):
    for i in range(2):
        out = renderer.render_image_fused(big, poses[i], tile_begin=b, n_tiles=n, return_stats=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(5):
        out = renderer.render_image_fused(big, poses[2 + i], tile_begin=b, n_tiles=n, return_stats=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    rays = n * 64
    print(f'{label}: {n} tiles, {dt * 1e3:.2f} ms, {rays / dt / 1e6:.1f} Mrays/s, rows {out["n_rows"]}, finite {bool(torch.isfinite(out["rgb"]).all())}')
