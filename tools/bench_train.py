#!/usr/bin/env python3
"""tools/bench_train.py -- InstantNGP training-iteration throughput through the drop-in modules (Trainer.py:79-94 sequence)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import bench
from nerficg_amd.raygen import generate_rays

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 2200
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda', 0)
model, renderer, cam, poses = bench.build_scene(dev)
rays = [generate_rays(cam.width, cam.height, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, p, device=dev, want_direction=False) for p in poses[:4]]
origin = torch.cat([r['origin'] for r in rays]); vdir = torch.cat([r['view_direction'] for r in rays])
g = torch.Generator(device='cpu').manual_seed(0)
perm = torch.randperm(origin.shape[0], generator=g).to(dev)
from nerficg_amd.apex_optimizers import FusedAdam
opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)  # Trainer.py:35
from nerficg_amd.amp import GradScaler
scaler = GradScaler(init_scale=128.0, growth_interval=10 ** 9)
target = torch.rand(origin.shape[0], 3, device=dev)

from nerficg_amd.instant_ngp import InstantNGPLoss
from nerficg_amd.ngp import gather_ray_batch
criterion = InstantNGPLoss(model)

def step(i):
    ids = perm[(i * n_rays) % (perm.numel() - n_rays):][:n_rays]
    batch = gather_ray_batch(ids, origin, vdir, target)       # RayPoolSampler.get: ray_pool[ids], every field in one launch
    with torch.amp.autocast('cuda'):
        bg = torch.rand(3, device=dev)
        out = renderer.render_rays(batch['origin'], batch['view_direction'], cam, train_mode=True, custom_bg_color=bg)
        loss = criterion(out, batch, bg)
    scaler.scale(loss).backward()
    scaler.step(opt); scaler.update(); opt.zero_grad()
    return int(out['rm_samples'].item())

for i in range(3):
    s = step(i)
if len(sys.argv) > 3:   # per-kernel times of the query forward / backward (stage timer), early and late in the run
    from nerficg_amd import _lib
    done = 3
    for label, skip in (('early', 0), ('late', int(sys.argv[3]))):
        for i in range(skip):
            step(done); done += 1
        with _lib.stage_timer() as st:
            for i in range(10):
                step(done); done += 1
        print(label, ' '.join(f'{k}={tot / 10 * 1e3:.1f}' for k, (tot, _) in sorted(st.by_name().items())))
torch.cuda.synchronize(); t0 = time.perf_counter(); tot = 0
for i in range(iters):
    tot += step(3 + i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
print(f'train iteration: {dt * 1e3:.2f} ms, {n_rays} rays, {tot / iters:.0f} samples/iter -> {n_rays / dt / 1e6:.3f} Mrays/s, {tot / iters / dt / 1e6:.1f} Msamples/s')
