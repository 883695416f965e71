#!/usr/bin/env python3
"""tools/bench_train_graph.py [n_rays] [iters] -- the recorded InstantNGP training iteration (nerficg_amd.graphs) on the bench scene; for
rocprofv3 --kernel-trace --stats: which kernels a replay is made of."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from nerficg_amd.apex_optimizers import FusedAdam
from nerficg_amd.graphs import instant_ngp_iteration
from nerficg_amd.raygen import generate_rays

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 2200
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device('cuda', 0)
model, renderer, cam, poses = bench.build_scene(dev)
rays = [generate_rays(cam.width, cam.height, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, p, device=dev, want_direction=False) for p in poses[:2]]
origin = torch.cat([r['origin'] for r in rays]); vdir = torch.cat([r['view_direction'] for r in rays])
perm = torch.randperm(origin.shape[0], generator=torch.Generator(device='cpu').manual_seed(0)).to(dev)
target = torch.rand(origin.shape[0], 3, device=dev)
opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False, capturable=True)
from nerficg_amd.amp import GradScaler
scaler = GradScaler(init_scale=128.0, growth_interval=10 ** 9)
step = instant_ngp_iteration(model, renderer, opt, scaler, cam, n_rays, 307200, ray_pool={'origin': origin, 'view_direction': vdir, 'rgb': target},
                             fold_weight_decay=True)
batch = lambda i: perm[(i * n_rays) % (perm.numel() - n_rays):][:n_rays]
for i in range(3):
    step(ids=batch(i))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(iters):
    out = step(ids=batch(3 + i))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
print(f'recorded iteration: {dt * 1e3:.3f} ms, {int(out["rm_samples"])} samples in the last one')
