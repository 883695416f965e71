#!/usr/bin/env python3
"""tools/bench_train_fused.py RAYS ITERS [prefetch=1] [graph=1] -- the fused InstantNGP training iteration (nerficg_amd.ngp_trainer) on the bench scene."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from nerficg_amd.raygen import generate_rays

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 2200
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
prefetch = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
graph = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
accumulate = bool(int(sys.argv[5])) if len(sys.argv) > 5 else True
fused_step = bool(int(sys.argv[6])) if len(sys.argv) > 6 else True
dev = torch.device('cuda', 0)
model, renderer, cam, poses = bench.build_scene(dev)
rays = [generate_rays(cam.width, cam.height, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, p, device=dev, want_direction=False) for p in poses[:2]]
origin = torch.cat([r['origin'] for r in rays]); vdir = torch.cat([r['view_direction'] for r in rays])
perm = torch.randperm(origin.shape[0], generator=torch.Generator(device='cpu').manual_seed(0)).to(dev)
target = torch.rand(origin.shape[0], 3, device=dev)
from nerficg_amd.apex_optimizers import FusedAdam
from nerficg_amd.amp import GradScaler
from nerficg_amd.ngp_trainer import FusedTrainingIteration
opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False, capturable=True)
scaler = GradScaler(init_scale=128.0, growth_interval=10 ** 9)
capacity = (int(1.15 * 264000 * n_rays / 2200) + 4095) // 4096 * 4096
it = FusedTrainingIteration(model, renderer, opt, scaler, cam, {'origin': origin, 'view_direction': vdir, 'rgb': target}, n_rays, capacity, order=perm,
                            prefetch=prefetch, graph=graph, fused_step=fused_step)
for i in range(4):
    out = it()
for block in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    marched = torch.zeros((), dtype=torch.int64, device=dev)
    for i in range(iters):
        out = it()
        if accumulate:
            marched += out['rm_samples']
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    print(f'fused iteration (prefetch={prefetch}, graph={graph}, accumulate={accumulate}, fused_step={fused_step}), block {block}: {dt * 1e3:.3f} ms, {int(marched) / iters:.0f} samples / iteration, loss {float(out["loss"]):.4f}, '
          f'overflow {int(out["sample_overflow"])}, graphs {len(it._graphs)}')
