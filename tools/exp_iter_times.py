#!/usr/bin/env python3
"""tools/exp_iter_times.py -- GPU time of every one of the first N training iterations (events), fused iteration vs the recorded op-by-op iteration."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from nerficg_amd.raygen import generate_rays
dev = torch.device('cuda', 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 80
model, renderer, cam, poses = bench.build_scene(dev)
rays = [generate_rays(cam.width, cam.height, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, p, device=dev, want_direction=False) for p in poses[:2]]
origin = torch.cat([r['origin'] for r in rays]); vdir = torch.cat([r['view_direction'] for r in rays])
perm = torch.randperm(origin.shape[0], generator=torch.Generator(device='cpu').manual_seed(0)).to(dev)
target = torch.rand(origin.shape[0], 3, device=dev)
from nerficg_amd.apex_optimizers import FusedAdam
from nerficg_amd.amp import GradScaler
from nerficg_amd.ngp_trainer import FusedTrainingIteration
from nerficg_amd.graphs import instant_ngp_iteration
saved = [p.detach().clone() for p in model.parameters()]
def restore():
    with torch.no_grad():
        for p, q in zip(model.parameters(), saved):
            p.copy_(q)
for mode in ('fused', 'recorded', 'fused_prefetch'):
    restore()
    opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False, capturable=True)
    scaler = GradScaler(init_scale=128.0, growth_interval=10 ** 9)
    pool = {'origin': origin, 'view_direction': vdir, 'rgb': target}
    if mode == 'recorded':
        step_ = instant_ngp_iteration(model, renderer, opt, scaler, cam, 2200, 307200, ray_pool=pool, fold_weight_decay=True)
        batch = lambda i: perm[(i * 2200) % (perm.numel() - 2200):][:2200]
        step = lambda i: step_(ids=batch(i))
    else:
        it = FusedTrainingIteration(model, renderer, opt, scaler, cam, pool, 2200, 307200, order=perm, prefetch=(mode == 'fused_prefetch'), graph=False)
        step = lambda i: it()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
    losses = []
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(N):
        out = step(i)
        ev[i + 1].record()
        if i % 10 == 9:
            losses.append(float(out['loss']))
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(N)]
    print(mode, 'us per iteration:', ' '.join(f'{t:.0f}' for t in ts))
    print(mode, 'loss every 10:', ' '.join(f'{l:.4f}' for l in losses))
