#!/usr/bin/env python3
"""tools/exp_fused_host.py -- where a call of the fused training iteration spends its time: host time of __call__, GPU time by events, cProfile."""
import sys, time, cProfile, pstats
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from nerficg_amd.raygen import generate_rays
dev = torch.device('cuda', 0)
model, renderer, cam, poses = bench.build_scene(dev)
rays = [generate_rays(cam.width, cam.height, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, p, device=dev, want_direction=False) for p in poses[:2]]
origin = torch.cat([r['origin'] for r in rays]); vdir = torch.cat([r['view_direction'] for r in rays])
perm = torch.randperm(origin.shape[0], generator=torch.Generator(device='cpu').manual_seed(0)).to(dev)
target = torch.rand(origin.shape[0], 3, device=dev)
from nerficg_amd.apex_optimizers import FusedAdam
from nerficg_amd.amp import GradScaler
from nerficg_amd.ngp_trainer import FusedTrainingIteration
saved = [p.detach().clone() for p in model.parameters()]
for prefetch, graph, accumulate in ((False, True, False), (False, False, False), (True, True, False), (True, False, False), (False, False, True), (True, False, True), (False, True, True)):
    with torch.no_grad():
        for p_, q_ in zip(model.parameters(), saved):
            p_.copy_(q_)
    opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False, capturable=True)
    scaler = GradScaler(init_scale=128.0, growth_interval=10 ** 9)
    it = FusedTrainingIteration(model, renderer, opt, scaler, cam, {'origin': origin, 'view_direction': vdir, 'rgb': target}, 2200, 307200, order=perm, prefetch=prefetch, graph=graph)
    for i in range(5):
        it()
    torch.cuda.synchronize()
    n = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    marched = torch.zeros((), dtype=torch.int64, device=dev)
    for i in range(n):
        out = it()
        if accumulate:
            marched += out['rm_samples']
    e1.record(); t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print(f'prefetch={prefetch} graph={graph} accumulate={accumulate}: host enqueue {t_host / n * 1e6:.0f} us / call, events {e0.elapsed_time(e1) / n * 1e3:.0f} us / call, wall {t_all / n * 1e6:.0f} us / call')
    if False:
        pr = cProfile.Profile(); pr.enable()
        for i in range(50):
            it()
        pr.disable(); torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats('tottime').print_stats(12)
