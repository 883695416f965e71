#!/usr/bin/env python3
"""tools/gs_host_cost.py -- what a rasterizer call costs on the HOST: a scene so small that the GPU is never the bottleneck (2 000 Gaussians),
timed per call, forward only and forward + backward, plus cProfile's top entries of the forward."""
import sys, time, cProfile, pstats
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench

dev = torch.device('cuda', 0)
gs = bench.build_gs_scene(dev, 2000)
t = gs['tensors']
m2d = torch.zeros_like(t['means3D'])
call = lambda: gs['rast'](means3D=t['means3D'], means2D=m2d, opacities=t['opacities'], shs=t['shs'], scales=t['scales'], rotations=t['rotations'])
with torch.no_grad():
    for _ in range(20):
        call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(500):
        call()
    torch.cuda.synchronize()
    print('forward, host-bound: %.1f us per call' % ((time.perf_counter() - t0) / 500 * 1e6))
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300):
        call()
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(14)
for k in ('means3D', 'opacities', 'shs', 'scales', 'rotations'):
    t[k].requires_grad_(True)
g = torch.rand(3, bench.GS_H, bench.GS_W, device=dev)
def step():
    c, _ = call()
    c.backward(g)
    for k in ('means3D', 'opacities', 'shs', 'scales', 'rotations'):
        t[k].grad = None
for _ in range(20):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(300):
    step()
torch.cuda.synchronize()
print('forward + backward, host-bound: %.1f us per step' % ((time.perf_counter() - t0) / 300 * 1e6))
