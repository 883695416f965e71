#!/usr/bin/env python3
"""tools/bench_gs_train.py -- one 3DGS optimisation step on the synthetic scene through the drop-in modules: rasterize (fwd) ->
0.8 L1 + 0.2 (1 - SSIM) (Loss.py:14-15) -> backward -> FusedAdam over the six parameter groups (Model.py:123-136)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from nerficg_amd.apex_optimizers import FusedAdam
from nerficg_amd.gaussian_splatting import training_loss

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda', 0)
gs = bench.build_gs_scene(dev, n)
t = {k: torch.nn.Parameter(v.clone()) for k, v in gs['tensors'].items()}
groups = [{'params': [t['means3D']], 'lr': 1.6e-4, 'name': 'positions'}, {'params': [t['shs']], 'lr': 2.5e-3, 'name': 'features'},
          {'params': [t['opacities']], 'lr': 0.05, 'name': 'opacities'}, {'params': [t['scales']], 'lr': 5e-3, 'name': 'scales'},
          {'params': [t['rotations']], 'lr': 1e-3, 'name': 'rotations'}]
opt = FusedAdam(groups, lr=0.0, eps=1e-15, adam_w_mode=False)
target = torch.rand(3, bench.GS_H, bench.GS_W, device=dev)


def step():
    m2d = torch.zeros_like(t['means3D'], requires_grad=True)
    color, radii = gs['rast'](means3D=t['means3D'], means2D=m2d, opacities=t['opacities'], shs=t['shs'], scales=t['scales'], rotations=t['rotations'])
    loss = training_loss(color, target)
    loss.backward()
    opt.step(); opt.zero_grad()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters):
    loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
print(f'3DGS training step: {dt * 1e3:.2f} ms for {n} Gaussians at {bench.GS_W}x{bench.GS_H} -> {n / dt / 1e6:.1f} Msplats/s, loss {loss.item():.4f}')
