#!/usr/bin/env python3
"""tools/bench_gs_step.py [n] [iters] [fuse] -- the 3DGS optimisation step of bench.py's dp_training.gs leg on its own (for rocprofv3)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import bench
from tests import scenes
from nerficg_amd.gaussian_splatting import Gaussians, PerspectiveCamera, render_image_training, training_loss

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda', 0)
sc = scenes.gs_random_scene(n, seed=0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
g = Gaussians(T(sc['means3D']), torch.log(T(sc['scales'])), T(sc['rotations']), torch.logit(T(sc['opacities']).clamp(1e-4, 1 - 1e-4))[:, None].contiguous(),
              T(sc['shs'][:, :1]), T(sc['shs'][:, 1:]))
g.training_setup(training_cameras_extent=4.5)
g.fuse_rest_step = len(sys.argv) > 3 and sys.argv[3] == 'fuse'      # the f_rest Adam step inside the preprocessing backward
cam = PerspectiveCamera(bench.GS_W, bench.GS_H, 1.2 * bench.GS_W, 1.2 * bench.GS_W, background_color=torch.zeros(3, device=dev))
target = torch.rand(3, bench.GS_H, bench.GS_W, device=dev)
poses = [torch.from_numpy(np.asarray(scenes.orbit_pose(0.8 + 0.7 * i, 0.35, 4.5), dtype=np.float32)).to(dev) for i in range(8)]

def step(i):
    out = render_image_training(g, cam, poses[i % 8])
    training_loss(out['rgb'], target).backward()
    with torch.no_grad():
        g.add_densification_stats(out['viewspace_points'], out['radii'])
    g.optimizer.step(); g.optimizer.zero_grad()

for i in range(3):
    step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(iters):
    step(i)
torch.cuda.synchronize()
print(f'3DGS step (fuse_rest_step={g.fuse_rest_step}): {(time.perf_counter() - t0) / iters * 1e3:.3f} ms')
