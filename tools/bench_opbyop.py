#!/usr/bin/env python3
"""tools/bench_opbyop.py -- times the reference-shaped op-by-op image loop (render_image) next to the fused pipeline."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
dev = torch.device('cuda', 0)
model, renderer, cam, poses = bench.build_scene(dev)
for name, fn in (('fused', lambda p: renderer.render_image_fused(cam, p)), ('op-by-op loop', lambda p: renderer.render_image(cam, p))):
    fn(poses[0]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3):
        fn(poses[1 + i])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f'{name}: {dt * 1e3:.1f} ms / image = {800 * 800 / dt / 1e6:.2f} Mrays/s')
