#!/usr/bin/env python3
"""tools/exp_frame_time.py [reps] -- whole-frame time of render_image_fused over the first poses of the bench orbit (A/B builds: NRC_LIB_PATH)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda', 0)
model, renderer, cam, poses = bench.build_scene(dev)
for p in poses[:3]:
    renderer.render_image_fused(cam, p)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    for p in poses[:8]:
        renderer.render_image_fused(cam, p)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / (reps * 8) * 1e3
print(f'{ms:.3f} ms per frame = {800 * 800 / ms / 1e3:.2f} Mrays/s (8 poses x {reps})')
