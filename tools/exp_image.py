"""Experiment: whole-image time of the fused InstantNGP pipeline over 20 poses (for library variants via NRC_LIB_PATH)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
dev = torch.device('cuda', 0)
model, renderer, cam, poses = bench.build_scene(dev)
for i in range(5):
    renderer.render_image_fused(cam, poses[i])
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20):
    renderer.render_image_fused(cam, poses[5 + i])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f'{dt * 1e3:.3f} ms per image = {800 * 800 / dt / 1e6:.2f} Mrays/s')
