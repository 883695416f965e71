#!/usr/bin/env python3
"""tools/gs_fwd_only.py [N] [FRAMES] [bwd] -- rasterizer frames back to back, forward only or (third argument `bwd`) forward + backward, nothing else
in the process (for tools/kseq.sh / kstats2.sh: what one frame is made of, and where the GPU idles between its kernels)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda', 0)
gs = bench.build_gs_scene(dev, n)
t = gs['tensors']
m2d = torch.zeros_like(t['means3D'])
with torch.no_grad():
    for _ in range(3):
        gs['rast'](means3D=t['means3D'], means2D=m2d, opacities=t['opacities'], shs=t['shs'], scales=t['scales'], rotations=t['rotations'])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(frames):
        gs['rast'](means3D=t['means3D'], means2D=m2d, opacities=t['opacities'], shs=t['shs'], scales=t['scales'], rotations=t['rotations'])
    torch.cuda.synchronize()
print('forward ms', (time.perf_counter() - t0) / frames * 1e3)
if len(sys.argv) > 3 and sys.argv[3] == 'bwd':
    for k in ('means3D', 'opacities', 'shs', 'scales', 'rotations'):
        t[k].requires_grad_(True)
    m2d.requires_grad_(True)
    g = torch.rand(3, bench.GS_H, bench.GS_W, device=dev)

    def step():
        color, _ = gs['rast'](means3D=t['means3D'], means2D=m2d, opacities=t['opacities'], shs=t['shs'], scales=t['scales'], rotations=t['rotations'])
        color.backward(g)
    for _ in range(3):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(frames):
        step()
    torch.cuda.synchronize()
    print('forward + backward ms', (time.perf_counter() - t0) / frames * 1e3)
