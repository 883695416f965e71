#!/usr/bin/env python3
"""tools/ab_gs.py -- prints checksums of the 3DGS forward image, radii and all gradients on the synthetic scene, to compare two builds
(NRC_LIB_PATH=<other .so> python tools/ab_gs.py N).  The forward is order-deterministic: image / n_contrib must agree bit for bit."""
import hashlib, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device('cuda', 0)
gs = bench.build_gs_scene(dev, n)
t = {k: v.detach().requires_grad_(True) for k, v in gs['tensors'].items()}
m2d = torch.zeros_like(t['means3D'], requires_grad=True)
color, radii = gs['rast'](means3D=t['means3D'], means2D=m2d, opacities=t['opacities'], shs=t['shs'], scales=t['scales'], rotations=t['rotations'])
g = torch.rand(color.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
color.backward(g)
h = lambda x: hashlib.sha1(x.detach().cpu().numpy().tobytes()).hexdigest()[:12]  # noqa: E731
print('image', h(color), 'radii', h(radii), 'sum', float(color.double().sum()))
for k, v in list(t.items()) + [('means2D', m2d)]:
    print(f'grad {k:10s} sum {float(v.grad.double().sum()):+.9e} abs {float(v.grad.double().abs().sum()):.9e}')
