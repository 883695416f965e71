#!/usr/bin/env python3
"""tools/exp_gs.py [n_gaussians] [reps] -- the 3DGS bench frame forward + backward `reps` times (for rocprofv3 --kernel-trace --stats A/B runs of
library variants: NRC_LIB_PATH=_ab/x.so), prints wall ms per forward and per forward+backward and gradient checksums."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda', 0)
gs = bench.build_gs_scene(dev, n)
res = bench.time_gs(gs, reps=reps)
t = {k: v.detach().requires_grad_(True) for k, v in gs['tensors'].items()}
m2d = torch.zeros_like(t['means3D'], requires_grad=True)
color, radii = gs['rast'](means3D=t['means3D'], means2D=m2d, opacities=t['opacities'], shs=t['shs'], scales=t['scales'], rotations=t['rotations'])
g = torch.rand(color.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
color.backward(g)
sums = ' '.join(f'{k}={float(v.grad.double().abs().sum()):.6e}' for k, v in list(t.items()) + [('means2D', m2d)])
print(f'n={n} fwd {res["ms_fwd"]} ms  fwd+bwd {res["ms_fwd_bwd"]} ms  img {float(color.double().sum()):.6f}  {sums}')
