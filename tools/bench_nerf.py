#!/usr/bin/env python3
"""tools/bench_nerf.py -- SURVEY 8(d) C1: vanilla NeRF (configs/nerf_lego.yaml: 8 x 256 MLP, 64 coarse + 192 fine samples, near 2, far 6,
white background) through nerficg_amd.nerf, the pure-PyTorch mirror of src/Methods/NeRF -- a 64x64 lego-intrinsics image (forward) and a
1024-ray batch forward + backward.  Runs on the host cores by default (the reference's CPU path, `GLOBAL.GPU_INDICES: null`); `--device cuda`
runs the same torch code on the GPU (rocBLAS GEMMs: plain library work, no kernels of this repository involved).
Usage: python tools/bench_nerf.py [--device cpu|cuda] [--threads N]"""
import argparse, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from nerficg_amd import nerf
from tests import scenes

ap = argparse.ArgumentParser()
ap.add_argument('--device', default='cpu')
ap.add_argument('--threads', type=int, default=min(os.cpu_count(), 32))  # torch's CPU GEMMs stop scaling (and regress) far below 256 threads
args = ap.parse_args()
torch.set_num_threads(args.threads)
dev = torch.device(args.device)
torch.manual_seed(0)
coarse, fine = nerf.NeRFBlock().to(dev), nerf.NeRFBlock().to(dev)
W = H = 64
fx, fy, cx, cy = scenes.lego_intrinsics(W, H)
c2w = np.eye(4); c2w[2, 3] = -4.0
o, d, vd = (torch.from_numpy(a).to(dev) for a in scenes.numpy_rays(W, H, c2w, fx, fy, cx, cy))
bg = torch.ones(3, device=dev)
sync = torch.cuda.synchronize if dev.type == 'cuda' else (lambda: None)


def image():
    with torch.no_grad():
        return nerf.render_rays(coarse, fine, o, d, vd, 2.0, 6.0, bg, ray_batch_size=8192, n_samples_coarse_nerf=64, n_samples_nerf=192)


image() if dev.type == 'cuda' else None
sync(); t0 = time.perf_counter(); out = image(); sync(); t_img = time.perf_counter() - t0
ids = torch.randperm(W * H)[:1024].to(dev)
opt = torch.optim.Adam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
target = torch.rand(1024, 3, device=dev)


def step():
    res = nerf.render_rays(coarse, fine, o[ids], d[ids], vd[ids], 2.0, 6.0, bg, ray_batch_size=8192, n_samples_coarse_nerf=64, n_samples_nerf=192,
                           randomize_samples=True)
    loss = torch.nn.functional.mse_loss(res['rgb'], target) + torch.nn.functional.mse_loss(res['rgb_coarse'], target)
    opt.zero_grad(); loss.backward(); opt.step()


step() if dev.type == 'cuda' else None
sync(); t0 = time.perf_counter(); step(); sync(); t_step = time.perf_counter() - t0
print(f'NeRF C1 on {dev.type} ({args.threads} threads): 64x64 image {t_img:.2f} s = {W * H / t_img / 1e3:.2f} Krays/s forward; '
      f'1024-ray training step {t_step:.2f} s = {1024 / t_step / 1e3:.2f} Krays/s forward + backward (320 samples/ray, 1.19 MFLOP/sample)')
