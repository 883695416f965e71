#!/usr/bin/env python3
"""tools/bench_gs_train_dp.py -- view-parallel 3DGS optimisation: replicated Gaussians, every rank rasterizes ANOTHER view per
step, per-Gaussian gradients meet in a visibility-sparse reduction (only rows seen by at least one rank travel), every rank applies the same
fused Adam step; densification statistics are summed and densify_and_prune runs redundantly from identical noise.
Launch: python -m torch.distributed.run --nproc-per-node N tools/bench_gs_train_dp.py [--backend nccl|gloo] [--gaussians P] [--iters K]
(gloo lets two ranks share one GPU to exercise the path; nccl = RCCL over xGMI on a multi-GPU node)."""
import argparse, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import torch.distributed as dist
from nerficg_amd import parallel
from nerficg_amd.gaussian_splatting import Gaussians, PerspectiveCamera, render_image_training, training_loss
from tests import scenes

ap = argparse.ArgumentParser()
ap.add_argument('--backend', default='nccl')
ap.add_argument('--gaussians', type=int, default=200_000)
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--dense', action='store_true', help='dense all-reduce of all gradient rows instead of the visibility-sparse one')
args = ap.parse_args()
local = int(os.environ.get('LOCAL_RANK', 0))
dev = torch.device('cuda', local if args.backend == 'nccl' else 0)
torch.cuda.set_device(dev)
rank, world = parallel.init_distributed(args.backend, dev)
W, H = 1297, 840
sc = scenes.gs_random_scene(args.gaussians, seed=0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
g = Gaussians(T(sc['means3D']), torch.log(T(sc['scales'])), T(sc['rotations']), torch.logit(T(sc['opacities']).clamp(1e-4, 1 - 1e-4))[:, None].contiguous(),
              T(sc['shs'][:, :1]), T(sc['shs'][:, 1:]))
g.training_setup(training_cameras_extent=4.5)
parallel.broadcast_parameters([grp['params'][0] for grp in g.optimizer.param_groups])
cam = PerspectiveCamera(W, H, 1.2 * W, 1.2 * W, background_color=torch.zeros(3, device=dev))
target = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
moved = [0, 0]


def step(i):
    pose = scenes.orbit_pose(0.8 + 0.7 * (i * world + rank), 0.35, 4.5)  # view (i * world + rank) of the orbit: another one on every rank
    out = render_image_training(g, cam, pose)
    training_loss(out['rgb'], target).backward()
    g.add_densification_stats(out['viewspace_points'], out['visibility_mask'])
    params = [grp['params'][0] for grp in g.optimizer.param_groups]
    if args.dense:
        parallel.allreduce_gradients(params, average=True)
        moved[0] += g.get_positions.shape[0]
    else:
        moved[0] += max(parallel.sparse_allreduce_gradients(params, out["visibility_mask"], average=True), 0)  # -1 = single process: nothing travels
    moved[1] += g.get_positions.shape[0]
    g.optimizer.step(); g.optimizer.zero_grad()


def densify(i):
    parallel.allreduce_densification_stats(g)
    return g.densify_and_prune(2e-4, 0.005, True, noise=parallel.synchronized_noise(2 * g.get_positions.shape[0], seed=i, device=dev))


for i in range(3):
    step(i)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
moved[:] = [0, 0]
t0 = time.perf_counter()
for i in range(args.iters):
    step(3 + i)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
dt = (time.perf_counter() - t0) / args.iters
info = densify(0)
step(1000)
flat = torch.cat([grp['params'][0].detach().reshape(-1) for grp in g.optimizer.param_groups])
sizes = torch.tensor([flat.numel()], device=dev)
ref_size = sizes.clone()
if world > 1:
    dist.broadcast(ref_size, src=0)
drift = float('inf')
if int(ref_size) == flat.numel():
    ref = flat.clone()
    if world > 1:
        dist.broadcast(ref, src=0)
    drift = float((flat - ref).abs().max())
stats = torch.tensor([dt, drift], device=dev, dtype=torch.float64)
if world > 1:
    dist.all_reduce(stats, op=dist.ReduceOp.MAX)
if rank == 0:
    print(f'3DGS view-parallel x{world} ({args.backend}, {"dense" if args.dense else "sparse"} reduction): {stats[0].item() * 1e3:.2f} ms/step for {world} views of '
          f'{args.gaussians} Gaussians -> {world * args.gaussians / stats[0].item() / 1e6:.1f} Msplats/s; gradient rows moved {moved[0] / max(moved[1], 1):.2f} of all; '
          f'densify {args.gaussians} -> {info["n_out"]}; max parameter drift between ranks after densify + 1 step {stats[1].item():.1e}')
if world > 1:
    dist.destroy_process_group()
