#!/usr/bin/env python3
"""tools/bench_train_dp.py -- data-parallel InstantNGP training iteration: every rank draws the SAME seeded batch of ray
ids, renders its share ray_ids[rank::world], the encoding / MLP gradients are averaged with one bucketed reduce-scatter + all-gather,
every rank applies the same Adam step.  Launch: python -m torch.distributed.run --nproc-per-node N tools/bench_train_dp.py [--backend nccl|gloo]
(gloo lets two ranks share one GPU to exercise the path; nccl = RCCL over xGMI on a multi-GPU node)."""
import argparse, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.distributed as dist
import bench
from nerficg_amd import parallel
from nerficg_amd.apex_optimizers import FusedAdam
from nerficg_amd.raygen import generate_rays

ap = argparse.ArgumentParser()
ap.add_argument('--backend', default='nccl')
ap.add_argument('--rays', type=int, default=2200, help='rays per rank and iteration (weak scaling)')
ap.add_argument('--iters', type=int, default=20)
args = ap.parse_args()
local = int(os.environ.get('LOCAL_RANK', 0))
dev = torch.device('cuda', local if args.backend == 'nccl' else 0)
torch.cuda.set_device(dev)
rank, world = parallel.init_distributed(args.backend, dev)
model, renderer, cam, poses = bench.build_scene(dev)
parallel.broadcast_parameters(model.parameters())
rays = [generate_rays(cam.width, cam.height, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, p, device=dev, want_direction=False) for p in poses[:2]]
origin = torch.cat([r['origin'] for r in rays]); vdir = torch.cat([r['view_direction'] for r in rays])
perm = torch.randperm(origin.shape[0], generator=torch.Generator(device='cpu').manual_seed(0)).to(dev)  # identical on every rank
opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)
scaler = torch.amp.GradScaler(init_scale=128.0, growth_interval=10 ** 9)
target = torch.rand(origin.shape[0], 3, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
n_global = args.rays * world


def step(i):
    ids = parallel.shard_ray_ids(perm[(i * n_global) % (perm.numel() - n_global):][:n_global])
    torch.manual_seed(1000 + i)  # the same random background on every rank
    with torch.amp.autocast('cuda'):
        bg = torch.rand(3, device=dev)
        out = renderer.render_rays(origin[ids], vdir[ids], cam, train_mode=True, custom_bg_color=bg)
        loss = torch.nn.functional.mse_loss(out['rgb'].float(), target[ids]) + 0.5e-6 * model.weight_decay_mlp()
    scaler.scale(loss).backward()
    parallel.allreduce_gradients(model.parameters(), average=True)
    scaler.step(opt); scaler.update(); opt.zero_grad()
    return int(out['rm_samples'].item())


for i in range(3):
    step(i)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
t0 = time.perf_counter(); tot = 0
for i in range(args.iters):
    tot += step(3 + i)
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
dt = (time.perf_counter() - t0) / args.iters
# replicas must stay bit-identical: same gradients (averaged), same update
flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
ref = flat.clone()
if world > 1:
    dist.broadcast(ref, src=0)
drift = float((flat - ref).abs().max())
stats = torch.tensor([dt, float(tot) / args.iters, drift], device=dev, dtype=torch.float64)
if world > 1:
    dist.all_reduce(stats, op=dist.ReduceOp.MAX)
if rank == 0:
    print(f'DP x{world} ({args.backend}): {stats[0].item() * 1e3:.2f} ms/iteration, {n_global} rays and ~{stats[1].item() * world:.0f} samples per global batch '
          f'-> {n_global / stats[0].item() / 1e6:.3f} Mrays/s; max parameter drift between ranks {stats[2].item():.1e}')
if world > 1:
    dist.destroy_process_group()
