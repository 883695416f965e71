#!/usr/bin/env python3
"""tools/exp_fused_dp.py -- the data-parallel fused trainer under torchrun (any backend): ms per iteration with / without the batch marched ahead, and the
time of the gradient collective alone.  `python -m torch.distributed.run --nproc-per-node 2 tools/exp_fused_dp.py gloo` runs both ranks on the one GPU of a box."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.distributed as dist
backend = sys.argv[1] if len(sys.argv) > 1 else 'nccl'
dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count())
torch.cuda.set_device(dev)
dist.init_process_group(backend, **({'device_id': dev} if backend == 'nccl' else {}))
import bench
from nerficg_amd import parallel
from nerficg_amd.raygen import generate_rays
from nerficg_amd.apex_optimizers import FusedAdam
from nerficg_amd.amp import GradScaler
from nerficg_amd.ngp_trainer import FusedTrainingIteration
rank, world = parallel.world_info()
model, renderer, cam, poses = bench.build_scene(dev)
rays = [generate_rays(cam.width, cam.height, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, p, device=dev, want_direction=False) for p in poses[:2]]
origin = torch.cat([r['origin'] for r in rays]); vdir = torch.cat([r['view_direction'] for r in rays])
perm = torch.randperm(origin.shape[0], generator=torch.Generator(device='cpu').manual_seed(0)).to(dev)
target = torch.rand(origin.shape[0], 3, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
for prefetch in (True, False):
    opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False, capturable=True)
    it = FusedTrainingIteration(model, renderer, opt, GradScaler(init_scale=128.0, growth_interval=10 ** 9), cam, {'origin': origin, 'view_direction': vdir, 'rgb': target},
                                2200, 307200, order=perm, seed=5, prefetch=prefetch)
    for _ in range(3):
        it()
    torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
    for _ in range(20):
        it()
    torch.cuda.synchronize(); dist.barrier(); dt = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(10):
        parallel.allreduce_flat(it.grads, average=True)
    torch.cuda.synchronize(); dc = (time.perf_counter() - t0) / 10
    if rank == 0:
        print(f'{backend} world {world} prefetch {prefetch}: {dt * 1e3:.2f} ms per iteration; collective alone {dc * 1e3:.2f} ms ({it.grads.numel() * 4 / 1e6:.1f} MB)', flush=True)
dist.destroy_process_group()
