#!/usr/bin/env python3
"""tools/soak_fused.py [ITERS] -- the fused training iteration for many iterations with everything on (next batch marched ahead on the side stream, dense
levels on the fork stream, a batch-size change every 16 iterations with the call in front of it not marching ahead, a new epoch's order when the old
one runs out; `occ` as second argument also updates the occupancy grid every 16 iterations -- on the bench's random targets the grid fills up and the
sample capacity then CUTS most batches, which exercises the capping for tens of thousands of iterations): the loss and the parameters must stay
finite, without `occ` no sample may be cut, and the steps the GradScaler skipped are reported."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from nerficg_amd.raygen import generate_rays
from nerficg_amd.apex_optimizers import FusedAdam
from nerficg_amd.amp import GradScaler
from nerficg_amd.ngp_trainer import FusedTrainingIteration

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
with_occ = len(sys.argv) > 2 and sys.argv[2] == 'occ'
dev = torch.device('cuda', 0)
model, renderer, cam, poses = bench.build_scene(dev)
rays = [generate_rays(cam.width, cam.height, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, p, device=dev, want_direction=False) for p in poses[:2]]
origin = torch.cat([r['origin'] for r in rays]); vdir = torch.cat([r['view_direction'] for r in rays])
target = torch.rand(origin.shape[0], 3, device=dev)
opt = FusedAdam(model.parameters(), lr=1e-3, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False, capturable=True)
scaler = GradScaler(init_scale=128.0, growth_interval=2000)
perm = lambda e: torch.randperm(origin.shape[0], generator=torch.Generator().manual_seed(e)).to(dev)
it = FusedTrainingIteration(model, renderer, opt, scaler, cam, {'origin': origin, 'view_direction': vdir, 'rgb': target}, 4096, 400_000, order=perm(0))
it.set_batch_size(2200)
marched = torch.zeros((), dtype=torch.int64, device=dev); cut = torch.zeros((), dtype=torch.int64, device=dev)
epoch, t0, sizes = 0, time.perf_counter(), [2200, 2048, 2304, 2176]
for i in range(iters):
    if i % 16 == 0 and i:
        if with_occ:
            renderer.update_occupancy_grid(warmup=False)
        it.set_batch_size(sizes[(i // 16) % 4])
    if it.remaining_batches() < 2:
        epoch += 1
        it.rewind(perm(epoch))
    out = it(prefetch=(i + 1) % 16 != 0)
    marched += out['rm_samples']; cut += out['sample_overflow']
    if i % 5000 == 4999:
        torch.cuda.synchronize()
        print(f'{i + 1} iterations, {(time.perf_counter() - t0) / (i + 1) * 1e3:.3f} ms each, loss {float(out["loss"]):.4f}, scale {float(scaler.get_scale())}, '
              f'samples {int(marched) / (i + 1):.0f} per iteration, cut {int(cut)}, epochs {epoch + 1}', flush=True)
torch.cuda.synchronize()
steps = opt.effective_step(opt.param_groups[0])
ok = bool(torch.isfinite(out['loss'])) and (with_occ or int(cut) == 0) and all(bool(torch.isfinite(p).all()) for p in model.parameters())
print('soak', 'OK' if ok else 'FAILED', f'-- {iters} iterations, optimizer steps {steps}, skipped {iters - steps}')
sys.exit(0 if ok else 1)
