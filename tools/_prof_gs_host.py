import sys, time, cProfile, pstats
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np, torch, bench
from tests import scenes
from nerficg_amd.gaussian_splatting import Gaussians, PerspectiveCamera, render_image_training, training_loss
dev = torch.device('cuda', 0)
sc = scenes.gs_random_scene(1_000_000, seed=0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
g = Gaussians(T(sc['means3D']), torch.log(T(sc['scales'])), T(sc['rotations']), torch.logit(T(sc['opacities']).clamp(1e-4, 1 - 1e-4))[:, None].contiguous(), T(sc['shs'][:, :1]), T(sc['shs'][:, 1:]))
g.training_setup(training_cameras_extent=4.5)
g.fuse_rest_step = len(sys.argv) > 1 and sys.argv[1] == 'fuse'
cam = PerspectiveCamera(bench.GS_W, bench.GS_H, 1.2 * bench.GS_W, 1.2 * bench.GS_W, background_color=torch.zeros(3, device=dev))
target = torch.rand(3, bench.GS_H, bench.GS_W, device=dev)
poses = [torch.from_numpy(np.asarray(scenes.orbit_pose(0.8 + 0.7 * i, 0.35, 4.5), dtype=np.float32)).to(dev) for i in range(8)]
def step(i):
    out = render_image_training(g, cam, poses[i % 8])
    training_loss(out['rgb'], target).backward()
    g.optimizer.step(); g.optimizer.zero_grad()
for i in range(5): step(i)
torch.cuda.synchronize()
# host time per step WITHOUT waiting for the GPU: enqueue 30 steps, time the enqueue only
t0 = time.perf_counter()
for i in range(30): step(i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'host enqueue {1e3 * (t1 - t0) / 30:.3f} ms per step; with GPU {1e3 * (t2 - t0) / 30:.3f} ms per step')
pr = cProfile.Profile(); pr.enable()
for i in range(30): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
