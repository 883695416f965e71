"""Experiment: where does the time of a replayed 3DGS step go?"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from tests import scenes
from nerficg_amd.gaussian_splatting import Gaussians, PerspectiveCamera, render_image_training, training_loss
from nerficg_amd import diff_gaussian_rasterization as dgr
from nerficg_amd.graphs import gaussian_splatting_step, GraphedIteration
dev = 'cuda'
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
mode = sys.argv[2] if len(sys.argv) > 2 else 'full'
W, H = 1297, 840
sc = scenes.gs_random_scene(n, seed=0)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
g = Gaussians(T(sc['means3D']), torch.log(T(sc['scales'])), T(sc['rotations']), torch.logit(T(sc['opacities']).clamp(1e-4, 1 - 1e-4))[:, None].contiguous(),
              T(sc['shs'][:, :1]), T(sc['shs'][:, 1:]))
g.training_setup(training_cameras_extent=4.5, capturable=(mode != 'switch'))
cam = PerspectiveCamera(W, H, 1.2 * W, 1.2 * W, background_color=torch.zeros(3, device=dev))
target = torch.rand(3, H, W, device=dev)
poses = [torch.from_numpy(np.asarray(scenes.orbit_pose(0.8 + 0.7 * i, 0.35, 4.5), dtype=np.float32)).to(dev) for i in range(12)]
out = render_image_training(g, cam, poses[0]); training_loss(out['rgb'], target).backward(); g.optimizer.zero_grad(); del out
n_inst, n_spans = dgr.last_counts().tolist()
caps = (int(1.3 * n_inst), int(1.3 * n_spans) + 65536)
print('counts', n_inst, n_spans, caps)

def body_fwd(c2w, target):
    with dgr.fixed_capacity(*caps):
        out = render_image_training(g, cam, c2w)
    return {'rgb': out['rgb'].detach()}

def body_opt(c2w, target):
    with dgr.fixed_capacity(*caps):
        out = render_image_training(g, cam, c2w)
        loss = out['rgb'].sum()
        loss.backward()
    g.optimizer.step(); g.optimizer.zero_grad()
    return {'loss': loss.detach()}

def body_fb(c2w, target):
    with dgr.fixed_capacity(*caps):
        out = render_image_training(g, cam, c2w)
        loss = training_loss(out['rgb'], target)
        loss.backward()
    g.optimizer.zero_grad()
    return {'loss': loss.detach()}

ex = {'c2w': torch.eye(4, device=dev), 'target': torch.zeros(3, H, W, device=dev)}
if mode == 'switch':
    for i in range(4):
        out = render_image_training(g, cam, poses[i]); training_loss(out['rgb'], target).backward(); g.optimizer.step(); g.optimizer.zero_grad()
    del out
    g.optimizer.capturable = True
if mode in ('full', 'switch'):
    st = gaussian_splatting_step(g, cam, *caps)
elif mode == 'opt':
    st = GraphedIteration(body_opt, ex, parameters=lambda: [grp['params'][0] for grp in g.optimizer.param_groups])
elif mode == 'fwd':
    st = GraphedIteration(body_fwd, ex)
else:
    st = GraphedIteration(body_fb, ex, parameters=lambda: [grp['params'][0] for grp in g.optimizer.param_groups])
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st(c2w=poses[i], target=target)
    torch.cuda.synchronize(); print(mode, i, f'{(time.perf_counter() - t0) * 1e3:.3f} ms', 'recorded' if st.recorded else 'eager')
import os
sep = os.environ.get('SEP', 'none')
torch.cuda.synchronize(); t0 = time.perf_counter()
ev = None
for i in range(40):
    if sep == 'event_wait' and ev is not None:
        torch.cuda.current_stream().wait_event(ev)
    if sep == 'event_sync' and ev is not None:
        ev.synchronize()
    if sep == 'sync':
        torch.cuda.synchronize()
    st(c2w=poses[i % 12], target=target)
    ev = torch.cuda.Event(); ev.record()
torch.cuda.synchronize(); print('ok', sep, (time.perf_counter() - t0) / 40 * 1e3, 'ms')
