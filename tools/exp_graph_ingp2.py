import sys, torch
sys.path.insert(0, '.')
from tests.test_gpu_graphs import _rays, _train_pair
from nerficg_amd.apex_optimizers import FusedAdam
from nerficg_amd.graphs import GraphedIteration
mode = sys.argv[1]
DEV = 'cuda'
cam, o, d = _rays()
n = 2048
model, renderer, scaler = _train_pair(seed=4)
opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False, capturable=True)
target = torch.tensor([0.8, 0.3, 0.1], device=DEV).expand(n, 3).contiguous()
g = torch.Generator(device=DEV).manual_seed(5)
renderer.sample_capacity = 400_000
def body(origin, view_direction, rgb):
    if mode == 'rand_only':
        r = torch.rand(3, device=DEV)
        return {'loss': r.sum()}
    if mode == 'fwd_nograd':
        with torch.no_grad(), torch.amp.autocast('cuda'):
            bg = torch.rand(3, device=DEV)
            out = renderer.render_rays(origin, view_direction, cam, train_mode=True, custom_bg_color=bg)
        return {'loss': out['rgb'].sum()}
    with torch.amp.autocast('cuda'):
        bg = torch.rand(3, device=DEV)
        out = renderer.render_rays(origin, view_direction, cam, train_mode=True, custom_bg_color=bg)
        loss = torch.nn.functional.mse_loss(out['rgb'].float(), rgb) + 0.5e-6 * model.weight_decay_mlp()
    if mode == 'fwd':
        return {'loss': loss.detach()}
    if mode == 'fwd_bwd':
        loss.backward(); opt.zero_grad()
        return {'loss': loss.detach()}
    if mode == 'scaled_bwd':
        scaler.scale(loss).backward(); opt.zero_grad()
        return {'loss': loss.detach()}
    if mode == 'noscaler':
        loss.backward(); opt.step(); opt.zero_grad()
        return {'loss': loss.detach()}
    scaler.scale(loss).backward()
    scaler.step(opt); scaler.update(); opt.zero_grad()
    return {'loss': loss.detach()}
step = GraphedIteration(body, dict(origin=o[:n].contiguous(), view_direction=d[:n].contiguous(), rgb=target))
for it in range(12):
    ids = torch.randint(0, o.shape[0], (n,), device=DEV, generator=g)
    out = step(origin=o[ids], view_direction=d[ids], rgb=target)
    torch.cuda.synchronize()
print('ok', mode, float(out['loss']))
