import sys, torch
sys.path.insert(0, '.')
from tests.test_gpu_graphs import _rays, _train_pair
from nerficg_amd.apex_optimizers import FusedAdam
from nerficg_amd.graphs import GraphedIteration, instant_ngp_iteration
mode = sys.argv[1]
DEV = 'cuda'
cam, o, d = _rays()
n = 2048
model, renderer, scaler = _train_pair(seed=4)
opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False, capturable=True)
target = torch.tensor([0.8, 0.3, 0.1], device=DEV).expand(n, 3).contiguous()
g = torch.Generator(device=DEV).manual_seed(5)
if mode == 'helper':
    step = instant_ngp_iteration(model, renderer, opt, scaler, cam, n_rays=n, sample_capacity=400_000)
    call = lambda ids: step(origin=o[ids], view_direction=d[ids], rgb=target)
else:
    renderer.sample_capacity = 400_000
    def body(origin, view_direction, rgb, bg, noise):
        with torch.amp.autocast('cuda'):
            if mode == 'rng_bg':
                bg = torch.rand(3, device=DEV)
            if mode == 'rng_noise':
                noise = None
            out = renderer.render_rays(origin, view_direction, cam, train_mode=True, custom_bg_color=bg, noise=noise)
            loss = torch.nn.functional.mse_loss(out['rgb'].float(), rgb) + 0.5e-6 * model.weight_decay_mlp()
        scaler.scale(loss).backward()
        scaler.step(opt); scaler.update(); opt.zero_grad()
        return {'loss': loss.detach(), 'rm_samples': out['rm_samples']}
    step = GraphedIteration(body, dict(origin=o[:n].contiguous(), view_direction=d[:n].contiguous(), rgb=target, bg=torch.rand(3, device=DEV), noise=torch.rand(n, device=DEV)))
    call = lambda ids: step(origin=o[ids], view_direction=d[ids], rgb=target, bg=torch.rand(3, device=DEV, generator=g), noise=torch.rand(n, device=DEV, generator=g))
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    ids = torch.randint(0, o.shape[0], (n,), device=DEV, generator=g)
    out = call(ids)
    torch.cuda.synchronize()
    print(mode, it, float(out['loss']), int(out['rm_samples']), flush=True)
print('ok', mode)
