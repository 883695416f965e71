#!/usr/bin/env python3
"""tools/bench_densify.py -- densify_and_prune on P synthetic Gaussians: the device plan + one gather (include/nerficg_hip.h group 10) next to
the mask-copy / torch.cat formulation the reference uses (Model.py:157-241 + adam_utils.py:21-61), written here with plain torch ops on the
same GPU tensors.  Usage: python tools/bench_densify.py [P] [repeats]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from nerficg_amd.gaussian_splatting import Gaussians, quaternion_to_rotation_matrix

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda', 0)
EXTENT, PD, THR, MIN_OP = 4.0, 0.01, 2e-4, 0.005
NAMES = ('positions', 'f_dc', 'f_rest', 'opacities', 'scales', 'rotations')


def fresh(seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    m = Gaussians(r(P, 3) * 2, r(P, 3) * 1.2 - 3.2, r(P, 4), r(P, 1) * 3 - 2, r(P, 1, 3), r(P, 15, 3) * 0.1)
    m.training_setup(PERCENT_DENSE=PD, training_cameras_extent=EXTENT)
    for grp in m.optimizer.param_groups:
        grp['params'][0].grad = torch.ones_like(grp['params'][0]) * 1e-3
    m.optimizer.step(); m.optimizer.zero_grad()
    m.densification_gradient_accum = torch.rand(P, 1, device=dev, generator=g) * 1.2e-3
    m.n_observations = torch.randint(0, 9, (P, 1), device=dev, generator=g, dtype=torch.int32)
    return m


def torch_style(m):
    """The reference's sequence with torch ops: clone -> cat; split -> cat, prune; final prune; each over 6 params x (1 + 2 moments)."""
    opt = m.optimizer
    def tensors():
        return {g['name']: g['params'][0] for g in opt.param_groups}
    def extend(extra):
        for g in opt.param_groups:
            old = g['params'][0]; st = opt.state.pop(old)
            new = torch.nn.Parameter(torch.cat((old.data, extra[g['name']]), 0))
            for k in ('exp_avg', 'exp_avg_sq'):
                st[k] = torch.cat((st[k], torch.zeros_like(extra[g['name']])), 0)
            opt.state[new] = st; g['params'][0] = new
    def prune(valid):
        for g in opt.param_groups:
            old = g['params'][0]; st = opt.state.pop(old)
            new = torch.nn.Parameter(old.data[valid])
            for k in ('exp_avg', 'exp_avg_sq'):
                st[k] = st[k][valid]
            opt.state[new] = st; g['params'][0] = new
    grads = m.densification_gradient_accum / m.n_observations.clamp_min(1)
    t = tensors()
    sel = (grads.norm(dim=-1) >= THR) & (t['scales'].exp().max(dim=1).values <= PD * EXTENT)
    extend({k: v.data[sel] for k, v in t.items()})
    t = tensors(); n = t['positions'].shape[0]
    padded = torch.zeros(n, device=dev); padded[:grads.shape[0]] = grads.squeeze()
    sel = (padded >= THR) & (t['scales'].exp().max(dim=1).values > PD * EXTENT)
    stds = t['scales'].data[sel].exp().repeat(2, 1)
    samples = torch.normal(torch.zeros_like(stds), stds)
    rots = quaternion_to_rotation_matrix(t['rotations'].data[sel]).repeat(2, 1, 1)
    extra = {k: v.data[sel].repeat(2, *([1] * (v.dim() - 1))) for k, v in t.items()}
    extra['positions'] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + extra['positions']
    extra['scales'] = torch.log(stds / 1.6)
    extend(extra)
    prune(~torch.cat((sel, torch.zeros(2 * int(sel.sum().item()), device=dev, dtype=torch.bool))))
    t = tensors()
    pm = (torch.sigmoid(t['opacities']).flatten() < MIN_OP) | (t['scales'].exp().max(dim=1).values > 0.1 * EXTENT)
    prune(~pm)
    return tensors()['positions'].shape[0]


def timed(fn):
    best = []
    for i in range(reps + 1):
        m = fresh(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = fn(m)
        torch.cuda.synchronize(); best.append(time.perf_counter() - t0)
    return out, float(np.median(best[1:])) * 1e3


with torch.no_grad():
    n_t, ms_t = timed(torch_style)
    info, ms_h = timed(lambda m: m.densify_and_prune(THR, MIN_OP, True))
rows_bytes = 59 * 4 * 3
print(f'densify_and_prune, P = {P}: device plan + gather {ms_h:.2f} ms ({info["n_out"]} rows out, {(P + info["n_out"]) * rows_bytes / ms_h / 1e6:.0f} GB/s of '
      f'parameter+moment rows), torch mask/cat formulation {ms_t:.2f} ms ({n_t} rows) -> x{ms_t / ms_h:.1f}')
