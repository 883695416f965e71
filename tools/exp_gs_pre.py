#!/usr/bin/env python3
"""tools/exp_gs_pre.py -- k_preprocess / k_preprocess_bw under the variations between the forward-only bench frame and the training step:
activated + concatenated SH (bench) vs raw parameters + split dc / rest (training), no_grad vs grad."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from nerficg_amd import _lib

dev = torch.device('cuda', 0)
gs = bench.build_gs_scene(dev, 1_000_000)
t = gs['tensors']
rast = gs['rast']
raw = dict(means3D=t['means3D'], opacities=torch.logit(t['opacities'].clamp(1e-4, 1 - 1e-4)), scales=torch.log(t['scales']), rotations=t['rotations'] * 1.7,
           shs=t['shs'][:, :1].contiguous(), shs_rest=t['shs'][:, 1:].contiguous())


def run(tag, grad, **kw):
    args = {k: v.detach().clone().requires_grad_(grad) for k, v in kw.items() if torch.is_tensor(v)}
    extra = {k: v for k, v in kw.items() if not torch.is_tensor(v)}
    m2d = torch.zeros_like(args['means3D'], requires_grad=grad)
    g = torch.rand(3, gs['h'], gs['w'], device=dev)
    for rep in range(3):
        with _lib.stage_timer() as st:
            with torch.set_grad_enabled(grad):
                color, radii = rast(means2D=m2d, **args, **extra)
            if grad:
                color.backward(g)
            torch.cuda.synchronize()
    by = st.by_name()
    print(tag, {k: round(v[0] * 1e3, 1) for k, v in by.items() if 'preprocess' in k or 'render' in k}, 'visible', int((radii > 0).sum()))


run('activated, concat SH, no grad', False, means3D=t['means3D'], opacities=t['opacities'], shs=t['shs'], scales=t['scales'], rotations=t['rotations'])
run('activated, concat SH, grad   ', True, means3D=t['means3D'], opacities=t['opacities'], shs=t['shs'], scales=t['scales'], rotations=t['rotations'])
run('activated, split SH, grad    ', True, means3D=t['means3D'], opacities=t['opacities'], shs=raw['shs'], shs_rest=raw['shs_rest'], scales=t['scales'], rotations=t['rotations'])
run('raw, split SH, grad          ', True, **raw, raw_parameters=True)
run('raw, split SH, no grad       ', False, **raw, raw_parameters=True)
