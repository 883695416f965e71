"""nerficg_amd.fused_ssim -- drop-in for the external package `fused_ssim` as nerficg imports it (src/Thirdparty/FusedSSIM.py:15:
`from fused_ssim import fused_ssim`) and calls it (src/Optim/Losses/DSSIM.py:11-18: `1.0 - fused_ssim(input, target)` on (B,3,H,W)).

Same call surface as github.com/rahul-goel/fused-ssim: fused_ssim(img1, img2, padding="same", train=True) -> scalar mean SSIM,
differentiable w.r.t. img1 only.  Kernels: nerficg_amd/csrc/ssim.hip through the C ABI (include/nerficg_hip.h group 7).
"""
from __future__ import annotations

import torch

from .. import _lib

__all__ = ['fused_ssim']

_ALLOWED_PADDING = ('same', 'valid')


def _planes(t: torch.Tensor) -> tuple[int, int, int]:
    return t.shape[0] * t.shape[1], t.shape[2], t.shape[3]


class _FusedSSIMMap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, C1, C2, img1, img2, padding='same', train=True):
        for t, name in ((img1, 'img1'), (img2, 'img2')):
            _lib.check_input(t, name, torch.float32)
        if img1.dim() != 4 or img1.shape != img2.shape:
            raise RuntimeError('fused_ssim: img1 and img2 must be (B, C, H, W) tensors of the same shape')
        lib = _lib.load()
        planes, h, w = _planes(img1)
        ssim_map = torch.empty_like(img1)
        d = [torch.empty_like(img1) for _ in range(3)] if train else [None, None, None]
        _lib.check(lib.nrc_ssim_forward(_lib.ptr(img1), _lib.ptr(img2), planes, h, w, float(C1), float(C2), _lib.ptr(ssim_map),
                                        _lib.ptr(d[0]), _lib.ptr(d[1]), _lib.ptr(d[2]), _lib.stream_of(img1)), 'ssim_forward')
        if padding == 'valid':
            ssim_map = ssim_map[:, :, 5:-5, 5:-5]
        if train:
            ctx.save_for_backward(img1.detach(), img2, *d)
        ctx.padding = padding
        ctx.train = train
        return ssim_map

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, opt_grad):
        if not ctx.train:
            raise RuntimeError('fused_ssim: backward needs train=True in the forward call')
        img1, img2, d1, d2, d3 = ctx.saved_tensors
        dL_dmap = opt_grad
        if ctx.padding == 'valid':
            dL_dmap = torch.zeros_like(img1)
            dL_dmap[:, :, 5:-5, 5:-5] = opt_grad
        dL_dmap = dL_dmap.to(torch.float32).contiguous()
        lib = _lib.load()
        planes, h, w = _planes(img1)
        grad = torch.empty_like(img1)
        _lib.check(lib.nrc_ssim_backward(_lib.ptr(img1), _lib.ptr(img2), planes, h, w, _lib.ptr(dL_dmap), _lib.ptr(d1), _lib.ptr(d2), _lib.ptr(d3),
                                         _lib.ptr(grad), _lib.stream_of(img1)), 'ssim_backward')
        return None, None, grad, None, None, None


def fused_ssim(img1: torch.Tensor, img2: torch.Tensor, padding: str = 'same', train: bool = True) -> torch.Tensor:
    if padding not in _ALLOWED_PADDING:
        raise ValueError(f'fused_ssim: padding must be one of {_ALLOWED_PADDING}')
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return _FusedSSIMMap.apply(C1, C2, img1.contiguous(), img2.contiguous(), padding, train).mean()
