"""nerficg_amd.fused_ssim -- drop-in for the external package `fused_ssim` as nerficg imports it (src/Thirdparty/FusedSSIM.py:15:
`from fused_ssim import fused_ssim`) and calls it (src/Optim/Losses/DSSIM.py:11-18: `1.0 - fused_ssim(input, target)` on (B,3,H,W)).

Same call surface as github.com/rahul-goel/fused-ssim: fused_ssim(img1, img2, padding="same", train=True) -> scalar mean SSIM,
differentiable w.r.t. img1 only.  Kernels: nerficg_amd/csrc/ssim.hip through the C ABI (include/nerficg_hip.h group 7).
"""
from __future__ import annotations

import torch

from .. import _lib

__all__ = ['fused_ssim', 'photometric_loss']

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


class _PhotometricLoss(torch.autograd.Function):
    """lambda_l1 * L1 + lambda_dssim * (1 - SSIM) as ONE node: stencil + reduction forward, one stencil backward (include/nerficg_hip.h,
    nrc_photometric_loss_*).  The upstream gradient of the loss value stays on the device (a GradScaler's scale, a weight of a larger loss)."""

    @staticmethod
    def forward(ctx, image, target, lambda_l1, lambda_dssim):
        lib = _lib.load()
        planes, h, w = _planes(image)
        train = ctx.needs_input_grad[0]
        d = [torch.empty_like(image) for _ in range(3)] if train else [None, None, None]
        ws = torch.empty(int(lib.nrc_photometric_loss_ws_floats(planes, h, w)), dtype=torch.float32, device=image.device)
        loss3 = torch.empty(3, dtype=torch.float32, device=image.device)
        _lib.check(lib.nrc_photometric_loss_forward(_lib.ptr(image), _lib.ptr(target), planes, h, w, 0.01 ** 2, 0.03 ** 2, float(lambda_l1), float(lambda_dssim),
                                                    _lib.ptr(d[0]), _lib.ptr(d[1]), _lib.ptr(d[2]), _lib.ptr(ws), _lib.ptr(loss3), _lib.stream_of(image)),
                   'photometric_loss_forward')
        if train:
            ctx.save_for_backward(image.detach(), target, *d)
        ctx.weights = (float(lambda_l1), float(lambda_dssim))
        ctx.terms = loss3      # [1] = mean |image - target|, [2] = mean SSIM (device; for logging without another pass)
        return loss3[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_loss):
        image, target, d1, d2, d3 = ctx.saved_tensors
        lib = _lib.load()
        planes, h, w = _planes(image)
        g = grad_loss.to(torch.float32).reshape(1).contiguous()
        grad = torch.empty_like(image)
        _lib.check(lib.nrc_photometric_loss_backward(_lib.ptr(image), _lib.ptr(target), planes, h, w, ctx.weights[0], ctx.weights[1], _lib.ptr(g),
                                                     _lib.ptr(d1), _lib.ptr(d2), _lib.ptr(d3), _lib.ptr(grad), _lib.stream_of(image)),
                   'photometric_loss_backward')
        return grad, None, None, None


def photometric_loss(image: torch.Tensor, target: torch.Tensor, lambda_l1: float = 0.8, lambda_dssim: float = 0.2) -> torch.Tensor:
    """GaussianSplattingLoss (src/Methods/GaussianSplatting/Loss.py:11-23) on (B, C, H, W) f32 images as one autograd node: the value of
    lambda_l1 * l1_loss(image, target) + lambda_dssim * (1 - fused_ssim(image, target)), differentiable w.r.t. `image` only (like fused_ssim)."""
    for t, name in ((image, 'image'), (target, 'target')):
        _lib.check_input(t, name, torch.float32)
    if image.dim() != 4 or image.shape != target.shape:
        raise RuntimeError('photometric_loss: image and target must be (B, C, H, W) tensors of the same shape')
    return _PhotometricLoss.apply(image, target, float(lambda_l1), float(lambda_dssim))
