"""CPU: pins oracle/ssim_oracle.c against a plain PyTorch statement of the same SSIM (conv2d with the 11x11 sigma-1.5 Gaussian window,
zero 'same' padding, C1 = 0.01^2, C2 = 0.03^2 -- the loss the reference's 3DGS trainer uses through fused-ssim, DSSIM.py:11-18) and its
autograd gradient."""
import numpy as np
import pytest
import torch

import oracle


def torch_ssim_map(img1, img2, C1=0.01 ** 2, C2=0.03 ** 2):
    g = torch.tensor([np.exp(-(x - 5) ** 2 / (2 * 1.5 ** 2)) for x in range(11)], dtype=img1.dtype)
    g = g / g.sum()
    c = img1.shape[1]
    win = (g[:, None] * g[None, :])[None, None].expand(c, 1, 11, 11).contiguous()
    blur = lambda x: torch.nn.functional.conv2d(x, win, padding=5, groups=c)
    mu1, mu2 = blur(img1), blur(img2)
    s1, s2, s12 = blur(img1 * img1) - mu1 * mu1, blur(img2 * img2) - mu2 * mu2, blur(img1 * img2) - mu1 * mu2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))


@pytest.mark.parametrize('shape', [(1, 3, 37, 53), (2, 1, 8, 8), (1, 3, 11, 64)])
def test_ssim_map_and_gradient_match_torch(shape):
    rng = np.random.default_rng(sum(shape))
    a = rng.random(shape).astype(np.float32)
    b = np.clip(a + 0.1 * rng.normal(size=shape), 0, 1).astype(np.float32)
    ta = torch.from_numpy(a).double().requires_grad_(True)
    ref = torch_ssim_map(ta, torch.from_numpy(b).double())
    w = torch.from_numpy(rng.normal(size=shape)).double()
    (ref * w).sum().backward()
    m, d1, d2, d3 = oracle.ssim_forward(a, b)
    np.testing.assert_allclose(m, ref.detach().numpy(), rtol=0, atol=2e-6)
    grad = oracle.ssim_backward(a, b, w.numpy().astype(np.float32), d1, d2, d3)
    np.testing.assert_allclose(grad, ta.grad.numpy(), rtol=0, atol=2e-5 * np.abs(ta.grad.numpy()).max())
    assert oracle.ssim_forward(a, a, train=False).min() > 0.999999  # identical images
