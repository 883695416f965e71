"""GPU: nerficg_amd.fused_ssim (HIP, C ABI group 7) against oracle/ssim_oracle.c.  f32 separable blur vs the oracle's double
accumulation: map within 2e-6 absolute, gradient within 2e-5 of its scale."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _pair(shape, seed):
    rng = np.random.default_rng(seed)
    a = rng.random(shape).astype(np.float32)
    b = np.clip(a + 0.1 * rng.normal(size=shape), 0, 1).astype(np.float32)
    return a, b, rng


@pytest.mark.parametrize('shape', [(1, 3, 70, 100), (2, 3, 16, 16), (1, 1, 5, 7), (1, 3, 33, 129), (3, 2, 48, 31)])
def test_ssim_forward_and_backward_match_oracle(shape):
    from nerficg_amd.fused_ssim import fused_ssim
    a, b, rng = _pair(shape, sum(shape))
    ta = torch.from_numpy(a).to(DEV).requires_grad_(True)
    tb = torch.from_numpy(b).to(DEV)
    val = fused_ssim(ta, tb)
    m, d1, d2, d3 = oracle.ssim_forward(a, b)
    assert abs(val.item() - float(m.astype(np.float64).mean())) < 2e-6
    (1.0 - val).backward()  # the reference's loss term: DSSIM = 1 - SSIM (DSSIM.py:18)
    ref = oracle.ssim_backward(a, b, np.full(shape, -1.0 / a.size, np.float32), d1, d2, d3)
    np.testing.assert_allclose(ta.grad.cpu().numpy(), ref, rtol=0, atol=2e-5 * np.abs(ref).max())


def test_ssim_map_weighted_gradient_and_valid_padding():
    from nerficg_amd.fused_ssim import _FusedSSIMMap, fused_ssim
    shape = (1, 3, 40, 56)
    a, b, rng = _pair(shape, 7)
    w = rng.normal(size=shape).astype(np.float32)
    ta = torch.from_numpy(a).to(DEV).requires_grad_(True)
    tb = torch.from_numpy(b).to(DEV)
    smap = _FusedSSIMMap.apply(0.01 ** 2, 0.03 ** 2, ta, tb, 'same', True)
    m, d1, d2, d3 = oracle.ssim_forward(a, b)
    np.testing.assert_allclose(smap.detach().cpu().numpy(), m, rtol=0, atol=2e-6)
    (smap * torch.from_numpy(w).to(DEV)).sum().backward()
    ref = oracle.ssim_backward(a, b, w, d1, d2, d3)
    np.testing.assert_allclose(ta.grad.cpu().numpy(), ref, rtol=0, atol=2e-5 * np.abs(ref).max())
    # 'valid' = the same map cropped by the window radius
    v = fused_ssim(ta.detach(), tb, padding='valid', train=False)
    assert abs(v.item() - float(m[:, :, 5:-5, 5:-5].astype(np.float64).mean())) < 2e-6
    assert abs(fused_ssim(tb, tb, train=False).item() - 1.0) < 1e-6


def test_ssim_input_checks():
    from nerficg_amd.fused_ssim import fused_ssim
    x = torch.rand(1, 3, 8, 8, device=DEV)
    with pytest.raises(RuntimeError):
        fused_ssim(x.cpu(), x)
    with pytest.raises(RuntimeError):
        fused_ssim(x, x[:, :2])
    with pytest.raises(ValueError):
        fused_ssim(x, x, padding='reflect')


@pytest.mark.parametrize('shape', [(1, 3, 70, 100), (2, 3, 16, 16), (1, 1, 5, 7), (1, 3, 840, 1297), (3, 2, 48, 31)])
def test_photometric_loss_is_l1_plus_dssim_against_oracle_and_tensor_operations(shape):
    """nerficg_amd.fused_ssim.photometric_loss (one node: lambda_l1 L1 + lambda_dssim (1 - SSIM), Loss.py:11-23) against (i) the oracle's SSIM map /
    gradient + numpy's L1 in float64 and (ii) the same loss composed of l1_loss and fused_ssim as tensor operations -- value and gradient, with an
    upstream factor that lives on the device, with ties (image == target on a block: sign 0) and in the shape of the 3DGS bench (1297x840)."""
    from nerficg_amd.fused_ssim import fused_ssim, photometric_loss
    from nerficg_amd.gaussian_splatting import training_loss
    a, b, rng = _pair(shape, sum(shape) + 1)
    a[..., :3, :4] = b[..., :3, :4]          # exact ties: |x| has subgradient 0 there, like torch's abs backward
    l1w, dw = 0.8, 0.2
    up = torch.tensor(3.7, device=DEV)
    ta = torch.from_numpy(a).to(DEV).requires_grad_(True)
    tb = torch.from_numpy(b).to(DEV)
    val = photometric_loss(ta, tb, l1w, dw)
    (val * up).backward()
    m, d1, d2, d3 = oracle.ssim_forward(a, b)
    n = a.size
    ref_val = l1w * np.abs(a.astype(np.float64) - b).mean() + dw * (1.0 - m.astype(np.float64).mean())
    assert abs(val.item() - ref_val) < 3e-6
    ref_grad = 3.7 * (oracle.ssim_backward(a, b, np.full(shape, -dw / n, np.float32), d1, d2, d3) + l1w / n * np.sign(a - b))
    np.testing.assert_allclose(ta.grad.cpu().numpy(), ref_grad, rtol=0, atol=2e-5 * np.abs(ref_grad).max())
    # the tensor-operation form of the same loss (what training_loss does with the fused node switched off)
    tc = torch.from_numpy(a).to(DEV).requires_grad_(True)
    val2 = l1w * torch.nn.functional.l1_loss(tc, tb) + dw * (1.0 - fused_ssim(tc, tb))
    (val2 * up).backward()
    assert abs(val.item() - val2.item()) < 2e-6
    np.testing.assert_allclose(ta.grad.cpu().numpy(), tc.grad.cpu().numpy(), rtol=0, atol=2e-6 * float(tc.grad.abs().max()))
    # training_loss takes the fused node for (3, H, W) images as the trainer passes them; no gradient wanted: value only
    if shape[0] == 1:
        v3 = training_loss(torch.from_numpy(a[0]).to(DEV), tb[0], l1w, dw)
        assert abs(v3.item() - val.item()) < 1e-7
