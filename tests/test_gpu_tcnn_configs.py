"""GPU: the tinycudann drop-in over the configurations the reference's yaml can ask for (src/Methods/InstantNGP/Model.py:18-29 forwards HASHGRID_N_LEVELS,
HASHGRID_N_FEATURES_PER_LEVEL and DIR_SH_ENCODING_DEGREE into the tcnn config, :58-114) -- not only the shipped 16 x 2 / degree 4 -- against oracle/tcnn_oracle.c,
forward and backward, and an InstantNGP model of such a configuration rendering and training through nerficg_amd.instant_ngp."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = 'cuda'
NET_D = {'otype': 'FullyFusedMLP', 'activation': 'ReLU', 'output_activation': 'None', 'n_neurons': 64, 'n_hidden_layers': 1}
NET_C = {'otype': 'FullyFusedMLP', 'activation': 'ReLU', 'output_activation': 'Sigmoid', 'n_neurons': 64, 'n_hidden_layers': 2}


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _half_np(t):
    return t.detach().float().cpu().numpy().astype(np.float16).astype(np.float32)


@pytest.mark.parametrize('levels,feats,log2_t', [(16, 2, 19), (8, 4, 19), (12, 2, 19), (8, 2, 15), (5, 4, 14), (16, 2, 14)])
def test_density_network_of_any_grid_configuration(levels, feats, log2_t):
    import nerficg_amd.tinycudann as tcnn
    pls = float(np.exp(np.log(2048 / 16) / max(levels - 1, 1)))
    grid = dict(n_levels=levels, log2_hashmap_size=log2_t, base_resolution=16, per_level_scale=pls)
    net = tcnn.NetworkWithInputEncoding(3, 16, {'otype': 'Grid', 'type': 'Hash', 'interpolation': 'Linear', 'n_features_per_level': feats, **grid}, NET_D, seed=7).to(DEV)
    total, offsets, _, _ = oracle.grid_layout(**grid)
    n_in = (levels * feats + 15) // 16 * 16
    assert net.n_in_padded == n_in and net.n_mlp_params == 64 * n_in + 16 * 64 and net.params.numel() == net.n_mlp_params + total * feats
    assert net.default_layout == (levels == 16 and feats == 2)
    with torch.no_grad():
        g = torch.Generator().manual_seed(3)
        net.params[net.n_mlp_params:] = ((torch.rand(total * feats, generator=g) * 2 - 1) * 0.5).to(DEV)
    rng = np.random.default_rng(levels * 10 + feats)
    m = 6001
    x = rng.random((m, 3)).astype(np.float32)
    x[0] = [0.0, 1.0, 0.5]
    net.zero_grad()
    out = net(T(x))
    g_out = (rng.normal(size=(m, 16)) * 0.01).astype(np.float16)
    out.backward(T(g_out))
    p = _half_np(net.params)
    W, table = p[:net.n_mlp_params], p[net.n_mlp_params:].reshape(-1, feats)
    enc = np.zeros((m, n_in), np.float32)
    enc[:, :levels * feats] = oracle.grid_encode_fw(x, table, **grid)
    ref, acts = oracle.mlp_fw(enc, W, n_in=n_in, n_hidden=1, out_act=0, want_acts=True)
    np.testing.assert_allclose(out.detach().float().cpu().numpy(), ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    dW, d_in = oracle.mlp_bw(enc, W, ref, acts, g_out.astype(np.float32) * 128.0, n_in=n_in, n_hidden=1, out_act=0)
    dW /= 128.0
    d_in /= 128.0
    got = net.params.grad.cpu().numpy()
    np.testing.assert_allclose(got[:net.n_mlp_params], dW, rtol=2e-2, atol=2e-3 * np.abs(dW).max())
    g_table = oracle.grid_encode_bw(x, np.ascontiguousarray(d_in[:, :levels * feats]), total, **grid)
    np.testing.assert_allclose(got[net.n_mlp_params:].reshape(-1, feats), g_table, rtol=3e-2, atol=3e-3 * np.abs(g_table).max())


@pytest.mark.parametrize('degree', [1, 2, 3, 4])
def test_color_network_of_any_sh_degree(degree):
    import nerficg_amd.tinycudann as tcnn
    enc = {'otype': 'Composite', 'nested': [{'n_dims_to_encode': 3, 'otype': 'SphericalHarmonics', 'degree': degree}, {'otype': 'Identity'}]}
    net = tcnn.NetworkWithInputEncoding(19, 3, enc, NET_C, seed=11).to(DEV)
    assert net.params.numel() == 7168 and net.default_layout == (degree == 4)      # [SH d^2 | identity 16] padded to 32: len(color_mlp.params) of Model.py:115 whatever the degree
    rng = np.random.default_rng(40 + degree)
    m = 3000
    d = rng.normal(size=(m, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    h = (rng.normal(size=(m, 16)) * 0.5).astype(np.float16)
    x = torch.cat([T(d * 0.5 + 0.5).half(), T(h)], dim=-1).requires_grad_(True)
    out = net(x)
    g_out = (rng.normal(size=(m, 3)) * 0.01).astype(np.float16)
    out.backward(T(g_out))
    p = _half_np(net.params)
    d01 = (d * np.float32(0.5) + np.float32(0.5)).astype(np.float16).astype(np.float32)
    n_sh = degree * degree
    cin = np.zeros((m, 32), np.float32)           # tiny-cuda-nn's composite: [SH coefficients | identity dims | padding]
    cin[:, :n_sh] = oracle.sh4_encode(d01)[:, :n_sh]
    cin[:, n_sh:n_sh + 16] = h.astype(np.float32)
    ref, acts = oracle.mlp_fw(cin, p, n_hidden=2, out_act=1, want_acts=True)
    np.testing.assert_allclose(out.detach().float().cpu().numpy(), ref[:, :3], rtol=0, atol=2e-3)
    g_pad = np.zeros((m, 16), np.float32)
    g_pad[:, :3] = g_out.astype(np.float32) * 128.0
    dW, d_in = oracle.mlp_bw(cin, p, ref, acts, g_pad, n_hidden=2, out_act=1)
    dW /= 128.0
    d_in /= 128.0
    dW0 = dW[:64 * 32].reshape(64, 32)
    dW0[:, n_sh + 16:] = 0.0                      # padding inputs are not inputs: their columns receive no gradient here (the oracle's are zero too: inputs 0)
    np.testing.assert_allclose(net.params.grad.cpu().numpy(), dW, rtol=2e-2, atol=2e-3 * np.abs(dW).max())
    gx = x.grad.float().cpu().numpy()
    assert np.all(gx[:, :3] == 0)
    np.testing.assert_allclose(gx[:, 3:], d_in[:, n_sh:n_sh + 16], rtol=2e-2, atol=2e-3 * np.abs(d_in).max())


def test_unsupported_grid_raises_by_key_name():
    import nerficg_amd.tinycudann as tcnn
    for levels, feats in ((16, 4), (9, 4), (16, 8), (4, 1)):
        with pytest.raises(RuntimeError, match='HASHGRID_N_LEVELS'):
            tcnn.NetworkWithInputEncoding(3, 16, {'otype': 'Grid', 'type': 'Hash', 'interpolation': 'Linear', 'n_levels': levels, 'n_features_per_level': feats,
                                                  'log2_hashmap_size': 15, 'base_resolution': 16, 'per_level_scale': 1.5}, NET_D)
    with pytest.raises(RuntimeError, match='DIR_SH_ENCODING_DEGREE'):
        tcnn.NetworkWithInputEncoding(19, 3, {'otype': 'Composite', 'nested': [{'n_dims_to_encode': 3, 'otype': 'SphericalHarmonics', 'degree': 5}, {'otype': 'Identity'}]}, NET_C)


def test_an_instant_ngp_model_of_another_yaml_configuration_renders_and_trains():
    """HASHGRID_N_LEVELS = 8, HASHGRID_N_FEATURES_PER_LEVEL = 4, DIR_SH_ENCODING_DEGREE = 3 through nerficg_amd.instant_ngp: the frame goes through the general
    path (drop-in modules), forty op-by-op iterations on a fixed batch lower the loss, the fused trainer refuses the configuration by key name."""
    from nerficg_amd.amp import GradScaler
    from nerficg_amd.apex_optimizers import FusedAdam
    from nerficg_amd.instant_ngp import InstantNGPLoss, InstantNGPModel, InstantNGPRenderer
    from nerficg_amd.ngp_trainer import FusedTrainingIteration
    from tests import scenes
    from tests.test_gpu_render_parity import make_camera
    model = InstantNGPModel(HASHGRID_N_LEVELS=8, HASHGRID_N_FEATURES_PER_LEVEL=4, HASHGRID_LOG2_SIZE=15, DIR_SH_ENCODING_DEGREE=3, device=DEV)
    assert model.n_params_encoding_mlp == model.encoding_xyz.n_mlp_params == 3072 and model.color_mlp_with_encoding.params.numel() == 7168
    with torch.no_grad():
        model.occupancy_bitfield.copy_(T(scenes.sphere_bitfield(128, 0.5, 0.35, model.cascades)))
        n_mlp = model.encoding_xyz.n_mlp_params
        model.encoding_xyz.params[n_mlp:] = (torch.rand(model.encoding_xyz.params.numel() - n_mlp, generator=torch.Generator().manual_seed(1)) - 0.5).to(DEV)
    renderer = InstantNGPRenderer(model)
    cam = make_camera(40, 32, bg=(1.0, 1.0, 1.0))
    c2w = scenes.orbit_pose(0.7, 0.4, scenes.LEGO_RADIUS)
    img = renderer.render_image(cam, c2w)
    assert img['rgb'].shape == (32, 40, 3) and bool(torch.isfinite(img['rgb']).all()) and float(img['alpha'].max()) > 0.05
    o, _, d = scenes.numpy_rays(40, 32, c2w)
    o, d = T(o), T(d / np.linalg.norm(d, axis=-1, keepdims=True))
    target = torch.tensor([0.2, 0.5, 0.7], device=DEV).expand(o.shape[0], 3).contiguous()
    opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)
    scaler = GradScaler(init_scale=128.0, growth_interval=10 ** 6)
    criterion = InstantNGPLoss(model)
    bg = torch.ones(3, device=DEV)
    noise = torch.full((o.shape[0],), 0.5, device=DEV)
    losses = []
    for _ in range(40):
        with torch.amp.autocast('cuda'):
            out = renderer.render_rays(o, d, cam, train_mode=True, custom_bg_color=bg, noise=noise)
            loss = torch.nn.functional.mse_loss(out['rgb'].float(), target) + 0.5e-6 * model.weight_decay_mlp()
        scaler.scale(loss).backward()
        scaler.step(opt); scaler.update(); opt.zero_grad()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.7 * losses[0], losses
    opt_c = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False, capturable=True)
    with pytest.raises(RuntimeError, match='HASHGRID_N_LEVELS'):
        FusedTrainingIteration(model, renderer, opt_c, scaler, cam, {'origin': o, 'view_direction': d, 'rgb': target}, 256, 65536)
