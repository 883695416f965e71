"""GPU: 3DGS densification bookkeeping in HIP (C ABI group 10, nerficg_amd.gaussian_splatting.Gaussians / nerficg_amd.adam_utils) against
oracle/gs_densify.py and the reference's golden vectors (tests/golden/gs_densify.npz).  Row bookkeeping (which Gaussian lands where, with
which Adam moments) is compared bit-exactly; the split children's positions / log-scales within f32 rounding of exp/log/normalise."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import gs_densify as og

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)
GOLDEN = Path(__file__).parent / 'golden' / 'gs_densify.npz'
ATTR = {'positions': '_positions', 'f_dc': '_features_dc', 'f_rest': '_features_rest', 'opacities': '_opacities', 'scales': '_scales',
        'rotations': '_rotations'}


def build(params, moments=None, extent=4.0, percent_dense=0.01, optimizer_class=None):
    from nerficg_amd.gaussian_splatting import Gaussians
    t = {k: torch.from_numpy(np.ascontiguousarray(params[k], np.float32)).to(DEV) for k in og.GROUPS}
    m = Gaussians(t['positions'], t['scales'], t['rotations'], t['opacities'], t['f_dc'], t['f_rest'], sh_degree=3)
    m.training_setup(PERCENT_DENSE=percent_dense, training_cameras_extent=extent, optimizer_class=optimizer_class)
    if moments is not None:
        for group in m.optimizer.param_groups:
            p = group['params'][0]
            st = m.optimizer.state[p]
            st['exp_avg'] = torch.from_numpy(np.ascontiguousarray(moments[group['name']][0])).to(DEV)
            st['exp_avg_sq'] = torch.from_numpy(np.ascontiguousarray(moments[group['name']][1])).to(DEV)
    return m


def state_of(m):
    out = {}
    for group in m.optimizer.param_groups:
        p = group['params'][0]
        assert getattr(m, ATTR[group['name']]) is p  # the model's attributes are the optimizer's parameters
        st = m.optimizer.state[p]
        out[group['name']] = (p.detach().cpu().numpy(), st['exp_avg'].cpu().numpy() if st else None, st['exp_avg_sq'].cpu().numpy() if st else None)
    return out


def compare(got, want_p, want_m):
    for k in og.GROUPS:
        p, ea, eas = got[k]
        assert p.shape == want_p[k].shape, k
        if k in ('positions', 'scales'):
            np.testing.assert_allclose(p, want_p[k], rtol=3e-6, atol=2e-6, err_msg=k)
        else:
            np.testing.assert_array_equal(p, want_p[k], err_msg=k)
        if want_m is not None:
            np.testing.assert_array_equal(ea, want_m[k][0], err_msg=k)
            np.testing.assert_array_equal(eas, want_m[k][1], err_msg=k)


def test_reference_golden_case_end_to_end():
    g = np.load(GOLDEN)
    m = build({k: g['in_' + k] for k in og.GROUPS}, {k: (g[f'in_{k}_exp_avg'], g[f'in_{k}_exp_avg_sq']) for k in og.GROUPS},
              float(g['extent']), float(g['percent_dense']))
    m.densification_gradient_accum = torch.from_numpy(g['in_accum']).to(DEV)
    m.n_observations = torch.from_numpy(g['in_n_obs']).to(DEV)
    vsp = torch.zeros(g['vsp_grad'].shape, device=DEV, requires_grad=True)
    vsp.grad = torch.from_numpy(g['vsp_grad']).to(DEV)
    m.add_densification_stats(vsp, torch.from_numpy(g['radii']).to(DEV) > 0)  # boolean mask as the reference's renderer hands it over
    np.testing.assert_allclose(m.densification_gradient_accum.cpu().numpy(), g['stats_accum'], rtol=2e-7, atol=0)
    np.testing.assert_array_equal(m.n_observations.cpu().numpy(), g['stats_n_obs'])
    info = m.densify_and_prune(float(g['grad_threshold']), float(g['min_opacity']), True, noise=torch.from_numpy(g['noise']).to(DEV))
    assert 2 * info['n_split'] == g['noise'].shape[0] and info['n_out'] == g['out_positions'].shape[0]
    compare(state_of(m), {k: g['out_' + k] for k in og.GROUPS}, {k: (g[f'out_{k}_exp_avg'], g[f'out_{k}_exp_avg_sq']) for k in og.GROUPS})
    assert m.densification_gradient_accum.shape == (info['n_out'], 1) and float(m.densification_gradient_accum.abs().max()) == 0.0
    assert m.n_observations.dtype == torch.int32 and int(m.n_observations.abs().max()) == 0
    # adam_utils.py:64-98 on the densified model, then Model.py:152-155
    from nerficg_amd import adam_utils
    order = torch.from_numpy(g['sort_order']).to(DEV)
    new = adam_utils.sort_param_groups(m.optimizer, order, ['positions', 'rotations'])
    adam_utils.reset_state(m.optimizer, ['positions'], torch.from_numpy(g['reset_idx']).to(DEV))
    np.testing.assert_allclose(new['positions'].detach().cpu().numpy(), g['sorted_positions'], rtol=3e-6, atol=2e-6)
    np.testing.assert_array_equal(m.optimizer.state[new['positions']]['exp_avg'].cpu().numpy(), g['sorted_positions_exp_avg'])
    np.testing.assert_array_equal(m.optimizer.state[new['rotations']]['exp_avg_sq'].cpu().numpy(), g['sorted_rotations_exp_avg_sq'])
    m.reset_opacities()
    np.testing.assert_allclose(m.optimizer.param_groups[3]['params'][0].detach().cpu().numpy(), g['reset_opacities'], rtol=2e-6, atol=2e-6)


def random_model(P, seed, with_state=True):
    rng = np.random.default_rng(seed)
    f = np.float32
    params = {'positions': rng.normal(size=(P, 3)).astype(f) * 2, 'f_dc': rng.normal(size=(P, 1, 3)).astype(f),
              'f_rest': rng.normal(size=(P, 15, 3)).astype(f) * f(0.1), 'opacities': (rng.normal(size=(P, 1)) * 3 - 2).astype(f),
              'scales': (rng.normal(size=(P, 3)) * 1.2 - 3.2).astype(f), 'rotations': rng.normal(size=(P, 4)).astype(f)}
    mom = {k: (rng.normal(size=v.shape).astype(f), rng.random(size=v.shape).astype(f)) for k, v in params.items()} if with_state else None
    accum = (rng.random(size=(P, 1)) * 1.2e-3).astype(f)
    n_obs = rng.integers(0, 9, size=(P, 1)).astype(np.int32)
    return params, mom, accum, n_obs, rng


@pytest.mark.parametrize('P,prune_large,with_state', [(1, True, True), (1023, False, True), (1025, True, False), (50_000, True, True), (300_001, False, True)])
def test_densify_and_prune_matches_oracle(P, prune_large, with_state):
    params, mom, accum, n_obs, rng = random_model(P, P, with_state)
    noise = rng.normal(size=(2 * P, 3)).astype(np.float32)
    want_p, want_m, n_split = og.densify_and_prune(params, mom, accum, n_obs, 2e-4, 0.005, prune_large, 0.01, 4.0, noise)
    m = build(params, mom, optimizer_class=None if with_state else torch.optim.Adam)
    m.densification_gradient_accum, m.n_observations = torch.from_numpy(accum).to(DEV), torch.from_numpy(n_obs).to(DEV)
    info = m.densify_and_prune(2e-4, 0.005, prune_large, noise=torch.from_numpy(noise).to(DEV))
    assert info['n_split'] == n_split and info['n_out'] == want_p['positions'].shape[0]
    compare(state_of(m), want_p, want_m)
    assert not m.optimizer.state[m._positions] if not with_state else True
    # the model is still trainable: one fused Adam step over the new tensors
    for group in m.optimizer.param_groups:
        group['params'][0].grad = torch.ones_like(group['params'][0])
    m.optimizer.step()


def test_nothing_selected_and_everything_pruned():
    params, mom, accum, n_obs, rng = random_model(4096, 11)
    m = build(params, mom)
    m.densification_gradient_accum, m.n_observations = torch.zeros(4096, 1, device=DEV), torch.ones(4096, 1, dtype=torch.int32, device=DEV)
    info = m.densify_and_prune(2e-4, 0.0, False)  # no gradient above the threshold, nothing below opacity 0: identity
    assert info == {'n_out': 4096, 'n_kept': 4096, 'n_cloned': 0, 'n_split': 0, 'n_children_kept': 0}
    compare(state_of(m), params, mom)
    info = m.densify_and_prune(2e-4, 2.0, False)  # every opacity < 2: empty model, tensors keep their trailing shapes
    assert info['n_out'] == 0 and m._features_rest.shape == (0, 15, 3) and m.optimizer.state[m._features_rest]['exp_avg'].shape == (0, 15, 3)
    with pytest.raises(RuntimeError):
        build(params, mom).densify_and_prune(0.0, 0.005, True)  # threshold <= 0 is outside the plan (NRC_ERR_UNSUPPORTED)


def test_prune_points_and_mask_compaction():
    from nerficg_amd import adam_utils
    params, mom, accum, n_obs, rng = random_model(70_001, 5)
    for n in (0, 1, 1024, 70_001):
        mask = rng.random(n) < 0.37
        idx = adam_utils.compact_mask(torch.from_numpy(mask).to(DEV))
        assert idx.dtype == torch.int32
        np.testing.assert_array_equal(idx.cpu().numpy(), np.nonzero(mask)[0])
    m = build(params, mom)
    m.densification_gradient_accum, m.n_observations = torch.from_numpy(accum).to(DEV), torch.from_numpy(n_obs).to(DEV)
    prune = rng.random(70_001) < 0.5
    m.prune_points(torch.from_numpy(prune).to(DEV))
    want_p, want_m = og.prune(params, mom, ~prune)
    compare(state_of(m), want_p, want_m)
    np.testing.assert_array_equal(m.densification_gradient_accum.cpu().numpy(), accum[~prune])
    np.testing.assert_array_equal(m.n_observations.cpu().numpy(), n_obs[~prune])
    with pytest.raises(RuntimeError):
        adam_utils.compact_mask(torch.zeros(4, device=DEV))  # not a boolean mask
    with pytest.raises(RuntimeError):
        adam_utils.gather_rows([torch.zeros(4, 3)], torch.zeros(2, dtype=torch.int32, device=DEV), 2)  # CPU tensor: no fallback


def test_bake_activations_orders_by_morton_code():
    import oracle
    params, _, _, _, rng = random_model(20_000, 9, with_state=False)
    m = build(params)
    m.bake_activations()
    opac = 1.0 / (1.0 + np.exp(-params['opacities'][:, 0].astype(np.float64)))
    keep = opac >= 0.00392156862 + 1e-7
    assert abs(m._positions.shape[0] - int(keep.sum())) <= 2  # 1-ulp sigmoid differences at the 1/255 threshold at most
    pos = m._positions.detach().cpu().numpy()
    codes = oracle.morton_encode(pos)
    assert np.all(np.diff(codes.astype(np.int64)) >= 0)  # Morton order (Model.py:262-269)
    # every kept row is one of the originals with its activated attributes
    src = {tuple(r): i for i, r in enumerate(params['positions'])}
    ids = np.array([src[tuple(r)] for r in pos])
    np.testing.assert_allclose(m._scales.detach().cpu().numpy(), np.exp(params['scales'][ids]), rtol=2e-6)
    q = params['rotations'][ids]
    np.testing.assert_allclose(m._rotations.detach().cpu().numpy(), q / np.linalg.norm(q, axis=1, keepdims=True), rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(m._features_rest.detach().cpu().numpy(), params['f_rest'][ids])
    assert m.baked and m.get_scales is m._scales and m.get_baked_covariances.shape == (pos.shape[0], 6) and not m.get_baked_covariances.requires_grad
