"""CPU: oracle/gs_densify.py (numpy restatement of Model.py:157-246 + adam_utils.py) pinned on tests/golden/gs_densify.npz, which the
reference's own Gaussians class and adam_utils produced on CPU (tests/golden/make_golden.py::make_gs_densify); plus the pure-torch host
pieces of the mirror that need no GPU (LR policy, ply dictionary, replace / extend / reset of optimizer state)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import gs_densify as og

GOLDEN = Path(__file__).parent / 'golden' / 'gs_densify.npz'
NAMES = {'positions': '_positions', 'f_dc': '_features_dc', 'f_rest': '_features_rest', 'opacities': '_opacities', 'scales': '_scales',
         'rotations': '_rotations'}


@pytest.fixture(scope='module')
def g():
    return np.load(GOLDEN)


def test_stats_oracle_matches_reference(g):
    acc, nobs = og.add_densification_stats(g['in_accum'], g['in_n_obs'], g['vsp_grad'], g['radii'])
    np.testing.assert_allclose(acc, g['stats_accum'], rtol=2e-7, atol=0)  # one sqrt of a two-term sum
    np.testing.assert_array_equal(nobs, g['stats_n_obs'])


def test_densify_and_prune_oracle_matches_reference(g):
    params = {k: g['in_' + k] for k in og.GROUPS}
    mom = {k: (g[f'in_{k}_exp_avg'], g[f'in_{k}_exp_avg_sq']) for k in og.GROUPS}
    p, m, n_split = og.densify_and_prune(params, mom, g['stats_accum'], g['stats_n_obs'], float(g['grad_threshold']), float(g['min_opacity']), True,
                                         float(g['percent_dense']), float(g['extent']), g['noise'])
    assert 2 * n_split == g['noise'].shape[0]
    for k in og.GROUPS:
        assert p[k].shape == g['out_' + k].shape, k
        if k in ('positions', 'scales'):  # split children: R z s + p and log(s / 1.6) -- f32 rounding of exp/log/normalise
            np.testing.assert_allclose(p[k], g['out_' + k], rtol=2e-6, atol=1e-6, err_msg=k)
        else:  # pure row bookkeeping: bit-exact
            np.testing.assert_array_equal(p[k], g['out_' + k], err_msg=k)
        np.testing.assert_array_equal(m[k][0], g[f'out_{k}_exp_avg'], err_msg=k)
        np.testing.assert_array_equal(m[k][1], g[f'out_{k}_exp_avg_sq'], err_msg=k)


def test_lr_decay_policy_matches_reference():
    from nerficg_amd.lr_utils import LRDecayPolicy
    g = np.load(GOLDEN.parent / 'misc.npz')
    pol = LRDecayPolicy(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_steps=100, lr_delay_mult=0.01, max_steps=30000)
    np.testing.assert_allclose([pol(int(i)) for i in g['lr_its']], g['lr_vals'], rtol=1e-12)
    assert LRDecayPolicy(lr_init=0.0, lr_final=0.0)(5) == 0.0 and pol(-1) == 0.0


def _cpu_model(g, prefix='in_'):
    from nerficg_amd.gaussian_splatting import Gaussians
    t = {k: torch.from_numpy(g[prefix + k].copy()) for k in og.GROUPS}
    return Gaussians(t['positions'], t['scales'], t['rotations'], t['opacities'], t['f_dc'], t['f_rest'], sh_degree=3)


def test_ply_dictionary_matches_reference(g):
    ply = _cpu_model(g).as_ply_dict()['vertex']
    assert list(ply.dtype.names) == [str(n) for n in g['ply_names']]
    rows = np.stack([ply[n] for n in ply.dtype.names], axis=1)
    np.testing.assert_allclose(rows, g['ply_rows'], rtol=1e-6, atol=1e-7)  # logit(sigmoid(x)), log(exp(x)), normalise: torch ops on both sides
    assert all(ply.dtype[n] == np.dtype('f4') for n in ply.dtype.names)


def test_state_surgery_without_row_moves(g):
    """replace_param_group_data / reset_state / extend_param_groups (adam_utils.py:6-18, 42-79) are pure torch in the mirror as well."""
    from nerficg_amd import adam_utils
    m = _cpu_model(g, 'out_')
    m.training_setup(optimizer_class=torch.optim.Adam)
    # Model.py:152-155 (before any step: the optimizer has no state yet, the values alone are compared)
    m.reset_opacities()
    np.testing.assert_allclose(m.optimizer.param_groups[3]['params'][0].detach().numpy(), g['reset_opacities'], rtol=1e-6, atol=1e-6)
    for group in m.optimizer.param_groups:
        group['params'][0].grad = torch.ones_like(group['params'][0])
    m.optimizer.step()
    m.reset_opacities()  # ... and with state: both moments are cleared
    new_op = m.optimizer.param_groups[3]['params'][0]
    assert float(m.optimizer.state[new_op]['exp_avg'].abs().max()) == 0.0 and float(m.optimizer.state[new_op]['exp_avg_sq'].abs().max()) == 0.0
    idx = torch.tensor([0, 5, 9])
    adam_utils.reset_state(m.optimizer, ['positions'], idx)
    st = m.optimizer.state[m.optimizer.param_groups[0]['params'][0]]
    assert float(st['exp_avg'][idx].abs().max()) == 0.0 and float(st['exp_avg'][1].abs().max()) > 0.0
    n0 = m._rotations.shape[0]
    new = adam_utils.extend_param_groups(m.optimizer, {'rotations': torch.ones(4, 4), 'scales': torch.zeros(4, 3)})
    assert set(new) == {'rotations', 'scales'} and new['rotations'].shape == (n0 + 4, 4)
    st = m.optimizer.state[new['rotations']]
    assert st['exp_avg'].shape == (n0 + 4, 4) and float(st['exp_avg'][n0:].abs().max()) == 0.0 and float(st['exp_avg'][:n0].abs().min()) > 0.0
    m.optimizer.param_groups[0]['params'].append(torch.nn.Parameter(torch.zeros(1)))
    with pytest.raises(NotImplementedError):
        adam_utils.reset_state(m.optimizer)
