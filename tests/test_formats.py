"""CPU: on-disk formats.  tests/golden/gs_reference_checkpoint.pt was written by the reference's own
GaussianSplattingModel.save after bake_activations (tests/golden/make_golden.py::make_gs_densify); the loader must read it, and what this
framework writes must carry exactly the same keys and value types."""
from pathlib import Path

import numpy as np
import pytest
import torch

from nerficg_amd import formats

GOLDEN = Path(__file__).parent / 'golden'


def test_reads_a_checkpoint_written_by_the_reference(tmp_path):
    g, meta = formats.gaussians_from_checkpoint(GOLDEN / 'gs_reference_checkpoint.pt', device='cpu')
    assert meta['model_name'] == 'golden' and meta['num_iterations_trained'] == 30000 and isinstance(meta['output_directory'], Path)
    assert g.baked and g.active_sh_degree == 3 and g.max_sh_degree == 3
    n = g.get_positions.shape[0]
    assert g._features_dc.shape == (n, 1, 3) and g._features_rest.shape == (n, 15, 3) and g.get_baked_covariances.shape == (n, 6)
    # trained checkpoints hold activated values: accessors are identities (Model.py:21-24), and the baked covariances are R S S^T R^T of them
    assert g.get_scales is g._scales and g.get_opacities is g._opacities and g.get_rotations is g._rotations
    assert float(g.get_opacities.detach().min()) >= 1 / 255 and float(g.get_opacities.detach().max()) <= 1.0 and float(g.get_scales.detach().min()) > 0.0
    np.testing.assert_allclose(g.get_rotations.norm(dim=1).detach().numpy(), 1.0, rtol=1e-6)
    np.testing.assert_allclose(g.get_covariances(1.0).detach().numpy(), g.get_baked_covariances.detach().numpy(), rtol=1e-5, atol=1e-9)
    raw = np.load(GOLDEN / 'gs_densify.npz')['ckpt_raw_opacities']
    kept = 1.0 / (1.0 + np.exp(-raw[:, 0].astype(np.float64)))
    np.testing.assert_allclose(g.get_opacities.detach().numpy()[:, 0], kept[kept >= 1 / 255], rtol=1e-6)
    # write it back: same dictionary keys, same state keys, same value types as the reference's file
    out = tmp_path / 'again.pt'
    formats.gaussians_to_checkpoint(g, out, model_name=meta['model_name'], creation_date=meta['creation_date'],
                                    num_iterations_trained=meta['num_iterations_trained'], output_directory=meta['output_directory'])
    ref, mine = formats.load_checkpoint(GOLDEN / 'gs_reference_checkpoint.pt'), formats.load_checkpoint(out)
    assert list(ref.keys()) == list(mine.keys()) and {k: type(v) for k, v in ref.items()} == {k: type(v) for k, v in mine.items()}
    assert list(ref['model_state_dict']) == list(mine['model_state_dict'])
    for k, v in ref['model_state_dict'].items():
        assert torch.equal(v, mine['model_state_dict'][k]) and v.dtype == mine['model_state_dict'][k].dtype, k
    assert all(ref[k] == mine[k] for k in ('model_name', 'creation_date', 'num_iterations_trained', 'output_directory', 'SH_DEGREE'))


def test_untrained_gaussians_round_trip_with_activations(tmp_path):
    from nerficg_amd.gaussian_splatting import Gaussians
    torch.manual_seed(0)
    g = Gaussians(torch.randn(9, 3), torch.randn(9, 3), torch.randn(9, 4), torch.randn(9, 1), torch.randn(9, 1, 3), torch.randn(9, 15, 3))
    formats.gaussians_to_checkpoint(g, tmp_path / 'u.pt', model_name='u')
    h, meta = formats.gaussians_from_checkpoint(tmp_path / 'u.pt', device='cpu')
    assert not h.baked and h.active_sh_degree == 0 and meta['num_iterations_trained'] == 0 and h.get_baked_covariances is None
    assert torch.equal(h.get_scales, torch.exp(g._scales)) and torch.equal(h._rotations, g._rotations)
    g.baked = True
    with pytest.raises(ValueError):
        formats.gaussians_to_checkpoint(g, tmp_path / 'bad.pt')  # activated values saved as "untrained" would be activated twice on load
    with pytest.raises(ValueError):
        formats.load_checkpoint(tmp_path / 'u.txt')


def test_instant_ngp_checkpoint_round_trip_and_shape_errors(tmp_path):
    from nerficg_amd.instant_ngp import InstantNGPModel
    m = InstantNGPModel(RESOLUTION=32, HASHGRID_LOG2_SIZE=14, SCALE=1.0, CENTER=(0.1, 0.0, -0.2), device='cpu', RANDOM_SEED=3)
    m.occupancy_grid.uniform_(-1, 1)
    m.occupancy_bitfield.random_(0, 255)
    formats.instant_ngp_to_checkpoint(m, tmp_path / 'ngp.pt', model_name='ngp', num_iterations_trained=123)
    ck = formats.load_checkpoint(tmp_path / 'ngp.pt')
    assert set(ck['model_state_dict']) == {'occupancy_grid', 'occupancy_bitfield', 'encoding_xyz.params', 'color_mlp_with_encoding.params'}  # Model.py:55-110
    assert set(formats.INSTANT_NGP_PARAMETERS) <= set(ck) and ck['SCALE'] == 1.0 and ck['CENTER'] == [0.1, 0.0, -0.2] and ck['ENABLE_JIT_FUSION'] is True
    assert ck['model_state_dict']['encoding_xyz.params'].dtype == torch.float32 and ck['model_state_dict']['encoding_xyz.params'].dim() == 1
    m2, meta = formats.instant_ngp_from_checkpoint(tmp_path / 'ngp.pt', device='cpu', RANDOM_SEED=99)
    assert meta['num_iterations_trained'] == 123 and m2.cascades == m.cascades == 2 and m2.RESOLUTION == 32
    for (k, a), b in zip(m.state_dict().items(), m2.state_dict().values()):
        assert torch.equal(a, b), k
    ck['HASHGRID_LOG2_SIZE'] = 15  # a table size the stored vector does not have
    torch.save(ck, tmp_path / 'bad.pt')
    with pytest.raises(ValueError, match='encoding_xyz.params'):
        formats.instant_ngp_from_checkpoint(tmp_path / 'bad.pt', device='cpu')


@pytest.mark.parametrize('use_ascii', [False, True])
def test_ply_export_layout_and_round_trip(tmp_path, use_ascii):
    g, _ = formats.gaussians_from_checkpoint(GOLDEN / 'gs_reference_checkpoint.pt', device='cpu')
    data = formats.gaussians_ply_dict(g)
    formats.write_ply(tmp_path / 'm.ply', data, use_ascii)
    raw = (tmp_path / 'm.ply').read_bytes()
    header = raw[:raw.index(b'end_header\n')].decode('ascii').split('\n')
    n = g.get_positions.shape[0]
    assert header[:5] == ['ply', f'format {"ascii" if use_ascii else "binary_little_endian"} 1.0', 'comment SplatRenderMode: default',
                          'comment Generated with NeRFICG/GaussianSplatting', f'element vertex {n}']
    props = [line.split() for line in header[5:] if line]
    assert all(p[0] == 'property' and p[1] == 'float' for p in props)
    assert [p[2] for p in props] == (['x', 'y', 'z', 'f_dc_0', 'f_dc_1', 'f_dc_2'] + [f'f_rest_{i}' for i in range(45)] + ['opacity', 'scale_0', 'scale_1',
                                                                                                                  'scale_2', 'rot_0', 'rot_1', 'rot_2', 'rot_3'])
    if not use_ascii:
        assert len(raw) == raw.index(b'end_header\n') + len(b'end_header\n') + n * 59 * 4
    back = formats.read_ply(tmp_path / 'm.ply')
    assert back['comments'] == data['comments']
    for name in data['vertex'].dtype.names:
        np.testing.assert_allclose(back['vertex'][name], data['vertex'][name], rtol=0 if not use_ascii else 1e-8)
    # viewers expect unactivated opacity / scale (Model.py:289-290): logit and log of the baked values; SH rest is channel-major
    np.testing.assert_allclose(back['vertex']['opacity'], torch.logit(g.get_opacities)[:, 0].detach().numpy(), rtol=1e-6)
    np.testing.assert_allclose(back['vertex']['f_rest_15'], g._features_rest[:, 0, 1].detach().numpy())


def test_empty_model_exports_nothing():
    from nerficg_amd.gaussian_splatting import Gaussians
    g = Gaussians(torch.zeros(0, 3), torch.zeros(0, 3), torch.zeros(0, 4), torch.zeros(0, 1), torch.zeros(0, 1, 3), torch.zeros(0, 15, 3))
    assert formats.gaussians_ply_dict(g) == {}
