"""Host-side mirrors (pure torch, CPU) against golden vectors generated from the reference's own Python
(tests/golden/make_golden.py): projection / rasterizer-settings marshalling, SH -> RGB, covariance construction,
quaternion -> rotation.  These are the in-tree pins of SURVEY.md 8(c) for a24 and a26."""
from pathlib import Path

import numpy as np
import pytest
import torch

from nerficg_amd import gaussian_splatting as gs


def test_projection_matrix_and_raster_settings_marshalling(golden_dir):
    g = np.load(golden_dir / 'projection.npz')
    w, h, fx, fy, cx, cy, near, far = g['intr']
    cam = gs.PerspectiveCamera(int(w), int(h), fx, fy, cx, cy, near, far)
    np.testing.assert_array_equal(gs.get_projection_matrix(cam).numpy(), g['P'])
    np.testing.assert_array_equal(gs.get_projection_matrix(cam, invert_z=True).numpy(), g['P_invz'])
    s = gs.make_raster_settings(cam, g['c2w'], 3, device='cpu')
    np.testing.assert_array_equal(s.viewmatrix.numpy(), g['viewmatrix'])
    np.testing.assert_allclose(s.projmatrix.numpy(), g['projmatrix'], rtol=1e-6, atol=1e-7)
    assert (s.tanfovx, s.tanfovy) == tuple(g['tanfov'])
    np.testing.assert_array_equal(s.campos.numpy(), g['campos'])


def test_sh_conversion_and_covariances(golden_dir):
    g = np.load(golden_dir / 'gs_utils.npz')
    sh, vd = torch.from_numpy(g['sh']), torch.from_numpy(g['view_dirs'])
    for deg in range(4):
        # f32, O(1) values; the basis-times-coefficients reduction adds in another order than the reference's running sum: a few ulp
        np.testing.assert_allclose(gs.convert_sh_features(sh.clone(), vd, deg).numpy(), g[f'rgb_deg{deg}'], rtol=1e-6, atol=5e-7)
    scales, quats = torch.from_numpy(g['scales']), torch.from_numpy(g['quats'])
    cov = gs.build_covariances(scales, quats)
    # f32 rotations carry ~1e-7 per entry whichever closed form is used; small off-diagonal entries of Sigma are differences of larger
    # products, so the bound is relative to the size of each matrix, not of each entry
    size = np.abs(g['cov']).max(axis=(1, 2), keepdims=True)
    assert (np.abs(cov.numpy() - g['cov']) <= 2e-6 * size).all()
    assert (np.abs(gs.extract_upper_triangular_matrix(cov).numpy() - g['cov_upper']) <= 2e-6 * size[:, 0]).all()
    np.testing.assert_allclose(gs.quaternion_to_rotation_matrix(quats, normalize=False).numpy(), g['rot'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(gs.quaternion_to_rotation_matrix(torch.from_numpy(g['rot_unnormalized_in'])).numpy(), g['rot_normalized'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gs.rgb_to_sh0(torch.linspace(0, 1, 7)).numpy(), g['rgb_to_sh0'], rtol=1e-6)


# ------------------------------------------------------------------------------------------------ vanilla NeRF path (config 1)
def test_frequency_encoding(golden_dir):
    from nerficg_amd import nerf
    g = np.load(golden_dir / 'freqenc.npz')
    x = torch.from_numpy(g['x'])
    np.testing.assert_allclose(nerf.FrequencyEncoding(10, True)(x).numpy(), g['pos10'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(nerf.FrequencyEncoding(4, True)(x).numpy(), g['dir4'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(nerf.FrequencyEncoding(6, False)(x).numpy(), g['noappend6'], rtol=1e-6, atol=1e-6)
    assert nerf.FrequencyEncoding(10, True).get_n_outputs(3) == 63 and nerf.FrequencyEncoding(4, True).get_n_outputs(3) == 27


def test_sampling_and_integration(golden_dir):
    from nerficg_amd import nerf
    g = np.load(golden_dir / 'nerf_sampling.npz')
    n = g['depth'].shape[0]
    depth = nerf.generate_samples(n, 64, 2.0, 6.0, False)
    np.testing.assert_array_equal(depth.numpy(), g['depth'])
    torch.manual_seed(1)
    np.testing.assert_allclose(nerf.generate_samples(n, 64, 2.0, 6.0, True).numpy(), g['depth_rand'], rtol=1e-6)
    rgb, dpt, alpha, w = nerf.integrate_samples(depth, torch.from_numpy(g['dirs']), torch.from_numpy(g['dens']), torch.from_numpy(g['cols']), torch.from_numpy(g['bg']))
    np.testing.assert_allclose(rgb.numpy(), g['rgb'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(dpt.numpy(), g['depth_out'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(alpha.numpy(), g['alpha'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(w.numpy(), g['weights'], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(nerf.generate_samples_from_pdf(depth, w, 192, False).numpy(), g['fine'], rtol=1e-6, atol=1e-6)
    torch.manual_seed(2)
    np.testing.assert_allclose(nerf.generate_samples_from_pdf(depth, w, 192, True).numpy(), g['fine_rand'], rtol=1e-6, atol=1e-6)


def test_nerf_block_and_hierarchical_renderer(golden_dir):
    from nerficg_amd import nerf
    g = np.load(golden_dir / 'nerf_render.npz')
    kw = dict(n_layers=8, n_color_layers=1, n_features=32, n_frequencies_position=10, n_frequencies_direction=4, encoding_append_input=True, input_skips=[5])
    coarse, fine = nerf.NeRFBlock(**kw), nerf.NeRFBlock(**kw)
    for name, block in (('coarse', coarse), ('fine', fine)):
        sd = {k[len(name) + 1:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(name + '.')}
        block.load_state_dict(sd, strict=True)  # identical module / parameter names as the reference
    with torch.no_grad():
        dens, col = fine(torch.from_numpy(g['block_pts']), torch.from_numpy(g['view_direction'][:1]).expand(40, 3).contiguous())
        out = nerf.render_rays(coarse, fine, torch.from_numpy(g['origin']), torch.from_numpy(g['direction']), torch.from_numpy(g['view_direction']),
                               2.0, 6.0, torch.ones(3), ray_batch_size=10, n_samples_coarse_nerf=16, n_samples_nerf=24)
    np.testing.assert_allclose(dens.numpy(), g['block_dens'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(col.numpy(), g['block_col'], rtol=1e-5, atol=1e-6)
    for k in ('rgb', 'alpha', 'depth', 'rgb_coarse', 'alpha_coarse', 'depth_coarse'):
        np.testing.assert_allclose(out[k].numpy(), g[f'out_{k}'], rtol=2e-5, atol=2e-6, err_msg=k)


def test_random_sequential_sampler_defines_the_ray_indices(golden_dir):
    from nerficg_amd.samplers import RandomSequentialSampler
    g = np.load(golden_dir / 'misc.npz')
    torch.manual_seed(0)
    s = RandomSequentialSampler(1000)
    draws = np.stack([s.get(300).numpy().copy() for _ in range(5)])
    np.testing.assert_array_equal(draws, g['sampler_draws'])  # includes the reshuffle on wrap-around


def test_ray_batch_container_and_pool_sampler():
    """RayBatch / RayCollection / RayPoolSampler semantics (Datasets/utils.py:537-690, DatasetSamplers.py:53-66) on CPU tensors."""
    from nerficg_amd.rays import RayBatch, RayCollection, RayPoolSampler
    n = 50
    o, d = torch.arange(n * 3, dtype=torch.float32).reshape(n, 3), torch.ones(n, 3)
    rgb = torch.rand(n, 3)
    b = RayBatch(origin=o, direction=d, rgb=rgb)
    assert len(b) == n and b.has_annotations and b.as_tensor.shape == (n, 9) and b[...] is b and len(b[3]) == 1
    ids = torch.tensor([4, 1, 7])
    sub = b[ids]
    assert torch.equal(sub.origin, o[ids]) and torch.equal(sub.rgb, rgb[ids]) and sub.view_direction is None
    parts = b.split(16)
    assert [len(p) for p in parts] == [16, 16, 16, 2] and torch.equal(RayBatch.cat(parts).origin, o)
    with pytest.raises(ValueError):
        RayBatch(origin=o, direction=d[:10])
    with pytest.raises(ValueError):
        RayBatch.cat([b, RayBatch(origin=o, direction=d)])
    col = RayCollection(rays=b, rays_per_view=(20, 30))
    assert len(col) == 2 and torch.equal(col[1].origin, o[20:]) and col.all_rays is b
    g = np.load(Path(__file__).parent / 'golden' / 'misc.npz')
    torch.manual_seed(0)
    pool = RayPoolSampler(RayBatch(origin=torch.zeros(1000, 3), direction=torch.zeros(1000, 3)))
    draws = np.stack([pool.get(300)['ray_ids'].numpy().copy() for _ in range(5)])
    np.testing.assert_array_equal(draws, g['sampler_draws'])  # the reference's RandomSequentialSampler draws define the ray ids


def test_project_points_matches_reference_view(golden_dir):
    from nerficg_amd.instant_ngp import Camera, project_points
    g = np.load(golden_dir / 'raygen.npz')
    w, h, fx, fy, cx, cy, near, far = g['proj_intr']
    cam = Camera(width=int(w), height=int(h), focal_x=float(fx), focal_y=float(fy), center_x=float(cx), center_y=float(cy), near_plane=float(near),
                 far_plane=float(far))
    xy, depth, inside = project_points(cam, g['proj_c2w'], torch.from_numpy(g['proj_pts']))
    np.testing.assert_array_equal(inside.numpy(), g['proj_in_frustum'])
    np.testing.assert_allclose(depth.numpy(), g['proj_depth'], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(xy.numpy()[g['proj_in_frustum']], g['proj_xy'][g['proj_in_frustum']], rtol=1e-5, atol=1e-4)
    assert 0 < g['proj_in_frustum'].sum() < len(g['proj_in_frustum'])
