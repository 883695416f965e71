"""Host-side mirrors (pure torch, CPU) against golden vectors generated from the reference's own Python
(tests/golden/make_golden.py): projection / rasterizer-settings marshalling, SH -> RGB, covariance construction,
quaternion -> rotation.  These are the in-tree pins of SURVEY.md 8(c) for a24 and a26."""
import numpy as np
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
        np.testing.assert_allclose(gs.convert_sh_features(sh.clone(), vd, deg).numpy(), g[f'rgb_deg{deg}'], rtol=1e-6, atol=1e-7)
    scales, quats = torch.from_numpy(g['scales']), torch.from_numpy(g['quats'])
    cov = gs.build_covariances(scales, quats)
    np.testing.assert_allclose(cov.numpy(), g['cov'], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(gs.extract_upper_triangular_matrix(cov).numpy(), g['cov_upper'], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(gs.quaternion_to_rotation_matrix(quats, normalize=False).numpy(), g['rot'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(gs.quaternion_to_rotation_matrix(torch.from_numpy(g['rot_unnormalized_in'])).numpy(), g['rot_normalized'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gs.rgb_to_sh0(torch.linspace(0, 1, 7)).numpy(), g['rgb_to_sh0'], rtol=1e-6)
