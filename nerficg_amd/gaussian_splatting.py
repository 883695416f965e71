"""nerficg_amd.gaussian_splatting -- host-side mirror of the reference's GaussianSplatting call sites into the rasterizer
(src/Methods/GaussianSplatting/Renderer.py:51-86,158-184), its activation accessors (Model.py:45-87) and the PyTorch
helpers that double as in-tree oracles for two rasterizer sub-steps (utils.py:10-59, Cameras/utils.py:180-208,
Cameras/Perspective.py:96-119).  Pure torch + nerficg_amd.diff_gaussian_rasterization; the helpers run on CPU too
(they are pinned against the reference's golden vectors in tests/test_host_golden.py).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

__all__ = ['PerspectiveCamera', 'get_projection_matrix', 'invert_3d_affine', 'make_raster_settings', 'quaternion_to_rotation_matrix',
           'build_covariances', 'extract_upper_triangular_matrix', 'convert_sh_features', 'rgb_to_sh0', 'sh0_to_rgb', 'Gaussians',
           'render_image_training', 'render_image_inference', 'training_loss']


@dataclass
class PerspectiveCamera:
    """Fields of src/Cameras/Perspective.py:16-37 + SharedCameraSettings that the 3DGS path reads."""
    width: int
    height: int
    focal_x: float
    focal_y: float
    center_x: float | None = None
    center_y: float | None = None
    near_plane: float = 0.01
    far_plane: float = 100.0
    background_color: torch.Tensor | None = None

    def __post_init__(self) -> None:
        if self.center_x is None:
            self.center_x = self.width / 2
        if self.center_y is None:
            self.center_y = self.height / 2
        if self.background_color is None:
            self.background_color = torch.zeros(3)


def get_projection_matrix(cam: PerspectiveCamera, invert_z: bool = False, device=None) -> torch.Tensor:
    """Cameras/Perspective.py:96-119 (OpenGL-style clip matrix, x/y/z in [-1,1] after the perspective division)."""
    half_width, half_height = cam.width * 0.5, cam.height * 0.5
    offset_x, offset_y = cam.center_x - half_width, cam.center_y - half_height
    z_sign = -1.0 if invert_z else 1.0
    f, n = cam.far_plane, cam.near_plane
    return torch.tensor([
        [cam.focal_x / half_width, 0.0, z_sign * offset_x / half_width, 0.0],
        [0.0, cam.focal_y / half_height, z_sign * offset_y / half_height, 0.0],
        [0.0, 0.0, z_sign * (f + n) / (f - n), -2.0 * f * n / (f - n)],
        [0.0, 0.0, z_sign, 0.0]], dtype=torch.float32, device=device)


def invert_3d_affine(transform: np.ndarray) -> np.ndarray:
    """Cameras/utils.py:211-222 for rigid transforms."""
    inv = np.eye(4, dtype=transform.dtype)
    inv[:3, :3] = transform[:3, :3].T
    inv[:3, 3] = transform[:3, :3].T @ -transform[:3, 3]
    return inv


def make_raster_settings(cam: PerspectiveCamera, c2w: np.ndarray, sh_degree: int, scale_modifier: float = 1.0, device='cuda'):
    """GaussianSplatting/Renderer.py:60-74: viewmatrix = w2c.T, projmatrix = w2c.T @ P.T, tanfov = size / focal / 2."""
    from .diff_gaussian_rasterization import GaussianRasterizationSettings
    c2w = np.asarray(c2w, dtype=np.float64)
    w2c = torch.as_tensor(invert_3d_affine(c2w), dtype=torch.float32, device=device).T
    return GaussianRasterizationSettings(
        image_height=cam.height, image_width=cam.width, tanfovx=cam.width / cam.focal_x * 0.5, tanfovy=cam.height / cam.focal_y * 0.5,
        bg=cam.background_color.to(device), scale_modifier=scale_modifier, viewmatrix=w2c,
        projmatrix=w2c @ get_projection_matrix(cam, device=device).T, sh_degree=sh_degree,
        campos=torch.as_tensor(c2w[:3, 3], dtype=torch.float32, device=device), prefiltered=False, debug=False)


def quaternion_to_rotation_matrix(quaternions: torch.Tensor, normalize: bool = True) -> torch.Tensor:
    """Cameras/utils.py:180-208 (torch branch)."""
    batch_dim_added = quaternions.ndim == 1
    if batch_dim_added:
        quaternions = quaternions[None]
    if normalize:
        quaternions = torch.nn.functional.normalize(quaternions)
    R = torch.empty((quaternions.shape[0], 3, 3), dtype=quaternions.dtype, device=quaternions.device)
    r, i, j, k = quaternions.T
    ii2, jj2, kk2 = i * i * 2, j * j * 2, k * k * 2
    ij2, ik2, jk2 = i * j * 2, i * k * 2, j * k * 2
    ri2, rj2, rk2 = r * i * 2, r * j * 2, r * k * 2
    R[:, 0, 0] = 1 - (jj2 + kk2); R[:, 0, 1] = ij2 - rk2; R[:, 0, 2] = ik2 + rj2
    R[:, 1, 0] = ij2 + rk2; R[:, 1, 1] = 1 - (ii2 + kk2); R[:, 1, 2] = jk2 - ri2
    R[:, 2, 0] = ik2 - rj2; R[:, 2, 1] = jk2 + ri2; R[:, 2, 2] = 1 - (ii2 + jj2)
    return R[0] if batch_dim_added else R


def build_covariances(scales: torch.Tensor, rotations: torch.Tensor) -> torch.Tensor:
    """GaussianSplatting/utils.py:10-18: R S (R S)^T."""
    R = quaternion_to_rotation_matrix(rotations, normalize=False)
    batch_dim_added = scales.dim() == 1
    if batch_dim_added:
        scales = scales[None]
    RS = R @ torch.diag_embed(scales)
    RSSR = RS @ RS.transpose(-2, -1)
    return RSSR[0] if batch_dim_added else RSSR


def extract_upper_triangular_matrix(matrix: torch.Tensor) -> torch.Tensor:
    """utils.py:70-73"""
    idx = torch.triu_indices(matrix.shape[-2], matrix.shape[-1])
    return matrix[..., idx[0], idx[1]]


def convert_sh_features(sh_features: torch.Tensor, view_directions: torch.Tensor, degree: int) -> torch.Tensor:
    """utils.py:21-59: SH (..., 3, 16) -> RGB, +0.5 and clamped at 0 from below (the rasterizer's fused SH step)."""
    result = 0.5 + 0.28209479177387814 * sh_features[..., 0]
    if degree == 0:
        return result.clamp_min(0.0)
    x, y, z = view_directions[..., 0:1], view_directions[..., 1:2], view_directions[..., 2:3]
    result = result + -0.48860251190291987 * y * sh_features[..., 1] + 0.48860251190291987 * z * sh_features[..., 2] + -0.48860251190291987 * x * sh_features[..., 3]
    if degree == 1:
        return result.clamp_min(0.0)
    x2, y2, z2 = x * x, y * y, z * z
    xy, yz, xz = x * y, y * z, x * z
    result = (result + 1.0925484305920792 * xy * sh_features[..., 4] + -1.0925484305920792 * yz * sh_features[..., 5]
              + (0.94617469575755997 * z2 - 0.31539156525251999) * sh_features[..., 6] + -1.0925484305920792 * xz * sh_features[..., 7]
              + 0.54627421529603959 * (x2 - y2) * sh_features[..., 8])
    if degree == 2:
        return result.clamp_min(0.0)
    result = (result + 0.59004358992664352 * y * (-3.0 * x2 + y2) * sh_features[..., 9] + 2.8906114426405538 * xy * z * sh_features[..., 10]
              + 0.45704579946446572 * y * (1.0 - 5.0 * z2) * sh_features[..., 11] + 0.3731763325901154 * z * (5.0 * z2 - 3.0) * sh_features[..., 12]
              + 0.45704579946446572 * x * (1.0 - 5.0 * z2) * sh_features[..., 13] + 1.4453057213202769 * z * (x2 - y2) * sh_features[..., 14]
              + 0.59004358992664352 * x * (-x2 + 3.0 * y2) * sh_features[..., 15])
    return result.clamp_min(0.0)


def rgb_to_sh0(rgb):
    return (rgb - 0.5) / 0.28209479177387814


def sh0_to_rgb(sh):
    return sh * 0.28209479177387814 + 0.5


class Gaussians(torch.nn.Module):
    """Parameter store + activation accessors of GaussianSplatting/Model.py:18-87 (exp scales, normalised quaternions, sigmoid
    opacities, cat[dc, rest] SH features of shape (P, 16, 3))."""

    def __init__(self, positions, log_scales, rotations, opacity_logits, features_dc, features_rest, sh_degree: int = 3) -> None:
        super().__init__()
        P = torch.nn.Parameter
        self._positions, self._scales, self._rotations = P(positions), P(log_scales), P(rotations)
        self._opacities, self._features_dc, self._features_rest = P(opacity_logits), P(features_dc), P(features_rest)
        self.active_sh_degree = sh_degree

    @property
    def get_positions(self): return self._positions
    @property
    def get_scales(self): return torch.exp(self._scales)
    @property
    def get_rotations(self): return torch.nn.functional.normalize(self._rotations)
    @property
    def get_opacities(self): return torch.sigmoid(self._opacities)
    @property
    def get_features(self): return torch.cat((self._features_dc, self._features_rest), dim=1)


def render_image_training(gaussians: Gaussians, cam: PerspectiveCamera, c2w: np.ndarray) -> dict[str, torch.Tensor]:
    """GaussianSplatting/Renderer.py:51-86."""
    from .diff_gaussian_rasterization import GaussianRasterizer
    positions = gaussians.get_positions
    viewspace_points = torch.zeros_like(positions, requires_grad=True) + 0
    viewspace_points.retain_grad()
    rasterizer = GaussianRasterizer(make_raster_settings(cam, c2w, gaussians.active_sh_degree, 1.0, positions.device))
    image, radii = rasterizer(means3D=positions, means2D=viewspace_points, shs=gaussians.get_features, opacities=gaussians.get_opacities,
                              scales=gaussians.get_scales, rotations=gaussians.get_rotations)
    return {'rgb': image, 'viewspace_points': viewspace_points, 'visibility_mask': radii > 0}


@torch.no_grad()
def render_image_inference(gaussians: Gaussians, cam: PerspectiveCamera, c2w: np.ndarray, scale_modifier: float = 1.0, to_chw: bool = False):
    """GaussianSplatting/Renderer.py:89-155 with the fused SH / covariance paths (USE_FUSED_* = True, the shipped defaults)."""
    from .diff_gaussian_rasterization import GaussianRasterizer
    positions = gaussians.get_positions
    rasterizer = GaussianRasterizer(make_raster_settings(cam, c2w, gaussians.active_sh_degree, scale_modifier, positions.device))
    image, _ = rasterizer(means3D=positions, means2D=torch.empty_like(positions), shs=gaussians.get_features,
                          opacities=gaussians.get_opacities, scales=gaussians.get_scales, rotations=gaussians.get_rotations)
    image.clamp_(0.0, 1.0)
    return {'rgb': image if to_chw else image.permute(1, 2, 0)}


def training_loss(image: torch.Tensor, target: torch.Tensor, lambda_l1: float = 0.8, lambda_dssim: float = 0.2) -> torch.Tensor:
    """GaussianSplattingLoss (src/Methods/GaussianSplatting/Loss.py:11-23, weights Trainer.py:34-35): lambda_l1 * L1 + lambda_dssim *
    (1 - SSIM) on (3, H, W) images, SSIM through the HIP kernels (nerficg_amd.fused_ssim; DSSIM.py:11-18 adds the batch dimension)."""
    from .fused_ssim import fused_ssim
    l1 = torch.nn.functional.l1_loss(image, target)
    a = image[None] if image.dim() == 3 else image
    b = target[None] if target.dim() == 3 else target
    return lambda_l1 * l1 + lambda_dssim * (1.0 - fused_ssim(a, b))
