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
           'build_covariances', 'extract_upper_triangular_matrix', 'sh_basis', 'convert_sh_features', 'rgb_to_sh0', 'sh0_to_rgb', 'Gaussians',
           'render_image_training', 'render_image_inference', 'training_loss', 'rest_step_schedule', 'RestStep']


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


_PROJECTIONS: dict = {}


def _projection_transposed(cam: PerspectiveCamera, device) -> torch.Tensor:
    """P.T of get_projection_matrix on `device`, built once per camera geometry (a host -> device copy per frame otherwise)."""
    key = (cam.width, cam.height, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, cam.near_plane, cam.far_plane, str(device))
    hit = _PROJECTIONS.get(key)
    if hit is None:
        if len(_PROJECTIONS) >= 64:
            _PROJECTIONS.pop(next(iter(_PROJECTIONS)))
        hit = _PROJECTIONS[key] = get_projection_matrix(cam, device=device).T.contiguous()
    return hit


def make_raster_settings(cam: PerspectiveCamera, c2w, sh_degree: int, scale_modifier: float = 1.0, device='cuda'):
    """GaussianSplatting/Renderer.py:60-74: viewmatrix = w2c.T, projmatrix = w2c.T @ P.T, tanfov = size / focal / 2.  `c2w`: a (4,4) / (3,4)
    array, or a device tensor -- the pose then never visits the host (w2c.T = [[R, 0], [-(R^T t)^T, 1]] is assembled by torch on the
    device), which is what a recorded training step needs: it re-reads the tensor on every replay."""
    from .diff_gaussian_rasterization import GaussianRasterizationSettings
    if torch.is_tensor(c2w) and c2w.is_cuda:
        # one launch writes the rasterizer's whole camera block (view | view @ P^T | position | background); the settings' tensors are views of it and
        # the rasterizer recognises the block (diff_gaussian_rasterization.adopt_camera_block) instead of concatenating its own
        from . import _lib
        from .diff_gaussian_rasterization import adopt_camera_block
        pose = c2w.to(device=device, dtype=torch.float32).contiguous()
        bg = cam.background_color.to(device=device, dtype=torch.float32).contiguous()
        block = torch.empty(38, dtype=torch.float32, device=pose.device)
        _lib.check(_lib.load().nrc_gs_camera_block(_lib.ptr(pose), _lib.ptr(_projection_transposed(cam, device)), _lib.ptr(bg), _lib.ptr(block), _lib.stream_of(block)),
                   'gs_camera_block')
        adopt_camera_block(block)
        return GaussianRasterizationSettings(
            image_height=cam.height, image_width=cam.width, tanfovx=cam.width / cam.focal_x * 0.5, tanfovy=cam.height / cam.focal_y * 0.5,
            bg=block[35:38], scale_modifier=scale_modifier, viewmatrix=block[:16].view(4, 4), projmatrix=block[16:32].view(4, 4), sh_degree=sh_degree,
            campos=block[32:35], prefiltered=False, debug=False)
    elif torch.is_tensor(c2w):
        pose = c2w.to(device=device, dtype=torch.float32)
        rot, pos = pose[:3, :3], pose[:3, 3]
        last_row = torch.cat([-(rot.T @ pos), pose.new_ones(1)])
        view = torch.cat([torch.cat([rot, pose.new_zeros(3, 1)], dim=1), last_row[None]], dim=0)
        campos = pos.contiguous()
    else:
        c2w = np.asarray(c2w, dtype=np.float64)
        view = torch.as_tensor(invert_3d_affine(c2w), dtype=torch.float32, device=device).T
        campos = torch.as_tensor(c2w[:3, 3], dtype=torch.float32, device=device)
    return GaussianRasterizationSettings(
        image_height=cam.height, image_width=cam.width, tanfovx=cam.width / cam.focal_x * 0.5, tanfovy=cam.height / cam.focal_y * 0.5,
        bg=cam.background_color.to(device), scale_modifier=scale_modifier, viewmatrix=view,
        projmatrix=view @ _projection_transposed(cam, device), sh_degree=sh_degree, campos=campos, prefiltered=False, debug=False)


def quaternion_to_rotation_matrix(quaternions: torch.Tensor, normalize: bool = True) -> torch.Tensor:
    """(..., 4) quaternions, real part first -> (..., 3, 3) rotation matrices (same convention and values as Cameras/utils.py:180-208):
    the nine entries of R = I + 2 (w [v]x + [v]x [v]x), v = vector part, stacked row by row."""
    q = torch.nn.functional.normalize(quaternions, dim=-1) if normalize else quaternions
    w, x, y, z = q.unbind(-1)
    rows = (1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y))
    return torch.stack(rows, dim=-1).reshape(*q.shape[:-1], 3, 3)


_UPPER_ROWS, _UPPER_COLS = (0, 0, 0, 1, 1, 2), (0, 1, 2, 1, 2, 2)


def build_covariances(scales: torch.Tensor, rotations: torch.Tensor) -> torch.Tensor:
    """Sigma = (R diag(s)) (R diag(s))^T for unit quaternions `rotations` (the rasterizer's cov3D step; GaussianSplatting/utils.py:10-18).
    Scaling the columns of R replaces the product with a diagonal matrix."""
    stretched = quaternion_to_rotation_matrix(rotations, normalize=False) * scales[..., None, :]
    return stretched @ stretched.mT


def extract_upper_triangular_matrix(matrix: torch.Tensor) -> torch.Tensor:
    """Row-major upper triangle of (..., n, n) matrices; (..., 6) for the 3x3 covariances the rasterizer takes as cov3D_precomp."""
    n = matrix.shape[-1]
    if n == 3:
        return matrix[..., _UPPER_ROWS, _UPPER_COLS]
    rows, cols = zip(*[(r, c) for r in range(n) for c in range(r, n)])
    return matrix[..., rows, cols]


# real spherical-harmonics basis up to degree 3 in the sign / order convention of 3DGS (GaussianSplatting/utils.py:27-58): constants first,
# then the polynomial each one multiplies
_SH_C = (0.28209479177387814,
         -0.48860251190291987, 0.48860251190291987, -0.48860251190291987,
         1.0925484305920792, -1.0925484305920792, 0.94617469575755997, -1.0925484305920792, 0.54627421529603959,
         0.59004358992664352, 2.8906114426405538, 0.45704579946446572, 0.3731763325901154, 0.45704579946446572, 1.4453057213202769, 0.59004358992664352)


def sh_basis(directions: torch.Tensor, degree: int) -> torch.Tensor:
    """(..., 3) unit directions -> (..., (degree+1)^2) basis values."""
    x, y, z = directions.unbind(-1)
    one = torch.ones_like(x)
    poly = [one]
    if degree >= 1:
        poly += [y, z, x]
    if degree >= 2:
        xx, yy, zz = x * x, y * y, z * z
        poly += [x * y, y * z, zz - 0.31539156525251999 / 0.94617469575755997, x * z, xx - yy]
    if degree >= 3:
        poly += [y * (yy - 3.0 * xx), x * y * z, y * (1.0 - 5.0 * zz), z * (5.0 * zz - 3.0), x * (1.0 - 5.0 * zz), z * (xx - yy), x * (3.0 * yy - xx)]
    basis = torch.stack(poly, dim=-1)
    return basis * torch.tensor(_SH_C[:len(poly)], dtype=basis.dtype, device=basis.device)


def convert_sh_features(sh_features: torch.Tensor, view_directions: torch.Tensor, degree: int) -> torch.Tensor:
    """SH coefficients (..., 3, 16) seen from view_directions (..., 1|3 broadcastable, 3) -> RGB = max(0, 0.5 + sum_k Y_k c_k): the rasterizer's
    colour step in torch (the reference's utils.py:21-59 is pinned against this in tests/test_host_golden.py)."""
    n = (degree + 1) ** 2
    basis = sh_basis(view_directions[..., 0, :] if view_directions.dim() == sh_features.dim() else view_directions, degree)
    return ((sh_features[..., :n] * basis[..., None, :]).sum(-1) + 0.5).clamp_min(0.0)


def rgb_to_sh0(rgb):
    return (rgb - 0.5) / _SH_C[0]


def sh0_to_rgb(sh):
    return sh * _SH_C[0] + 0.5


class Gaussians(torch.nn.Module):
    """Parameter store + activation accessors of GaussianSplatting/Model.py:18-87 (exp scales, normalised quaternions, sigmoid
    opacities, cat[dc, rest] SH features of shape (P, 16, 3)) and the training-time bookkeeping of Model.py:94-284: optimizer setup,
    densification statistics, densify_and_prune, opacity reset, activation baking, ply export.  The bookkeeping runs on the device through
    include/nerficg_hip.h group 10 (one classification + scan, one gather over all parameters and Adam moments)."""

    def __init__(self, positions, log_scales, rotations, opacity_logits, features_dc, features_rest, sh_degree: int = 3) -> None:
        super().__init__()
        P = torch.nn.Parameter
        self._positions, self._scales, self._rotations = P(positions), P(log_scales), P(rotations)
        self._opacities, self._features_dc, self._features_rest = P(opacity_logits), P(features_dc), P(features_rest)
        self.active_sh_degree = sh_degree
        self.max_sh_degree = sh_degree
        self.optimizer = None
        self.percent_dense = 0.0
        self.training_cameras_extent = 1.0
        self.baked = False
        self._baked_covariances = None
        n, dev = positions.shape[0], positions.device
        self.densification_gradient_accum = torch.zeros((n, 1), dtype=torch.float32, device=dev)
        self.n_observations = torch.zeros((n, 1), dtype=torch.int32, device=dev)

    @classmethod
    def from_point_cloud(cls, positions: torch.Tensor, colors: torch.Tensor | None = None, sh_degree: int = 3) -> 'Gaussians':
        """initialize_from_point_cloud (Model.py:94-119): isotropic scales from the RMS distance to the 3 nearest neighbours
        (Optim/knn_utils.py:29-40 through the HIP kNN), identity rotations, opacity 0.1, DC colour from rgb."""
        from .simple_knn import distCUDA2
        xyz = positions.to(torch.float32).contiguous()
        n, dev = xyz.shape[0], xyz.device
        n_coeff = (sh_degree + 1) ** 2
        base_colour = torch.full((n, 3), 0.5, device=dev) if colors is None else colors.to(dev, torch.float32)
        dc = rgb_to_sh0(base_colour).reshape(n, 1, 3).contiguous()                       # (P, 1, 3): coefficient-major like the rasterizer reads it
        rest = torch.zeros(n, n_coeff - 1, 3, device=dev)
        spacing = distCUDA2(xyz).clamp_min(1e-7).sqrt()                                  # RMS distance to the three nearest neighbours
        log_scales = spacing.log().reshape(n, 1).expand(n, 3).contiguous()
        identity = torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev).expand(n, 4).contiguous()
        opacity_logits = torch.full((n, 1), float(np.log(0.1 / 0.9)), device=dev)        # logit(0.1)
        out = cls(xyz, log_scales, identity, opacity_logits, dc, rest, sh_degree)
        out.active_sh_degree = 0
        return out

    @property
    def get_positions(self): return self._positions
    @property
    def get_scales(self): return self._scales if self.baked else torch.exp(self._scales)
    @property
    def get_rotations(self): return self._rotations if self.baked else torch.nn.functional.normalize(self._rotations)
    @property
    def get_opacities(self): return self._opacities if self.baked else torch.sigmoid(self._opacities)
    @property
    def get_features_dc(self): return self._features_dc
    @property
    def get_features_rest(self): return self._features_rest
    @property
    def get_features(self): return torch.cat((self._features_dc, self._features_rest), dim=1)
    @property
    def get_baked_covariances(self): return self._baked_covariances

    def get_covariances(self, scale_modifier: float) -> torch.Tensor:
        """Model.py:84-86"""
        return extract_upper_triangular_matrix(build_covariances(self.get_scales * scale_modifier, self.get_rotations))

    def increase_used_sh_degree(self) -> None:
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ---------------------------------------------------------------- optimizer (Model.py:121-150)
    _GROUP_OF = {'positions': '_positions', 'f_dc': '_features_dc', 'f_rest': '_features_rest', 'opacities': '_opacities', 'scales': '_scales',
                 'rotations': '_rotations'}

    def training_setup(self, LEARNING_RATE_POSITION_INIT: float = 0.00016, LEARNING_RATE_POSITION_FINAL: float = 0.0000016,
                       LEARNING_RATE_POSITION_MAX_STEPS: int = 30000, LEARNING_RATE_FEATURE: float = 0.0025, LEARNING_RATE_OPACITY: float = 0.05,
                       LEARNING_RATE_SCALING: float = 0.005, LEARNING_RATE_ROTATION: float = 0.001, PERCENT_DENSE: float = 0.01,
                       training_cameras_extent: float | None = None, optimizer_class=None, capturable: bool = False) -> None:
        """Six single-tensor groups in the reference's order and names; FusedAdam(lr=0, eps=1e-15, adam_w_mode=False) over the HIP step.
        capturable: step counters and learning rates on the device, for steps recorded in a HIP graph (nerficg_amd.graphs)."""
        from .lr_utils import LRDecayPolicy
        if training_cameras_extent is not None:
            self.training_cameras_extent = training_cameras_extent
        self.percent_dense = PERCENT_DENSE
        ext = self.training_cameras_extent
        rates = {'positions': LEARNING_RATE_POSITION_INIT * ext, 'f_dc': LEARNING_RATE_FEATURE, 'f_rest': LEARNING_RATE_FEATURE / 20.0,
                 'opacities': LEARNING_RATE_OPACITY, 'scales': LEARNING_RATE_SCALING, 'rotations': LEARNING_RATE_ROTATION}
        # one single-tensor group per attribute, in the order and under the names the reference's optimizer-state surgery looks up (Model.py:121-130)
        param_groups = [{'name': name, 'lr': rates[name], 'params': [getattr(self, attr)]} for name, attr in self._GROUP_OF.items()]
        if optimizer_class is None:
            from .apex_optimizers import FusedAdam
            self.optimizer = FusedAdam(param_groups, lr=0.0, eps=1e-15, adam_w_mode=False, capturable=capturable)
        else:
            self.optimizer = optimizer_class(param_groups, lr=0.0, eps=1e-15)
        self.position_lr_scheduler = LRDecayPolicy(lr_init=LEARNING_RATE_POSITION_INIT * ext, lr_final=LEARNING_RATE_POSITION_FINAL * ext,
                                                   max_steps=LEARNING_RATE_POSITION_MAX_STEPS)

    def update_learning_rate(self, iteration: int) -> None:
        rate = self.position_lr_scheduler(iteration)
        for group in self.optimizer.param_groups:
            if group['name'] == 'positions':
                group['lr'] = rate
        # the reference's trainer calls this first thing in iteration `iteration - 1` (Trainer.py:84): with a schedule installed the in-backward f_rest step is
        # switched per iteration, off where densify_and_prune will run between the backward pass and optimizer.step() (rest_step_schedule below)
        schedule = getattr(self, 'fuse_rest_schedule', None)
        if schedule is not None:
            self.fuse_rest_step = bool(schedule(iteration - 1))

    def _adopt(self, tensors: dict[str, torch.Tensor]) -> None:
        for name, attr in self._GROUP_OF.items():
            setattr(self, attr, tensors[name])

    # ---------------------------------------------------------------- densification (Model.py:152-246)
    def reset_opacities(self) -> None:
        from .adam_utils import replace_param_group_data
        capped = torch.special.logit(self.get_opacities.clamp(max=0.01))  # Model.py:152-155
        replace_param_group_data(self.optimizer, capped, 'opacities')

    def prune_points(self, prune_mask: torch.Tensor) -> None:
        from .adam_utils import compact_mask, gather_param_groups, gather_rows
        idx = compact_mask(~prune_mask)
        self._adopt(gather_param_groups(self.optimizer, idx, idx.numel(), None, None, 'prune_param_groups'))
        self.densification_gradient_accum, n_obs = gather_rows([self.densification_gradient_accum, self.n_observations.view(torch.float32)],
                                                               idx, idx.numel())
        self.n_observations = n_obs.view(torch.int32)

    @torch.no_grad()
    def add_densification_stats(self, viewspace_point_tensor: torch.Tensor, visibility: torch.Tensor) -> None:
        """Model.py:243-246.  `visibility`: the rasterizer's int32 radii (used as they are) or the boolean mask radii > 0."""
        from . import _lib
        grad = viewspace_point_tensor.grad if viewspace_point_tensor.grad is not None else viewspace_point_tensor
        radii = visibility if visibility.dtype == torch.int32 else visibility.to(torch.int32)
        _lib.check_input(grad, 'viewspace gradient', torch.float32)
        _lib.check_input(radii, 'radii', torch.int32)
        lib = _lib.load()
        _lib.check(lib.nrc_gs_densify_stats(_lib.ptr(grad), grad.shape[1], _lib.ptr(radii), grad.shape[0], _lib.ptr(self.densification_gradient_accum),
                                            _lib.ptr(self.n_observations), _lib.stream_of(grad)), 'gs_densify_stats')

    @torch.no_grad()
    def densify_and_prune(self, grad_threshold: float, min_opacity: float, prune_large_gaussians: bool, noise: torch.Tensor | None = None) -> dict[str, int]:
        """Model.py:226-241 (duplicate, split, prune) as one device plan + one gather.  `noise`: optional (>= 2 * n_split, 3) standard
        normal draws; by default torch.randn((2 * n_split, 3)) -- the draws torch.normal(mean=0, std=stds) consumes in the reference
        (normal_(0, 1) into the output, then * std)."""
        from . import _lib
        from .adam_utils import gather_param_groups
        lib = _lib.load()
        P, dev = self._positions.shape[0], self._positions.device
        src = torch.empty(2 * max(P, 1), dtype=torch.int32, device=dev)
        kind, aux = torch.empty_like(src), torch.empty_like(src)
        counts = torch.zeros(5, dtype=torch.int32, device=dev)
        ws = torch.empty(int(lib.nrc_gs_densify_plan_ws_bytes(P)), dtype=torch.uint8, device=dev)
        scales, opac = self._scales.data.contiguous(), self._opacities.data.contiguous()
        max_scale = float(np.float32(0.1 * self.training_cameras_extent)) if prune_large_gaussians else 0.0
        _lib.check(lib.nrc_gs_densify_plan(_lib.ptr(self.densification_gradient_accum), _lib.ptr(self.n_observations), _lib.ptr(scales), _lib.ptr(opac), P,
                                           grad_threshold, self.percent_dense * self.training_cameras_extent, min_opacity, max_scale, _lib.ptr(src),
                                           _lib.ptr(kind), _lib.ptr(aux), _lib.ptr(counts), _lib.ptr(ws), _lib.stream_of(src)), 'gs_densify_plan')
        n_out, n_keep, n_dup, n_child, n_split = counts.tolist()  # the one host read: sizes of the new tensors
        if noise is None:
            noise = torch.randn((2 * n_split, 3), dtype=torch.float32, device=dev)
        elif noise.shape[0] < 2 * n_split:
            raise RuntimeError(f'densify_and_prune: noise has {noise.shape[0]} rows, {2 * n_split} needed')
        old_pos, old_rot = self._positions.data, self._rotations.data.contiguous()
        new = gather_param_groups(self.optimizer, src, n_out, kind, None, 'extend_param_groups')
        if n_child > 0:
            noise = noise.to(torch.float32).contiguous()
            _lib.check(lib.nrc_gs_densify_split_children(_lib.ptr(src), _lib.ptr(kind), _lib.ptr(aux), n_out, _lib.ptr(old_pos), _lib.ptr(scales),
                                                         _lib.ptr(old_rot), _lib.ptr(noise), _lib.ptr(new['positions'].data), _lib.ptr(new['scales'].data),
                                                         _lib.stream_of(src)), 'gs_densify_split_children')
        self._adopt(new)
        self.densification_gradient_accum = torch.zeros((n_out, 1), dtype=torch.float32, device=dev)
        self.n_observations = torch.zeros((n_out, 1), dtype=torch.int32, device=dev)
        return {'n_out': n_out, 'n_kept': n_keep, 'n_cloned': n_dup, 'n_split': n_split, 'n_children_kept': 2 * n_child}

    # ---------------------------------------------------------------- export (Model.py:248-318)
    @torch.no_grad()
    def bake_activations(self) -> None:
        """Model.py:248-273: activations folded into the parameters, never-visible Gaussians pruned, Morton order, covariances baked."""
        from .adam_utils import compact_mask, gather_rows
        from .MortonEncoding import morton_encode
        rot, opa, sca = self.get_rotations, self.get_opacities, self.get_scales
        keep = compact_mask(~(opa.flatten() < 0.00392156862))
        names = ('_positions', '_rotations', '_features_dc', '_features_rest', '_scales', '_opacities')
        vals = gather_rows([self._positions.data, rot.contiguous(), self._features_dc.data, self._features_rest.data, sca.contiguous(), opa.contiguous()],
                           keep, keep.numel())
        order = torch.argsort(morton_encode(vals[0])).to(torch.int32)
        vals = gather_rows(vals, order, order.numel())
        for name, v in zip(names, vals):
            setattr(self, name, torch.nn.Parameter(v, requires_grad=getattr(self, name).requires_grad))
        self.baked = True
        self._baked_covariances = torch.nn.Parameter(self.get_covariances(1.0), requires_grad=False)

    @torch.no_grad()
    def as_ply_dict(self) -> dict[str, np.ndarray]:
        """The 'vertex' element of the standard 3DGS .ply (Model.py:275-318): every property f4 -- x y z, f_dc_0..2, f_rest_0.. (channel-major),
        opacity (logit), scale_0..2 (log), rot_0..3 (unit quaternion)."""
        n = self._positions.shape[0]
        if n == 0:
            return {}
        channel_major = lambda t: t.detach().permute(0, 2, 1).reshape(n, -1)
        columns = {'x y z': self._positions.detach(), 'f_dc': channel_major(self._features_dc), 'f_rest': channel_major(self._features_rest),
                   'opacity': torch.special.logit(self.get_opacities).detach(), 'scale': torch.log(self.get_scales).detach(), 'rot': self.get_rotations.detach()}
        names: list[str] = []
        for key, block in columns.items():
            width = block.shape[1]
            names += key.split() if ' ' in key else ([key] if (width == 1 and key == 'opacity') else [f'{key}_{i}' for i in range(width)])
        table = torch.cat([block.reshape(n, -1).float() for block in columns.values()], dim=1).cpu().numpy()
        vertex = np.empty(n, dtype=[(name, 'f4') for name in names])
        for column, name in enumerate(names):
            vertex[name] = table[:, column]
        return {'vertex': vertex}


def rest_step_schedule(densify_start_iteration: int, densify_end_iteration: int, densification_interval: int):
    """iteration -> may the f_rest optimizer step of that iteration run inside the backward pass (Gaussians.fuse_rest_step) without leaving the reference's trajectory?

    The reference's trainer runs `densify` (priority 90) between `loss.backward()` (priority 100) and `optimizer.step()` (priority 70), Trainer.py:76-128.  A callback
    fires when start <= iteration <= end and (iteration - start) % stride == 0 (Base/Trainer.py:238-243); `densify` returns at once in its first and last iteration
    (Trainer.py:104-105) and otherwise calls densify_and_prune, whose prune_points replaces ALL six parameter tensors by fresh nn.Parameters without .grad
    (Model.py:157-167, 242-252): the optimizer then finds no gradient and that iteration's update is dropped.  A step applied inside the backward pass has already
    happened by then, so those iterations keep the plain order.  `reset_opacities` (priority 80) replaces the opacity tensor only: the f_rest step is not affected.
    Install with `gaussians.fuse_rest_schedule = rest_step_schedule(DENSIFY_START_ITERATION, DENSIFY_END_ITERATION, DENSIFICATION_INTERVAL)`; the trainer's own
    `update_learning_rate(iteration + 1)` call then sets `fuse_rest_step` for the iteration."""
    a, b, k = int(densify_start_iteration), int(densify_end_iteration), int(densification_interval)
    if k <= 0:
        raise ValueError('densification_interval must be positive')

    def clean(iteration: int) -> bool:
        return not (a < iteration < b and (iteration - a) % k == 0)
    return clean


class RestStep:
    """The optimizer step of the model's `f_rest` group, handed to the rasterizer's backward pass (which applies it where the gradient rows are: C ABI
    nrc_gs_backward_rest_step).  take() is called once per backward: it advances the group's step counter like FusedAdam.step would (apex: one counter per
    group, advanced whenever the group has gradients) and returns the tensors and host scalars of this step.  The optimizer's own step() then finds no
    gradient on the tensor and leaves it alone.  Plain (non-capturable) FusedAdam only; eps / betas / learning rate as the group has them."""

    def __init__(self, gaussians: 'Gaussians') -> None:
        opt = gaussians.optimizer
        self.opt = opt
        self.group = next(g for g in opt.param_groups if g.get('name') == 'f_rest')
        self.param = self.group['params'][0]
        from .apex_optimizers import FusedAdam
        if not isinstance(opt, FusedAdam) or opt.capturable or self.group.get('weight_decay', 0.0) != 0.0 or opt.adam_w_mode:
            raise RuntimeError('RestStep: a plain (non-capturable) nerficg_amd FusedAdam in L2 mode without weight decay is expected (Model.py:131-136)')

    def take(self):
        g, p = self.group, self.param
        state = self.opt.state[p]
        if len(state) == 0:
            state['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        g['step'] = g.get('step', 0) + 1
        beta1, beta2 = g['betas']
        bc = (1.0 - beta1 ** g['step'], 1.0 - beta2 ** g['step']) if g.get('bias_correction', True) else (1.0, 1.0)
        return p, state['exp_avg'], state['exp_avg_sq'], float(g['lr']), float(beta1), float(beta2), float(g['eps']), bc[0], bc[1]


def render_image_training(gaussians: Gaussians, cam: PerspectiveCamera, c2w: np.ndarray, fuse_rest_step: bool | None = None) -> dict[str, torch.Tensor]:
    """GaussianSplatting/Renderer.py:51-86.  fuse_rest_step (default: gaussians.fuse_rest_step, False unless set): the backward pass of this frame applies the
    optimizer's step to the `f_rest` SH tensor itself -- for loops that run exactly one backward pass and one optimizer.step() per frame on one GPU (the
    view-parallel exchange needs the gradient on the wire, gradient accumulation needs it in .grad: both leave this off)."""
    from .diff_gaussian_rasterization import GaussianRasterizer
    positions = gaussians.get_positions
    # the carrier of the screen-space gradient (Renderer.py:56-58: zeros_like + 0, retain_grad): the rasterizer never reads its VALUES, only hands it a
    # gradient -- a leaf of uninitialised memory receives the same .grad without a 12 MB fill and a 24 MB add per step
    viewspace_points = torch.empty_like(positions).requires_grad_(True)
    rasterizer = GaussianRasterizer(make_raster_settings(cam, c2w, gaussians.active_sh_degree, 1.0, positions.device))
    if gaussians.baked:  # a baked model holds activated values
        image, radii = rasterizer(means3D=positions, means2D=viewspace_points, shs=gaussians.get_features_dc, shs_rest=gaussians.get_features_rest,
                                  opacities=gaussians.get_opacities, scales=gaussians.get_scales, rotations=gaussians.get_rotations)
    else:  # raw parameters straight into the kernels: no get_features concatenation, no separate exp / sigmoid / normalize passes (a25)
        fuse = getattr(gaussians, 'fuse_rest_step', False) if fuse_rest_step is None else bool(fuse_rest_step)
        rest_step = RestStep(gaussians) if fuse and not getattr(gaussians.optimizer, 'capturable', False) and not torch.cuda.is_current_stream_capturing() and torch.is_grad_enabled() and gaussians._features_rest.requires_grad and gaussians._features_rest.shape[1] > 0 and torch.is_tensor(c2w) else None
        image, radii = rasterizer(means3D=positions, means2D=viewspace_points, shs=gaussians._features_dc, shs_rest=gaussians._features_rest,
                                  opacities=gaussians._opacities, scales=gaussians._scales, rotations=gaussians._rotations, raw_parameters=True, rest_step=rest_step)
    return _TrainingOutputs({'rgb': image, 'viewspace_points': viewspace_points, 'radii': radii})


class _TrainingOutputs(dict):
    """The outputs of Renderer.py:83-86; 'visibility_mask' (radii > 0) is built when somebody asks for it -- the densification statistics and the view-parallel
    exchange take `radii` itself, so the plain step does not pay the launch."""

    def __missing__(self, key):
        if key == 'visibility_mask':
            self[key] = self['radii'] > 0
            return self[key]
        raise KeyError(key)

    def __contains__(self, key):
        return key == 'visibility_mask' or super().__contains__(key)


@torch.no_grad()
def render_image_inference(gaussians: Gaussians, cam: PerspectiveCamera, c2w: np.ndarray, scale_modifier: float = 1.0, to_chw: bool = False,
                           use_baked_covariance: bool = True):
    """GaussianSplatting/Renderer.py:89-155 with the fused SH path (USE_FUSED_SH_CONVERSION = True, the shipped default); a baked model
    (trained checkpoint, Model.py:248-273) is rasterized from its baked covariances like the reference's USE_BAKED_COVARIANCE branch
    (Renderer.py:129-139), everything else through the in-kernel covariance computation."""
    from .diff_gaussian_rasterization import GaussianRasterizer
    positions = gaussians.get_positions
    rasterizer = GaussianRasterizer(make_raster_settings(cam, c2w, gaussians.active_sh_degree, scale_modifier, positions.device))
    covariances = gaussians.get_baked_covariances if (use_baked_covariance and gaussians.baked) else None
    if covariances is not None and covariances.shape[0] != positions.shape[0]:
        covariances = None  # "Baked covariance requested but not available"
    if covariances is not None:
        image, _ = rasterizer(means3D=positions, means2D=torch.empty_like(positions), shs=gaussians.get_features, opacities=gaussians.get_opacities,
                              cov3D_precomp=covariances)
    else:
        image, _ = rasterizer(means3D=positions, means2D=torch.empty_like(positions), shs=gaussians.get_features,
                              opacities=gaussians.get_opacities, scales=gaussians.get_scales, rotations=gaussians.get_rotations)
    image.clamp_(0.0, 1.0)
    return {'rgb': image if to_chw else image.permute(1, 2, 0)}


FUSED_PHOTOMETRIC_LOSS = True   # training_loss as one node (nerficg_amd.fused_ssim.photometric_loss); False: l1_loss + fused_ssim as tensor operations


def training_loss(image: torch.Tensor, target: torch.Tensor, lambda_l1: float = 0.8, lambda_dssim: float = 0.2) -> torch.Tensor:
    """GaussianSplattingLoss (src/Methods/GaussianSplatting/Loss.py:11-23, weights Trainer.py:34-35): lambda_l1 * L1 + lambda_dssim *
    (1 - SSIM) on (3, H, W) images, SSIM through the HIP kernels (nerficg_amd.fused_ssim; DSSIM.py:11-18 adds the batch dimension)."""
    from .fused_ssim import fused_ssim, photometric_loss
    a = image[None] if image.dim() == 3 else image
    b = target[None] if target.dim() == 3 else target
    if FUSED_PHOTOMETRIC_LOSS and a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 4 and a.shape == b.shape:
        # one autograd node, three launches (stencil + reduction, stencil) instead of ~22 tensor operations around the SSIM kernels
        return photometric_loss(a.contiguous(), b.contiguous(), lambda_l1, lambda_dssim)
    l1 = torch.nn.functional.l1_loss(image, target)
    return lambda_l1 * l1 + lambda_dssim * (1.0 - fused_ssim(a, b))
