"""nerficg_amd.raygen -- device ray generation for undistorted perspective cameras.

Mirrors PerspectiveCamera.compute_local_ray_directions (src/Cameras/Perspective.py:64-94) + View.get_rays /
cam_to_world (src/Datasets/utils.py:1033-1074): pixel centres at +0.5, inclusive torch.linspace end points, rows of
(H*W,3) in y-major order, direction = local @ R^T, view_direction = normalize(direction).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib

__all__ = ['generate_rays']


def generate_rays(width: int, height: int, focal_x: float, focal_y: float, center_x: float, center_y: float,
                  c2w: np.ndarray, device: torch.device | str = 'cuda', want_direction: bool = True,
                  want_view_direction: bool = True) -> dict[str, torch.Tensor]:
    c2w = np.ascontiguousarray(np.asarray(c2w, dtype=np.float64))
    if c2w.shape == (3, 4):
        c2w = np.vstack([c2w, np.array([0.0, 0.0, 0.0, 1.0])])
    if c2w.shape != (4, 4):
        raise RuntimeError(f'c2w must have shape (4, 4) or (3, 4), got {c2w.shape}')
    dev = torch.device(device)
    n = width * height
    origin = torch.empty(n, 3, dtype=torch.float32, device=dev)
    direction = torch.empty(n, 3, dtype=torch.float32, device=dev) if want_direction else None
    view_dir = torch.empty(n, 3, dtype=torch.float32, device=dev) if want_view_direction else None
    intr = (ctypes.c_double * 4)(focal_x, focal_y, center_x, center_y)
    mat = (ctypes.c_double * 16)(*c2w.reshape(-1).tolist())
    _lib.check(_lib.load().nrc_generate_rays(int(width), int(height), ctypes.cast(intr, ctypes.c_void_p),
                                             ctypes.cast(mat, ctypes.c_void_p), _lib.ptr(origin), _lib.ptr(direction),
                                             _lib.ptr(view_dir), _lib.stream_of(origin)), 'generate_rays')
    return {'origin': origin, 'direction': direction, 'view_direction': view_dir}
