"""Synthetic, seeded inputs shared by the oracle tests, the GPU parity tests, smoke() and bench.py."""
from __future__ import annotations

import math

import numpy as np


def expand_bits(v):
    v = np.asarray(v, dtype=np.uint64)
    v = (v * 0x00010001) & 0xFF0000FF
    v = (v * 0x00000101) & 0x0F00F00F
    v = (v * 0x00000011) & 0xC30C30C3
    v = (v * 0x00000005) & 0x49249249
    return v


def morton3d_np(x, y, z):
    return (expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2)).astype(np.int64)


def sphere_bitfield(grid_size: int = 128, scale: float = 0.5, radius: float = 0.35, cascades: int = 1, shell: float = 0.0):
    """Occupancy bitfield: cells of cascade c whose centre lies inside the sphere (optionally only a shell), Morton order."""
    g = np.arange(grid_size)
    x, y, z = np.meshgrid(g, g, g, indexing='ij')
    bits = np.zeros(cascades * grid_size ** 3, dtype=bool)
    for c in range(cascades):
        bound = min(2.0 ** (c - 1), scale)
        cx = ((x + 0.5) / grid_size * 2 - 1) * bound
        cy = ((y + 0.5) / grid_size * 2 - 1) * bound
        cz = ((z + 0.5) / grid_size * 2 - 1) * bound
        rr = np.sqrt(cx * cx + cy * cy + cz * cz)
        occ = (rr < radius) & (rr >= shell)
        idx = morton3d_np(x, y, z)
        bits[c * grid_size ** 3 + idx.reshape(-1)] = occ.reshape(-1)
    return np.packbits(bits, bitorder='little')


def orbit_pose(theta: float, phi: float, radius: float) -> np.ndarray:
    """c2w (4x4 f64), camera on an orbit looking at the origin; x right, y down, z forward."""
    pos = np.array([radius * math.cos(phi) * math.cos(theta), radius * math.sin(phi), radius * math.cos(phi) * math.sin(theta)])
    fwd = -pos / np.linalg.norm(pos)
    up = np.array([0.0, -1.0, 0.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, pos
    return c2w


LEGO_FOV_X = 0.6911112070083618  # camera_angle_x of the NeRF-synthetic lego scene
LEGO_RADIUS = 4.031128874 / 3.0  # camera orbit radius after the loader's cube normalisation to SCALE 0.5 (bbox +-1.5)


def lego_intrinsics(width: int, height: int):
    focal = 0.5 / math.tan(0.5 * LEGO_FOV_X) * width
    return focal, focal, width / 2, height / 2


def numpy_rays(width, height, c2w, fx=None, fy=None, cx=None, cy=None):
    """float64-free numpy restatement of ray generation used only to FEED tests (not a parity oracle)."""
    if fx is None:
        fx, fy, cx, cy = lego_intrinsics(width, height)
    xs = np.linspace((0.5 - cx) / fx, (width - 0.5 - cx) / fx, width, dtype=np.float32)
    ys = np.linspace((0.5 - cy) / fy, (height - 0.5 - cy) / fy, height, dtype=np.float32)
    local = np.stack([np.broadcast_to(xs[None, :], (height, width)), np.broadcast_to(ys[:, None], (height, width)),
                      np.ones((height, width), np.float32)], -1).reshape(-1, 3)
    R = c2w[:3, :3].astype(np.float32)
    d = local @ R.T
    o = np.broadcast_to(c2w[:3, 3].astype(np.float32), d.shape).copy()
    vd = d / np.maximum(np.linalg.norm(d, axis=-1, keepdims=True), 1e-12)
    return o, d.astype(np.float32), vd.astype(np.float32)


def random_ragged_rays(rng, n_rays, max_len, empty_frac=0.15):
    """rays_a (n,3) i64 in ray order with ragged segment lengths (some empty), total sample count."""
    lens = rng.integers(1, max_len + 1, size=n_rays)
    lens[rng.random(n_rays) < empty_frac] = 0
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    rays_a = np.stack([np.arange(n_rays), starts, lens], -1).astype(np.int64)
    return rays_a, int(lens.sum())
