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


def layered_bitfield(scale: float, cascades: int, grid_size: int = 128) -> np.ndarray:
    """Occupancy of a scene with content at three distances: a solid ball r < 0.3 (lives in cascade 0), a shell 0.62 <= r < 0.9 (beyond the
    first cascade's box) and a shell 1.45 <= r < 1.8 with a polar cap removed (beyond the second); every cascade marks the cells whose CENTRE
    lies in that set, at its own cell size -- all cascades are populated, and rays cross empty stretches in each of them."""
    g = np.arange(grid_size)
    x, y, z = np.meshgrid(g, g, g, indexing='ij')
    idx = morton3d_np(x, y, z).reshape(-1)
    bits = np.zeros(cascades * grid_size ** 3, dtype=bool)
    for c in range(cascades):
        bound = min(2.0 ** (c - 1), scale)
        cx, cy, cz = (((a + 0.5) / grid_size * 2 - 1) * bound for a in (x, y, z))
        r = np.sqrt(cx * cx + cy * cy + cz * cz)
        occ = (r < 0.3) | ((r >= 0.62) & (r < 0.9) & (cx > -0.5)) | ((r >= 1.45) & (r < 1.8) & (cy < 1.2))
        bits[c * grid_size ** 3 + idx] = occ.reshape(-1)
        assert occ.any(), c
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


# ------------------------------------------------------------------------------------------------ 3DGS synthetic scenes
def gs_camera(width, height, c2w, fx=None, fy=None, near=0.01, far=100.0):
    """Settings marshalling of GaussianSplatting/Renderer.py:60-74 in numpy: returns viewmatrix (= w2c.T), projmatrix
    (= w2c.T @ P.T), tanfovx, tanfovy, campos.  P = PerspectiveCamera.get_projection_matrix (Cameras/Perspective.py:96-119)."""
    fx = 1.2 * width if fx is None else fx
    fy = fx if fy is None else fy
    cx, cy = width / 2, height / 2
    R, t = c2w[:3, :3], c2w[:3, 3]
    w2c = np.eye(4)
    w2c[:3, :3] = R.T
    w2c[:3, 3] = R.T @ -t
    hw, hh = width * 0.5, height * 0.5
    P = np.array([[fx / hw, 0.0, (cx - hw) / hw, 0.0], [0.0, fy / hh, (cy - hh) / hh, 0.0],
                  [0.0, 0.0, (far + near) / (far - near), -2.0 * far * near / (far - near)], [0.0, 0.0, 1.0, 0.0]], dtype=np.float32)
    vm = w2c.astype(np.float32).T
    pm = vm @ P.T
    return dict(viewmatrix=np.ascontiguousarray(vm), projmatrix=np.ascontiguousarray(pm.astype(np.float32)), tanfovx=width / fx * 0.5,
                tanfovy=height / fy * 0.5, campos=t.astype(np.float32), width=width, height=height)


def gs_random_scene(n, seed=0, extent=1.5, log_scale_mean=math.log(0.01), log_scale_std=0.5, sh_degree=3):
    """SURVEY 8(d) C3 synthetic Gaussians: positions U([-extent,extent]^3) + a ground-plane cluster, log-scales N(log 0.01, 0.5^2),
    unit quaternions, opacity logits N(0, 2^2), SH dc U(-1,1), rest N(0, 0.1^2)."""
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-extent, extent, size=(n, 3)).astype(np.float32)
    k = n // 4
    pos[:k, 1] = (extent * 0.6 + rng.normal(size=k) * 0.02).astype(np.float32)  # ground plane (y down)
    scales = np.exp(rng.normal(size=(n, 3)) * log_scale_std + log_scale_mean).astype(np.float32)
    q = rng.normal(size=(n, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    opac = (1.0 / (1.0 + np.exp(-rng.normal(size=n) * 2.0))).astype(np.float32)
    shs = np.zeros((n, 16, 3), np.float32)
    shs[:, 0] = rng.uniform(-1, 1, size=(n, 3))
    shs[:, 1:] = rng.normal(size=(n, 15, 3)) * 0.1
    return dict(means3D=pos, scales=scales, rotations=q, opacities=opac, shs=shs, sh_degree=sh_degree)
