"""nerficg_amd.VolumeRenderingV2 -- drop-in for the reference's CUDA extension module `VolumeRenderingV2`
(src/Methods/InstantNGP/VolumeRenderingV2/csrc/binding.cpp:234-250) and its autograd layer
(src/Methods/InstantNGP/VolumeRenderingV2/custom_functions.py), backed by libnerficg_hip.so.

Same function names, argument order, dtypes, in-place behaviour and error behaviour (RuntimeError on non-device /
non-contiguous input, csrc/include/utils.h:4-6).  Outputs are allocated here ("callee allocates" stays true at the
Python level); the C ABI underneath takes raw pointers.  Differences, all documented in DESIGN.md:
  * raymarching_train returns sample arrays with exactly counter[0] rows (the reference returns n_rays*max_samples
    zero-filled rows and its only caller slices them, custom_functions.py:114-119); rays_a is in ray order.
  * launches go to torch's CURRENT stream (the reference uses the legacy default stream).
"""
from __future__ import annotations

import os

import torch
from torch.amp import custom_bwd, custom_fwd

from .. import _lib

__all__ = [
    'ray_aabb_intersect', 'ray_sphere_intersect', 'morton3D', 'morton3D_invert', 'packbits', 'raymarching_train',
    'raymarching_test', 'composite_train_fw', 'composite_train_bw', 'composite_test_fw', 'distortion_loss_fw',
    'distortion_loss_bw', 'RayAABBIntersector', 'RaySphereIntersector', 'RayMarcher', 'VolumeRenderer', 'TruncExp',
    'DistortionLoss',
]

_f32, _i32, _i64, _u8 = torch.float32, torch.int32, torch.int64, torch.uint8
_PARKED_MARCH = os.environ.get('NRC_TRAIN_PARK', '1') != '0'  # 0: raymarching_train marches twice instead of expanding parked positions


def _chk(*pairs) -> None:
    for t, name, dtype in pairs:
        _lib.check_input(t, name, dtype)


# ----------------------------------------------------------------------------------------------- raw ops (binding.cpp)
def ray_aabb_intersect(rays_o, rays_d, centers, half_sizes, max_hits: int):
    """binding.cpp:4-16. Returns [hit_cnt (N) i32, hits_t (N,max_hits,2) f32, hits_voxel_idx (N,max_hits) i64]."""
    _chk((rays_o, 'rays_o', _f32), (rays_d, 'rays_d', _f32), (centers, 'centers', _f32), (half_sizes, 'half_sizes', _f32))
    n, v = rays_o.shape[0], centers.shape[0]
    dev = rays_o.device
    hit_cnt = torch.empty(n, dtype=_i32, device=dev)
    hits_t = torch.empty(n, max_hits, 2, dtype=_f32, device=dev)
    hits_idx = torch.empty(n, max_hits, dtype=_i64, device=dev)
    _lib.check(_lib.load().nrc_ray_aabb_intersect(
        _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(centers), _lib.ptr(half_sizes), n, v, int(max_hits),
        _lib.ptr(hit_cnt), _lib.ptr(hits_t), _lib.ptr(hits_idx), _lib.stream_of(rays_o)), 'ray_aabb_intersect')
    return [hit_cnt, hits_t, hits_idx]


def ray_sphere_intersect(rays_o, rays_d, centers, radii, max_hits: int):
    """binding.cpp:19-31."""
    _chk((rays_o, 'rays_o', _f32), (rays_d, 'rays_d', _f32), (centers, 'centers', _f32), (radii, 'radii', _f32))
    n, v = rays_o.shape[0], centers.shape[0]
    dev = rays_o.device
    hit_cnt = torch.empty(n, dtype=_i32, device=dev)
    hits_t = torch.empty(n, max_hits, 2, dtype=_f32, device=dev)
    hits_idx = torch.empty(n, max_hits, dtype=_i64, device=dev)
    _lib.check(_lib.load().nrc_ray_sphere_intersect(
        _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(centers), _lib.ptr(radii), n, v, int(max_hits),
        _lib.ptr(hit_cnt), _lib.ptr(hits_t), _lib.ptr(hits_idx), _lib.stream_of(rays_o)), 'ray_sphere_intersect')
    return [hit_cnt, hits_t, hits_idx]


def morton3D(coords):
    """binding.cpp:46-50. coords (N,3) i32 -> (N) i32."""
    _chk((coords, 'coords', _i32))
    out = torch.empty(coords.shape[0], dtype=_i32, device=coords.device)
    _lib.check(_lib.load().nrc_morton3D(_lib.ptr(coords), coords.shape[0], _lib.ptr(out), _lib.stream_of(coords)), 'morton3D')
    return out


def morton3D_invert(indices):
    """binding.cpp:53-57. indices (N) i32 -> (N,3) i32."""
    _chk((indices, 'indices', _i32))
    out = torch.empty(indices.shape[0], 3, dtype=_i32, device=indices.device)
    _lib.check(_lib.load().nrc_morton3D_invert(_lib.ptr(indices), indices.shape[0], _lib.ptr(out), _lib.stream_of(indices)),
               'morton3D_invert')
    return out


def packbits(density_grid, density_threshold: float, density_bitfield) -> None:
    """binding.cpp:34-43. In place on density_bitfield (u8, numel = grid.numel()/8)."""
    _chk((density_grid, 'density_grid', None), (density_bitfield, 'density_bitfield', _u8))
    if density_grid.dtype == _f32:
        dt = 0
    elif density_grid.dtype == torch.float16:
        dt = 1
    else:
        raise RuntimeError(f'packbits: unsupported density_grid dtype {density_grid.dtype}')
    _lib.check(_lib.load().nrc_packbits(_lib.ptr(density_grid), dt, density_bitfield.shape[0], float(density_threshold),
                                        _lib.ptr(density_bitfield), _lib.stream_of(density_grid)), 'packbits')


COUNT_MAILBOX = True   # raymarching_train: the sample count reaches the host through a host mailbox (False: device-to-host copy of counter[0])


def raymarching_train(rays_o, rays_d, hits_t, density_bitfield, cascades: int, scale: float, exp_step_factor: float, noise,
                      grid_size: int, max_samples: int, sample_capacity: int | None = None, return_overflow: bool = False):
    """binding.cpp:60-81. Returns [rays_a (N,3) i64, xyzs (M,3), dirs (M,3), deltas (M), ts (M), counter (2) i32], M = counter[0].
    With `sample_capacity` (not in the reference; for graph capture) M = sample_capacity and nothing is read back: rows past counter[0] are
    inert samples no ray refers to, and counter[0] > sample_capacity tells that rays were cut short (nrc_raymarching_train_cap); with
    `return_overflow` a seventh entry, the device int64 max(counter[0] - sample_capacity, 0), is appended to the list."""
    _chk((rays_o, 'rays_o', _f32), (rays_d, 'rays_d', _f32), (hits_t, 'hits_t', _f32),
         (density_bitfield, 'density_bitfield', _u8), (noise, 'noise', _f32))
    lib = _lib.load()
    n = rays_o.shape[0]
    dev = rays_o.device
    st = _lib.stream_of(rays_o)
    rays_a = torch.empty(n, 3, dtype=_i64, device=dev)
    counter = torch.empty(2, dtype=_i32, device=dev)
    ws = torch.empty(max(int(lib.nrc_raymarching_train_ws_bytes(n, int(max_samples))), 1), dtype=_u8, device=dev)
    args = (_lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(hits_t), _lib.ptr(density_bitfield), int(cascades), float(scale),
            float(exp_step_factor), _lib.ptr(noise), int(grid_size), int(max_samples), n)
    if sample_capacity is not None and _PARKED_MARCH and 0 < n <= 32768 and int(sample_capacity) > 0:
        # small batch, fixed capacity (a recorded training iteration): count, cut and write as three launches
        total = int(sample_capacity)
        xyzs = torch.empty(total, 3, dtype=_f32, device=dev)
        dirs = torch.empty(total, 3, dtype=_f32, device=dev)
        deltas = torch.empty(total, dtype=_f32, device=dev)
        ts = torch.empty(total, dtype=_f32, device=dev)
        overflow = torch.empty((), dtype=_i64, device=dev)
        _lib.check(lib.nrc_raymarching_train_capped(*args, total, _lib.ptr(rays_a), _lib.ptr(counter), _lib.ptr(xyzs), _lib.ptr(dirs), _lib.ptr(deltas),
                                                    _lib.ptr(ts), _lib.ptr(overflow), _lib.ptr(ws), st), 'raymarching_train(capped)')
        return [rays_a, xyzs, dirs, deltas, ts, counter] + ([overflow] if return_overflow else [])
    if sample_capacity is None and torch.cuda.is_current_stream_capturing():
        raise RuntimeError('raymarching_train: sizing the sample buffers reads counter[0] on the host, which a stream capture cannot do -- '
                           'pass sample_capacity (InstantNGPRenderer.sample_capacity / nerficg_amd.graphs.instant_ngp_iteration)')
    total = None
    mailbox = _lib.HostMailbox.for_device(dev) if (sample_capacity is None and _PARKED_MARCH and COUNT_MAILBOX and 0 < n <= 32768) else None
    if mailbox is not None:
        # the count reaches the host through mapped host memory that the scan kernel writes and this thread polls: no device-to-host copy, no
        # stream synchronisation (35 us of idle GPU per training iteration through the copy; the same mechanism as the image pipeline's row count)
        with mailbox.lock:
            ticket = mailbox.next_ticket()
            _lib.check(lib.nrc_raymarching_train_count_posted(*args, _lib.ptr(rays_a), _lib.ptr(counter), _lib.ptr(ws), mailbox.ptr, ticket, st),
                       'raymarching_train(count, posted)')
            got = mailbox.counts(ticket, dev)
        if got is not None:
            total = int(got[0])
    else:
        _lib.check(lib.nrc_raymarching_train_count(*args, _lib.ptr(rays_a), _lib.ptr(counter), _lib.ptr(ws), st), 'raymarching_train(count)')
    if sample_capacity is None:
        if total is None:
            total = int(counter[0].item())  # same host sync the reference pays when slicing by counter[0] (custom_functions.py:112-119)
    else:
        total = int(sample_capacity)
    xyzs = torch.empty(total, 3, dtype=_f32, device=dev)
    dirs = torch.empty(total, 3, dtype=_f32, device=dev)
    deltas = torch.empty(total, dtype=_f32, device=dev)
    ts = torch.empty(total, dtype=_f32, device=dev)
    overflow = None
    if sample_capacity is not None:
        overflow = torch.empty((), dtype=_i64, device=dev)
        _lib.check(lib.nrc_raymarching_train_cap_overflow(n, total, _lib.ptr(counter), _lib.ptr(rays_a), _lib.ptr(xyzs), _lib.ptr(dirs), _lib.ptr(deltas),
                                                          _lib.ptr(ts), _lib.ptr(overflow), st), 'raymarching_train(cap)')
    _lib.check(lib.nrc_raymarching_train_write(*args, _lib.ptr(rays_a), _lib.ptr(xyzs), _lib.ptr(dirs), _lib.ptr(deltas),
                                               _lib.ptr(ts), _lib.ptr(ws) if _PARKED_MARCH else None, st), 'raymarching_train(write)')
    return [rays_a, xyzs, dirs, deltas, ts, counter] + ([overflow] if overflow is not None and return_overflow else [])


def raymarching_test(rays_o, rays_d, hits_t, alive_indices, density_bitfield, cascades: int, scale: float,
                     exp_step_factor: float, grid_size: int, max_samples: int, N_samples: int):
    """binding.cpp:84-106. hits_t is advanced in place. Returns [xyzs (A,S,3), dirs (A,S,3), deltas (A,S), ts (A,S), N_eff (A) i32]."""
    _chk((rays_o, 'rays_o', _f32), (rays_d, 'rays_d', _f32), (hits_t, 'hits_t', _f32), (alive_indices, 'alive_indices', _i64),
         (density_bitfield, 'density_bitfield', _u8))
    a = alive_indices.shape[0]
    dev = rays_o.device
    xyzs = torch.empty(a, N_samples, 3, dtype=_f32, device=dev)
    dirs = torch.empty(a, N_samples, 3, dtype=_f32, device=dev)
    deltas = torch.empty(a, N_samples, dtype=_f32, device=dev)
    ts = torch.empty(a, N_samples, dtype=_f32, device=dev)
    n_eff = torch.empty(a, dtype=_i32, device=dev)
    _lib.check(_lib.load().nrc_raymarching_test(
        _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(hits_t), _lib.ptr(alive_indices), a, _lib.ptr(density_bitfield),
        int(cascades), float(scale), float(exp_step_factor), int(grid_size), int(max_samples), int(N_samples),
        _lib.ptr(xyzs), _lib.ptr(dirs), _lib.ptr(deltas), _lib.ptr(ts), _lib.ptr(n_eff), _lib.stream_of(rays_o)), 'raymarching_test')
    return [xyzs, dirs, deltas, ts, n_eff]


def composite_train_fw(sigmas, rgbs, deltas, ts, rays_a, T_threshold: float):
    """binding.cpp:109-126. Returns [total_samples (N) i64, opacity (N), depth (N), rgb (N,3), ws (M)]."""
    _chk((sigmas, 'sigmas', _f32), (rgbs, 'rgbs', _f32), (deltas, 'deltas', _f32), (ts, 'ts', _f32), (rays_a, 'rays_a', _i64))
    n, m = rays_a.shape[0], sigmas.shape[0]
    dev = sigmas.device
    total = torch.empty(n, dtype=_i64, device=dev)
    opacity = torch.empty(n, dtype=_f32, device=dev)
    depth = torch.empty(n, dtype=_f32, device=dev)
    rgb = torch.empty(n, 3, dtype=_f32, device=dev)
    ws = torch.empty(m, dtype=_f32, device=dev)
    _lib.check(_lib.load().nrc_composite_train_fw(
        _lib.ptr(sigmas), _lib.ptr(rgbs), _lib.ptr(deltas), _lib.ptr(ts), _lib.ptr(rays_a), n, m, float(T_threshold),
        _lib.ptr(total), _lib.ptr(opacity), _lib.ptr(depth), _lib.ptr(rgb), _lib.ptr(ws), _lib.stream_of(sigmas)), 'composite_train_fw')
    return [total, opacity, depth, rgb, ws]


def composite_train_bw(dL_dopacity, dL_ddepth, dL_drgb, dL_dws, sigmas, rgbs, ws, deltas, ts, rays_a, opacity, depth, rgb,
                       T_threshold: float):
    """binding.cpp:129-163. Returns [dL_dsigmas (M), dL_drgbs (M,3)]."""
    _chk((dL_dopacity, 'dL_dopacity', _f32), (dL_ddepth, 'dL_ddepth', _f32), (dL_drgb, 'dL_drgb', _f32), (dL_dws, 'dL_dws', _f32),
         (sigmas, 'sigmas', _f32), (rgbs, 'rgbs', _f32), (ws, 'ws', _f32), (deltas, 'deltas', _f32), (ts, 'ts', _f32),
         (rays_a, 'rays_a', _i64), (opacity, 'opacity', _f32), (depth, 'depth', _f32), (rgb, 'rgb', _f32))
    n, m = rays_a.shape[0], sigmas.shape[0]
    dev = sigmas.device
    dL_dsigmas = torch.empty(m, dtype=_f32, device=dev)
    dL_drgbs = torch.empty(m, 3, dtype=_f32, device=dev)
    _lib.check(_lib.load().nrc_composite_train_bw(
        _lib.ptr(dL_dopacity), _lib.ptr(dL_ddepth), _lib.ptr(dL_drgb), _lib.ptr(dL_dws), _lib.ptr(sigmas), _lib.ptr(rgbs),
        _lib.ptr(ws), _lib.ptr(deltas), _lib.ptr(ts), _lib.ptr(rays_a), _lib.ptr(opacity), _lib.ptr(depth), _lib.ptr(rgb), n, m,
        float(T_threshold), _lib.ptr(dL_dsigmas), _lib.ptr(dL_drgbs), _lib.stream_of(sigmas)), 'composite_train_bw')
    return [dL_dsigmas, dL_drgbs]


def composite_test_fw(sigmas, rgbs, deltas, ts, hits_t, alive_indices, T_threshold: float, N_eff_samples, opacity, depth, rgb) -> None:
    """binding.cpp:166-194. In place on opacity/depth/rgb and alive_indices. `hits_t` is accepted and unused, as in the reference kernel."""
    _chk((sigmas, 'sigmas', _f32), (rgbs, 'rgbs', _f32), (deltas, 'deltas', _f32), (ts, 'ts', _f32), (hits_t, 'hits_t', _f32),
         (alive_indices, 'alive_indices', _i64), (N_eff_samples, 'N_eff_samples', _i32), (opacity, 'opacity', _f32),
         (depth, 'depth', _f32), (rgb, 'rgb', _f32))
    a = alive_indices.shape[0]
    n_samples = sigmas.shape[1] if sigmas.dim() == 2 else 1
    _lib.check(_lib.load().nrc_composite_test_fw(
        _lib.ptr(sigmas), _lib.ptr(rgbs), _lib.ptr(deltas), _lib.ptr(ts), _lib.ptr(alive_indices), a, int(n_samples),
        float(T_threshold), _lib.ptr(N_eff_samples), _lib.ptr(opacity), _lib.ptr(depth), _lib.ptr(rgb), _lib.stream_of(sigmas)),
        'composite_test_fw')


def distortion_loss_fw(ws, deltas, ts, rays_a):
    """binding.cpp:197-209. Returns [loss (N), ws_inclusive_scan (M), wts_inclusive_scan (M)]."""
    _chk((ws, 'ws', _f32), (deltas, 'deltas', _f32), (ts, 'ts', _f32), (rays_a, 'rays_a', _i64))
    n, m = rays_a.shape[0], ws.shape[0]
    dev = ws.device
    loss = torch.empty(n, dtype=_f32, device=dev)
    ws_i = torch.empty(m, dtype=_f32, device=dev)
    wts_i = torch.empty(m, dtype=_f32, device=dev)
    _lib.check(_lib.load().nrc_distortion_loss_fw(_lib.ptr(ws), _lib.ptr(deltas), _lib.ptr(ts), _lib.ptr(rays_a), n, m,
                                                  _lib.ptr(loss), _lib.ptr(ws_i), _lib.ptr(wts_i), _lib.stream_of(ws)), 'distortion_loss_fw')
    return [loss, ws_i, wts_i]


def distortion_loss_bw(dL_dloss, ws_inclusive_scan, wts_inclusive_scan, ws, deltas, ts, rays_a):
    """binding.cpp:212-231. Returns dL_dws (M)."""
    _chk((dL_dloss, 'dL_dloss', _f32), (ws_inclusive_scan, 'ws_inclusive_scan', _f32), (wts_inclusive_scan, 'wts_inclusive_scan', _f32),
         (ws, 'ws', _f32), (deltas, 'deltas', _f32), (ts, 'ts', _f32), (rays_a, 'rays_a', _i64))
    n, m = rays_a.shape[0], ws.shape[0]
    out = torch.empty(m, dtype=_f32, device=ws.device)
    _lib.check(_lib.load().nrc_distortion_loss_bw(
        _lib.ptr(dL_dloss), _lib.ptr(ws_inclusive_scan), _lib.ptr(wts_inclusive_scan), _lib.ptr(ws), _lib.ptr(deltas),
        _lib.ptr(ts), _lib.ptr(rays_a), n, m, _lib.ptr(out), _lib.stream_of(ws)), 'distortion_loss_bw')
    return out


# ----------------------------------------------------------------------------------------------- autograd layer
# The six classes the reference star-imports from custom_functions.py (names, positional `apply` arguments and outputs are the interface the
# reference's Renderer.py:41-78 and Optim/Losses/Distortion.py:5-9 call).  Everything behind `apply` is this package's own: every backward is
# one C-ABI call (the ray gradients of the march included -- no torch_scatter), outputs that carry no gradient are marked so, and RayMarcher
# takes an optional trailing `noise` so that data-parallel ranks can slice ONE seeded jitter vector (nerficg_amd.parallel).
def _dense(grad, like, shape=None):
    """Upstream gradient as a contiguous f32 tensor; autograd hands over None for outputs nothing depended on."""
    if grad is None:
        return torch.zeros(shape if shape is not None else like.shape, dtype=_f32, device=like.device)
    return grad.to(_f32).contiguous()


class _Intersector(torch.autograd.Function):
    """Ray / primitive hit lists (custom_functions.py:8-58): pure index / interval outputs, nothing differentiable."""
    op = None

    @classmethod
    def _run(cls, ctx, *args):
        found = tuple(cls.op(*args))
        ctx.mark_non_differentiable(*found)
        return found

    @staticmethod
    def backward(ctx, *_):
        return (None,) * 5


class RayAABBIntersector(_Intersector):
    """apply(rays_o, rays_d, center, half_size, max_hits) -> (hit_cnt, hits_t, hits_voxel_idx)"""
    op = staticmethod(ray_aabb_intersect)

    @staticmethod
    @custom_fwd(cast_inputs=_f32, device_type='cuda')
    def forward(ctx, *rays_boxes_and_cap):
        return RayAABBIntersector._run(ctx, *rays_boxes_and_cap)


class RaySphereIntersector(_Intersector):
    """apply(rays_o, rays_d, center, radii, max_hits) -> (hit_cnt, hits_t, hits_voxel_idx)"""
    op = staticmethod(ray_sphere_intersect)

    @staticmethod
    @custom_fwd(cast_inputs=_f32, device_type='cuda')
    def forward(ctx, *rays_spheres_and_cap):
        return RaySphereIntersector._run(ctx, *rays_spheres_and_cap)


class RayMarcher(torch.autograd.Function):
    """Training-batch sample generation (custom_functions.py:61-137): apply(rays_o, rays_d, hits_t, density_bitfield, cascades, scale,
    exp_step_factor, grid_size, max_samples[, noise]) -> (rays_a, xyzs, dirs, deltas, ts, total_samples).  The first sample of a ray is
    jittered by dt * noise; `noise` defaults to one torch.rand draw per ray on the rays' device."""

    @staticmethod
    @custom_fwd(cast_inputs=_f32, device_type='cuda')
    def forward(ctx, origins, directions, spans, bitfield, *march_cfg):
        n_cascades, box_scale, step_growth, resolution, sample_cap = march_cfg[:5]
        jitter = march_cfg[5] if len(march_cfg) > 5 and march_cfg[5] is not None else torch.rand(origins.shape[0], dtype=_f32, device=origins.device)
        *samples, counter = raymarching_train(origins, directions, spans, bitfield, n_cascades, box_scale, step_growth, jitter.contiguous(),
                                              resolution, sample_cap)
        segments, _, _, step_sizes, depths = samples
        n_marched = counter[0]
        ctx.mark_non_differentiable(segments, step_sizes, depths, n_marched)
        ctx.save_for_backward(segments, depths)
        ctx.n_inputs = 4 + len(march_cfg)
        return (*samples, n_marched)

    @staticmethod
    @custom_bwd(device_type='cuda')
    def backward(ctx, _g_segments, g_positions, g_directions, *_unused):
        segments, depths = ctx.saved_tensors
        n, m = segments.shape[0], depths.shape[0]
        g_o = torch.zeros(n, 3, dtype=_f32, device=depths.device)
        g_d = torch.zeros(n, 3, dtype=_f32, device=depths.device)
        if n and m and (g_positions is not None or g_directions is not None):
            _lib.check(_lib.load().nrc_raymarching_train_bw(
                _lib.ptr(_dense(g_positions, depths, (m, 3))), _lib.ptr(None if g_directions is None else _dense(g_directions, depths)), _lib.ptr(depths),
                _lib.ptr(segments), n, m, _lib.ptr(g_o), _lib.ptr(g_d), _lib.stream_of(depths)), 'raymarching_train_bw')
        return (g_o, g_d) + (None,) * (ctx.n_inputs - 2)


class VolumeRenderer(torch.autograd.Function):
    """Front-to-back compositing of a training batch (custom_functions.py:140-194): apply(sigmas, rgbs, deltas, ts, rays_a, T_threshold) ->
    (n composited samples, opacity, depth, rgb, ws)."""

    @staticmethod
    @custom_fwd(cast_inputs=_f32, device_type='cuda')
    def forward(ctx, *sample_batch):
        density, radiance, step_sizes, depths, segments, cutoff = sample_batch
        per_ray_count, *pixel, weights = composite_train_fw(*sample_batch)
        n_composited = per_ray_count.sum()
        ctx.mark_non_differentiable(n_composited)
        ctx.cutoff = float(cutoff)
        # saved in the argument order of composite_train_bw (binding.cpp:129-163), behind the four upstream gradients
        ctx.save_for_backward(density, radiance, weights, step_sizes, depths, segments, *pixel)
        return (n_composited, *pixel, weights)

    @staticmethod
    @custom_bwd(device_type='cuda')
    def backward(ctx, _g_count, *upstream):
        saved = ctx.saved_tensors
        pixel, weights = saved[6:9], saved[2]
        grads = [_dense(g, like) for g, like in zip(upstream, (*pixel, weights))]
        g_density, g_radiance = composite_train_bw(*grads, *saved, ctx.cutoff)
        return g_density, g_radiance, None, None, None, None


class TruncExp(torch.autograd.Function):
    """Density activation (custom_functions.py:197-208): exp forward; the backward evaluates exp on the argument clamped to [-15, 15], which
    keeps fp16-scaled gradients finite."""
    LIMIT = 15.0

    @staticmethod
    @custom_fwd(cast_inputs=_f32, device_type='cuda')
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return x.exp()

    @staticmethod
    @custom_bwd(device_type='cuda')
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return x.clamp(-TruncExp.LIMIT, TruncExp.LIMIT).exp_().mul_(g)


class DistortionLoss(torch.autograd.Function):
    """Mip-NeRF 360 distortion regulariser per ray (custom_functions.py:211-252): apply(ws, deltas, ts, rays_a) -> loss (N); the gradient
    reaches the weights only."""

    @staticmethod
    def forward(ctx, *weights_and_segments):
        loss, *scans = distortion_loss_fw(*weights_and_segments)
        ctx.save_for_backward(*scans, *weights_and_segments)  # = the argument order of distortion_loss_bw behind dL_dloss
        return loss

    @staticmethod
    def backward(ctx, g_loss):
        saved = ctx.saved_tensors
        return distortion_loss_bw(_dense(g_loss, saved[2], (saved[-1].shape[0],)), *saved), None, None, None
