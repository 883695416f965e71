"""oracle -- Python face of the CPU oracle (oracle/*.c built into oracle/liboracle.so by oracle/Makefile).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg. The
product package nerficg_amd never imports this module.  All functions take and return numpy arrays.
"""
from __future__ import annotations

import ctypes
import subprocess
from functools import lru_cache
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / 'liboracle.so'

f32, i32, i64, u8 = np.float32, np.int32, np.int64, np.uint8
_P = ctypes.c_void_p


def build(force: bool = False) -> Path:
    srcs = sorted(HERE.glob('*.c'))
    if force or not LIB.exists() or any(s.stat().st_mtime > LIB.stat().st_mtime for s in srcs + [HERE / 'Makefile']):
        subprocess.run(['make', '-C', str(HERE), '-B', 'liboracle.so'], check=True, capture_output=True)
    return LIB


@lru_cache(maxsize=1)
def lib() -> ctypes.CDLL:
    build()
    return ctypes.CDLL(str(LIB))


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _p(a):
    return None if a is None else a.ctypes.data_as(_P)


def _call(name, *args, restype=None):
    fn = getattr(lib(), name)
    fn.restype = restype
    conv = []
    for a in args:
        if isinstance(a, np.ndarray):
            conv.append(_p(a))
        elif isinstance(a, float):
            conv.append(ctypes.c_float(a))
        elif isinstance(a, (int, np.integer)):
            conv.append(ctypes.c_int64(int(a)))
        elif a is None:
            conv.append(None)
        else:
            conv.append(a)
    return fn(*conv)


def _i(v):  # C `int` argument
    return ctypes.c_int(int(v))


# ------------------------------------------------------------------------------------------------ VolumeRenderingV2
def morton3D(coords):
    coords = _c(coords, i32)
    out = np.empty(coords.shape[0], i32)
    _call('oracle_morton3D', coords, coords.shape[0], out)
    return out


def morton3D_invert(indices):
    indices = _c(indices, i32)
    out = np.empty((indices.shape[0], 3), i32)
    _call('oracle_morton3D_invert', indices, indices.shape[0], out)
    return out


def packbits(grid, thr):
    grid = _c(grid, f32).reshape(-1)
    out = np.empty(grid.size // 8, u8)
    _call('oracle_packbits', grid, out.size, float(thr), out)
    return out


def occupancy_update(grid, cell_indices, densities, decay, density_threshold):
    """InstantNGPRenderer.update_occupancy_grid after the density query (src/Methods/InstantNGP/Renderer.py:258-272): scratch grid <- densities
    by index (the reference's index_put keeps an arbitrary one of several samples of a cell; the maximum is taken here, entries with
    index < 0 are padding), grid = where(grid < 0, grid, max(grid * decay, scratch)), mean over cells > 0 (f64 accumulation), bitfield at
    min(mean, density_threshold) with Python's min() NaN behaviour.  Returns (grid, bitfield, threshold_used, mean)."""
    grid = _c(grid, f32).copy()
    C, N = grid.shape
    scratch = np.zeros_like(grid)
    idx = np.asarray(cell_indices).reshape(C, -1)
    den = np.asarray(densities).astype(f32).reshape(C, -1)
    for c in range(C):
        ok = idx[c] >= 0
        np.maximum.at(scratch[c], idx[c][ok], den[c][ok])
    upd = np.maximum(grid * f32(decay), scratch)
    grid = np.where(grid < 0, grid, upd).astype(f32)
    pos = grid[grid > 0]
    mean = f32(pos.astype(np.float64).mean()) if pos.size else f32(np.nan)
    thr = f32(density_threshold) if f32(density_threshold) < mean else mean
    bits = np.zeros(C * N // 8, u8) if np.isnan(thr) else packbits(grid, float(thr))
    return grid, bits, thr, mean


def ray_aabb_intersect(rays_o, rays_d, centers, half_sizes, max_hits):
    rays_o, rays_d, centers, half_sizes = _c(rays_o, f32), _c(rays_d, f32), _c(centers, f32), _c(half_sizes, f32)
    n = rays_o.shape[0]
    cnt, ht, hv = np.empty(n, i32), np.empty((n, max_hits, 2), f32), np.empty((n, max_hits), i64)
    _call('oracle_ray_aabb_intersect', rays_o, rays_d, centers, half_sizes, n, centers.shape[0], _i(max_hits), cnt, ht, hv)
    return cnt, ht, hv


def ray_sphere_intersect(rays_o, rays_d, centers, radii, max_hits):
    rays_o, rays_d, centers, radii = _c(rays_o, f32), _c(rays_d, f32), _c(centers, f32), _c(radii, f32)
    n = rays_o.shape[0]
    cnt, ht, hv = np.empty(n, i32), np.empty((n, max_hits, 2), f32), np.empty((n, max_hits), i64)
    _call('oracle_ray_sphere_intersect', rays_o, rays_d, centers, radii, n, centers.shape[0], _i(max_hits), cnt, ht, hv)
    return cnt, ht, hv


def raymarching_train(rays_o, rays_d, hits_t, bitfield, cascades, scale, exp_step_factor, noise, grid_size, max_samples):
    rays_o, rays_d, hits_t, noise = _c(rays_o, f32), _c(rays_d, f32), _c(hits_t, f32), _c(noise, f32)
    bitfield = _c(bitfield, u8)
    n = rays_o.shape[0]
    common = (rays_o, rays_d, hits_t, bitfield, _i(cascades), float(scale), float(exp_step_factor), noise, _i(grid_size),
              _i(max_samples), n)
    total = _call('oracle_raymarching_train', *common, None, None, None, None, None, None, restype=ctypes.c_int64)
    rays_a = np.empty((n, 3), i64)
    xyzs, dirs = np.zeros((total, 3), f32), np.zeros((total, 3), f32)
    deltas, ts = np.zeros(total, f32), np.zeros(total, f32)
    counter = np.zeros(2, i32)
    _call('oracle_raymarching_train', *common, rays_a, xyzs, dirs, deltas, ts, counter, restype=ctypes.c_int64)
    return rays_a, xyzs, dirs, deltas, ts, counter


def raymarching_test(rays_o, rays_d, hits_t, alive, bitfield, cascades, scale, exp_step_factor, grid_size, max_samples, N_samples):
    """hits_t is modified in place (must be a contiguous float32 array)."""
    rays_o, rays_d, alive, bitfield = _c(rays_o, f32), _c(rays_d, f32), _c(alive, i64), _c(bitfield, u8)
    assert hits_t.dtype == f32 and hits_t.flags.c_contiguous
    a = alive.shape[0]
    xyzs, dirs = np.zeros((a, N_samples, 3), f32), np.zeros((a, N_samples, 3), f32)
    deltas, ts = np.zeros((a, N_samples), f32), np.zeros((a, N_samples), f32)
    n_eff = np.zeros(a, i32)
    _call('oracle_raymarching_test', rays_o, rays_d, hits_t, alive, a, bitfield, _i(cascades), float(scale),
          float(exp_step_factor), _i(grid_size), _i(max_samples), _i(N_samples), xyzs, dirs, deltas, ts, n_eff)
    return xyzs, dirs, deltas, ts, n_eff


def composite_train_fw(sigmas, rgbs, deltas, ts, rays_a, T_threshold):
    sigmas, rgbs, deltas, ts, rays_a = _c(sigmas, f32), _c(rgbs, f32), _c(deltas, f32), _c(ts, f32), _c(rays_a, i64)
    n, m = rays_a.shape[0], sigmas.shape[0]
    total, opacity, depth, rgb, ws = np.empty(n, i64), np.empty(n, f32), np.empty(n, f32), np.empty((n, 3), f32), np.empty(m, f32)
    _call('oracle_composite_train_fw', sigmas, rgbs, deltas, ts, rays_a, n, m, float(T_threshold), total, opacity, depth, rgb, ws)
    return total, opacity, depth, rgb, ws


def composite_train_bw(dL_dopacity, dL_ddepth, dL_drgb, dL_dws, sigmas, rgbs, ws, deltas, ts, rays_a, opacity, depth, rgb, T_threshold):
    a = [_c(x, f32) for x in (dL_dopacity, dL_ddepth, dL_drgb, dL_dws, sigmas, rgbs, ws, deltas, ts)]
    rays_a = _c(rays_a, i64)
    b = [_c(x, f32) for x in (opacity, depth, rgb)]
    n, m = rays_a.shape[0], a[4].shape[0]
    ds, dr = np.empty(m, f32), np.empty((m, 3), f32)
    _call('oracle_composite_train_bw', *a, rays_a, *b, n, m, float(T_threshold), ds, dr)
    return ds, dr


def composite_test_fw(sigmas, rgbs, deltas, ts, alive, T_threshold, n_eff, opacity, depth, rgb):
    """alive / opacity / depth / rgb are modified in place."""
    sigmas, rgbs, deltas, ts, n_eff = _c(sigmas, f32), _c(rgbs, f32), _c(deltas, f32), _c(ts, f32), _c(n_eff, i32)
    for arr, dt in ((alive, i64), (opacity, f32), (depth, f32), (rgb, f32)):
        assert arr.dtype == dt and arr.flags.c_contiguous
    _call('oracle_composite_test_fw', sigmas, rgbs, deltas, ts, alive, alive.shape[0], _i(sigmas.shape[1]), float(T_threshold),
          n_eff, opacity, depth, rgb)


def distortion_loss_fw(ws, deltas, ts, rays_a):
    ws, deltas, ts, rays_a = _c(ws, f32), _c(deltas, f32), _c(ts, f32), _c(rays_a, i64)
    n, m = rays_a.shape[0], ws.shape[0]
    loss, wi, wti = np.empty(n, f32), np.empty(m, f32), np.empty(m, f32)
    _call('oracle_distortion_loss_fw', ws, deltas, ts, rays_a, n, m, loss, wi, wti)
    return loss, wi, wti


def distortion_loss_bw(dL_dloss, ws_incl, wts_incl, ws, deltas, ts, rays_a):
    a = [_c(x, f32) for x in (dL_dloss, ws_incl, wts_incl, ws, deltas, ts)]
    rays_a = _c(rays_a, i64)
    out = np.empty(a[3].shape[0], f32)
    _call('oracle_distortion_loss_bw', *a, rays_a, rays_a.shape[0], a[3].shape[0], out)
    return out


def morton_encode(positions):
    positions = _c(positions, f32)
    out = np.empty(positions.shape[0], i64)
    _call('oracle_morton_encode', positions, positions.shape[0], out)
    return out


# ------------------------------------------------------------------------------------------------ tinycudann subset
def round_half(a):
    """fp16 round trip (numpy's float16 conversion is IEEE round-to-nearest-even, same as oracle_round_to_half)."""
    return np.asarray(a, f32).astype(np.float16).astype(f32)


def round_half_c(a, soft=False):
    """The C library's fp16 round trip: F16C instructions when compiled in (has_f16c()), `soft=True` the bit-twiddled definition."""
    a = _c(a, f32)
    out = np.empty_like(a)
    _call('oracle_round_to_half_soft' if soft else 'oracle_round_to_half', a, a.size, out)
    return out


def has_f16c() -> bool:
    fn = lib().oracle_has_f16c
    fn.restype = ctypes.c_int
    return bool(fn())


def grid_layout(n_levels=16, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.3819128800):
    offsets = np.zeros(n_levels + 1, np.uint32)
    scales = np.zeros(n_levels, f32)
    res = np.zeros(n_levels, np.uint32)
    fn = lib().oracle_grid_layout
    fn.restype = ctypes.c_uint32
    total = fn(_i(n_levels), _i(log2_hashmap_size), _i(base_resolution), ctypes.c_float(per_level_scale), _p(offsets), _p(scales), _p(res))
    return int(total), offsets, scales, res


def _half_flag(accumulate):
    if accumulate not in ('float', 'half'):
        raise ValueError("accumulate must be 'float' (the definition the HIP kernels are held to) or 'half' (model of upstream's __half sums)")
    return _i(accumulate == 'half')


def grid_encode_fw(x01, table, n_levels=16, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.3819128800, accumulate='float'):
    """x01 (M,3) f32 in [0,1]; table (entries,F) fp16-representable f32 -> (M, F*n_levels) fp16-rounded f32, level-major (F = table.shape[1]:
    n_features_per_level).  accumulate='half': fp16 weights and fp16 running sums (oracle_grid_encode_fwF)."""
    x01, table = _c(x01, f32), _c(table, f32)
    n_feat = int(table.shape[1]) if table.ndim == 2 else 2
    out = np.empty((x01.shape[0], n_feat * n_levels), f32)
    _call('oracle_grid_encode_fwF', x01, x01.shape[0], table, _i(n_levels), _i(n_feat), _i(log2_hashmap_size), _i(base_resolution),
          float(per_level_scale), _half_flag(accumulate), out)
    return out


def grid_encode_bw(x01, d_out, n_entries, n_levels=16, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.3819128800):
    """d_out (M, F*n_levels) -> gradient table (n_entries, F) f32; F = d_out.shape[1] / n_levels."""
    x01, d_out = _c(x01, f32), _c(d_out, f32)
    n_feat = d_out.shape[1] // n_levels
    grad = np.zeros((n_entries, n_feat), f32)
    _call('oracle_grid_encode_bwF', x01, x01.shape[0], d_out, _i(n_levels), _i(n_feat), _i(log2_hashmap_size), _i(base_resolution),
          float(per_level_scale), grad)
    return grad


def sh4_encode(d01):
    d01 = _c(d01, f32)
    out = np.empty((d01.shape[0], 16), f32)
    _call('oracle_sh4_encode', d01, d01.shape[0], out)
    return out


def mlp_fw(x, W, n_in=32, width=64, n_hidden=1, n_out_pad=16, out_act=0, want_acts=False, accumulate='float'):
    """x (M,n_in), W flat -- both fp16-representable f32. Returns out (M,n_out_pad) [, acts (n_hidden,M,width)].
    accumulate='half': running sums rounded to fp16 after every multiply-add (oracle_mlp_fw2)."""
    x, W = _c(x, f32), _c(W, f32)
    m = x.shape[0]
    out = np.empty((m, n_out_pad), f32)
    acts = np.empty((n_hidden, m, width), f32) if want_acts else None
    _call('oracle_mlp_fw2', x, m, W, _i(n_in), _i(width), _i(n_hidden), _i(n_out_pad), _i(out_act), _half_flag(accumulate), out, acts)
    return (out, acts) if want_acts else out


def mlp_bw(x, W, out, acts, d_out, n_in=32, width=64, n_hidden=1, n_out_pad=16, out_act=0):
    x, W, out, acts, d_out = _c(x, f32), _c(W, f32), _c(out, f32), _c(acts, f32), _c(d_out, f32)
    m = x.shape[0]
    dW = np.empty_like(W)
    d_in = np.empty((m, n_in), f32)
    _call('oracle_mlp_bw', x, m, W, _i(n_in), _i(width), _i(n_hidden), _i(n_out_pad), _i(out_act), out, acts, d_out, dW, d_in)
    return dW, d_in


def ngp_query_staged(xyz01, dirs, Wd, Wc, table, accumulate='float', **grid_kw):
    """InstantNGPRayRenderingComponent.query_model (src/Methods/InstantNGP/Renderer.py:48-53) assembled from the stage functions:
    h = density_net(grid(x)); sigma = exp(h[:,0]); rgb = color_net([SH4(fp16(d*.5+.5)) | h])[:, :3].
    accumulate='half': every running sum of the encoder and of the two MLPs in fp16 (the model of upstream tiny-cuda-nn's arithmetic)."""
    enc = grid_encode_fw(xyz01, table, accumulate=accumulate, **grid_kw)
    h = mlp_fw(enc, Wd, n_hidden=1, out_act=0, accumulate=accumulate)
    sigma = np.exp(h[:, 0].astype(f32))
    d01 = round_half(_c(dirs, f32) * f32(0.5) + f32(0.5))
    cin = np.concatenate([sh4_encode(d01), h], axis=1)
    rgb = mlp_fw(cin, Wc, n_hidden=2, out_act=1, accumulate=accumulate)[:, :3]
    return sigma.astype(f32), rgb, h


def ngp_query(xyz01, dirs, Wd, Wc, table, accumulate='float', n_levels=16, log2_hashmap_size=19, base_resolution=16, per_level_scale=1.3819128800):
    """The same composition as ngp_query_staged in ONE C call (oracle_ngp_query: one parallel loop over blocks of samples, no numpy glue
    between the stages) -- what the tests and bench.py's cpu_baseline use; sigma differs from the staged form by libm-expf-vs-numpy-exp only
    where those differ (tests/test_oracle_tcnn.py pins the two against each other)."""
    xyz01, dirs, Wd, Wc, table = _c(xyz01, f32), _c(dirs, f32), _c(Wd, f32), _c(Wc, f32), _c(table, f32)
    m = xyz01.shape[0]
    if Wd.size != 32 * 64 + 64 * 16 or Wc.size != 32 * 64 + 64 * 64 + 64 * 16 or n_levels != 16:
        raise ValueError('ngp_query: the fused call is written for the shipped configuration (16 levels x 2, 64 neurons, 16 density outputs)')
    sigma, rgb, h = np.empty(m, f32), np.empty((m, 3), f32), np.empty((m, 16), f32)
    _call('oracle_ngp_query', xyz01, dirs, m, Wd, Wc, table, _i(n_levels), _i(log2_hashmap_size), _i(base_resolution), float(per_level_scale),
          _half_flag(accumulate), sigma, rgb, h)
    return sigma, rgb, h


# ------------------------------------------------------------------------------------------------ 3DGS rasterizer
class GSState:
    """Everything the rasterizer keeps between forward and backward (geometry / binning / image state)."""


def gs_forward(means3D, opacities, viewmatrix, projmatrix, campos, tan_fovx, tan_fovy, W, H, bg, sh_degree=3, shs=None,
               colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0, dtype=np.float32):
    """GaussianRasterizer.forward (call sites src/Methods/GaussianSplatting/Renderer.py:60-81). viewmatrix/projmatrix are the
    (4,4) arrays the reference passes (w2c.T and w2c.T @ P.T), shs (P,16,3). Returns (color (3,H,W), radii (P), state)."""
    pre = 'gsf_' if dtype == np.float32 else 'gsd_'
    ft = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    c = lambda a: None if a is None else np.ascontiguousarray(a, dtype=dtype)
    means3D, opacities, viewmatrix, projmatrix, campos, bg = c(means3D), c(opacities).reshape(-1), c(viewmatrix), c(projmatrix), c(campos), c(bg)
    shs, colors_precomp, scales, rotations, cov3D_precomp = c(shs), c(colors_precomp), c(scales), c(rotations), c(cov3D_precomp)
    P = means3D.shape[0]
    M = 0 if shs is None else shs.shape[1]
    st = GSState()
    st.dtype, st.P, st.M, st.D, st.W, st.H = dtype, P, M, sh_degree, W, H
    st.radii = np.zeros(P, i32); st.depths = np.zeros(P, dtype); st.points_xy = np.zeros((P, 2), dtype)
    st.conic_opacity = np.zeros((P, 4), dtype); st.rgb = np.zeros((P, 3), dtype); st.clamped = np.zeros((P, 3), u8)
    st.cov3D = np.zeros((P, 6), dtype); st.tiles_touched = np.zeros(P, np.uint32)
    fn = getattr(lib(), pre + 'preprocess')
    fn.restype = ctypes.c_int64
    n = fn(_i(P), _i(sh_degree), _i(M), _i(W), _i(H), _p(means3D), _p(shs), _p(colors_precomp), _p(opacities), _p(scales), ft(scale_modifier),
           _p(rotations), _p(cov3D_precomp), _p(viewmatrix), _p(projmatrix), _p(campos), ft(tan_fovx), ft(tan_fovy), _p(st.radii), _p(st.depths),
           _p(st.points_xy), _p(st.conic_opacity), _p(st.rgb), _p(st.clamped), _p(st.cov3D), _p(st.tiles_touched))
    st.num_rendered = int(n)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    st.point_list = np.zeros(max(st.num_rendered, 1), i32); st.ranges = np.zeros((gx * gy, 2), np.uint32)
    st.n_contrib = np.zeros(H * W, np.uint32); st.final_T = np.zeros(H * W, dtype)
    color = np.zeros((3, H, W), dtype)
    fn = getattr(lib(), pre + 'bin_and_render')
    fn.restype = None
    fn(_i(P), _i(W), _i(H), _p(bg), _p(st.radii), _p(st.depths), _p(st.points_xy), _p(st.conic_opacity), _p(st.rgb), ctypes.c_int64(st.num_rendered),
       _p(st.point_list), _p(st.ranges), _p(color), _p(st.n_contrib), _p(st.final_T))
    st.inputs = dict(means3D=means3D, shs=shs, colors_precomp=colors_precomp, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp,
                     viewmatrix=viewmatrix, projmatrix=projmatrix, campos=campos, bg=bg, tan_fovx=tan_fovx, tan_fovy=tan_fovy,
                     scale_modifier=scale_modifier)
    return color, st.radii.copy(), st


def gs_render_source_order(st, bg):
    """(color (3,H,W), n_contrib (H*W), final_T (H*W)) of the tile lists in `st` (gs_forward's state) with the blend in the published source's order --
    no fused multiply-add, T (1 - alpha), colour + c alpha T (gs_oracle_impl.h: render_source_order).  The fused statement the kernels and gs_forward share is
    held against this one with a bounded-mismatch criterion."""
    pre = 'gsf_' if st.dtype == np.float32 else 'gsd_'
    color = np.zeros((3, st.H, st.W), st.dtype)
    n_contrib = np.zeros(st.H * st.W, np.uint32)
    final_T = np.zeros(st.H * st.W, st.dtype)
    fn = getattr(lib(), pre + 'render_source_order')
    fn.restype = None
    fn(_i(st.W), _i(st.H), _p(np.ascontiguousarray(bg, dtype=st.dtype)), _p(st.points_xy), _p(st.conic_opacity), _p(st.rgb), _p(st.point_list), _p(st.ranges),
       _p(color), _p(n_contrib), _p(final_T))
    return color, n_contrib, final_T


def usable_cpus() -> int:
    """CPUs this process may actually use: the affinity mask capped by the cgroup's CPU quota.  The GPU boxes of the pool show 256 logical CPUs and
    grant 16 (cpu.max = 1600000 100000): 256 OpenMP threads on a 16-CPU quota are throttled and run SLOWER than 32 (measured round 5, oracle.mlp_fw:
    1 / 8 / 32 / 64 / 128 / 256 threads = 1.7 / 13.0 / 41.9 / 22.7 / 10.3 / 3.7 Msamples/s) -- which is what rounds 3-4 reported as "8.5 x on 256 threads"."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]            # cgroup v2
        if quota != 'max':
            n = min(n, max(1, math.ceil(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())              # cgroup v1
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota > 0:
                n = min(n, max(1, math.ceil(quota / period)))
        except (OSError, ValueError):
            pass
    return n


def set_threads(n: int) -> int:
    """OpenMP threads of the oracle library (0 = every CPU this process may use, usable_cpus()); returns the previous setting."""
    fn = lib().oracle_set_threads
    fn.restype = ctypes.c_int
    return int(fn(ctypes.c_int(int(n) if int(n) > 0 else usable_cpus())))


def gs_backward(st, dL_dpix, threads: int = 1):
    """Returns dict of gradients: mean2D (P,3), conic (P,4), opacity (P), color (P,3), mean3D (P,3), cov3D (P,6), sh (P,M,3), scale (P,3), rot (P,4).
    threads = 1 (default): the serial order of the float additions, reproducible bit for bit; 0 = all host cores (CPU-baseline timing)."""
    before = set_threads(threads)
    try:
        return _gs_backward(st, dL_dpix)
    finally:
        set_threads(before)


def _gs_backward(st, dL_dpix):
    dtype = st.dtype
    pre = 'gsf_' if dtype == np.float32 else 'gsd_'
    ft = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    g = np.ascontiguousarray(dL_dpix, dtype=dtype)
    P, M = st.P, max(st.M, 1)
    out = dict(mean2D=np.zeros((P, 3), dtype), conic=np.zeros((P, 4), dtype), opacity=np.zeros(P, dtype), color=np.zeros((P, 3), dtype),
               mean3D=np.zeros((P, 3), dtype), cov3D=np.zeros((P, 6), dtype), sh=np.zeros((P, M, 3), dtype), scale=np.zeros((P, 3), dtype),
               rot=np.zeros((P, 4), dtype))
    a = st.inputs
    fn = getattr(lib(), pre + 'backward')
    fn.restype = None
    fn(_i(P), _i(st.D), _i(st.M), _i(st.W), _i(st.H), _p(a['bg']), _p(a['means3D']), _p(a['shs']), _p(a['colors_precomp']), _p(a['scales']),
       ft(a['scale_modifier']), _p(a['rotations']), _p(a['cov3D_precomp']), _p(a['viewmatrix']), _p(a['projmatrix']), _p(a['campos']),
       ft(a['tan_fovx']), ft(a['tan_fovy']), _p(st.radii), _p(st.points_xy), _p(st.conic_opacity), _p(st.rgb), _p(st.clamped), _p(st.cov3D),
       _p(st.point_list), _p(st.ranges), _p(st.n_contrib), _p(st.final_T), _p(g), _p(out['mean2D']), _p(out['conic']), _p(out['opacity']),
       _p(out['color']), _p(out['mean3D']), _p(out['cov3D']), _p(out['sh']), _p(out['scale']), _p(out['rot']))
    return out


# ------------------------------------------------------------------------------------------------ SSIM (3DGS loss; ssim_oracle.c)
def ssim_forward(img1, img2, C1=0.01 ** 2, C2=0.03 ** 2, train=True):
    """img (..., H, W) f32 -> ssim_map and (train) the maps d ssim / d mu1, d sigma1^2, d sigma12 (same shape)."""
    a, b = _c(img1, f32), _c(img2, f32)
    H, W = a.shape[-2:]
    planes = a.size // (H * W)
    out = [np.empty_like(a) for _ in range(4 if train else 1)]
    _call('oracle_ssim_forward', a, b, planes, H, W, float(C1), float(C2), out[0], *(out[1:] if train else (None, None, None)))
    return out if train else out[0]


def ssim_backward(img1, img2, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12):
    a, b = _c(img1, f32), _c(img2, f32)
    H, W = a.shape[-2:]
    out = np.empty_like(a)
    _call('oracle_ssim_backward', a, b, a.size // (H * W), H, W, _c(dL_dmap, f32), _c(dm_dmu1, f32), _c(dm_dsigma1_sq, f32), _c(dm_dsigma12, f32), out)
    return out


# ------------------------------------------------------------------------------------------------ Adam (adam_oracle.c)
def adam_step(p, g, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, adam_w_mode=False, grad_scale=1.0, found_inf=False):
    """In-place on copies; returns (p, m, v) after one step number `step` (1-based)."""
    p, m, v = _c(p, f32).copy(), _c(m, f32).copy(), _c(v, f32).copy()
    bc1, bc2 = 1.0 - betas[0] ** step, 1.0 - betas[1] ** step
    _call('oracle_adam_step', p, _c(g, f32), m, v, p.size, float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay),
          int(bool(adam_w_mode)), float(bc1), float(bc2), float(grad_scale), int(bool(found_inf)))
    return p, m, v


# ------------------------------------------------------------------------------------------------ 3-NN (knn_oracle.c)
def knn3_mean_sq_dist(points):
    p = _c(points, f32)
    out = np.empty(p.shape[0], f32)
    _call('oracle_knn3_mean_sq_dist', p, p.shape[0], out)
    return out
