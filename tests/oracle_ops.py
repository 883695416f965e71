"""TEST INFRASTRUCTURE ONLY: the native ops of the InstantNGP plugin (VolumeRenderingV2's twelve functions, tinycudann.NetworkWithInputEncoding.forward)
implemented by the CPU oracle (oracle/*.c) behind the ops' own Python signatures, on CPU torch tensors, with a call trace.

What it is for (tests/test_shims.py, tests/golden/make_golden.py: make_ingp_orchestration): the reference's OWN host code --
src/Methods/InstantNGP/Renderer.py:30-138 (InstantNGPRayRenderingComponent: box test, training march, query_model, alive-ray inference loop) and
VolumeRenderingV2/custom_functions.py -- cannot execute on a GPU box (it never travels) and has no CPU path of its own; with these stand-ins
patched over the shims it runs in the build container, and its outputs + the sequence of native calls it makes become a committed fixture
(tests/golden/ingp_orchestration.npz).  The GPU mirror, nerficg_amd/instant_ngp.py, is then compared with THAT on the same rays
(tests/test_gpu_render_parity.py): the reference's orchestration is pinned by execution, not by reading.
Nothing under nerficg_amd/ imports this module.
"""
from __future__ import annotations

import numpy as np
import torch

import oracle

TRACE: list[tuple] = []   # (op name, ((shape, dtype) | scalar, ...)) in call order


def _sig(args):
    out = []
    for a in args:
        if torch.is_tensor(a):
            out.append((tuple(a.shape), str(a.dtype).replace('torch.', '')))
        elif isinstance(a, (int, float, bool)):
            out.append(round(float(a), 6) if isinstance(a, float) else a)
        else:
            out.append(type(a).__name__)
    return tuple(out)


def _traced(name):
    def deco(fn):
        def wrapper(*args):
            TRACE.append((name, _sig(args)))
            return fn(*args)
        wrapper.__name__ = name
        return wrapper
    return deco


def _np(t, dtype=None):
    a = t.detach().cpu().contiguous().numpy()
    return a if dtype is None else np.ascontiguousarray(a, dtype=dtype)


T = torch.from_numpy


@_traced('ray_aabb_intersect')
def ray_aabb_intersect(rays_o, rays_d, centers, half_sizes, max_hits):
    cnt, ht, hv = oracle.ray_aabb_intersect(_np(rays_o), _np(rays_d), _np(centers), _np(half_sizes), int(max_hits))
    return [T(cnt), T(ht), T(hv)]


@_traced('raymarching_train')
def raymarching_train(rays_o, rays_d, hits_t, density_bitfield, cascades, scale, exp_step_factor, noise, grid_size, max_samples):
    rays_a, xyzs, dirs, deltas, ts, counter = oracle.raymarching_train(_np(rays_o), _np(rays_d), _np(hits_t), _np(density_bitfield), cascades, scale,
                                                                       exp_step_factor, _np(noise), grid_size, max_samples)
    return [T(rays_a), T(xyzs), T(dirs), T(deltas), T(ts), T(counter)]


@_traced('raymarching_test')
def raymarching_test(rays_o, rays_d, hits_t, alive_indices, density_bitfield, cascades, scale, exp_step_factor, grid_size, max_samples, N_samples):
    ht = hits_t.detach().numpy()          # advanced IN PLACE, like the native op (binding.cpp:84-106): must be the caller's storage
    assert ht.flags.c_contiguous and ht.dtype == np.float32
    xyzs, dirs, deltas, ts, n_eff = oracle.raymarching_test(_np(rays_o), _np(rays_d), ht, _np(alive_indices), _np(density_bitfield), cascades, scale,
                                                            exp_step_factor, grid_size, max_samples, N_samples)
    return [T(xyzs), T(dirs), T(deltas), T(ts), T(n_eff)]


@_traced('composite_train_fw')
def composite_train_fw(sigmas, rgbs, deltas, ts, rays_a, T_threshold):
    total, opacity, depth, rgb, ws = oracle.composite_train_fw(_np(sigmas, np.float32), _np(rgbs, np.float32), _np(deltas), _np(ts), _np(rays_a), T_threshold)
    return [T(total), T(opacity), T(depth), T(rgb), T(ws)]


@_traced('composite_train_bw')
def composite_train_bw(dL_dopacity, dL_ddepth, dL_drgb, dL_dws, sigmas, rgbs, ws, deltas, ts, rays_a, opacity, depth, rgb, T_threshold):
    ds, dr = oracle.composite_train_bw(_np(dL_dopacity), _np(dL_ddepth), _np(dL_drgb), _np(dL_dws), _np(sigmas, np.float32), _np(rgbs, np.float32), _np(ws),
                                       _np(deltas), _np(ts), _np(rays_a), _np(opacity), _np(depth), _np(rgb), T_threshold)
    return [T(ds), T(dr)]


@_traced('composite_test_fw')
def composite_test_fw(sigmas, rgbs, deltas, ts, hits_t, alive_indices, T_threshold, N_eff_samples, opacity, depth, rgb):
    for t in (alive_indices, opacity, depth, rgb):      # written in place
        assert t.is_contiguous()
    oracle.composite_test_fw(_np(sigmas, np.float32), _np(rgbs, np.float32), _np(deltas), _np(ts), alive_indices.numpy(), T_threshold, _np(N_eff_samples),
                             opacity.numpy(), depth.numpy(), rgb.numpy())


def network_forward(module, x: torch.Tensor) -> torch.Tensor:
    """nerficg_amd.tinycudann.NetworkWithInputEncoding.forward on the oracle: the fp16 outputs of the grid network (encoding 0) or of the
    [SH4(d01) | identity] network (encoding 1), from the fp16-rounded parameters the module holds -- returned as FLOAT32 tensors holding fp16 values:
    on the GPU the reference's autograd classes widen their inputs to f32 under autocast (custom_fwd(cast_inputs=torch.float32): TruncExp computes
    exp in f32), which torch does for device tensors only; with an f32 carrier the CPU run does the arithmetic the GPU run does.  The one fp16
    rounding the host code itself applies, (d * 0.5 + 0.5).to(h.dtype) (Renderer.py:52), is applied where the colour network reads it."""
    TRACE.append(('network_forward:' + ('grid' if module.encoding == 0 else 'sh_identity'), _sig((x,))))
    p = oracle.round_half(_np(module.params))
    n_mlp = module.n_mlp_params
    xin = _np(x.float())
    if module.encoding == 0:
        g = module.grid_cfg
        enc = oracle.grid_encode_fw(xin, p[n_mlp:].reshape(-1, 2), n_levels=g['n_levels'], log2_hashmap_size=g['log2_hashmap_size'],
                                    base_resolution=g['base_resolution'], per_level_scale=float(g['per_level_scale']))
        out = oracle.mlp_fw(enc, p[:n_mlp], n_hidden=module.n_hidden, out_act=0)
    else:
        cin = np.concatenate([oracle.sh4_encode(oracle.round_half(xin[:, :3])), oracle.round_half(xin[:, 3:])], axis=1)
        out = oracle.mlp_fw(cin, p[:n_mlp], n_hidden=module.n_hidden, out_act=1)
    return T(np.ascontiguousarray(out[:, :module.n_output_dims], dtype=np.float32))


OPS = {f.__name__: f for f in (ray_aabb_intersect, raymarching_train, raymarching_test, composite_train_fw, composite_train_bw, composite_test_fw)}


def install(shim_module, package_module, network_class) -> None:
    """Patch the oracle-backed ops over the shim module `VolumeRenderingV2` (what custom_functions.py calls), over the reference's package
    Methods.InstantNGP.VolumeRenderingV2 (which copied the names at import) and over the drop-in network class's forward."""
    for name, fn in OPS.items():
        setattr(shim_module, name, fn)
        setattr(package_module, name, fn)
    network_class.forward = lambda self, x: network_forward(self, x)
