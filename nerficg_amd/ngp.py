"""nerficg_amd.ngp -- MI355X-native fast paths for the InstantNGP hot loop that go beyond the reference's op granularity.

The reference composes the loop from separate ops with tensors materialised in between
(src/Methods/InstantNGP/Renderer.py:48-138).  The functions here fuse across those boundaries while producing the same
results as the drop-in modules (tests/test_gpu_tcnn_parity.py, tests/test_gpu_render_parity.py).
"""
from __future__ import annotations

import torch

from . import _lib

__all__ = ['query_fused', 'query_train', 'query_modules', 'default_layout', 'composite_over_background', 'weight_decay_mlp', 'instant_ngp_loss']


def default_layout(density_net, color_net) -> bool:
    """Are both networks the configuration the fused paths of this module (and of nerficg_amd.instant_ngp / ngp_trainer) are built for -- 16 x 2 hash grid,
    degree-4 SH?  Any other configuration the yaml can ask for (HASHGRID_N_LEVELS, HASHGRID_N_FEATURES_PER_LEVEL, DIR_SH_ENCODING_DEGREE) runs through the
    drop-in modules' own forward / backward (query_modules), which is what the reference's Renderer.py:48-53 does."""
    return bool(getattr(density_net, 'default_layout', True) and getattr(color_net, 'default_layout', True))


def query_modules(density_net, color_net, xyz01: torch.Tensor, dirs: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """query_model exactly as src/Methods/InstantNGP/Renderer.py:48-53 composes it from the two tcnn modules (differentiable through them): the path of
    every configuration the fused kernels are not built for."""
    h = density_net(xyz01)
    sigmas = torch.exp(h[:, 0].float())       # TruncExp forward (custom_functions.py:201-204); its clamp only acts in the backward
    if torch.is_grad_enabled() and h.requires_grad:
        from .VolumeRenderingV2 import TruncExp
        sigmas = TruncExp.apply(h[:, 0])
    rgbs = color_net(torch.cat([(dirs * 0.5 + 0.5).to(h.dtype), h], dim=-1)).float()
    return sigmas, rgbs


def query_fused(density_net, color_net, xyz01: torch.Tensor, dirs: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """query_model (Renderer.py:48-53) as one encode + one MLP kernel per chunk: xyz01 (M,3) f32 in [0,1], dirs (M,3) unit f32 -> sigmas (M) f32, rgbs (M,3) f32."""
    if not default_layout(density_net, color_net):
        with torch.no_grad():
            return query_modules(density_net, color_net, xyz01, dirs)
    _lib.check_input(xyz01, 'xyz01', torch.float32)
    _lib.check_input(dirs, 'dirs', torch.float32)
    m = xyz01.shape[0]
    wd = density_net._half_params()
    wc = color_net._half_params()
    g = density_net.grid_cfg
    sig = torch.empty(m, dtype=torch.float32, device=xyz01.device)
    rgb = torch.empty(m, 3, dtype=torch.float32, device=xyz01.device)
    ws = torch.empty(int(_lib.load().nrc_ngp_query_ws_bytes(m)), dtype=torch.uint8, device=xyz01.device)
    _lib.check(_lib.load().nrc_ngp_query_fused(
        _lib.ptr(xyz01), _lib.ptr(dirs), m, _lib.ptr(wd), _lib.ptr(wc), _lib.ptr(density_net._table16()), g['n_levels'],
        g['log2_hashmap_size'], g['base_resolution'], float(g['per_level_scale']), _lib.ptr(sig), _lib.ptr(rgb), _lib.ptr(ws),
        _lib.stream_of(sig)), 'ngp_query_fused')
    return sig, rgb


# ---- the weight-decay term of InstantNGPLoss (Loss.py:15: 0.5e-6 * Model.weight_decay_mlp()) without its dense gradients -----------------------------
# weight_decay_mlp() touches 10 240 MLP weights, but through torch's slicing / square / sum its backward builds a zero-filled gradient of the whole
# 12.2 M-element parameter vector and adds it to the networks' gradient (~14 launches, a 49 MB fill and a 49 MB add per iteration).  Here the term is
# one launch forward; backward it launches NOTHING: it leaves a "seed" -- (dL/dterm on the device, 2 / n, how many leading weights) -- for the
# parameter, and the backward of the training query, which runs later in the same backward pass and has to clear its gradient buffers anyway,
# starts the MLP part of them at seed * w instead of zero (nrc_clear_seed_two).  A seed nobody picked up by the end of the backward pass (the query was
# not part of this graph) is added to p.grad the ordinary way by a callback the engine runs when the pass ends -- the result is the term's gradient
# either way.
_GRAD_SEEDS: dict[int, tuple] = {}   # parameter storage address -> (weak reference to the parameter, upstream 0-d f32 on the device, coefficient, count)


def _take_seed(param):
    hit = _GRAD_SEEDS.get(param.data_ptr())
    if hit is None or hit[0]() is not param:
        return None
    del _GRAD_SEEDS[param.data_ptr()]
    return hit


def _flush_seeds() -> None:
    """End of a backward pass: seeds that the training query did not consume become ordinary gradient contributions."""
    while _GRAD_SEEDS:
        _, (ref, up, coeff, count) = _GRAD_SEEDS.popitem()
        p = ref()
        if p is None:
            continue
        with torch.no_grad():
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            p.grad[:count].add_(p.detach()[:count] * (coeff * up.to(p.dtype)))


def weight_decay_mlp(density_net, color_net, n_density_mlp: int, n_total: int) -> torch.Tensor:
    """Model.weight_decay_mlp (Model.py:38-44): mean squared MLP weight over both networks, the hash table excluded -- one launch forward, no launch
    and no dense gradient backward (see _GRAD_SEEDS).  Falls back to the torch expression for parameters that are not plain f32 device vectors."""
    pd, pc = density_net.params, color_net.params
    plain = all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in (pd, pc))
    if not plain or not torch.is_grad_enabled() or not (pd.requires_grad and pc.requires_grad):
        return (pd[:n_density_mlp].square().sum() + pc.square().sum()) / n_total
    return _WeightDecayNode.apply(pd, pc, int(n_density_mlp), int(n_total))


class _WeightDecayNode(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, params_d, params_c, n_d, n_total):
        _GRAD_SEEDS.clear()      # a seed that survived until the NEXT forward pass belongs to a backward pass that never finished: it must not reach this one's query
        out = torch.empty(1, dtype=torch.float32, device=params_d.device)
        _lib.check(_lib.load().nrc_sum_squares_two(_lib.ptr(params_d), n_d, _lib.ptr(params_c), params_c.numel(), 1.0 / n_total, _lib.ptr(out),
                                                   _lib.stream_of(out)), 'sum_squares_two')
        ctx.save_for_backward(params_d, params_c)
        ctx.n_d, ctx.n_total = n_d, n_total
        return out.reshape(())

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        params_d, params_c = ctx.saved_tensors
        up = g.detach().to(torch.float32).reshape(1).contiguous()
        _leave_seeds(((params_d, ctx.n_d), (params_c, params_c.numel())), up, 2.0 / ctx.n_total)
        return None, None, None, None


def _leave_seeds(params_and_counts, up, coeff: float) -> None:
    """register (up * coeff * w) as the pending gradient of the leading weights of each parameter (see _GRAD_SEEDS)"""
    import weakref
    for p, count in params_and_counts:
        prev = _GRAD_SEEDS.get(p.data_ptr())
        if prev is not None and prev[0]() is p:      # a second term on the same parameter in one pass: settle the first the ordinary way
            _GRAD_SEEDS.pop(p.data_ptr())
            with torch.no_grad():
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                p.grad[:prev[3]].add_(p.detach()[:prev[3]] * (prev[2] * prev[1]))
        _GRAD_SEEDS[p.data_ptr()] = (weakref.ref(p), up, coeff, count)
    # ALWAYS queue the end-of-pass settlement (it pops whatever is left and is a no-op on an empty registry): "only when the registry was empty" left a seed of a pass
    # that raised before its callbacks ran in place for ever -- no callback was queued for the next pass, whose query then consumed the stale seed (advisor, round 5)
    torch.autograd.Variable._execution_engine.queue_callback(_flush_seeds)


class _NGPLoss(torch.autograd.Function):
    """InstantNGPLoss.forward (Loss.py:18-26): mse_loss(rgb, colour) + 0.5e-6 * weight_decay_mlp() as one node -- one launch forward, one backward."""

    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda', cast_inputs=torch.float32)
    def forward(ctx, pred, target, params_d, params_c, n_d, n_total, wd_weight):
        _GRAD_SEEDS.clear()      # (see _WeightDecayNode.forward)
        pred, target = pred.contiguous(), target.contiguous()
        if pred.shape != target.shape:
            raise RuntimeError(f'instant_ngp_loss: prediction {tuple(pred.shape)} vs target {tuple(target.shape)}')
        out3 = torch.empty(3, dtype=torch.float32, device=pred.device)
        _lib.check(_lib.load().nrc_ngp_loss_forward(pred.numel(), _lib.ptr(pred), _lib.ptr(target), _lib.ptr(params_d), n_d, _lib.ptr(params_c), params_c.numel(),
                                                    1.0 / n_total, float(wd_weight), _lib.ptr(out3), _lib.stream_of(pred)), 'ngp_loss_forward')
        ctx.save_for_backward(pred, target, params_d, params_c)
        ctx.n_d, ctx.n_total, ctx.wd_weight = n_d, n_total, float(wd_weight)
        _NGPLoss.parts = out3      # (loss, mse, weight decay) of the latest call, for logging: not an output of the node
        return out3[0]

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        pred, target, params_d, params_c = ctx.saved_tensors
        up = g.detach().to(torch.float32).reshape(1).contiguous()
        grad = torch.empty_like(pred)
        _lib.check(_lib.load().nrc_mse_scaled_backward(pred.numel(), _lib.ptr(pred), _lib.ptr(target), None, _lib.ptr(up), None, _lib.ptr(grad),
                                                       _lib.stream_of(pred)), 'mse_scaled_backward')
        if ctx.wd_weight != 0.0 and params_d.requires_grad:
            _leave_seeds(((params_d, ctx.n_d), (params_c, params_c.numel())), up, ctx.wd_weight * 2.0 / ctx.n_total)
        return grad, None, None, None, None, None, None


def instant_ngp_loss(pred_rgb: torch.Tensor, target_rgb: torch.Tensor, density_net, color_net, n_density_mlp: int, n_total: int,
                     weight_decay_weight: float = 0.5e-6) -> tuple[torch.Tensor, torch.Tensor]:
    """-> (loss, parts) with loss = mse_loss(pred_rgb, target_rgb) + weight_decay_weight * weight_decay_mlp() (0-d, differentiable w.r.t. the prediction and,
    through the gradient seeds, the MLP weights) and parts = (loss, mse, weight decay) as a non-differentiable (3,) tensor for logging."""
    loss = _NGPLoss.apply(pred_rgb, target_rgb, density_net.params, color_net.params, int(n_density_mlp), int(n_total), float(weight_decay_weight))
    return loss, _NGPLoss.parts.detach()


class _QueryTrain(torch.autograd.Function):
    """query_model as ONE autograd node: the reference's sequence (normalise x, grid net, TruncExp, d*0.5+0.5, cat, colour net -- ~25
    launches forward and ~30 backward through the op-by-op modules) becomes 4 + 6 kernels.  Same arithmetic per sample; gradients reach
    both parameter vectors, not the sample positions / directions (data, as in the reference)."""

    @staticmethod
    def forward(ctx, xyzs, dirs, params_d, params_c, density_net, color_net, xyz_min, xyz_size):
        lib = _lib.load()
        m, dev = xyzs.shape[0], xyzs.device
        g = density_net.grid_cfg
        f16, f32 = torch.float16, torch.float32
        x01 = torch.empty(m, 3, dtype=f32, device=dev)
        h = torch.empty(m, 16, dtype=f16, device=dev)
        rgb16 = torch.empty(m, 4, dtype=f16, device=dev)
        sigmas = torch.empty(m, dtype=f32, device=dev)
        rgbs = torch.empty(m, 3, dtype=f32, device=dev)
        rows = (m + 31) // 32 * 32   # nrc_nwie_save_rows: the saved state is laid out in whole 32-sample tiles
        save = [torch.empty(rows, 32, dtype=f16, device=dev), torch.empty(1, rows, 64, dtype=f16, device=dev),
                torch.empty(rows, 32, dtype=f16, device=dev), torch.empty(2, rows, 64, dtype=f16, device=dev)]
        ws = torch.empty(int(lib.nrc_ngp_train_query_ws_bytes(m)), dtype=torch.uint8, device=dev)
        wd, wc = density_net._half_params(), color_net._half_params()
        _lib.check(lib.nrc_ngp_train_query_forward(
            _lib.ptr(xyzs), _lib.ptr(dirs), m, _lib.ptr(xyz_min), _lib.ptr(xyz_size), _lib.ptr(wd), _lib.ptr(wc), _lib.ptr(density_net._table16()),
            g['n_levels'], g['log2_hashmap_size'], g['base_resolution'], float(g['per_level_scale']), _lib.ptr(x01), _lib.ptr(h), _lib.ptr(rgb16),
            _lib.ptr(sigmas), _lib.ptr(rgbs), _lib.ptr(save[0]), _lib.ptr(save[1]), _lib.ptr(save[2]), _lib.ptr(save[3]), _lib.ptr(ws), None,
            _lib.stream_of(sigmas)), 'ngp_train_query_forward')
        ctx.save_for_backward(x01, h, rgb16, *save, wd, wc)
        ctx.nets = (density_net, color_net)
        return sigmas, rgbs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_sigmas, d_rgbs):
        x01, h, rgb16, sd_in, sd_acts, sc_in, sc_acts, wd, wc = ctx.saved_tensors
        density_net, color_net = ctx.nets
        lib = _lib.load()
        m, dev = x01.shape[0], x01.device
        g = density_net.grid_cfg
        # uninitialised: the call sets them (the owners of the hashed levels' slices write every entry, the rest is zeroed in one launch)
        gd = torch.empty(density_net.params.numel(), dtype=torch.float32, device=dev)
        gc = torch.empty(color_net.params.numel(), dtype=torch.float32, device=dev)
        scratch = torch.empty(int(lib.nrc_ngp_train_query_scratch_bytes(m)) if m > 0 else 16, dtype=torch.uint8, device=dev)
        common = (_lib.ptr(d_sigmas.to(torch.float32).contiguous()), _lib.ptr(d_rgbs.to(torch.float32).contiguous()), m, _lib.ptr(x01), _lib.ptr(wd), _lib.ptr(wc),
                  g['n_levels'], g['log2_hashmap_size'], g['base_resolution'], float(g['per_level_scale']), _lib.ptr(h), _lib.ptr(rgb16), _lib.ptr(sd_in),
                  _lib.ptr(sd_acts), _lib.ptr(sc_in), _lib.ptr(sc_acts), float(density_net.loss_scale), _lib.ptr(gd), _lib.ptr(gc), density_net.n_mlp_params,
                  gd.numel(), gc.numel(), _lib.ptr(scratch))
        # a weight-decay term waiting for these parameters (weight_decay_mlp): its gradient is where the buffers start instead of zero
        seed_d, seed_c = _take_seed(density_net.params), _take_seed(color_net.params)
        if seed_d is not None and seed_c is not None and seed_d[1] is seed_c[1] and seed_d[2] == seed_c[2]:
            n_clear = int(lib.nrc_ngp_train_query_clear_floats(m, g['n_levels'], g['log2_hashmap_size'], g['base_resolution'], float(g['per_level_scale']),
                                                               density_net.n_mlp_params, gd.numel()))
            _lib.check(lib.nrc_clear_seed_two(_lib.ptr(gd), n_clear, _lib.ptr(density_net.params), min(seed_d[3], n_clear), _lib.ptr(gc), gc.numel(),
                                              _lib.ptr(color_net.params), seed_c[3], _lib.ptr(seed_d[1]), seed_d[2], _lib.stream_of(gd)), 'clear_seed_two')
            _lib.check(lib.nrc_ngp_train_query_backward_cleared(*common, None, None, _lib.stream_of(gd)), 'ngp_train_query_backward_cleared')
        else:
            for seed in (seed_d, seed_c):      # only one of the two (or mismatched): back into the registry, the end-of-pass callback settles it
                if seed is not None:
                    _GRAD_SEEDS[seed[0]().data_ptr()] = seed
            _lib.check(lib.nrc_ngp_train_query_backward_set(*common, _lib.stream_of(gd)), 'ngp_train_query_backward_set')
        return None, None, gd, gc, None, None, None, None


def query_train(density_net, color_net, xyzs: torch.Tensor, dirs: torch.Tensor, xyz_min: torch.Tensor, xyz_size: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """Differentiable query_model for training batches: world positions / unit directions (M,3) f32 -> sigmas (M), rgbs (M,3) f32.
    xyz_min / xyz_size: HOST tensors (3,) f32 (the model box)."""
    xyzs = xyzs.detach().to(torch.float32).contiguous()
    dirs = dirs.detach().to(torch.float32).contiguous()
    _lib.check_input(xyzs, 'xyzs', torch.float32)
    _lib.check_input(dirs, 'dirs', torch.float32)
    if not default_layout(density_net, color_net):
        x01 = (xyzs - xyz_min.to(xyzs.device).reshape(1, 3)) / xyz_size.to(xyzs.device).reshape(1, 3)
        return query_modules(density_net, color_net, x01, dirs)
    return _QueryTrain.apply(xyzs, dirs, density_net.params, color_net.params, density_net, color_net, xyz_min, xyz_size)


class _CompositeOverBackground(torch.autograd.Function):
    """Compositing of a training batch and the pixel arithmetic behind it (Renderer.py:78-84: rgb + (1 - alpha) * bg, depth / (alpha + 1e-6))
    as ONE autograd node: two launches forward, two backward, instead of the compositing node plus ~5 + ~8 element-wise torch kernels."""

    @staticmethod
    @torch.amp.custom_fwd(cast_inputs=torch.float32, device_type='cuda')
    def forward(ctx, sigmas, rgbs, deltas, ts, rays_a, bg, cutoff):
        from . import VolumeRenderingV2 as vr
        _, opacity, depth, rgb, ws = vr.composite_train_fw(sigmas, rgbs.contiguous(), deltas, ts, rays_a, cutoff)
        n = rays_a.shape[0]
        bg = bg.to(device=opacity.device, dtype=torch.float32).contiguous()
        rgb_out, depth_out = torch.empty_like(rgb), torch.empty_like(depth)
        _lib.check(_lib.load().nrc_ngp_train_pixels_fw(n, _lib.ptr(opacity), _lib.ptr(depth), _lib.ptr(rgb), _lib.ptr(bg), _lib.ptr(rgb_out),
                                                       _lib.ptr(depth_out), _lib.stream_of(opacity)), 'ngp_train_pixels_fw')
        ctx.save_for_backward(sigmas, rgbs, ws, deltas, ts, rays_a, opacity, depth, rgb, bg)
        ctx.cutoff = float(cutoff)
        ctx.set_materialize_grads(False)   # an output the loss does not use arrives as None in backward, not as a zero-filled tensor (one launch each)
        return rgb_out, opacity, depth_out

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, g_rgb, g_alpha, g_depth):
        sigmas, rgbs, ws, deltas, ts, rays_a, opacity, depth, rgb, bg = ctx.saved_tensors
        lib = _lib.load()
        n, m = rays_a.shape[0], sigmas.shape[0]
        f32 = torch.float32
        dense = lambda g: None if g is None else g.to(f32).contiguous()
        g_rgb, g_alpha, g_depth = dense(g_rgb), dense(g_alpha), dense(g_depth)
        if g_rgb is None and g_alpha is None and g_depth is None:
            return None, None, None, None, None, None, None
        d_op, d_depth = torch.empty(n, dtype=f32, device=opacity.device), torch.empty(n, dtype=f32, device=opacity.device)
        st = _lib.stream_of(opacity)
        _lib.check(lib.nrc_ngp_train_pixels_bw(n, _lib.ptr(g_rgb), _lib.ptr(g_alpha), _lib.ptr(g_depth), _lib.ptr(opacity), _lib.ptr(depth), _lib.ptr(bg),
                                               _lib.ptr(d_op), _lib.ptr(d_depth), st), 'ngp_train_pixels_bw')
        if g_rgb is None:
            g_rgb = torch.zeros(n, 3, dtype=f32, device=opacity.device)
        d_sigmas = torch.empty(m, dtype=f32, device=opacity.device)
        d_rgbs = torch.empty(m, 3, dtype=f32, device=opacity.device)
        _lib.check(lib.nrc_composite_train_bw(_lib.ptr(d_op), _lib.ptr(d_depth), _lib.ptr(g_rgb), None, _lib.ptr(sigmas), _lib.ptr(rgbs), _lib.ptr(ws),
                                              _lib.ptr(deltas), _lib.ptr(ts), _lib.ptr(rays_a), _lib.ptr(opacity), _lib.ptr(depth), _lib.ptr(rgb), n, m,
                                              ctx.cutoff, _lib.ptr(d_sigmas), _lib.ptr(d_rgbs), st), 'composite_train_bw')
        return d_sigmas, d_rgbs, None, None, None, None, None


def composite_over_background(sigmas, rgbs, deltas, ts, rays_a, bg, T_threshold: float):
    """-> (rgb over `bg`, alpha, depth / (alpha + 1e-6)) of a training batch; differentiable w.r.t. sigmas / rgbs."""
    return _CompositeOverBackground.apply(sigmas, rgbs, deltas, ts, rays_a, bg, T_threshold)


class _ScaledMSE(torch.autograd.Function):
    """mean((pred - target)^2) and the same times the GradScaler's scale, one launch each way (nrc_mse_scaled_forward / _backward)."""

    @staticmethod
    @torch.amp.custom_fwd(cast_inputs=torch.float32, device_type='cuda')
    def forward(ctx, pred, target, scale):
        pred, target = pred.contiguous(), target.contiguous()
        _lib.check_input(pred, 'pred', torch.float32)
        _lib.check_input(target, 'target', torch.float32)
        if pred.shape != target.shape:
            raise RuntimeError(f'scaled_mse_loss: pred {tuple(pred.shape)} vs target {tuple(target.shape)}')
        out2 = torch.empty(2, dtype=torch.float32, device=pred.device)
        _lib.check(_lib.load().nrc_mse_scaled_forward(pred.numel(), _lib.ptr(pred), _lib.ptr(target), _lib.ptr(scale), _lib.ptr(out2), _lib.stream_of(pred)),
                   'mse_scaled_forward')
        ctx.save_for_backward(pred, target, scale)
        ctx.set_materialize_grads(False)
        return out2[0], out2[1]

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, g_loss, g_scaled):
        pred, target, scale = ctx.saved_tensors
        if g_loss is None and g_scaled is None:
            return None, None, None
        f = lambda g: None if g is None else g.to(torch.float32).contiguous()
        g_loss, g_scaled = f(g_loss), f(g_scaled)
        grad = torch.empty_like(pred)
        _lib.check(_lib.load().nrc_mse_scaled_backward(pred.numel(), _lib.ptr(pred), _lib.ptr(target), _lib.ptr(scale), _lib.ptr(g_loss), _lib.ptr(g_scaled),
                                                       _lib.ptr(grad), _lib.stream_of(pred)), 'mse_scaled_backward')
        return grad, None, None


def scaled_mse_loss(pred: torch.Tensor, target: torch.Tensor, scale: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """-> (mse_loss(pred, target), the same * scale) as 0-d tensors; `scale`: the GradScaler's device scalar (Trainer.py:87-89 computes the two with
    mse_loss + scaler.scale: seven small launches forward + backward where this takes two).  Differentiable w.r.t. pred through either output."""
    return _ScaledMSE.apply(pred, target, scale.detach().reshape(1))


def gather_ray_batch(ids: torch.Tensor, origin: torch.Tensor, view_direction: torch.Tensor, rgb: torch.Tensor | None = None,
                     alpha: torch.Tensor | None = None) -> dict[str, torch.Tensor]:
    """pool[ids] for every field of a resident ray pool in one launch (DatasetSamplers.py:53-66 does one fancy-index gather per field)."""
    _lib.check_input(ids, 'ids', torch.int64)
    n, n_pool, dev = ids.shape[0], origin.shape[0], ids.device
    fields = {'origin': (origin, 3), 'view_direction': (view_direction, 3), 'rgb': (rgb, 3), 'alpha': (alpha, 1)}
    out = {}
    for k, (v, w) in fields.items():
        if v is None:
            continue
        _lib.check_input(v, k, torch.float32)
        if v.shape[0] != n_pool or v.numel() != n_pool * w:
            raise RuntimeError(f'gather_ray_batch: {k} has shape {tuple(v.shape)}, expected ({n_pool}, {w})')
        out[k] = torch.empty((n, 3) if w == 3 else (n,), dtype=torch.float32, device=dev)
    p = lambda k: _lib.ptr(fields[k][0]) if fields[k][0] is not None else None
    q = lambda k: _lib.ptr(out[k]) if k in out else None
    _lib.check(_lib.load().nrc_gather_ray_batch(_lib.ptr(ids), n, n_pool, p('origin'), p('view_direction'), p('rgb'), p('alpha'), q('origin'),
                                                q('view_direction'), q('rgb'), q('alpha'), _lib.stream_of(ids)), 'gather_ray_batch')
    return out
