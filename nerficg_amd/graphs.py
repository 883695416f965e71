"""nerficg_amd.graphs -- whole optimisation iterations recorded once in a HIP graph and replayed.

The reference's trainers issue an iteration op by op (src/Methods/InstantNGP/Trainer.py:79-94, src/Methods/GaussianSplatting/Trainer.py): on an
MI355X the InstantNGP iteration is ~1.0 ms of kernels behind ~90 launches and one host read, i.e. bound by the host.  Every entry point of
the C ABI is asynchronous on the caller's stream, so the whole iteration can be captured -- what has to change is the few places where the
op-by-op code sizes a buffer from a device count:

  * ray marching: `InstantNGPRenderer.sample_capacity` fixes the number of sample rows (nrc_raymarching_train_cap makes the unused tail
    inert and cuts rays that would cross it; `counter[0] > capacity` reports the latter);
  * rasterizer: `diff_gaussian_rasterization.fixed_capacity(instances, spans)` fixes the length of the per-tile lists and of the row-span
    workspace (the count comes back in `num_rendered`, nothing is read on the host), and the camera travels as a device block;
  * FusedAdam(capturable=True): step counter and learning rate live on the device.

`GraphedIteration` is the generic piece (static input buffers, eager warm-up calls, capture, replay); `instant_ngp_iteration` and
`gaussian_splatting_step` build the two iterations of the reference on top of it.  The HIP graph is torch.cuda.CUDAGraph (hipGraph on ROCm):
kernels launched through ctypes on torch's current stream are recorded like torch's own.
"""
from __future__ import annotations

import weakref
from typing import Callable

import torch

__all__ = ['GraphedIteration', 'instant_ngp_iteration', 'gaussian_splatting_step']


class GraphedIteration:
    """body(**inputs) -> dict of tensors, run on fixed input buffers.  The first `eager_calls` calls execute `body` as it is (they create
    lazily built state: optimizer moments, GradScaler scalars, compute copies); the next call records it; every later call copies the
    new inputs into the buffers and replays.  Inputs must keep shape and dtype; outputs are the same tensor objects on every call
    (overwritten by the next one).  `before_replay()` runs ahead of every replay (host-side hyper-parameter pushes).
    `parameters`: the leaf tensors `body` differentiates.  Autograd pins a leaf's gradient accumulator to the stream that first used it; the
    eager calls run on the caller's stream and the recording on torch's capture stream, so the accumulators are dropped before the capture
    (requires_grad off / on) -- otherwise the recorded backward hops over to the eager stream for every accumulation (torch warns about
    exactly this) and the graph carries cross-stream edges it does not need.

    Two findings on ROCm 7.2 that shaped this (tools/exp_gs_graph.py, tools/exp_graph_ingp.py): (1) graph MEMSET nodes -- a hipMemsetAsync
    inside the capture -- let a replay finish after the launch stream considered it done, so back-to-back replays overran their own
    predecessor (GPU memory faults from half-updated index buffers); the library clears memory with a kernel of its own (nrc_zero_async,
    csrc/common.h) and recorded graphs hold kernel nodes only.  (2) Running the eager calls on a private side stream that is then used for
    the capture produced the same faults; eager calls stay on the caller's stream."""

    def __init__(self, body: Callable[..., dict], example_inputs: dict[str, torch.Tensor], eager_calls: int = 1,
                 before_replay: Callable[[], None] | None = None, parameters=None) -> None:
        if eager_calls < 1:
            raise ValueError('at least one eager call is needed to build lazily created state before the capture')
        self.body = body
        self.inputs = {k: v.detach().clone() for k, v in example_inputs.items()}
        for k, v in self.inputs.items():
            if not v.is_cuda:
                raise RuntimeError(f'GraphedIteration: the buffer of input {k!r} must live on the GPU (later calls may pass host tensors)')
        self.eager_calls = eager_calls
        self.before_replay = before_replay
        self.parameters = None if parameters is None else (parameters if callable(parameters) else list(parameters))
        self.calls = 0
        self.graph: torch.cuda.CUDAGraph | None = None
        self.outputs: dict[str, torch.Tensor] | None = None
        self.on_close: tuple = ()

    def _load(self, inputs: dict[str, torch.Tensor]) -> None:
        if inputs.keys() != self.inputs.keys():
            raise KeyError(f'expected inputs {sorted(self.inputs)}, got {sorted(inputs)}')
        for k, v in inputs.items():
            buf = self.inputs[k]
            if v.shape != buf.shape or v.dtype != buf.dtype:
                raise RuntimeError(f'input {k!r}: {tuple(v.shape)} {v.dtype} does not match the recorded {tuple(buf.shape)} {buf.dtype}')
            if v.data_ptr() != buf.data_ptr():
                buf.copy_(v, non_blocking=True)

    def _record(self) -> None:
        leaves = self.parameters() if callable(self.parameters) else (self.parameters or [])
        for p in leaves:
            if p.requires_grad and p.is_leaf:
                p.grad = None
                p.requires_grad_(False).requires_grad_(True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):   # records, does not execute
            self.outputs = self.body(**self.inputs)

    def __call__(self, **inputs: torch.Tensor) -> dict[str, torch.Tensor]:
        self._load(inputs)
        self.calls += 1
        if self.calls <= self.eager_calls:
            return self.body(**self.inputs)
        if self.graph is None:
            self._record()
        if self.before_replay is not None:
            self.before_replay()
        self.graph.replay()
        return self.outputs

    @property
    def recorded(self) -> bool:
        return self.graph is not None

    def close(self) -> None:
        """Drop the recording and run the `on_close` hooks (state an iteration installed outside itself, e.g. the folded weight decay)."""
        self.graph = None
        self.outputs = None
        for hook in self.on_close:
            hook()
        self.on_close = ()

    def __del__(self):
        try:
            self.close()
        except Exception:   # interpreter shutdown
            pass


def instant_ngp_iteration(model, renderer, optimizer, scaler, camera, n_rays: int, sample_capacity: int, loss_fn: Callable | None = None,
                          with_alpha: bool = False, ray_pool: dict[str, torch.Tensor] | None = None, eager_calls: int = 1,
                          fold_weight_decay: bool = False) -> GraphedIteration:
    """The training iteration of src/Methods/InstantNGP/Trainer.py:79-94 for batches of `n_rays` rays as a GraphedIteration:
    call(origin=(n,3), view_direction=(n,3), rgb=(n,3)[, alpha=(n,)]) -> {'loss', 'rm_samples', 'sample_overflow'}; with
    `ray_pool` = {'origin', 'view_direction', 'rgb'[, 'alpha']} (all rays of the training set resident on the GPU, like the reference's
    RayPoolSampler) the call is call(ids=(n,) int64) and the gather of the batch is part of the recording.
    loss_fn(outputs, rgb, alpha | None, bg) defaults to InstantNGPLoss (MSE on the colours over the random background + 0.5e-6 * mean squared
    MLP weight).  `sample_capacity` rows are marched / queried per iteration whatever the occupancy; 'sample_overflow' (device, > 0 when rays
    were cut) is for the caller to look at every now and then -- e.g. where the reference reads rm_samples to adapt its batch size.
    The optimizer must be FusedAdam(capturable=True).
    fold_weight_decay (default loss only): the 0.5e-6 * mean(w^2) term leaves the loss and its gradient, 1e-6 / n_mlp * w on the MLP weights, is
    added inside the Adam kernel (FusedAdam.set_l2_slice) -- the same update without the dense 12 M-element gradients autograd builds for a
    term that touches 10 240 weights (~16 launches less per iteration); the reported loss is then the colour term alone.  The slices live in
    the optimizer while this iteration object lives: `close()` (or dropping the object) removes them, so a later op-by-op step with the default
    loss -- which contains the term -- does not decay twice."""
    if not getattr(optimizer, 'capturable', False):
        raise RuntimeError('instant_ngp_iteration: build the optimizer as FusedAdam(..., capturable=True)')
    dev = model.center.device
    if fold_weight_decay:
        if loss_fn is not None:
            raise ValueError('fold_weight_decay replaces the weight-decay term of the DEFAULT loss; a custom loss_fn decides for itself')
        coeff = 1e-6 / model.n_mlp_params
        installed = [(p, optimizer.set_l2_slice(p, n, coeff)) for p, n in
                     ((model.encoding_xyz.params, model.n_params_encoding_mlp),
                      (model.color_mlp_with_encoding.params, model.color_mlp_with_encoding.params.numel()))]

    def default_loss(out, rgb, alpha, bg):
        target = rgb if alpha is None else rgb * alpha[:, None] + (1 - alpha)[:, None] * bg
        colour = torch.nn.functional.mse_loss(out['rgb'].float(), target)
        return colour if fold_weight_decay else colour + 0.5e-6 * model.weight_decay_mlp()

    criterion = loss_fn or default_loss
    # The default colour loss with the weight decay folded away is ONE expression: mean squared error, times the scaler's scale.  With a scaler that
    # exposes its device scalar (nerficg_amd.amp.GradScaler.scale_tensor) it runs as one launch forward and one backward
    # (nerficg_amd.ngp.scaled_mse_loss) instead of mse_loss + scaler.scale(loss) + their autograd nodes (seven), and backward starts from a
    # resident one instead of a freshly filled tensor.
    fused_loss = loss_fn is None and fold_weight_decay and hasattr(scaler, 'scale_tensor') and scaler.is_enabled()
    one = torch.ones((), dtype=torch.float32, device=dev)

    def iteration(origin, view_direction, rgb, alpha=None):
        from .ngp import scaled_mse_loss
        renderer.sample_capacity = int(sample_capacity)
        try:
            with torch.amp.autocast('cuda'):
                bg = torch.rand(3, device=dev)
                out = renderer.render_rays(origin, view_direction, camera, train_mode=True, custom_bg_color=bg)
                if fused_loss and alpha is None:
                    loss, scaled = scaled_mse_loss(out['rgb'], rgb, scaler.scale_tensor(dev))
                else:
                    loss = criterion(out, rgb, alpha, bg)
                    scaled = None
            if scaled is not None:
                torch.autograd.backward(scaled, grad_tensors=one)
            else:
                scaler.scale(loss).backward()
            scaler.step(optimizer)
            scaler.update()
            optimizer.zero_grad()
        finally:
            renderer.sample_capacity = None
        marched = out['rm_samples']
        cut = out['sample_overflow'] if 'sample_overflow' in out else (marched - int(sample_capacity)).clamp_(min=0)
        return {'loss': loss.detach(), 'rm_samples': marched, 'sample_overflow': cut}

    if ray_pool is not None:
        pool = {k: v.contiguous() for k, v in ray_pool.items()}
        one_launch = set(pool) <= {'origin', 'view_direction', 'rgb', 'alpha'} and {'origin', 'view_direction'} <= set(pool) and \
            all(v.dtype == torch.float32 and v.is_cuda for v in pool.values())

        def body(ids):
            if one_launch:   # every field of the batch in one gather launch
                from .ngp import gather_ray_batch
                return iteration(**gather_ray_batch(ids, pool['origin'], pool['view_direction'], pool.get('rgb'), pool.get('alpha')))
            return iteration(**{k: v[ids] for k, v in pool.items()})
        example = {'ids': torch.zeros(n_rays, dtype=torch.int64, device=dev)}
    else:
        body = iteration
        example = {'origin': torch.zeros(n_rays, 3, device=dev), 'view_direction': torch.zeros(n_rays, 3, device=dev), 'rgb': torch.zeros(n_rays, 3, device=dev)}
        example['view_direction'][:, 2] = 1.0
        if with_alpha:
            example['alpha'] = torch.zeros(n_rays, device=dev)
    it = GraphedIteration(body, example, eager_calls=eager_calls, before_replay=optimizer.sync_hyperparameters, parameters=model.parameters)
    if fold_weight_decay:
        # only what THIS iteration installed: a successor built on the same optimizer before this object goes (the usual rebuild with a new
        # n_rays) has replaced the entries, and its slices must survive this one's close() / __del__
        def drop_own_slices(opt_ref=weakref.ref(optimizer), installed=[(weakref.ref(p), tok) for p, tok in installed]):
            opt = opt_ref()
            for p_ref, tok in installed:
                p = p_ref()
                if opt is not None and p is not None:
                    opt.remove_l2_slice(p, tok)
        it.on_close = (drop_own_slices,)
    return it


def gaussian_splatting_step(gaussians, camera, instance_capacity: int, span_capacity: int = 0, loss_fn: Callable | None = None,
                            densification_stats: bool = True, eager_calls: int = 1) -> GraphedIteration:
    """The optimisation step of src/Methods/GaussianSplatting/Trainer.py (render_image_training -> 0.8 L1 + 0.2 DSSIM -> backward ->
    densification statistics -> FusedAdam) as a GraphedIteration: call(c2w=(4,4) f32, target=(3,H,W) f32) -> {'loss', 'radii', 'counts'}.
    The pose is a device tensor (make_raster_settings assembles the camera on the device), the rasterizer runs with fixed list / span
    capacities ('counts' = the DEVICE int64[2] the frame needed: compare with the capacities every now and then), and the optimizer must
    have been built with training_setup(capturable=True); `update_learning_rate` keeps working (the new rate is pushed to the device
    ahead of every replay).  Whatever replaces the parameter tensors -- densify_and_prune, reset_opacities -- invalidates the recording:
    build a new step afterwards (the reference densifies every 100 iterations; a capture costs about as much as two eager steps)."""
    from .diff_gaussian_rasterization import fixed_capacity, last_counts
    from .gaussian_splatting import render_image_training, training_loss
    optimizer = gaussians.optimizer
    if not getattr(optimizer, 'capturable', False):
        raise RuntimeError('gaussian_splatting_step: build the optimizer with Gaussians.training_setup(capturable=True)')
    criterion = loss_fn or training_loss
    dev = gaussians.get_positions.device

    def body(c2w, target):
        with fixed_capacity(int(instance_capacity), int(span_capacity)):
            out = render_image_training(gaussians, camera, c2w)
            loss = criterion(out['rgb'], target)
            loss.backward()
        counts = last_counts()
        if densification_stats:
            with torch.no_grad():
                gaussians.add_densification_stats(out['viewspace_points'], out['radii'])
        optimizer.step()
        optimizer.zero_grad()
        return {'loss': loss.detach(), 'radii': out['radii'], 'counts': counts}

    example = {'c2w': torch.eye(4, device=dev), 'target': torch.zeros(3, camera.height, camera.width, device=dev)}
    return GraphedIteration(body, example, eager_calls=eager_calls, before_replay=optimizer.sync_hyperparameters,
                            parameters=lambda: [grp['params'][0] for grp in optimizer.param_groups])
