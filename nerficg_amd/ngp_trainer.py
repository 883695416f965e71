"""nerficg_amd.ngp_trainer -- the InstantNGP training iteration on device-resident state (include/nerficg_hip.h group 13).

The reference's iteration (src/Methods/InstantNGP/Trainer.py:79-94) is: draw a batch from the ray pool -> random background -> render_rays(train) ->
MSE + weight decay -> GradScaler.scale / backward -> GradScaler.step(FusedAdam) -> GradScaler.update -> zero_grad.  Through the drop-in modules that
is ~90 launches and a host read; recorded op by op in a HIP graph (nerficg_amd.graphs.instant_ngp_iteration) 33 launches, 17 of them 4-5 us
element-wise or fill kernels.  `FusedTrainingIteration` issues the same arithmetic as FOUR library calls / 13 launches on buffers that exist once:

    nrc_ngp_train_march                   batch out of the resident pool + box clipping + jitter / background draws + march (2 launches)
    nrc_ngp_train_query_forward           encode + both networks                                                          (3)
    nrc_ngp_train_loss                    compositing, pixel, MSE, x scale, and back to dL/dsigma, dL/drgb; clears what the backward adds into  (1)
    nrc_ngp_train_backward_step           both networks backward (they flag inf / NaN), step counter + scale update, hash-grid backward whose slice
                                          owners apply Adam to the hashed levels themselves, Adam on the rest                 (7)
    (fused_step=False -- what data-parallel ranks take, who need the gradients on the wire:
     nrc_ngp_train_query_backward_cleared (5) + nrc_amp_adam_step: inf / NaN check + step counter + scale update, Adam       (2))

and -- since the march of batch i + 1 depends on nothing iteration i updates (rays, the occupancy bitfield, a counter-based generator) -- runs that
march on a forked stream NEXT TO iteration i's backward pass and Adam (`prefetch`): two sets of batch buffers, consumed alternately.  The caller says
when the bitfield is about to change (`step(prefetch=False)` before an occupancy update), so every batch is marched against the bitfield the
reference's loop would have used.  The calls only enqueue (85-110 us of host time per iteration for ~0.35-0.4 ms of kernels), so the iteration
needs no recording; with `graph=True` each (buffer set, inline / prefetched march, prefetch on / off) combination is recorded once in a HIP graph
and replayed anyway (measured SLOWER than the eager calls on ROCm 7.2: 0.478 against 0.447 ms -- a replay pays a few microseconds per kernel node).

Parity: tests/test_gpu_ngp_trainer.py -- the same batches, backgrounds and jitter through this class (eager and recorded, with and without
prefetch) and through the op-by-op loop leave the same parameters within the op-by-op loop's own run-to-run spread.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

__all__ = ['FusedTrainingIteration']


class _Batch:
    """The buffers one marched batch lives in (two of these when the next batch is marched ahead)."""

    def __init__(self, n_rays: int, n_samples: int, dev) -> None:
        f32 = dict(dtype=torch.float32, device=dev)
        self.rays_o, self.rays_d, self.target = (torch.zeros(n_rays, 3, **f32) for _ in range(3))
        self.hits_t = torch.zeros(n_rays, 2, **f32)
        self.bg = torch.zeros(3, **f32)
        self.rays_a = torch.zeros(n_rays, 3, dtype=torch.int64, device=dev)
        self.counter = torch.zeros(2, dtype=torch.int32, device=dev)
        self.overflow = torch.zeros(1, dtype=torch.int64, device=dev)
        self.xyzs, self.dirs = torch.zeros(n_samples, 3, **f32), torch.zeros(n_samples, 3, **f32)
        self.deltas, self.ts = torch.zeros(n_samples, **f32), torch.zeros(n_samples, **f32)

    def tensors(self):
        return (self.rays_o, self.rays_d, self.target, self.hits_t, self.bg, self.rays_a, self.counter, self.overflow, self.xyzs, self.dirs, self.deltas, self.ts)


class FusedTrainingIteration:
    """call() -> {'loss', 'rm_samples', 'sample_overflow', 'rgb', 'alpha', 'bg'} (device tensors, overwritten by later calls).

    model / renderer: nerficg_amd.instant_ngp.  optimizer: FusedAdam(capturable=True) over model.parameters() (ONE group holding both parameter
    vectors, as Trainer.py:35 builds it); its moments, step counter and learning-rate scalar are used in place, so checkpoints and schedulers keep
    working (`optimizer.param_groups[0]['lr']` is pushed to the device ahead of every call).  scaler: torch.amp.GradScaler or nerficg_amd.amp.GradScaler
    (its scale / growth tracker are updated on the device by the same rule), or None.
    ray_pool: {'origin', 'view_direction', 'rgb'[, 'alpha']} f32 on the device (dataset.get_all_rays()).  Batches: `order` (i64 on the device: the
    sampler's permutation) is consumed front to back from a device cursor, `ray_capacity` rows per call of which `n_rays` are live
    (`set_batch_size`); `rewind(order)` installs the next epoch's permutation.  Or pass `ids=` (i64, ray_capacity) to a call.
    weight_decay: coefficient of Loss.py:15's 0.5e-6 * mean(w^2) over the MLP weights, applied as FusedAdam's L2 slice (gradient 2 * weight_decay / n * w
    inside the Adam kernel); the reported loss is the colour term."""

    T_THRESHOLD = 1e-4

    def __init__(self, model, renderer, optimizer, scaler, camera, ray_pool: dict, ray_capacity: int, sample_capacity: int, order: torch.Tensor | None = None,
                 seed: int = 0, weight_decay: float = 0.5e-6, prefetch: bool = True, graph: bool = False, ray_offset: int | None = None, fused_step: bool = True,
                 fork_dense_levels: bool = True, data_parallel: bool | None = None, sharded: bool | None = None, prefetch_at: str | None = None,
                 dp_timing: bool = False, wire_dtype: torch.dtype = torch.float32) -> None:
        if not getattr(optimizer, 'capturable', False):
            raise RuntimeError('FusedTrainingIteration: build the optimizer as FusedAdam(..., capturable=True)')
        if len(optimizer.param_groups) != 1:
            raise RuntimeError('FusedTrainingIteration: one parameter group holding both parameter vectors is expected (Trainer.py:35)')
        lib = _lib.load()
        from .ngp import default_layout
        if not default_layout(model.encoding_xyz, model.color_mlp_with_encoding):
            raise RuntimeError('FusedTrainingIteration is built for the default HASHGRID_N_LEVELS = 16, HASHGRID_N_FEATURES_PER_LEVEL = 2, DIR_SH_ENCODING_DEGREE = 4; '
                               'other configurations train through the op-by-op loop (renderer.render_rays + InstantNGPLoss + FusedAdam)')
        self.model, self.renderer, self.optimizer, self.scaler, self.camera = model, renderer, optimizer, scaler, camera
        # Data parallel (BASELINE configs[3]): every rank holds the whole pool and the same global order / seed; rank r marches rows
        # [r n, (r + 1) n) of every global batch of W n rays (parallel.rank_batch_order) with the jitter of their GLOBAL indices and the iteration's one
        # background colour, so W ranks do the arithmetic one rank would do on W n rays; the two gradient vectors live in ONE flat buffer that is
        # averaged by one reduce-scatter + all-gather over RCCL (parallel.allreduce_flat) between the backward pass and the step -- an inf / NaN reaches
        # every rank through the sum, so all replicas skip the same steps.  The gradients have to exist on the wire: fused_step and recording are off.
        # `sharded` (round 6, the default whenever the table divides by the world size): the step of SURVEY 8(e) -- see _update_sharded.
        from . import parallel
        self.rank, self.world = parallel.world_info()
        self.data_parallel = (self.world > 1) if data_parallel is None else bool(data_parallel)
        if self.data_parallel:
            fused_step, graph = False, False
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_backend() == 'gloo':
                # gloo's device all-reduce waits for the whole device, the side stream's march included, and the two then take turns:
                # 129 ms per iteration against 14 (two ranks on one GPU, tools/exp_fused_dp.py).  RCCL's collectives are stream-ordered.
                prefetch = False
            if ray_offset is None:
                ray_offset = self.rank * int(ray_capacity)
            if order is not None:
                order = parallel.rank_batch_order(order.to(model.center.device), int(ray_capacity), self.rank, self.world)
        ray_offset = 0 if ray_offset is None else ray_offset
        self.dev = dev = model.center.device
        self.n_cap, self.m_cap = int(ray_capacity), int(sample_capacity)
        self.ray_offset = int(ray_offset)
        self.use_graph, self.prefetch_default, self.fused_step = bool(graph), bool(prefetch), bool(fused_step)
        self._gb_side = torch.cuda.Stream(device=dev) if fork_dense_levels else None
        self.pool = {k: ray_pool[k].contiguous() for k in ('origin', 'view_direction', 'rgb', 'alpha') if ray_pool.get(k) is not None}
        for k, v in self.pool.items():
            _lib.check_input(v, k, torch.float32)
        self.n_pool = self.pool['origin'].shape[0]
        self.order = None
        self.cursor = torch.zeros(1, dtype=torch.int64, device=dev)
        self._cursor_host = 0
        if order is not None:
            self.rewind(order)
        self.n_rays = self.n_cap
        self.n_rays_dev = torch.full((1,), self.n_cap, dtype=torch.int32, device=dev)
        self.rng = torch.tensor([int(seed), 0], dtype=torch.int64, device=dev)
        self.ids = torch.zeros(self.n_cap, dtype=torch.int64, device=dev)
        self.bg_in, self.noise_in = torch.zeros(3, device=dev), torch.zeros(self.n_cap, device=dev)
        self.sets = [_Batch(self.n_cap, self.m_cap, dev), _Batch(self.n_cap, self.m_cap, dev) if prefetch else None]
        ws_bytes = int(lib.nrc_ngp_train_march_ws_bytes(self.n_cap, renderer.MAX_SAMPLES))
        if ws_bytes < 0:
            raise RuntimeError(f'FusedTrainingIteration: ray_capacity {self.n_cap} is outside the wave-per-ray march (1 .. 32768)')
        self.march_ws = [torch.zeros(ws_bytes, dtype=torch.uint8, device=dev) for s in self.sets if s is not None]
        # the networks' forward state / outputs, the sample gradients, the parameter gradients: allocated once
        m, f16, f32 = self.m_cap, torch.float16, torch.float32
        rows = int(lib.nrc_nwie_save_rows(m))
        e = lambda *shape, dtype=f32: torch.empty(*shape, dtype=dtype, device=dev)
        self.x01, self.h, self.rgb16 = e(m, 3), e(m, 16, dtype=f16), e(m, 4, dtype=f16)
        self.sigmas, self.rgbs = e(m), e(m, 3)
        self.save = [e(rows, 32, dtype=f16), e(1, rows, 64, dtype=f16), e(rows, 32, dtype=f16), e(2, rows, 64, dtype=f16)]
        self.fwd_ws = e(int(lib.nrc_ngp_train_query_ws_bytes(m)), dtype=torch.uint8)
        self.d_sigmas, self.d_rgbs = e(m), e(m, 3)
        self.ray_rgb, self.ray_alpha = torch.zeros(self.n_cap, 3, device=dev), torch.zeros(self.n_cap, device=dev)
        self.loss2 = torch.zeros(2, device=dev)
        self.loss_ws = torch.zeros(int(lib.nrc_ngp_train_loss_ws_bytes(self.n_cap)), dtype=torch.uint8, device=dev)
        dn, cn = model.encoding_xyz, model.color_mlp_with_encoding
        # one buffer, in the order a data-parallel iteration finishes its pieces: [colour MLP | aux | density MLP | hash table] (parallel.ShardedStepLayout)
        self.layout = L = parallel.ShardedStepLayout(cn.params.numel(), dn.n_mlp_params, dn.params.numel() - dn.n_mlp_params, self.rank, self.world)
        self.grads = torch.zeros(L.total, dtype=f32, device=dev)
        self.gc, self.aux, self.gd = self.grads[:L.off_aux], self.grads[L.off_aux:L.off_density], self.grads[L.off_density:]
        self.sharded = self.data_parallel and L.sharded and (sharded is None or bool(sharded))
        if sharded and not self.sharded:
            raise RuntimeError(f'FusedTrainingIteration(sharded=True): needs data_parallel and a table of {L.n_table} floats that divides by the world size {self.world}')
        self.prefetch_at = prefetch_at or ('collective' if self.sharded else 'forward')
        if self.prefetch_at not in ('forward', 'collective') or (self.prefetch_at == 'collective' and not self.sharded):
            raise ValueError("prefetch_at: 'forward' (march the next batch beside this iteration's forward / backward pass) or, sharded step only, 'collective'")
        # wire_dtype = torch.float16 (sharded step only, off by default): the table gradient crosses the wire as saturating fp16 -- 21.3 instead of 42.7 MB per GPU at
        # N = 8 -- and is summed over the ranks in fp16 (tiny-cuda-nn's own gradient precision under the same loss scale); produced and applied in f32
        if wire_dtype not in (torch.float32, torch.float16) or (wire_dtype != torch.float32 and not self.sharded):
            raise ValueError('wire_dtype: torch.float32, or torch.float16 with the sharded data-parallel step')
        self.wire = torch.empty(L.n_table, dtype=torch.float16, device=dev) if wire_dtype == torch.float16 else None
        self.wire_saturated = torch.zeros(1, dtype=torch.int64, device=dev) if self.wire is not None else None
        self._comm = torch.cuda.Stream(device=dev) if self.sharded else None
        self._ev = [torch.cuda.Event(), torch.cuda.Event()] if self.sharded else None
        self._master_stale = False       # sharded step: the fp32 master / moments of the table are current in this rank's shard only (gather_state())
        self.dp_timing, self._dp_events = bool(dp_timing), []
        self.bwd_scratch = e(int(lib.nrc_ngp_train_query_scratch_bytes(m)), dtype=torch.uint8)
        g = dn.grid_cfg
        self.n_clear = int(lib.nrc_ngp_train_query_clear_floats(m, g['n_levels'], g['log2_hashmap_size'], g['base_resolution'], float(g['per_level_scale']),
                                                                dn.n_mlp_params, dn.params.numel()))
        if self.n_clear < 0:
            raise RuntimeError('nrc_ngp_train_query_clear_floats refused the grid configuration')
        self.amp_state = torch.zeros(4, device=dev)
        self.amp_ticket = torch.zeros(17 * 16, dtype=torch.int32, device=dev)    # NRC_TICKET_WORDS
        n_mlp = model.n_mlp_params
        self.l2 = ((2.0 * float(weight_decay) / n_mlp, model.n_params_encoding_mlp), (2.0 * float(weight_decay) / n_mlp, cn.params.numel())) if weight_decay else ((0.0, 0), (0.0, 0))
        self._side = torch.cuda.Stream(device=dev) if prefetch else None
        self._ready: int | None = None      # index of the set that holds the batch marched ahead
        self._graphs: dict = {}
        self._signature = None
        self._explicit = (False, False, False)
        self.calls = 0

    # ------------------------------------------------------------------------------------------------ sampling state
    def rewind(self, order: torch.Tensor) -> None:
        """Install a (new) sampling order -- RandomSequentialSampler's permutation, Samplers/utils.py:30-33 -- and put the cursor at its start.
        A batch marched ahead from the old order is dropped.  (Data parallel: pass this rank's order, parallel.rank_batch_order(global_order, n).)"""
        order = order.to(device=self.dev, dtype=torch.int64).contiguous()
        if self.order is not None and self.order.shape == order.shape:
            self.order.copy_(order)       # recorded iterations read this buffer
        else:
            self.order = order.clone()
            self._graphs = {}
        self.cursor.zero_()
        self._cursor_host = 0
        self._ready = None

    def set_batch_size(self, n_rays: int) -> None:
        """Live rays per call from now on (<= ray_capacity): what Trainer.update_batch_size (:70-75) changes every 16 iterations; recorded
        iterations read it from the device.  A batch already marched ahead keeps its size."""
        n = int(n_rays)
        if getattr(self, 'data_parallel', False) and n != self.n_cap:
            raise RuntimeError('data-parallel ranks consume fixed slices of a global batch: the batch size is the ray capacity')
        if not 1 <= n <= self.n_cap:
            raise ValueError(f'batch size {n} outside 1 .. ray_capacity = {self.n_cap}')
        if n != self.n_rays:
            self.n_rays = n
            self.n_rays_dev.fill_(n)

    # ------------------------------------------------------------------------------------------------ the five calls
    def _state_tensors(self, check: bool = True):
        """Everything whose ADDRESS a recorded iteration holds: when one of them is replaced (load_state_dict, a new occupancy buffer, ...) the
        recordings are dropped and made again."""
        opt, m = self.optimizer, self.model
        group = opt.param_groups[0]
        dn, cn = m.encoding_xyz, m.color_mlp_with_encoding
        if check and self._master_stale and dn._half_key != dn._key_of(dn.params):
            # the fp16 table would be rebuilt from an fp32 master that is current in this rank's shard only
            raise RuntimeError('FusedTrainingIteration (sharded step): the density parameters were written outside the trainer while their fp32 master was '
                               'sharded over the ranks -- call gather_state() on every rank before editing / loading parameters')
        for p in (dn.params, cn.params):
            st = opt.state[p]
            if len(st) == 0:
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        bc, _skipped, step_dev, lr_dev = opt._device_scalars(0, group, self.dev)
        scale = tracker = None
        if self.scaler is not None and self.scaler.is_enabled():
            if self.scaler._scale is None:
                self.scaler._lazy_init_scale_growth_tracker(self.dev)
            scale, tracker = self.scaler._scale, self.scaler._growth_tracker
        return dict(pd=dn.params, pc=cn.params, hd=dn._half_for_optimizer(dn.params), hc=cn._half_for_optimizer(cn.params),
                    md=opt.state[dn.params]['exp_avg'], vd=opt.state[dn.params]['exp_avg_sq'], mc=opt.state[cn.params]['exp_avg'],
                    vc=opt.state[cn.params]['exp_avg_sq'], bc=bc, step=step_dev, lr=lr_dev, scale=scale, tracker=tracker, bitfield=m.occupancy_bitfield)

    def _march(self, k: int, st: dict, explicit) -> None:
        b, m, r, cam = self.sets[k], self.model, self.renderer, self.camera
        use_ids, use_bg, use_noise = explicit
        center, half = r._scene_box()
        p = _lib.ptr
        _lib.check(_lib.load().nrc_ngp_train_march(
            p(self.ids) if use_ids else None, p(self.order), p(self.cursor), p(self.n_rays_dev), self.n_cap, self.n_pool, self.ray_offset,
            p(self.pool['origin']), p(self.pool['view_direction']), p(self.pool.get('rgb')), p(self.pool.get('alpha')), p(center), p(half),
            float(cam.near_plane), float(cam.far_plane), p(st['bitfield']), m.cascades, float(m.SCALE), 1 / 256 if r.EXPONENTIAL_STEPS else 0.0, m.RESOLUTION,
            r.MAX_SAMPLES, p(self.rng), p(self.bg_in) if use_bg else None, p(self.noise_in) if use_noise else None, self.m_cap, p(b.rays_o), p(b.rays_d),
            p(b.hits_t), p(b.target), p(b.bg), p(b.rays_a), p(b.counter), p(b.xyzs), p(b.dirs), p(b.deltas), p(b.ts), p(b.overflow), p(self.march_ws[k]),
            _lib.stream_of(b.rays_o)), 'ngp_train_march')

    def _update(self, k: int, st: dict, fork_prefetch=None) -> None:
        b, m, lib, p = self.sets[k], self.model, _lib.load(), _lib.ptr
        dn, cn = m.encoding_xyz, m.color_mlp_with_encoding
        g = dn.grid_cfg
        grid = (g['n_levels'], g['log2_hashmap_size'], g['base_resolution'], float(g['per_level_scale']))
        mn, sz = self.renderer._box()
        stream = _lib.stream_of(b.rays_o)
        M = self.m_cap
        _lib.check(lib.nrc_ngp_train_query_forward(
            p(b.xyzs), p(b.dirs), M, p(mn), p(sz), p(st['hd']), p(st['hc']), p(st['hd'][dn.n_mlp_params:]), *grid, p(self.x01), p(self.h), p(self.rgb16),
            p(self.sigmas), p(self.rgbs), p(self.save[0]), p(self.save[1]), p(self.save[2]), p(self.save[3]), p(self.fwd_ws), p(b.counter), stream), 'ngp_train_query_forward')
        _lib.check(lib.nrc_ngp_train_loss(
            p(self.sigmas), p(self.rgbs), p(b.deltas), p(b.ts), p(b.rays_a), p(b.counter), self.n_cap, M, self.T_THRESHOLD, p(b.bg), p(b.target), p(st['scale']),
            p(self.ray_rgb), p(self.ray_alpha), None, p(self.loss2), p(self.d_sigmas), p(self.d_rgbs), p(self.gd), self.n_clear, p(self.gc), self.gc.numel() + self.aux.numel(),
            p(self.loss_ws), stream), 'ngp_train_loss')
        group = self.optimizer.param_groups[0]
        beta1, beta2 = group['betas']
        sc = self.scaler
        backward = (p(self.d_sigmas), p(self.d_rgbs), M, p(self.x01), p(st['hd']), p(st['hc']), *grid, p(self.h), p(self.rgb16), p(self.save[0]), p(self.save[1]),
                    p(self.save[2]), p(self.save[3]), float(dn.loss_scale), p(self.gd), p(self.gc), dn.n_mlp_params, self.gd.numel(), self.gc.numel(),
                    p(self.bwd_scratch), p(b.counter))
        hyper = (float(group['lr']), p(st['lr']), float(beta1), float(beta2), float(group['eps']), float(group['weight_decay']), self.optimizer.adam_w_mode,
                 p(st['step']), p(st['bc']), p(st['scale']), p(st['tracker']), float(sc.get_growth_factor()) if st['scale'] is not None else 2.0,
                 float(sc.get_backoff_factor()) if st['scale'] is not None else 0.5, int(sc.get_growth_interval()) if st['scale'] is not None else 1, p(self.amp_state))
        fork = ctypes.c_void_p(self._gb_side.cuda_stream) if self._gb_side is not None else None
        if self.fused_step:
            _lib.check(lib.nrc_ngp_train_backward_step(
                *backward, p(st['pd']), p(st['md']), p(st['vd']), p(st['hd']), self.l2[0][0], self.l2[0][1], p(st['pc']), p(st['mc']), p(st['vc']), p(st['hc']),
                self.l2[1][0], self.l2[1][1], *hyper, fork, stream), 'ngp_train_backward_step')
            return
        if self.sharded:
            return self._update_sharded(st, backward, hyper, fork, stream, fork_prefetch)
        _lib.check(lib.nrc_ngp_train_query_backward_cleared(*backward, fork, stream), 'ngp_train_query_backward_cleared')
        if self.data_parallel:
            from . import parallel
            parallel.allreduce_flat(self.grads, average=True)
        _lib.check(lib.nrc_amp_adam_step(
            p(st['pd']), p(self.gd), p(st['md']), p(st['vd']), p(st['hd']), self.gd.numel(), self.l2[0][0], self.l2[0][1],
            p(st['pc']), p(self.gc), p(st['mc']), p(st['vc']), p(st['hc']), self.gc.numel(), self.l2[1][0], self.l2[1][1],
            *hyper, p(self.amp_ticket), None, stream), 'amp_adam_step')

    def _update_sharded(self, st: dict, backward: tuple, hyper: tuple, fork, stream, fork_prefetch) -> None:
        """The data-parallel step of SURVEY 8(e), wire time beside compute where the dependencies allow it (DESIGN 5 has the byte / time model):

            main stream                                   communication stream
            networks backward (flags inf / NaN) --E0-->   all-reduce [colour MLP | flag | density MLP] gradients (41 KB; hides under the grid backward)
            grid backward                       --E1-->   reduce-scatter (in place) of the table gradient: 7/8 x 48.8 MB leave each GPU at N = 8
            [next batch's march forks here]               settle: flag -> found_inf, step counter, 1 / (scale x world), scale rule       (1 thread)
                                                          Adam on the MLP weights (every rank, redundantly) and on THIS RANK'S table shard (1 / N of 12.2 M)
                                                          all-gather (in place) of the fp16 table the kernels read: 7/8 x 24.4 MB
            wait  <------------------------------------   done

        The fp32 master and the moments of the table stay sharded (gather_state() before a checkpoint).  Nothing here reads a value back."""
        from . import parallel
        lib, p, L = _lib.load(), _lib.ptr, self.layout
        dn, cn = self.model.encoding_xyz, self.model.color_mlp_with_encoding
        main = torch.cuda.current_stream(self.dev)
        (lr, lr_dev, beta1, beta2, eps, wd, adam_w, step_dev, bc, scale, tracker, growth, backoff, interval, amp_state) = hyper
        _lib.check(lib.nrc_ngp_train_networks_backward(*backward, p(self.aux), stream), 'ngp_train_networks_backward')
        self._ev[0].record(main)
        # (backward = dL_dsigmas, dL_drgbs, M, x01, weights d / c, grid (4), h, rgb16, saves (4), loss_scale, gd, gc, n_mlp, n_d, n_c, scratch, counter)
        M, x01, wd16, wc16 = backward[2], backward[3], backward[4], backward[5]
        _lib.check(lib.nrc_ngp_train_grid_backward(M, x01, wd16, wc16, *backward[6:10], *backward[17:24], fork, stream), 'ngp_train_grid_backward')
        self._ev[1].record(main)
        if fork_prefetch is not None:
            fork_prefetch()
        comm = self._comm
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if self.dp_timing else None
        comm.wait_event(self._ev[0])
        with torch.cuda.stream(comm):
            cs = ctypes.c_void_p(comm.cuda_stream)
            n_dm = L.n_density_mlp
            at = lambda t, off: ctypes.c_void_p(t.data_ptr() + off * t.element_size())
            slices = lambda *a: _lib.check(lib.nrc_amp_adam_slices(*a, lr, lr_dev, beta1, beta2, eps, wd, adam_w, bc, amp_state, cs), 'amp_adam_slices')

            def settle(aux):
                _lib.check(lib.nrc_amp_settle(p(aux), float(self.world), beta1, beta2, step_dev, bc, scale, tracker, growth, backoff, interval, amp_state, None, cs), 'amp_settle')

            def adam_small():
                slices(p(st['pd']), p(self.gd), p(st['md']), p(st['vd']), p(st['hd']), n_dm, self.l2[0][0], min(self.l2[0][1], n_dm),
                       p(st['pc']), p(self.gc), p(st['mc']), p(st['vc']), p(st['hc']), self.gc.numel(), self.l2[1][0], self.l2[1][1])

            def adam_table(begin, count):
                b = n_dm + begin
                slices(at(st['pd'], b), at(self.gd, b), at(st['md'], b), at(st['vd'], b), at(st['hd'], b), count, 0.0, 0, None, None, None, None, None, 0, 0.0, 0)

            pack = unpack = None
            if self.wire is not None:
                def pack():
                    _lib.check(lib.nrc_wire_pack_f16(at(self.gd, n_dm), p(self.wire), L.n_table, p(self.wire_saturated), cs), 'wire_pack_f16')

                def unpack(begin, count):
                    _lib.check(lib.nrc_wire_unpack_f16(at(self.wire, begin), at(self.gd, n_dm + begin), count, cs), 'wire_unpack_f16')
            parallel.sharded_step(L, self.grads, st['hd'][n_dm:], settle, adam_small, adam_table, before_table=lambda: comm.wait_event(self._ev[1]),
                                  mark=(lambda k: ev[k].record(comm)) if ev else None, wire=self.wire, pack=pack, unpack=unpack)
        main.wait_stream(comm)
        self._master_stale = True
        if ev:
            ev[4].record(main)           # the iteration continues here: [1]..[4] is what the step adds behind the backward pass
            self._dp_events.append(ev)

    def dp_times(self, clear: bool = True) -> dict | None:
        """dp_timing=True: mean milliseconds per sharded step since the last call -- `reduce_scatter_ms`, `adam_ms` (settle + both Adam launches), `all_gather_ms`,
        `exposed_ms` (end of the backward pass -> the main stream may continue).  Synchronises the device."""
        if not self._dp_events:
            return None
        torch.cuda.synchronize(self.dev)
        rows = [(e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]), e[2].elapsed_time(e[3]), e[0].elapsed_time(e[4])) for e in self._dp_events]
        if clear:
            self._dp_events = []
        mean = [sum(r[k] for r in rows) / len(rows) for k in range(4)]
        return {'reduce_scatter_ms': mean[0], 'adam_ms': mean[1], 'all_gather_ms': mean[2], 'exposed_ms': mean[3], 'iterations': len(rows)}

    def gather_state(self) -> None:
        """Sharded step: all-gathers the table's fp32 master parameters and both Adam moments (each rank has kept only its shard current), so that
        model.state_dict() / optimizer.state_dict() hold the whole state on every rank.  A collective: every rank calls it (before a checkpoint, before
        anything outside this class reads or edits `params`).  3 x 7/8 x 48.8 MB per GPU at N = 8, off the iteration path."""
        if not self.sharded or not self._master_stale:
            return
        from . import parallel
        st, n_dm = self._state_tensors(check=False), self.layout.n_density_mlp
        for t in (st['pd'], st['md'], st['vd']):
            parallel.all_gather_(t.detach()[n_dm:], self.rank, self.world)
        self._master_stale = False

    def _enqueue(self, cur: int, inline: bool, prefetch: bool, st: dict, explicit) -> None:
        if inline:
            self._march(cur, st, explicit)

        def fork():
            main = torch.cuda.current_stream(self.dev)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                self._march(1 - cur, st, explicit)
        at_collective = prefetch and self.prefetch_at == 'collective'
        if prefetch and not at_collective:
            # the fork sits in FRONT of the forward pass (measured against forks behind the loss and in front of Adam: 0.380 / 0.395 / 0.407 ms per
            # iteration) and behind the inline march: both marches move the cursor and the generator.  (Sharded data-parallel step: behind the grid
            # backward instead, where the GPU would otherwise wait for the wire -- prefetch_at.)
            fork()
        self._update(cur, st, fork if at_collective else None)
        if prefetch:
            torch.cuda.current_stream(self.dev).wait_stream(self._side)

    # ------------------------------------------------------------------------------------------------ one iteration
    def __call__(self, ids: torch.Tensor | None = None, bg: torch.Tensor | None = None, noise: torch.Tensor | None = None,
                 prefetch: bool | None = None) -> dict[str, torch.Tensor]:
        """One training iteration.  ids / bg / noise: explicit batch rows, background colour and march jitter instead of the resident order and
        the generator's draws (they apply to the batch marched BY this call: with prefetch on that is the NEXT iteration's batch, so callers
        that pass explicit values run with prefetch=False).  prefetch=False: do not march ahead -- what the caller says in the iteration in
        front of an occupancy-grid update."""
        explicit = (ids is not None, bg is not None, noise is not None)
        prefetch = (self.prefetch_default and not any(explicit)) if prefetch is None else bool(prefetch)
        if prefetch and self._side is None:
            raise RuntimeError('FusedTrainingIteration was built with prefetch=False')
        if any(explicit) and (prefetch or self._ready is not None):
            if prefetch:
                raise ValueError('explicit ids / bg / noise belong to the batch this call marches: pass prefetch=False')
            self._ready = None      # a batch marched ahead from the resident order is dropped in favour of the explicit one
        if not explicit[0] and self.order is None:
            raise RuntimeError('FusedTrainingIteration: no sampling order installed (rewind(order)) and no ids given')
        if explicit[0]:
            self.ids.copy_(ids.to(torch.int64), non_blocking=True)
        if explicit[1]:
            self.bg_in.copy_(bg, non_blocking=True)
        if explicit[2]:
            self.noise_in.copy_(noise, non_blocking=True)
        inline = self._ready is None
        cur = 0 if inline else self._ready
        self.optimizer.sync_hyperparameters()
        st = self._state_tensors()
        signature = tuple(None if t is None else t.data_ptr() for t in st.values()) + (self.order.data_ptr() if self.order is not None else 0,)
        if signature != self._signature:
            self._graphs, self._signature = {}, signature
        if not explicit[0]:      # the host mirrors the device cursor: it knows when the order runs out (Samplers/utils.py:22-23)
            marches = int(inline) + int(prefetch)
            if self._cursor_host + marches * self.n_rays > self.order.numel():
                raise RuntimeError('FusedTrainingIteration: the sampling order is used up -- rewind(new_order) first (RandomSequentialSampler.reset)')
            self._cursor_host += marches * self.n_rays
        self.calls += 1
        if self.use_graph and self.calls > 1:     # the first call runs eagerly: lazily created state (scaler scalars, fp16 copies) exists afterwards
            key = (cur, inline, prefetch, explicit)
            graph = self._graphs.get(key)
            if graph is None:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):     # records, does not execute
                    self._enqueue(cur, inline, prefetch, st, explicit)
                self._graphs[key] = graph
            graph.replay()
        else:
            self._enqueue(cur, inline, prefetch, st, explicit)
        self._ready = (1 - cur) if prefetch else None
        # the kernels wrote the parameters (and their fp16 copies) through raw pointers: tell autograd and every version-keyed cache
        for net in (self.model.encoding_xyz, self.model.color_mlp_with_encoding):
            torch.autograd.graph.increment_version(net.params)
            net._half_written_by_optimizer(net.params)
        b = self.sets[cur]
        return {'loss': self.loss2[0], 'rm_samples': b.counter[0], 'sample_overflow': b.overflow[0], 'rgb': self.ray_rgb, 'alpha': self.ray_alpha, 'bg': b.bg}

    def remaining_batches(self) -> int:
        """Full batches of the current size left in the installed order (host arithmetic, no device read)."""
        return 0 if self.order is None else (self.order.numel() - self._cursor_host) // self.n_rays
