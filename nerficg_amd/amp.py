"""nerficg_amd.amp -- torch.amp.GradScaler with the inf / NaN check as ONE streaming kernel per gradient tensor.

The reference trains InstantNGP under `torch.amp.GradScaler(init_scale=128, growth_interval=...)` (src/Methods/InstantNGP/Trainer.py:44,89-93).
With an optimizer that applies the scale itself (FusedAdam: `_step_supports_amp_scaling`) the scaler's only work per iteration is the check
of the gradients for inf / NaN -- torch runs it as a multi-tensor "unscale by 1.0 and check" pass that READS AND WRITES every gradient (37 us
for the 48.8 MB hash-table gradient on an MI355X, plus four small kernels around it).  This subclass does the same check read-only at the
rate the HBM delivers (nrc_nonfinite_check, include/nerficg_hip.h group 8).  Same constructor, same state dict, same update() rule; the
reference's trainer keeps working with torch's own class -- this one is what nerficg_amd's own loops (graphs.py, bench.py) use.
"""
from __future__ import annotations

import torch

from . import _lib

__all__ = ['GradScaler']


_GRAD_SCALER_KWARG: dict = {}


def _takes_grad_scaler(optimizer_type) -> bool:
    """Does this optimizer class's step() take the (deprecated) `grad_scaler` keyword?  (torch inspects the signature on every call; once per class here)"""
    hit = _GRAD_SCALER_KWARG.get(optimizer_type)
    if hit is None:
        import inspect
        hit = _GRAD_SCALER_KWARG[optimizer_type] = 'grad_scaler' in inspect.signature(optimizer_type.step).parameters
    return hit


class GradScaler(torch.amp.GradScaler):
    """`single_optimizer` (keyword, default True): the scaler serves ONE optimizer per iteration, as the reference's trainers do (Trainer.py:44,89-93), so the
    first step() of an iteration may apply the scale rule on the device together with the Adam update (nrc_amp_adam_step).  A scaler shared by several
    optimizers cannot be recognised at that first step() -- torch's per-optimizer table is rebuilt by every update() -- so such setups say
    `single_optimizer=False` and take torch's general path; stepping or unscaling a SECOND optimizer behind a fused step raises instead of unscaling with the
    already-updated scale."""

    def __init__(self, *args, single_optimizer: bool = True, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._single_optimizer = bool(single_optimizer)
        self._fused_update_done = False
        self._fused_optimizer_id = None

    def _refuse_second_optimizer(self, optimizer, what: str) -> None:
        if self._fused_update_done and id(optimizer) != self._fused_optimizer_id:
            raise RuntimeError(f'nerficg_amd.amp.GradScaler.{what}: another optimizer has already been stepped in this iteration through the fused step + update '
                               '(the scale rule has been applied on the device); a scaler shared by several optimizers must be built with single_optimizer=False')

    def unscale_(self, optimizer) -> None:
        self._refuse_second_optimizer(optimizer, 'unscale_')
        return super().unscale_(optimizer)

    def scale_tensor(self, device) -> torch.Tensor:
        """The scale as the device scalar `scale()` multiplies by (created on first use, as `scale()` does): for callers that fold the
        multiplication into their own loss kernel (nerficg_amd.ngp.scaled_mse_loss) instead of calling scale(loss)."""
        if not self._enabled:
            raise RuntimeError('scale_tensor: the scaler is disabled')
        if self._scale is None:
            self._lazy_init_scale_growth_tracker(torch.device(device))
        return self._scale

    def step(self, optimizer, *args, **kwargs):
        """torch.amp.GradScaler.step for an optimizer that applies the scale itself (`_step_supports_amp_scaling`), minus two launches:
        torch builds `found_inf` as `sum([...])` (0 + t: an add) and `grad_scale` as `scale * 1` (a multiplication) on every call; with one
        device and no scale the caller has put on the optimizer these are the tensors themselves.  Everything else defers to torch."""
        from torch.amp.grad_scaler import OptState
        self._refuse_second_optimizer(optimizer, 'step')
        if (not self._enabled or 'closure' in kwargs or not getattr(optimizer, '_step_supports_amp_scaling', False)
                or hasattr(optimizer, 'grad_scale') or _takes_grad_scaler(type(optimizer))):
            return super().step(optimizer, *args, **kwargs)
        self._check_scale_growth_tracker('step')
        state = self._per_optimizer_states[id(optimizer)]
        if state['stage'] is OptState.STEPPED:
            raise RuntimeError('step() has already been called since the last update().')
        if (state['stage'] is OptState.READY and type(self)._check_inf_per_device is GradScaler._check_inf_per_device and not args and not kwargs
                and hasattr(optimizer, '_amp_fused_step') and self._single_optimizer and len(self._per_optimizer_states) == 1):
            # the whole of step() + update() as one library call: check, step counter, scale rule, Adam (nrc_amp_adam_step).  update() then has
            # nothing left to launch.  (Subclasses that hook the check -- the data-parallel scaler's agreement -- and scalers built with single_optimizer=False take the general path.)
            if optimizer._amp_fused_step(self._scale, self._growth_tracker, self._growth_factor, self._backoff_factor, self._growth_interval):
                state['stage'] = OptState.STEPPED
                self._fused_update_done = True
                self._fused_optimizer_id = id(optimizer)
                return None
        if state['stage'] is OptState.READY:
            self._check_inf_per_device(optimizer)
        scale = self._get_scale_async()
        found = [t.to(scale.device, non_blocking=True) for t in state['found_inf_per_device'].values()]
        optimizer.found_inf = found[0] if len(found) == 1 else sum(found)
        optimizer.grad_scale = None if state['stage'] == OptState.UNSCALED else scale
        try:
            retval = optimizer.step(*args, **kwargs)
        finally:
            del optimizer.grad_scale
            del optimizer.found_inf
        state['stage'] = OptState.STEPPED
        return retval

    def update(self, new_scale=None):
        """torch.amp.GradScaler.update; after a fused step (see step()) the scale rule has been applied on the device already: only the per-optimizer
        bookkeeping is reset.  An explicit new_scale is honoured as in torch."""
        if getattr(self, '_fused_update_done', False):
            self._fused_update_done = False
            self._fused_optimizer_id = None
            if new_scale is None:
                from collections import defaultdict
                from torch.amp.grad_scaler import _refresh_per_optimizer_state
                self._per_optimizer_states = defaultdict(_refresh_per_optimizer_state)
                return
            # (an explicit scale overrides whatever the rule did)
            _scale, _ = self._check_scale_growth_tracker('update')
            if isinstance(new_scale, float):
                self._scale.fill_(new_scale)
            else:
                self._scale.copy_(new_scale)
            from collections import defaultdict
            from torch.amp.grad_scaler import _refresh_per_optimizer_state
            self._per_optimizer_states = defaultdict(_refresh_per_optimizer_state)
            return
        return super().update(new_scale)

    def _check_inf_per_device(self, optimizer):
        grads = [p.grad for group in optimizer.param_groups for p in group['params'] if p.grad is not None]
        fast = bool(grads) and all(g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and not g.is_sparse and g.device == grads[0].device
                                   for g in grads)
        if not fast:
            return super()._check_inf_per_device(optimizer)
        _scale, _ = self._check_scale_growth_tracker('_check_inf_per_device')
        dev = grads[0].device
        found = torch.zeros((), dtype=torch.float32, device=dev)
        lib = _lib.load()
        for k in range(0, len(grads), 4):   # four tensors per launch
            four = grads[k:k + 4] + [None] * (4 - len(grads[k:k + 4]))
            args = []
            for g in four:
                args += [_lib.ptr(g), g.numel() if g is not None else 0]
            _lib.check(lib.nrc_nonfinite_check4(*args, _lib.ptr(found), _lib.stream_of(found)), 'nonfinite_check4')
        per_device = {dev: found} if _scale.device == dev else {dev: found, _scale.device: torch.zeros((), dtype=torch.float32, device=_scale.device)}
        self._per_optimizer_states[id(optimizer)]['found_inf_per_device'] = per_device
        return per_device
