"""nerficg_amd.apex_optimizers -- drop-in for `apex.optimizers.FusedAdam` as nerficg imports it (src/Thirdparty/Apex.py:17) and builds
it (src/Methods/InstantNGP/Trainer.py:33-38; src/Methods/GaussianSplatting/Model.py:131-136): FusedAdam(params | param_groups, lr, eps,
betas, adam_w_mode, weight_decay, bias_correction).

* state per parameter = {'exp_avg', 'exp_avg_sq'} (same names as apex / torch.optim.Adam), the step counter lives in the parameter
  group like in apex -- so the reference's optimizer-state surgery (src/Optim/adam_utils.py:6-98: prune / extend / sort / reset of
  single-parameter groups) works unchanged;
* `_step_supports_amp_scaling`: torch.amp.GradScaler hands over its scale and found-inf tensors and the kernel applies them on the
  device (apex's FusedAdam makes the scaler unscale in a separate pass and sync on found_inf).  With apex an overflow-skipped step does not
  advance the step counter (the scaler never calls step()); here `group['step']` is advanced on the host every call and a device counter
  of skipped steps is subtracted inside nrc_adam_prepare, so the bias corrections follow the reference's trajectory without a host sync
  (`effective_step(group)` reads the corrected count back);
* a parameter that belongs to a nerficg_amd.tinycudann module gets its fp16 compute copy rewritten by the same kernel and its version
  counter bumped, so the next forward can neither see stale weights nor pay a separate conversion pass.
Kernel: nerficg_amd/csrc/adam.hip through the C ABI (include/nerficg_hip.h group 8).
"""
from __future__ import annotations

import torch

from .. import _lib

__all__ = ['FusedAdam']


class FusedAdam(torch.optim.Optimizer):
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, bias_correction=True, betas=(0.9, 0.999), eps=1e-8, adam_w_mode=True, weight_decay=0.0, amsgrad=False,
                 capturable=False, master_weights=False, set_grad_none=True):
        if amsgrad:
            raise RuntimeError('FusedAdam does not support the AMSGrad variant.')  # same restriction as apex
        if capturable or master_weights:
            raise RuntimeError('nerficg_amd FusedAdam: capturable / master_weights are not implemented')
        defaults = dict(lr=lr, bias_correction=bias_correction, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.adam_w_mode = 1 if adam_w_mode else 0
        self.set_grad_none = set_grad_none
        self._amp = {}  # group index -> (skipped-step counter i32[1], bias corrections f32[2]) on the device, GradScaler runs only

    def effective_step(self, group) -> int:
        """Step count that entered the bias corrections: group['step'] minus the overflow-skipped steps (host read, for tests / logging)."""
        entry = self._amp.get(self.param_groups.index(group))
        return group.get('step', 0) - (int(entry[0].item()) if entry is not None else 0)

    def zero_grad(self, set_to_none: bool | None = None):
        super().zero_grad(set_to_none=self.set_grad_none if set_to_none is None else set_to_none)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        grad_scale = getattr(self, 'grad_scale', None)
        found_inf = getattr(self, 'found_inf', None)
        for gi, group in enumerate(self.param_groups):
            if not any(p.grad is not None for p in group['params']):
                continue
            # like apex: one step counter per group, advanced whenever the group has gradients
            group['step'] = group.get('step', 0) + 1
            beta1, beta2 = group['betas']
            if group['bias_correction']:
                bc1, bc2 = 1.0 - beta1 ** group['step'], 1.0 - beta2 ** group['step']
            else:
                bc1 = bc2 = 1.0
            bc_dev = None
            if found_inf is not None and group['bias_correction']:
                entry = self._amp.get(gi)
                if entry is None or entry[0].device != found_inf.device:
                    entry = self._amp[gi] = (torch.zeros(1, dtype=torch.int32, device=found_inf.device),
                                             torch.ones(2, dtype=torch.float32, device=found_inf.device))
                bc_dev = entry[1]
                _lib.check(lib.nrc_adam_prepare(int(group['step']), float(beta1), float(beta2), _lib.ptr(found_inf), _lib.ptr(entry[0]),
                                                _lib.ptr(bc_dev), _lib.stream_of(found_inf)), 'adam_prepare')
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError('FusedAdam does not support sparse gradients, please consider SparseAdam instead')
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise RuntimeError('nerficg_amd FusedAdam: only float32 parameters / gradients are implemented')
                state = self.state[p]
                if len(state) == 0:
                    state['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                for t, name in ((p, 'param'), (state['exp_avg'], 'exp_avg'), (state['exp_avg_sq'], 'exp_avg_sq')):
                    _lib.check_input(t, name, torch.float32)
                owner = getattr(p, '_nrc_half_owner', None)
                owner = owner() if owner is not None else None
                half = owner._half_for_optimizer(p) if owner is not None else None
                _lib.check(lib.nrc_adam_step(
                    _lib.ptr(p), _lib.ptr(g), _lib.ptr(state['exp_avg']), _lib.ptr(state['exp_avg_sq']), p.numel(), float(group['lr']), float(beta1),
                    float(beta2), float(group['eps']), float(group['weight_decay']), self.adam_w_mode, float(bc1), float(bc2), _lib.ptr(bc_dev),
                    _lib.ptr(grad_scale), _lib.ptr(found_inf), _lib.ptr(half), _lib.stream_of(p)), 'adam_step')
                # the kernel writes through a raw pointer: tell autograd (and every version-keyed cache) that p changed
                torch.autograd.graph.increment_version(p)
                if owner is not None:
                    owner._half_written_by_optimizer(p)
        return loss
