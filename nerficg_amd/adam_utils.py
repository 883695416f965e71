"""nerficg_amd.adam_utils -- the optimizer-state surgery of src/Optim/adam_utils.py:6-98 (same names, arguments and error behaviour) on
top of ONE multi-tensor row gather (include/nerficg_hip.h group 10) instead of one boolean-mask copy per tensor and Adam moment.

Works with any torch.optim.Optimizer whose state uses 'exp_avg' / 'exp_avg_sq' (torch.optim.Adam, nerficg_amd.apex_optimizers.FusedAdam)
and whose groups hold a single f32 parameter each and carry a 'name' (the reference's convention).
"""
from __future__ import annotations

import ctypes
import math

import torch

from . import _lib

__all__ = ['replace_param_group_data', 'prune_param_groups', 'extend_param_groups', 'reset_state', 'sort_param_groups', 'gather_param_groups',
           'gather_rows', 'compact_mask']

_MOMENTS = ('exp_avg', 'exp_avg_sq')


def _single_param(group, who: str) -> torch.Tensor:
    if len(group['params']) != 1:
        raise NotImplementedError(f'"{who}" only implemented for single-parameter groups.')
    return group['params'][0]


def gather_rows(tensors: list[torch.Tensor], src: torch.Tensor, n_out: int, kind: torch.Tensor | None = None,
                zero_new: list[bool] | None = None) -> list[torch.Tensor]:
    """out[t][o] = tensors[t][src[o]] for o < n_out (rows = everything behind dim 0), all tensors in one launch (24 per launch).
    With `kind`, tensors flagged in `zero_new` get zeros in rows whose kind != 0."""
    lib = _lib.load()
    outs: list[torch.Tensor] = []
    ins: list[torch.Tensor] = []
    for t in tensors:
        _lib.check_input(t, 'gather_rows tensor', torch.float32)
        ins.append(t)
        outs.append(torch.empty((n_out, *t.shape[1:]), dtype=torch.float32, device=t.device))
    if not ins or n_out == 0:
        return outs
    _lib.check_input(src, 'src', torch.int32)
    flags = [False] * len(ins) if zero_new is None else list(zero_new)
    for b in range(0, len(ins), 24):
        chunk_in, chunk_out, chunk_flag = ins[b:b + 24], outs[b:b + 24], flags[b:b + 24]
        n = len(chunk_in)
        rows = [max(1, math.prod(t.shape[1:])) for t in chunk_in]
        a_in = (ctypes.c_void_p * n)(*[t.data_ptr() for t in chunk_in])
        a_out = (ctypes.c_void_p * n)(*[t.data_ptr() for t in chunk_out])
        a_row = (ctypes.c_int32 * n)(*rows)
        a_zero = (ctypes.c_int32 * n)(*[int(f) for f in chunk_flag])
        _lib.check(lib.nrc_gather_rows(ctypes.cast(a_in, ctypes.c_void_p), ctypes.cast(a_out, ctypes.c_void_p), ctypes.cast(a_row, ctypes.c_void_p),
                                       ctypes.cast(a_zero, ctypes.c_void_p), n, _lib.ptr(src), _lib.ptr(kind), n_out, _lib.stream_of(src)),
                   'gather_rows')
    return outs


def compact_mask(mask: torch.Tensor) -> torch.Tensor:
    """Ascending int32 indices of the set entries of a boolean mask (device scan; one host read of the count)."""
    lib = _lib.load()
    m = mask.reshape(-1)
    if m.dtype != torch.bool:
        raise RuntimeError('mask must be a boolean tensor')
    _lib.check_input(m, 'mask')
    n = m.numel()
    idx = torch.empty(n, dtype=torch.int32, device=m.device)
    count = torch.zeros(1, dtype=torch.int32, device=m.device)
    ws = torch.empty(int(lib.nrc_compact_mask_ws_bytes(n)), dtype=torch.uint8, device=m.device)
    _lib.check(lib.nrc_compact_mask(_lib.ptr(m), n, _lib.ptr(idx), _lib.ptr(count), _lib.ptr(ws), _lib.stream_of(m)), 'compact_mask')
    return idx[:int(count.item())]


def gather_param_groups(optimizer: torch.optim.Optimizer, src: torch.Tensor, n_out: int, kind: torch.Tensor | None = None,
                        group_names: list[str] | None = None, who: str = 'gather_param_groups') -> dict[str, torch.Tensor]:
    """Rebuilds every (selected) single-parameter group as rows `src` of the old one -- parameter and both Adam moments of all groups
    in one kernel launch; rows with kind != 0 start with zero moments.  Returns {group name: new Parameter}."""
    groups = []
    for group in optimizer.param_groups:
        if group_names is not None and group['name'] not in group_names:
            continue
        groups.append((group, _single_param(group, who)))
    tensors, zero_new = [], []
    for _, p in groups:
        tensors.append(p.data)
        zero_new.append(False)
        state = optimizer.state.get(p)
        if state:
            for key in _MOMENTS:
                tensors.append(state[key])
                zero_new.append(True)
    outs = iter(gather_rows(tensors, src, n_out, kind, zero_new))
    new_params = {}
    for group, old_param in groups:
        new_param = torch.nn.Parameter(next(outs))
        state = optimizer.state.get(old_param)
        if state:
            for key in _MOMENTS:
                state[key] = next(outs)
            optimizer.state.pop(old_param)
            optimizer.state[new_param] = state
        group['params'][0] = new_param
        new_params[group['name']] = new_param
    return new_params


def replace_param_group_data(optimizer: torch.optim.Optimizer, new_values: torch.Tensor, group_name: str, reset_state: bool = True) -> None:
    """adam_utils.py:6-18"""
    for group in optimizer.param_groups:
        if group['name'] == group_name:
            param = _single_param(group, 'replace_param_group_data')
            param.data = new_values
            if reset_state:
                state = optimizer.state.get(param)
                if state:
                    for key in _MOMENTS:
                        state[key].zero_()


def prune_param_groups(optimizer: torch.optim.Optimizer, mask: torch.Tensor, group_names: list[str] | None = None) -> dict[str, torch.Tensor]:
    """adam_utils.py:21-39: keeps the rows where `mask` is set."""
    idx = compact_mask(mask)
    return gather_param_groups(optimizer, idx, idx.numel(), None, group_names, 'prune_param_groups')


def sort_param_groups(optimizer: torch.optim.Optimizer, ordering: torch.Tensor, group_names: list[str] | None = None) -> dict[str, torch.Tensor]:
    """adam_utils.py:81-98: rows in the given order."""
    idx = ordering.to(torch.int32).contiguous()
    return gather_param_groups(optimizer, idx, idx.numel(), None, group_names, 'sort_param_groups')


def extend_param_groups(optimizer: torch.optim.Optimizer, additional_params: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
    """adam_utils.py:42-61: appends rows (zero Adam moments for them)."""
    new_params = {}
    for group in optimizer.param_groups:
        old_param = _single_param(group, 'extend_param_groups')
        extension = additional_params.get(group['name'], None)
        if extension is None:
            continue
        state = optimizer.state.get(old_param)
        new_param = torch.nn.Parameter(torch.cat((old_param.data, extension), dim=0))
        if state:
            for key in _MOMENTS:
                grown = torch.zeros_like(new_param.data)
                grown[:old_param.shape[0]] = state[key]
                state[key] = grown
            optimizer.state.pop(old_param)
            optimizer.state[new_param] = state
        group['params'][0] = new_param
        new_params[group['name']] = new_param
    return new_params


def reset_state(optimizer: torch.optim.Optimizer, group_names: list[str] | None = None, indices: torch.Tensor | None = None) -> None:
    """adam_utils.py:64-79"""
    for group in optimizer.param_groups:
        if group_names is not None and group['name'] not in group_names:
            continue
        param = _single_param(group, 'reset_state')
        state = optimizer.state.get(param)
        if state:
            for key in _MOMENTS:
                if indices is not None:
                    state[key][indices] = 0
                else:
                    state[key].zero_()
