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
Optimizer = torch.optim.Optimizer
Names = list[str] | None          # group names to act on (None: all groups)
Rebuilt = dict[str, torch.Tensor]  # group name -> the Parameter that replaced the old one


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


def _selected(optimizer, names, who: str):
    """(group, its single parameter, its Adam state or None) for the groups whose 'name' is in `names` (all groups for None)."""
    for group in optimizer.param_groups:
        if names is None or group['name'] in names:
            tensor = _single_param(group, who)
            yield group, tensor, (optimizer.state.get(tensor) or None)


def _install(optimizer, group, old: torch.Tensor, new: torch.nn.Parameter, state) -> None:
    """`new` takes the place of `old` in its group; the state entry (already holding tensors of the new size) moves with it."""
    if state is not None:
        del optimizer.state[old]
        optimizer.state[new] = state
    group['params'][0] = new


def _moments(state):
    return [] if state is None else [state[key] for key in _MOMENTS]


def gather_param_groups(optimizer: Optimizer, src: torch.Tensor, n_out: int, kind: torch.Tensor | None = None,
                        group_names: Names = None, who: str = 'gather_param_groups') -> Rebuilt:
    """Rebuilds every (selected) single-parameter group as rows `src` of the old one -- parameter and both Adam moments of all groups
    in one kernel launch; rows with kind != 0 start with zero moments.  Returns {group name: new Parameter}."""
    todo = list(_selected(optimizer, group_names, who))
    tensors, fresh_rows_are_zero = [], []
    for _, tensor, state in todo:
        moments = _moments(state)
        tensors += [tensor.data, *moments]
        fresh_rows_are_zero += [False] + [True] * len(moments)
    gathered = iter(gather_rows(tensors, src, n_out, kind, fresh_rows_are_zero))
    rebuilt = {}
    for group, tensor, state in todo:
        replacement = torch.nn.Parameter(next(gathered))
        if state is not None:
            for key in _MOMENTS:
                state[key] = next(gathered)
        _install(optimizer, group, tensor, replacement, state)
        rebuilt[group['name']] = replacement
    return rebuilt


def replace_param_group_data(optimizer: Optimizer, new_values: torch.Tensor, group_name: str, reset_state: bool = True) -> None:
    """New values (same shape) for the parameter of group `group_name`, by default with cleared Adam moments (adam_utils.py:6-18)."""
    for _, tensor, state in _selected(optimizer, (group_name,), 'replace_param_group_data'):
        tensor.data = new_values
        if reset_state:
            for moment in _moments(state):
                moment.zero_()


def prune_param_groups(optimizer: Optimizer, mask: torch.Tensor, group_names: Names = None) -> Rebuilt:
    """Keeps the rows where `mask` is set (adam_utils.py:21-39)."""
    rows = compact_mask(mask)
    return gather_param_groups(optimizer, rows, rows.numel(), None, group_names, 'prune_param_groups')


def sort_param_groups(optimizer: Optimizer, ordering: torch.Tensor, group_names: Names = None) -> Rebuilt:
    """Rows in the given order (adam_utils.py:81-98)."""
    rows = ordering.to(torch.int32).contiguous()
    return gather_param_groups(optimizer, rows, rows.numel(), None, group_names, 'sort_param_groups')


def extend_param_groups(optimizer: Optimizer, additional_params: dict[str, torch.Tensor]) -> Rebuilt:
    """Appends rows to the groups named in `additional_params`; the new rows start with zero Adam moments (adam_utils.py:42-61)."""
    rebuilt = {}
    for group, tensor, state in _selected(optimizer, None, 'extend_param_groups'):
        extra = additional_params.get(group['name'])
        if extra is None:
            continue
        replacement = torch.nn.Parameter(torch.cat((tensor.data, extra)))
        if state is not None:
            n_old = tensor.shape[0]
            for key in _MOMENTS:
                padded = state[key].new_zeros(replacement.shape)
                padded[:n_old].copy_(state[key])
                state[key] = padded
        _install(optimizer, group, tensor, replacement, state)
        rebuilt[group['name']] = replacement
    return rebuilt


def reset_state(optimizer: Optimizer, group_names: Names = None, indices: torch.Tensor | None = None) -> None:
    """Clears the Adam moments of the selected groups, entirely or at `indices` (adam_utils.py:64-79)."""
    for _, _, state in _selected(optimizer, group_names, 'reset_state'):
        for moment in _moments(state):
            if indices is None:
                moment.zero_()
            else:
                moment[indices] = 0
