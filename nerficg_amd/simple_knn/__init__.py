"""nerficg_amd.simple_knn -- drop-in for `simple_knn._C.distCUDA2` as nerficg binds it (src/Thirdparty/SimpleKNN.py:17-18) and calls it
(src/Optim/knn_utils.py:34-38): points (N,3) f32 -> (N,) mean squared distance to the 3 nearest neighbours (exact).
Morton codes and the 3-NN search run in HIP (include/nerficg_hip.h groups 2 and 9); the sort between them is torch.sort."""
from __future__ import annotations

import torch

from .. import _lib
from ..MortonEncoding import morton_encode

__all__ = ['distCUDA2', '_C']


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    pts = points.detach().to(torch.float32).contiguous()
    _lib.check_input(pts, 'points', torch.float32)
    if pts.dim() != 2 or pts.shape[1] != 3:
        raise RuntimeError('distCUDA2: points must have shape (N, 3)')
    n = pts.shape[0]
    if n == 0:
        return torch.empty(0, dtype=torch.float32, device=pts.device)
    order = torch.argsort(morton_encode(pts), stable=True)
    sorted_pts = pts[order].contiguous()
    lib = _lib.load()
    out_sorted = torch.empty(n, dtype=torch.float32, device=pts.device)
    ws = torch.empty(int(lib.nrc_knn3_ws_bytes(n)), dtype=torch.uint8, device=pts.device)
    _lib.check(lib.nrc_knn3_mean_sq_dist(_lib.ptr(sorted_pts), n, _lib.ptr(out_sorted), _lib.ptr(ws), _lib.stream_of(pts)), 'knn3_mean_sq_dist')
    out = torch.empty_like(out_sorted)
    out[order] = out_sorted
    return out


class _C:  # `from simple_knn import _C; _C.distCUDA2(points)`
    distCUDA2 = staticmethod(distCUDA2)
