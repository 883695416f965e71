"""nerficg_amd.MortonEncoding -- drop-in for src/CudaUtils/MortonEncoding (morton_encoding.py:4-6 -> _C.morton_encode,
morton_encoding.cu:48-74): 63-bit Morton codes of (N,3) f32 positions normalised to their bounding cube."""
from __future__ import annotations

import torch

from .. import _lib

__all__ = ['morton_encode']


def morton_encode(positions: torch.Tensor) -> torch.Tensor:
    """Computes the morton codes for a set of 3D positions. Same four checks as morton_encoding.cu:49-52."""
    if not positions.is_cuda:
        raise RuntimeError('positions must be a CUDA tensor')
    if positions.dtype != torch.float32:
        raise RuntimeError('positions must be float32')
    if not positions.is_contiguous():
        raise RuntimeError('positions must be contiguous')
    if positions.dim() != 2 or positions.shape[1] != 3:
        raise RuntimeError('positions must have shape (N, 3)')
    lib = _lib.load()
    n = positions.shape[0]
    codes = torch.empty(n, dtype=torch.int64, device=positions.device)
    ws = torch.empty(int(lib.nrc_morton_encode_ws_bytes(n)), dtype=torch.uint8, device=positions.device)
    _lib.check(lib.nrc_morton_encode(_lib.ptr(positions), n, _lib.ptr(codes), _lib.ptr(ws), _lib.stream_of(positions)), 'morton_encode')
    return codes
