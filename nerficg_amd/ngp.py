"""nerficg_amd.ngp -- MI355X-native fast paths for the InstantNGP hot loop that go beyond the reference's op granularity.

The reference composes the loop from separate ops with tensors materialised in between
(src/Methods/InstantNGP/Renderer.py:48-138).  The functions here fuse across those boundaries while producing the same
results as the drop-in modules (tests/test_gpu_tcnn_parity.py, tests/test_gpu_render_parity.py).
"""
from __future__ import annotations

import torch

from . import _lib

__all__ = ['query_fused']


def query_fused(density_net, color_net, xyz01: torch.Tensor, dirs: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    """query_model (Renderer.py:48-53) as one encode + one MLP kernel per chunk: xyz01 (M,3) f32 in [0,1], dirs (M,3) unit f32 -> sigmas (M) f32, rgbs (M,3) f32."""
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
