"""CPU-only checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/nerficg_hip.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import subprocess
from pathlib import Path

import pytest

from nerficg_amd import _lib

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope='module')
def lib():
    if not _lib.LIB_PATH.exists():
        from nerficg_amd.build import build
        build(verbose=False)
    return _lib.load()


def test_header_parses_and_every_symbol_is_exported(lib):
    protos = _lib.parse_header()
    assert len(protos) >= 19
    for name in protos:
        assert hasattr(lib, name), name
    assert lib.nrc_abi_version() == _lib.header_abi_version() >= 3   # the loader refuses any other pairing (round-3 advisor finding)
    assert b'gfx950' in lib.nrc_build_info()


def test_no_undeclared_exports(lib):
    """Everything the .so exports with the nrc_ prefix is declared in the header (the header IS the boundary)."""
    out = subprocess.run(['nm', '-D', '--defined-only', str(_lib.LIB_PATH)], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ' T ' in ln and ln.split()[-1].startswith('nrc_')}
    assert exported == set(_lib.parse_header())


def test_code_object_targets_gfx950_only():
    """The fat binary embedded in the .so carries gfx950 code objects and nothing else (no multi-arch / compat builds)."""
    import re
    blob = _lib.LIB_PATH.read_bytes()
    targets = set(re.findall(rb'amdgcn-amd-amdhsa--(gfx[0-9a-z]+)', blob))
    assert targets == {b'gfx950'}, targets


def test_argument_validation_without_gpu(lib):
    """Host-side validation paths return NRC_ERR_INVALID before any HIP call."""
    assert lib.nrc_morton3D(None, -1, None, None) == -1
    assert lib.nrc_packbits(None, 7, 8, 0.0, None, None) == -1
    assert lib.nrc_ray_aabb_intersect(None, None, None, None, 4, 1, 0, None, None, None, None) == -1
    assert lib.nrc_raymarching_train_ws_bytes(1000, 1024) >= 1000 * 4 + 1000 * 1024 * 4
    assert lib.nrc_morton3D(None, 0, None, None) == 0


def test_group_13_entry_points_validate_before_they_launch(lib):
    """The fused training iteration's entry points (include/nerficg_hip.h group 13) called with null pointers / zero sizes on a machine WITHOUT a GPU:
    every one returns a status (NRC_ERR_INVALID, or NRC_OK for an empty call) before any HIP call -- none dereferences, none launches."""
    protos = _lib.parse_header()
    prefixes = ('nrc_ngp_train_', 'nrc_amp_adam_step', 'nrc_grid_backward_live', 'nrc_sum_squares_two', 'nrc_clear_seed_two', 'nrc_ngp_loss_forward',
                'nrc_raymarching_train_count_posted')
    names = [n for n in protos if n.startswith(prefixes) and not n.endswith('_bytes') and not n.endswith('_floats')]
    assert len(names) >= 12, names
    status = {}
    for n in names:
        args = [None if ('*' in t or t == 'nrc_stream_t') else (0.0 if t in ('float', 'double') else 0) for t, _ in protos[n][1]]
        status[n] = getattr(lib, n)(*args)
    assert set(status.values()) <= {0, -1}, status
    for n in ('nrc_ngp_train_march', 'nrc_ngp_train_loss', 'nrc_ngp_train_backward_step', 'nrc_ngp_train_query_backward_cleared', 'nrc_amp_adam_step',
              'nrc_ngp_train_query_forward', 'nrc_grid_backward_live', 'nrc_raymarching_train_count_posted'):
        assert status[n] == -1, (n, status[n])
    # sizes: the wave-per-ray march takes 1 .. 32 768 rays; the encoder's brick must tile the 8 x 8 footprint
    assert lib.nrc_ngp_train_march_ws_bytes(0, 1024) == -1 and lib.nrc_ngp_train_march_ws_bytes(32769, 1024) == -1
    assert lib.nrc_ngp_train_march_ws_bytes(4096, 1024) >= 4096 * 1024 * 4
    assert lib.nrc_ngp_set_encoder_shape(3, 1) == 0 and lib.nrc_ngp_set_encoder_shape(2, 2) == 0 and lib.nrc_ngp_set_encoder_shape(-1, -1) == 0
    assert lib.nrc_ngp_set_encoder_shape(0, 0) == -1 and lib.nrc_ngp_set_encoder_shape(3, 3) == -1 and lib.nrc_ngp_set_encoder_shape(4, 0) == -1


def test_python_mirror_exposes_reference_names():
    import nerficg_amd.VolumeRenderingV2 as vr
    # csrc/binding.cpp:234-250 + custom_functions.py classes
    for name in ['ray_aabb_intersect', 'ray_sphere_intersect', 'morton3D', 'morton3D_invert', 'packbits', 'raymarching_train',
                 'raymarching_test', 'composite_train_fw', 'composite_train_bw', 'composite_test_fw', 'distortion_loss_fw',
                 'distortion_loss_bw', 'RayAABBIntersector', 'RaySphereIntersector', 'RayMarcher', 'VolumeRenderer', 'TruncExp',
                 'DistortionLoss']:
        assert hasattr(vr, name), name
    from nerficg_amd.MortonEncoding import morton_encode  # noqa: F401


def test_ops_fail_loudly_on_cpu_tensors():
    import torch
    import nerficg_amd.VolumeRenderingV2 as vr
    with pytest.raises(RuntimeError, match='must be a CUDA tensor'):
        vr.morton3D(torch.zeros(4, 3, dtype=torch.int32))


def test_loader_refuses_a_library_of_another_abi_version(lib, monkeypatch):
    """A stale .so under newer bindings must fail at load time, not in an out-of-bounds write of a resized workspace."""
    monkeypatch.setattr(_lib, 'header_abi_version', lambda *a: 999)
    _lib.load.cache_clear()
    try:
        with pytest.raises(_lib.NativeLibraryError, match='ABI version'):
            _lib.load()
    finally:
        monkeypatch.undo()
        _lib.load.cache_clear()
        _lib.load()
