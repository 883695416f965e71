"""CPU: the import shims (nerficg_amd/shims) expose the names the reference tree imports; when the reference is present (build
container only -- it never travels to the GPU box) its own extension front-ends are imported against them in a subprocess."""
import importlib
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
SHIMS = ROOT / 'nerficg_amd' / 'shims'
REF = Path('/root/reference/src')


def test_shim_modules_export_the_expected_names(monkeypatch):
    monkeypatch.syspath_prepend(str(SHIMS))
    for name in ('VolumeRenderingV2', 'MortonEncoding', 'tinycudann', 'diff_gaussian_rasterization', 'fused_ssim', 'apex.optimizers', 'torch_scatter'):
        sys.modules.pop(name, None)
    vr = importlib.import_module('VolumeRenderingV2')
    assert sorted(vr.__all__) == sorted(['ray_aabb_intersect', 'ray_sphere_intersect', 'packbits', 'morton3D', 'morton3D_invert', 'raymarching_train',
                                         'raymarching_test', 'composite_train_fw', 'composite_train_bw', 'composite_test_fw', 'distortion_loss_fw',
                                         'distortion_loss_bw'])  # binding.cpp:234-250
    assert callable(importlib.import_module('MortonEncoding')._C.morton_encode)
    tcnn = importlib.import_module('tinycudann')
    assert hasattr(tcnn, 'NetworkWithInputEncoding') and hasattr(tcnn, 'free_temporary_memory')
    dgr = importlib.import_module('diff_gaussian_rasterization')
    assert hasattr(dgr, 'GaussianRasterizationSettings') and hasattr(dgr, 'GaussianRasterizer')
    assert callable(importlib.import_module('fused_ssim').fused_ssim)
    assert importlib.import_module('apex.optimizers').FusedAdam.__name__ == 'FusedAdam'
    assert callable(importlib.import_module('simple_knn')._C.distCUDA2)
    seg = importlib.import_module('torch_scatter').segment_csr
    src = torch.arange(12, dtype=torch.float32).reshape(6, 2)
    out = seg(src, torch.tensor([0, 2, 2, 6]))
    assert torch.equal(out, torch.stack([src[0:2].sum(0), torch.zeros(2), src[2:6].sum(0)]))
    for name in ('VolumeRenderingV2', 'MortonEncoding', 'MortonEncoding._C', 'tinycudann', 'diff_gaussian_rasterization', 'fused_ssim', 'apex', 'apex.optimizers',
                 'torch_scatter', 'simple_knn'):
        sys.modules.pop(name, None)


@pytest.mark.skipif(not REF.exists(), reason='reference tree only exists in the build container')
def test_reference_extension_frontends_import_against_the_shims():
    code = r'''
import sys, types, torch
sys.path[:0] = [%r, %r, %r]
sys.path.insert(0, %r)
from make_golden import install_shims
install_shims()
import Framework
Framework.config = Framework.ConfigWrapper.fromDict({'GLOBAL': {'RANDOM_SEED': 0, 'ANOMALY_DETECTION': False, 'GPU_INDICES': None,
    'DEFAULT_DEVICE': torch.device('cpu'), 'METHOD_TYPE': 'InstantNGP'}, 'TRAINING': {'WANDB': {'ACTIVATE': False}}})
import Thirdparty.TinyCudaNN as tcnn
import Thirdparty.DiffGaussianRasterization as dgr
import Thirdparty.FusedSSIM as fs
import Thirdparty.Apex as apex
import Thirdparty.TorchScatter as ts
import Thirdparty.SimpleKNN as knn
assert callable(knn.compute_mean_squared_knn_distances)
import CudaUtils.MortonEncoding as me
import Methods.InstantNGP.VolumeRenderingV2 as vr
assert tcnn.NetworkWithInputEncoding.__module__.startswith('nerficg_amd')
assert dgr.GaussianRasterizer.__module__.startswith('nerficg_amd') and fs.fused_ssim.__module__.startswith('nerficg_amd')
assert apex.FusedAdam.__module__.startswith('nerficg_amd') and callable(me.morton_encode)
# the reference's OWN autograd layer on top of the HIP ops
assert vr.RayMarcher.__module__.endswith('custom_functions') and vr.raymarching_train.__module__.startswith('nerficg_amd')
print('ok')
''' % (str(SHIMS), str(ROOT), str(REF), str(ROOT / 'tests' / 'golden'))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env={**os.environ, 'PYTHONPATH': ''})
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stderr[-2000:]
