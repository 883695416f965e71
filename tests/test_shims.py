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


def _run_in_reference(body: str):
    """Runs `body` in a subprocess with the reference tree importable on top of nerficg_amd/shims (CPU mode, Framework configured)."""
    code = r'''
import sys, types, math, torch, numpy as np
sys.path[:0] = [%r, %r, %r]
sys.path.insert(0, %r)
from make_golden import install_shims
install_shims()
import Framework
Framework.config = Framework.ConfigWrapper.fromDict({'GLOBAL': {'RANDOM_SEED': 1618033989, 'ANOMALY_DETECTION': False, 'GPU_INDICES': None,
    'DEFAULT_DEVICE': torch.device('cpu'), 'METHOD_TYPE': 'InstantNGP'}, 'TRAINING': {'WANDB': {'ACTIVATE': False}, 'MODEL_NAME': 'probe'}, 'MODEL': {},
    'RENDERER': {}})
import torchmetrics.functional.image as _tfi   # three more names of absent metric packages that the method packages import at module level
for _n in ('structural_similarity_index_measure', 'multiscale_structural_similarity_index_measure', 'learned_perceptual_image_patch_similarity'):
    setattr(_tfi, _n, lambda *a, **k: None)
''' % (str(SHIMS), str(ROOT), str(REF), str(ROOT / 'tests' / 'golden')) + body + "\nprint('ok')\n"
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env={**os.environ, 'PYTHONPATH': ''})
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.skipif(not REF.exists(), reason='reference tree only exists in the build container')
def test_reference_instant_ngp_model_builds_over_the_shims():
    """The reference's OWN InstantNGPModel(name).build() (src/Methods/InstantNGP/Model.py:46-123) constructed on the drop-in tinycudann: the
    flat parameter vectors have the sizes and the MLP-first layout its weight decay slices on (:38-44, :80-89: 3072; :115: 7168), the buffers
    the sizes the ray marcher expects, and yaml-settable architecture keys reach the replacement (or are refused by name)."""
    _run_in_reference(r'''
from Methods.InstantNGP.Model import InstantNGPModel
import Thirdparty.TinyCudaNN as tcnn
m = InstantNGPModel('probe').build()
assert type(m.encoding_xyz).__module__.startswith('nerficg_amd')
assert m.n_params_encoding_mlp == 3072 and len(m.color_mlp_with_encoding.params) == 7168 and m.n_mlp_params == 10240
assert m.encoding_xyz.params.dtype == torch.float32 and m.encoding_xyz.params.dim() == 1
n_table = sum(min(2 ** 19, -(-(math.ceil(16 * 1.3819128799677760 ** l - 1) + 1) ** 3 // 8) * 8) for l in range(16))
assert n_table == 6098120 and m.encoding_xyz.params.numel() == 3072 + 2 * n_table          # SURVEY App. C.1
assert m.encoding_xyz.n_output_dims == 16 and m.color_mlp_with_encoding.n_output_dims == 3
assert m.occupancy_grid.shape == (1, 128 ** 3) and m.occupancy_bitfield.shape == (128 ** 3 // 8,) and m.cascades == 1
assert m.encoding_xyz.jit_fusion and m.color_mlp_with_encoding.jit_fusion                    # Model.py:116-120
wd = m.weight_decay_mlp()                                                                    # the reference's own slice arithmetic on our layout
ref = (m.encoding_xyz.params[:3072].double().pow(2).sum() + m.color_mlp_with_encoding.params.double().pow(2).sum()) / 10240
assert abs(float(wd) - float(ref)) < 1e-6 * float(ref)
sd = m.state_dict()
assert set(sd) == {'occupancy_grid', 'occupancy_bitfield', 'encoding_xyz.params', 'color_mlp_with_encoding.params'}
# a yaml that changes the architecture: supported keys take effect ...
Framework.config.MODEL = Framework.ConfigWrapper.fromDict({'SCALE': 2.0, 'RESOLUTION': 64, 'HASHGRID_LOG2_SIZE': 17, 'HASHGRID_TARGET_RESOLUTION': 1024,
                                                           'N_COLOR_LAYERS': 1})
m2 = InstantNGPModel('probe2').build()
assert m2.cascades == 3 and m2.occupancy_grid.shape == (3, 64 ** 3) and len(m2.color_mlp_with_encoding.params) == 64 * 32 + 16 * 64
assert m2.encoding_xyz.grid_cfg['log2_hashmap_size'] == 17
# ... also the grid / SH keys (round 6): 8 levels x 4 features, SH degree 3 -- the reference's own slice arithmetic (Model.py:80-89,115) on the module's layout
Framework.config.MODEL = Framework.ConfigWrapper.fromDict({'HASHGRID_N_LEVELS': 8, 'HASHGRID_N_FEATURES_PER_LEVEL': 4, 'DIR_SH_ENCODING_DEGREE': 3, 'HASHGRID_LOG2_SIZE': 15})
m3 = InstantNGPModel('probe_keys').build()
assert m3.encoding_xyz.grid_cfg['n_levels'] == 8 and m3.encoding_xyz.n_features == 4 and m3.color_mlp_with_encoding.sh_degree == 3
assert m3.n_params_encoding_mlp == m3.encoding_xyz.n_mlp_params == 64 * 32 + 16 * 64 and len(m3.color_mlp_with_encoding.params) == 7168
assert m3.encoding_xyz.params.numel() == m3.encoding_xyz.n_mlp_params + 4 * m3.encoding_xyz.grid_offsets[-1]
assert not m3.encoding_xyz.default_layout and not m3.color_mlp_with_encoding.default_layout
Framework.config.MODEL = Framework.ConfigWrapper.fromDict({'HASHGRID_N_LEVELS': 8, 'HASHGRID_N_FEATURES_PER_LEVEL': 2})      # 16 encoded inputs: a (64, 16) first layer in the master
m4 = InstantNGPModel('probe_narrow').build()
assert m4.n_params_encoding_mlp == m4.encoding_xyz.n_mlp_params == 64 * 16 + 16 * 64
Framework.config.MODEL = Framework.ConfigWrapper.fromDict({'HASHGRID_N_FEATURES_PER_LEVEL': 4})      # 16 x 4 = 64 encoded inputs: refused by key name
try:
    InstantNGPModel('probe_wide').build()
    raise SystemExit('a 64-input first layer was accepted')
except RuntimeError as e:
    assert 'HASHGRID_N_LEVELS' in str(e) and 'HASHGRID_N_FEATURES_PER_LEVEL' in str(e), e
# ... and an unsupported one is refused with the key's name, not silently ignored
Framework.config.MODEL = Framework.ConfigWrapper.fromDict({'N_DENSITY_NEURONS': 128})
try:
    InstantNGPModel('probe3').build()
    raise SystemExit('a 128-neuron density network was accepted')
except RuntimeError as e:
    assert 'n_neurons' in str(e), e
''')


@pytest.mark.skipif(not REF.exists(), reason='reference tree only exists in the build container')
def test_reference_method_plugins_resolve_and_construct_over_the_shims():
    """SURVEY 8(b) "Method plugin": the reference's own registry (src/Implementations.py:27-65) imports all three method packages over the
    shims, every package exposes MODEL / RENDERER / TRAINING_INSTANCE, and get_model / get_renderer construct the InstantNGP pair on CPU
    (constructors need no kernel); the 3DGS renderer refuses CPU mode exactly like the reference (Renderer.py:32-33)."""
    _run_in_reference(r'''
import Implementations
assert {'InstantNGP', 'GaussianSplatting', 'NeRF'} <= set(Implementations.Methods.options)
for method in ('InstantNGP', 'GaussianSplatting', 'NeRF'):
    pkg = Implementations.Methods.import_method(method)
    assert all(hasattr(pkg, a) for a in ('MODEL', 'RENDERER', 'TRAINING_INSTANCE')), method
model = Implementations.Methods.get_model('InstantNGP', name='probe')                       # = MODEL(name).build()
assert type(model).__name__ == 'InstantNGPModel' and model.n_mlp_params == 10240
Framework.config.GLOBAL.GPU_INDICES = []   # CPU mode as a list: InstantNGP's component takes len() of it (Renderer.py:26)
renderer = Implementations.Methods.get_renderer('InstantNGP', model)                       # = RENDERER(model)
assert abs(renderer.density_threshold - 0.01 * 1024 / 3 ** 0.5) < 1e-9 and renderer.ray_rendering_component.model is model
import Methods.InstantNGP.Renderer as R
assert R.VolumeRenderingCuda.raymarching_train.__module__.startswith('nerficg_amd')          # the renderer's native module IS this library
Framework.config.GLOBAL.METHOD_TYPE = 'GaussianSplatting'
gs_model = Implementations.Methods.get_model('GaussianSplatting', name='probe')
try:
    Implementations.Methods.get_renderer('GaussianSplatting', gs_model)
    raise SystemExit('3DGS renderer constructed in CPU mode')
except Framework.RendererError:
    pass
''')


@pytest.mark.skipif(not REF.exists(), reason='reference tree only exists in the build container')
def test_reference_gaussians_and_raster_settings_over_the_shims():
    """The reference's own Gaussians container (Model.py:18-150) with its optimizer built on the drop-in FusedAdam, and the settings tuple its
    renderer marshals (Renderer.py:60-74) accepted by the drop-in rasterizer's settings type."""
    _run_in_reference(r'''
Framework.config.GLOBAL.METHOD_TYPE = 'GaussianSplatting'
import Thirdparty.Apex as apex
import Thirdparty.DiffGaussianRasterization as dgr
assert apex.FusedAdam.__module__.startswith('nerficg_amd')
from Methods.GaussianSplatting.Model import Gaussians
g = Gaussians(3, False)
n = 500
gen = torch.Generator().manual_seed(0)
g._positions = torch.nn.Parameter(torch.rand(n, 3, generator=gen))
g._features_dc = torch.nn.Parameter(torch.rand(n, 1, 3, generator=gen)); g._features_rest = torch.nn.Parameter(torch.zeros(n, 15, 3))
g._scales = torch.nn.Parameter(torch.full((n, 3), -4.0)); g._rotations = torch.nn.Parameter(torch.tensor([[1.0, 0, 0, 0]]).repeat(n, 1))
g._opacities = torch.nn.Parameter(torch.zeros(n, 1))
class Args: pass
a = Args()
a.LEARNING_RATE_POSITION_INIT, a.LEARNING_RATE_POSITION_FINAL, a.LEARNING_RATE_POSITION_MAX_STEPS = 0.00016, 0.0000016, 30000
a.LEARNING_RATE_FEATURE, a.LEARNING_RATE_OPACITY, a.LEARNING_RATE_SCALING, a.LEARNING_RATE_ROTATION, a.PERCENT_DENSE = 0.0025, 0.05, 0.005, 0.001, 0.01
g.training_cameras_extent = 4.0
g.training_setup(a)
assert type(g.optimizer).__module__.startswith('nerficg_amd')
assert [grp['name'] for grp in g.optimizer.param_groups] == ['positions', 'f_dc', 'f_rest', 'opacities', 'scales', 'rotations']
assert g.get_features.shape == (n, 16, 3) and torch.allclose(g.get_opacities, torch.full((n, 1), 0.5)) and torch.allclose(g.get_scales, torch.full((n, 3), math.exp(-4.0)))
g.update_learning_rate(10)
s = dgr.GaussianRasterizationSettings(image_height=64, image_width=96, tanfovx=0.5, tanfovy=0.33, bg=torch.zeros(3), scale_modifier=1.0,
                                      viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=3, campos=torch.zeros(3), prefiltered=False, debug=False)
r = dgr.GaussianRasterizer(raster_settings=s)
assert r.raster_settings.image_width == 96
try:
    r(means3D=g.get_positions, means2D=torch.zeros_like(g.get_positions), shs=g.get_features, opacities=g.get_opacities, scales=g.get_scales, rotations=g.get_rotations)
    raise SystemExit('the rasterizer ran on CPU tensors')
except RuntimeError as e:
    assert 'CUDA' in str(e)     # no CPU fallback: the product path fails loudly without a GPU
''')


@pytest.mark.skipif(not REF.exists(), reason='reference tree only exists in the build container')
def test_reference_autograd_layer_sits_on_the_hip_ops():
    """The reference's own custom_functions.py classes bound to this library's ops: applying one to CPU tensors reaches nerficg_amd's input check."""
    _run_in_reference(r'''
import Methods.InstantNGP.VolumeRenderingV2 as vr
assert vr.VolumeRenderer.__module__.endswith('custom_functions') and vr.composite_train_fw.__module__.startswith('nerficg_amd')
z = torch.zeros
try:
    vr.RayAABBIntersector.apply(z(4, 3), z(4, 3), z(1, 3), z(1, 3), 1)
    raise SystemExit('ray_aabb_intersect ran on CPU tensors')
except RuntimeError as e:
    assert 'must be a CUDA tensor' in str(e)
''')


@pytest.mark.skipif(not REF.exists(), reason='reference tree only exists in the build container')
def test_reference_host_code_runs_over_the_shims_and_reproduces_the_orchestration_fixture():
    """The reference's OWN InstantNGPRenderer / InstantNGPRayRenderingComponent / custom_functions.py (src/Methods/InstantNGP/Renderer.py:30-180) executed
    on CPU with the native ops patched to the oracle behind their own signatures (tests/oracle_ops.py): (1) the committed fixture
    tests/golden/ingp_orchestration.npz is what this run produces, bit for bit -- the fixture IS the reference's orchestration, not a restatement;
    (2) the native calls it makes are the ones the GPU mirror (nerficg_amd/instant_ngp.py, compared with the fixture in tests/test_gpu_render_parity.py)
    has to account for: one box test, then march -> grid network -> SH network -> compositing for a training batch, and the same four per round of the
    alive-ray loop, with the argument shapes / dtypes / scalars of binding.cpp:234-250."""
    code = r'''
import sys, json
import numpy as np
sys.path[:0] = [%r, %r]
import make_golden
blob = make_golden.run_ingp_orchestration()
ref = np.load(%r)
for k in ref.files:
    a, b = ref[k], blob[k]
    if k.startswith('trace_'):
        assert str(a) == str(b), k
    else:
        assert np.array_equal(np.asarray(a), np.asarray(b)), (k, np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())
tt = json.loads(str(blob['trace_train'])); te = json.loads(str(blob['trace_eval']))
n = int(blob['origin'].shape[0]); m = int(blob['train_rm_samples'])
assert [t[0] for t in tt] == ['ray_aabb_intersect', 'raymarching_train', 'network_forward:grid', 'network_forward:sh_identity', 'composite_train_fw']
assert tt[0][1] == [[[n, 3], 'float32'], [[n, 3], 'float32'], [[1, 3], 'float32'], [[1, 3], 'float32'], 1]
assert tt[1][1][2] == [[n, 2], 'float32'] and tt[1][1][4:7] == [1, 0.5, 0.0] and tt[1][1][7] == [[n], 'float32'] and tt[1][1][8:] == [128, 1024]
assert tt[2][1] == [[[m, 3], 'float32']] and tt[3][1][0][0] == [m, 19] and tt[4][1][:2] == [[[m], 'float32'], [[m, 3], 'float32']] and tt[4][1][5] == 1e-4
assert te[0][0] == 'ray_aabb_intersect' and (len(te) - 1) %% 4 in (0, 1)     # a last march that finds no sample ends the loop (Renderer.py:118-120)
if (len(te) - 1) %% 4 == 1:
    assert te[-1][0] == 'raymarching_test'
rounds = [te[1 + 4 * r: 5 + 4 * r] for r in range((len(te) - 1) // 4)]
assert all([c[0] for c in r] == ['raymarching_test', 'network_forward:grid', 'network_forward:sh_identity', 'composite_test_fw'] for r in rounds)
samples = [r[0][1][-1] for r in rounds]            # N_samples per round: max(min(n_rays // n_alive, 64), 1) -- grows as rays finish (Renderer.py:108)
assert samples[0] == 1 and samples == sorted(samples) and max(samples) <= 64 and sum(samples) <= 1024 + 64
alive = [r[0][1][3][0][0] for r in rounds]           # alive rays per round shrink
assert alive[0] == n and alive == sorted(alive, reverse=True)
''' % (str(ROOT / 'tests' / 'golden'), str(ROOT), str(ROOT / 'tests' / 'golden' / 'ingp_orchestration.npz'))
    r = subprocess.run([sys.executable, '-c', code + "\nprint('ok')\n"], capture_output=True, text=True, env={**os.environ, 'PYTHONPATH': ''})
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), (r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.skipif(not REF.exists(), reason='reference tree only exists in the build container')
def test_rest_step_schedule_follows_the_reference_trainers_own_callbacks():
    """nerficg_amd.gaussian_splatting.rest_step_schedule against the reference's code: the callbacks of GaussianSplattingTrainer with their registered priorities /
    start / end / stride (Trainer.py:76-128), the skip rule of the training loop (Base/Trainer.py:238-243, restated), and the reference's OWN `densify` and
    `reset_opacities` bodies executed on a recording stand-in for the model -- an iteration is 'clean' when nothing that runs between loss.backward() (priority 100)
    and optimizer.step() (priority 70) replaces the f_rest parameter, i.e. when densify_and_prune is not called."""
    _run_in_reference(r'''
Framework.config.GLOBAL.METHOD_TYPE = 'GaussianSplatting'
from Methods.GaussianSplatting.Trainer import GaussianSplattingTrainer as T
from nerficg_amd.gaussian_splatting import rest_step_schedule
cfg = dict(T.get_default_parameters()) if not isinstance(T.get_default_parameters(), dict) else T.get_default_parameters()
cfg = {k: cfg[k] for k in ('DENSIFY_START_ITERATION', 'DENSIFY_END_ITERATION', 'DENSIFICATION_INTERVAL', 'OPACITY_RESET_INTERVAL', 'NUM_ITERATIONS', 'DENSIFY_GRAD_THRESHOLD')}
assert (cfg['DENSIFY_START_ITERATION'], cfg['DENSIFY_END_ITERATION'], cfg['DENSIFICATION_INTERVAL'], cfg['OPACITY_RESET_INTERVAL']) == (500, 15000, 100, 3000)
between = [f for f in (getattr(T, n) for n in dir(T)) if callable(f) and getattr(f, 'callback_type', None) == 0 and 70 < f.priority < 100]
assert sorted(f.__name__ for f in between) == ['densify', 'reset_opacities', 'reset_opacities_white_background']
assert T.training_iteration.priority == 100 and T.perform_optimizer_step.priority == 70
calls = []
gaussians = types.SimpleNamespace(densify_and_prune=lambda *a: calls.append('densify_and_prune'), reset_opacities=lambda: calls.append('reset_opacities'))
me = types.SimpleNamespace(model=types.SimpleNamespace(gaussians=gaussians), **cfg)
value = lambda v: cfg[v] if isinstance(v, str) else v
clean = rest_step_schedule(cfg['DENSIFY_START_ITERATION'], cfg['DENSIFY_END_ITERATION'], cfg['DENSIFICATION_INTERVAL'])
dirty = 0
dataset = types.SimpleNamespace(default_camera=types.SimpleNamespace(background_color=torch.zeros(3)))
for it in range(cfg['NUM_ITERATIONS']):
    calls.clear()
    for f in sorted(between, key=lambda f: -f.priority):
        start, end, stride = value(f.start_iteration), value(f.end_iteration), value(f.iteration_stride)
        if (start is not None and it < start) or (end is not None and it > end) or (stride is not None and (it - (start or 0)) % stride != 0):
            continue
        f(me, it, dataset)
    replaced_f_rest = 'densify_and_prune' in calls          # prune_points rebuilds all six groups (Model.py:157-167); reset_opacities only the opacities (:152-155)
    assert clean(it) == (not replaced_f_rest), it
    dirty += replaced_f_rest
assert dirty == 144        # iterations 600, 700, ..., 14 900
''')
