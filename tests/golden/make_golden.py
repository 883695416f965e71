#!/usr/bin/env python3
"""Generate golden input/output vectors from the *reference's own* pure-PyTorch code.

Runs ONLY in the build container (needs /root/reference). The reference never travels to the GPU box;
what travels are the small .npz fixtures this script writes next to itself. Recipe = SURVEY.md Appendix B.

Vectors produced (reference symbol -> fixture):
  PerspectiveCamera.compute_local_ray_directions / View.get_rays   -> raygen.npz          (a1, a2)
  PerspectiveCamera.get_projection_matrix + GS settings marshalling -> projection.npz      (a24)
  FrequencyEncoding.forward                                         -> freqenc.npz         (a5)
  generate_samples / generate_samples_from_pdf / integrate_samples  -> nerf_sampling.npz   (a7-a9)
  convert_sh_features / build_covariances / quaternion_to_rotation  -> gs_utils.npz        (a26)
  LRDecayPolicy, apply_background_color, RandomSequentialSampler    -> misc.npz
  NeRFBlock.forward / NeRFRayRenderingComponent.forward (tiny model) -> nerf_render.npz         (a6, a10)
  Gaussians.densify_and_prune / add_densification_stats / as_ply_dict, adam_utils -> gs_densify.npz (8f rank 3, 4)
  torch.autograd through integrate_samples (dL/dsigma, dL/drgb)       -> composite_bw.npz    (a17 backward)
  View.project_points + GS settings marshalling of ONE off-centre camera -> gs_projection.npz (a22: screen positions / depths of the rasterizer)
  GaussianSplattingModel.save after bake_activations                 -> gs_reference_checkpoint.pt (8f rank 4)
"""
import importlib.util
import math
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path('/root/reference/src')
OUT = Path(__file__).resolve().parent


def install_shims():
    class Munch(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v

        def __delattr__(self, k):
            del self[k]

        @classmethod
        def fromDict(cls, d):
            out = cls()
            for k, v in d.items():
                out[k] = cls.fromDict(v) if isinstance(v, dict) else v
            return out

        def toDict(self):
            return {k: (v.toDict() if isinstance(v, Munch) else v) for k, v in self.items()}

        def copy(self):
            return type(self).fromDict(self.toDict())

    m = types.ModuleType('munch')
    m.Munch = Munch
    sys.modules['munch'] = m

    def stub(name, **attrs):
        mod = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(mod, k, v)
        sys.modules[name] = mod
        return mod

    stub('natsort', natsorted=sorted)
    stub('plyfile', PlyData=object, PlyElement=object)
    tv = stub('torchvision')
    tv.io = stub('torchvision.io')
    tv.utils = stub('torchvision.utils', _normalized_flow_to_image=None)
    tm = stub('torchmetrics', Metric=type('Metric', (), {}))
    tm.image = stub('torchmetrics.image')
    tm.functional = stub('torchmetrics.functional', peak_signal_noise_ratio=None)
    tm.functional.image = stub('torchmetrics.functional.image', peak_signal_noise_ratio=None)


def main():
    install_shims()
    sys.path.insert(0, str(REF))
    import Framework
    defaults = dict(RANDOM_SEED=1618033989, ANOMALY_DETECTION=False)
    Framework.config = Framework.ConfigWrapper.fromDict({
        'GLOBAL': {**defaults, 'GPU_INDICES': None, 'DEFAULT_DEVICE': torch.device('cpu'), 'METHOD_TYPE': 'NeRF'},
        'TRAINING': {'WANDB': {'ACTIVATE': False}},
    })
    from Cameras.Perspective import PerspectiveCamera
    from Cameras.utils import SharedCameraSettings, fov_to_focal, quaternion_to_rotation_matrix
    from Datasets.utils import View, apply_background_color
    from Methods.NeRF.utils import FrequencyEncoding, generate_samples, generate_samples_from_pdf, integrate_samples
    from Optim.lr_utils import LRDecayPolicy
    from Optim.Samplers.utils import RandomSequentialSampler

    spec = importlib.util.spec_from_file_location('gs_utils', REF / 'Methods/GaussianSplatting/utils.py')
    gs_utils = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gs_utils)

    g = torch.Generator().manual_seed(0)

    # ---------------------------------------------------------------- raygen (a1, a2)
    def lookat_pose(theta, phi, radius):
        # camera on an orbit, looking at the origin; colmap convention: x right, y down, z forward
        pos = np.array([radius * math.cos(phi) * math.cos(theta), radius * math.sin(phi), radius * math.cos(phi) * math.sin(theta)])
        fwd = -pos / np.linalg.norm(pos)
        up = np.array([0.0, -1.0, 0.0])
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        c2w = np.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, pos
        return c2w.astype(np.float64)

    ray_cases = {}
    for tag, (w, h) in {'64': (64, 64), '800': (800, 800), '53x31': (53, 31)}.items():
        settings = SharedCameraSettings(background_color=torch.tensor([1.0, 1.0, 1.0]), near_plane=2.0, far_plane=6.0)
        focal = fov_to_focal(0.6911112070083618) * w
        cam = PerspectiveCamera(shared_settings=settings, width=w, height=h, focal_x=focal, focal_y=focal)
        if tag == '53x31':
            cam = PerspectiveCamera(shared_settings=settings, width=w, height=h, focal_x=focal, focal_y=focal * 1.1,
                                    center_x=w / 2 + 1.25, center_y=h / 2 - 0.75)
        c2w = lookat_pose(0.7, 0.4, 4.0311)
        view = View(camera=cam, camera_index=0, frame_idx=0, global_frame_idx=0, c2w=c2w)
        rays = view.get_rays()
        local = cam.compute_local_ray_directions()
        if tag == '800':  # keep fixture small: corner / centre / a strided subset
            idx = torch.cat([torch.tensor([0, 799, 800 * 799, 800 * 800 - 1, 800 * 400 + 400]), torch.arange(0, 640000, 6397)])
        else:
            idx = torch.arange(w * h)
        ray_cases[f'{tag}_intr'] = np.array([w, h, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y], dtype=np.float64)
        ray_cases[f'{tag}_c2w'] = c2w
        ray_cases[f'{tag}_idx'] = idx.numpy()
        ray_cases[f'{tag}_local'] = local[idx].numpy()
        ray_cases[f'{tag}_origin'] = rays.origin[idx].numpy()
        ray_cases[f'{tag}_direction'] = rays.direction[idx].numpy()
        ray_cases[f'{tag}_view_direction'] = rays.view_direction[idx].numpy()
    # View.project_points (Datasets/utils.py:1040-1044 -> Perspective.cam_to_screen :39-52), used by carve_occupancy_grid (a19)
    settings = SharedCameraSettings(background_color=torch.tensor([1.0, 1.0, 1.0]), near_plane=0.2, far_plane=1000.0)
    cam = PerspectiveCamera(shared_settings=settings, width=53, height=31, focal_x=60.0, focal_y=66.0, center_x=53 / 2 + 1.25, center_y=31 / 2 - 0.75)
    c2w = lookat_pose(2.1, -0.3, 1.7)
    view = View(camera=cam, camera_index=0, frame_idx=0, global_frame_idx=0, c2w=c2w)
    pts = (torch.rand(400, 3, generator=g) * 2 - 1) * 1.5
    xy, depth, inside = view.project_points(pts)
    ray_cases.update(proj_intr=np.array([53, 31, 60.0, 66.0, cam.center_x, cam.center_y, 0.2, 1000.0]), proj_c2w=c2w, proj_pts=pts.numpy(),
                     proj_xy=xy.numpy(), proj_depth=depth.numpy(), proj_in_frustum=inside.numpy())
    np.savez_compressed(OUT / 'raygen.npz', **ray_cases)

    # ---------------------------------------------------------------- projection / GS settings (a24)
    settings = SharedCameraSettings(background_color=torch.tensor([0.0, 0.0, 0.0]), near_plane=0.01, far_plane=100.0)
    cam = PerspectiveCamera(shared_settings=settings, width=1297, height=840, focal_x=1160.3, focal_y=1158.9, center_x=650.1, center_y=418.7)
    c2w = lookat_pose(1.1, 0.3, 3.5)
    view = View(camera=cam, camera_index=0, frame_idx=0, global_frame_idx=0, c2w=c2w)
    P = cam.get_projection_matrix()
    w2cT = view.w2c.T
    np.savez_compressed(
        OUT / 'projection.npz',
        intr=np.array([cam.width, cam.height, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, 0.01, 100.0], dtype=np.float64),
        c2w=c2w, P=P.numpy(), P_invz=cam.get_projection_matrix(invert_z=True).numpy(),
        viewmatrix=w2cT.numpy(), projmatrix=(w2cT @ P.T).numpy(),
        tanfov=np.array([cam.width / cam.focal_x * 0.5, cam.height / cam.focal_y * 0.5], dtype=np.float64),
        campos=view.position.numpy(), viewport=cam.get_viewport_transform().numpy(),
    )

    # ---------------------------------------------------------------- frequency encoding (a5)
    x = torch.rand(97, 3, generator=g) * 4 - 2
    np.savez_compressed(
        OUT / 'freqenc.npz', x=x.numpy(),
        pos10=FrequencyEncoding(10, True)(x).numpy(), dir4=FrequencyEncoding(4, True)(x).numpy(),
        noappend6=FrequencyEncoding(6, False)(x).numpy(),
    )

    # ---------------------------------------------------------------- NeRF sampling / integration (a7-a9)
    class _R:  # the minimal RayBatch surface generate_samples touches
        def __init__(self, n):
            self.dtype, self.device, self.n = torch.float32, torch.device('cpu'), n

        def __len__(self):
            return self.n

    n_rays, n_coarse, n_fine = 37, 64, 192
    depth = generate_samples(_R(n_rays), n_coarse, 2.0, 6.0, False)
    torch.manual_seed(1)
    depth_rand = generate_samples(_R(n_rays), n_coarse, 2.0, 6.0, True)
    torch.manual_seed(1)
    depth_rand_u = torch.rand(n_rays, n_coarse)
    dirs = torch.randn(n_rays, 3, generator=g)
    dens = torch.rand(n_rays, n_coarse, generator=g) * 3
    dens[3] = 0.0            # empty ray (T stays 1 -> depth 0 branch)
    dens[5] *= 40.0          # saturating ray
    cols = torch.rand(n_rays, n_coarse, 3, generator=g)
    bg = torch.tensor([1.0, 0.5, 0.25])
    rgb, dpt, alpha, w = integrate_samples(depth, dirs, dens, cols, bg)
    rgb_nobg, _, _, _ = integrate_samples(depth, dirs, dens, cols, None)
    fine = generate_samples_from_pdf(depth, w, n_fine, False)
    torch.manual_seed(2)
    fine_rand = generate_samples_from_pdf(depth, w, n_fine, True)
    torch.manual_seed(2)
    fine_rand_u = torch.rand(n_rays, n_fine)
    np.savez_compressed(
        OUT / 'nerf_sampling.npz', depth=depth.numpy(), depth_rand=depth_rand.numpy(), depth_rand_u=depth_rand_u.numpy(),
        dirs=dirs.numpy(), dens=dens.numpy(), cols=cols.numpy(), bg=bg.numpy(), rgb=rgb.numpy(), rgb_nobg=rgb_nobg.numpy(),
        depth_out=dpt.numpy(), alpha=alpha.numpy(), weights=w.numpy(), fine=fine.numpy(), fine_rand=fine_rand.numpy(),
        fine_rand_u=fine_rand_u.numpy(),
    )

    # ---------------------------------------------------------------- GS utils (a26)
    n = 211
    sh = torch.randn(n, 3, 16, generator=g) * 0.4
    vd = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)
    scales = torch.exp(torch.randn(n, 3, generator=g) * 0.5 + math.log(0.01))
    quats = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1)
    cov = gs_utils.build_covariances(scales, quats)
    out = dict(sh=sh.numpy(), view_dirs=vd.numpy(), scales=scales.numpy(), quats=quats.numpy(), cov=cov.numpy(),
               cov_upper=gs_utils.extract_upper_triangular_matrix(cov).numpy(),
               rot=quaternion_to_rotation_matrix(quats, normalize=False).numpy(),
               rot_unnormalized_in=(quats * 1.7).numpy(), rot_normalized=quaternion_to_rotation_matrix(quats * 1.7, normalize=True).numpy())
    for deg in range(4):
        out[f'rgb_deg{deg}'] = gs_utils.convert_sh_features(sh.clone(), vd, deg).numpy()
    out['rgb_to_sh0'] = gs_utils.rgb_to_sh0(torch.linspace(0, 1, 7)).numpy()
    np.savez_compressed(OUT / 'gs_utils.npz', **out)

    # ---------------------------------------------------------------- misc host logic
    pol = LRDecayPolicy(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_steps=100, lr_delay_mult=0.01, max_steps=30000)
    its = np.array([0, 1, 50, 99, 100, 1000, 15000, 29999, 30000, 40000], dtype=np.int64)
    torch.manual_seed(0)
    sampler = RandomSequentialSampler(num_elements=1000)
    draws = [sampler.get(300).numpy().copy() for _ in range(5)]  # wraps (reshuffle) on the 4th draw
    raw = torch.rand(5, 3, generator=g)
    al = torch.rand(5, 1, generator=g)
    np.savez_compressed(
        OUT / 'misc.npz', lr_its=its, lr_vals=np.array([pol(int(i)) for i in its]),
        sampler_draws=np.stack(draws), bgc_raw=raw.numpy(), bgc_alpha=al.numpy(),
        bgc_out=apply_background_color(raw, al, torch.tensor([0.2, 0.4, 0.6]), is_chw=False).numpy(),
    )
    # ---------------------------------------------------------------- NeRF block + hierarchical renderer (a6, a10), tiny width
    from Methods.NeRF.Model import NeRFBlock
    from Methods.NeRF.Renderer import NeRFRayRenderingComponent
    from Datasets.utils import RayBatch
    torch.manual_seed(0)
    kw = dict(n_layers=8, n_color_layers=1, n_features=32, n_frequencies_position=10, n_frequencies_direction=4, encoding_append_input=True,
              input_skips=[5], activation_function='relu')
    coarse, fine = NeRFBlock(**kw), NeRFBlock(**kw)
    nr = 23
    origin = torch.tensor([0.0, 0.0, -4.0]).expand(nr, 3).contiguous()
    direction = torch.cat([torch.randn(nr, 2, generator=g) * 0.15, torch.ones(nr, 1)], dim=-1)
    vdir = torch.nn.functional.normalize(direction, dim=-1)
    rays = RayBatch(origin=origin, direction=direction, view_direction=vdir, timestamp=torch.zeros(nr, 1))
    settings = SharedCameraSettings(background_color=torch.tensor([1.0, 1.0, 1.0]), near_plane=2.0, far_plane=6.0)
    cam = PerspectiveCamera(shared_settings=settings, width=8, height=8, focal_x=10.0, focal_y=10.0)
    comp = NeRFRayRenderingComponent(coarse, fine)
    with torch.no_grad():
        out = comp(rays, cam, ray_batch_size=10, n_samples_coarse_nerf=16, n_samples_nerf=24, randomize_samples=False, random_noise_density=0.0)
        pts = torch.randn(40, 3, generator=g)
        dens, col = fine(pts, vdir[:1].expand(40, 3).contiguous())
    blob = {f'coarse.{k}': v.numpy() for k, v in coarse.state_dict().items()}
    blob.update({f'fine.{k}': v.numpy() for k, v in fine.state_dict().items()})
    blob.update(origin=origin.numpy(), direction=direction.numpy(), view_direction=vdir.numpy(), block_pts=pts.numpy(), block_dens=dens.numpy(),
                block_col=col.numpy(), **{f'out_{k}': v.numpy() for k, v in out.items()})
    np.savez_compressed(OUT / 'nerf_render.npz', **blob)

    print('golden fixtures written to', OUT)
    for f in sorted(OUT.glob('*.npz')):
        print(f'  {f.name}: {f.stat().st_size} B')


def make_gs_densify():
    """3DGS densification bookkeeping -> gs_densify.npz: the reference's own Gaussians class (Model.py:18-284) and
    Optim/adam_utils.py run on CPU.  Model.py hard-codes device='cuda' in split(); `torch.zeros` is redirected to the CPU for the call and
    `torch.normal(mean, std)` is replaced by its definition mean + z * std with the standard-normal draws z recorded (they are an INPUT of
    the kernels: RNG streams are not part of the parity contract)."""
    install_shims()
    if str(REF) not in sys.path:
        sys.path.insert(0, str(REF))
    import Framework
    if getattr(Framework, 'config', None) is None or 'GLOBAL' not in Framework.config:
        Framework.config = Framework.ConfigWrapper.fromDict({
            'GLOBAL': {'RANDOM_SEED': 1618033989, 'ANOMALY_DETECTION': False, 'GPU_INDICES': None, 'DEFAULT_DEVICE': torch.device('cpu'),
                       'METHOD_TYPE': 'GaussianSplatting'}, 'TRAINING': {'WANDB': {'ACTIVATE': False}}})
    # the class only needs these two at densification time: the Morton extension and the proprietary kNN are absent here
    pk = types.ModuleType('CudaUtils'); pk.__path__ = []
    me = types.ModuleType('CudaUtils.MortonEncoding'); me.morton_encode = None
    sys.modules.setdefault('CudaUtils', pk); sys.modules.setdefault('CudaUtils.MortonEncoding', me)
    sys.modules.setdefault('Thirdparty.SimpleKNN', None)
    import Methods  # noqa: F401
    gp = types.ModuleType('Methods.GaussianSplatting'); gp.__path__ = [str(REF / 'Methods/GaussianSplatting')]  # skip the package __init__ (imports the rasterizer)
    sys.modules['Methods.GaussianSplatting'] = gp
    from Methods.GaussianSplatting.Model import Gaussians
    from Optim import adam_utils

    g = torch.Generator().manual_seed(20240612)
    P = 1000
    extent, percent_dense = 4.0, 0.01
    gs = Gaussians(3, False)
    gs.training_cameras_extent, gs.percent_dense = extent, percent_dense
    rnd = lambda *shape: torch.randn(*shape, generator=g)  # noqa: E731
    gs._positions = torch.nn.Parameter(rnd(P, 3) * 2.0)
    gs._scales = torch.nn.Parameter(torch.log(torch.exp(rnd(P, 3) * 1.2 - 3.2)))       # exp(scale): 0.003 .. 1.5, threshold 0.04, large > 0.4
    gs._rotations = torch.nn.Parameter(rnd(P, 4) * 1.5)
    gs._opacities = torch.nn.Parameter(rnd(P, 1) * 3.0 - 2.0)                           # sigmoid: some below 0.005
    gs._features_dc = torch.nn.Parameter(rnd(P, 1, 3))
    gs._features_rest = torch.nn.Parameter(rnd(P, 15, 3) * 0.1)
    names = {'positions': '_positions', 'f_dc': '_features_dc', 'f_rest': '_features_rest', 'opacities': '_opacities', 'scales': '_scales',
             'rotations': '_rotations'}
    gs.optimizer = torch.optim.Adam([{'params': [getattr(gs, a)], 'lr': 1e-3, 'name': n} for n, a in names.items()], lr=0.0, eps=1e-15)
    for a in names.values():
        getattr(gs, a).grad = rnd(*getattr(gs, a).shape) * 1e-2
    gs.optimizer.step()   # creates exp_avg / exp_avg_sq
    gs.optimizer.zero_grad()
    gs.densification_gradient_accum = torch.rand(P, 1, generator=g) * 1.2e-3
    gs.n_observations = torch.randint(0, 9, (P, 1), generator=g, dtype=torch.int32)
    blob = {f'in_{n}': getattr(gs, a).detach().numpy().copy() for n, a in names.items()}
    for n, a in names.items():
        st = gs.optimizer.state[getattr(gs, a)]
        blob[f'in_{n}_exp_avg'], blob[f'in_{n}_exp_avg_sq'] = st['exp_avg'].numpy().copy(), st['exp_avg_sq'].numpy().copy()
    blob['in_accum'], blob['in_n_obs'] = gs.densification_gradient_accum.numpy().copy(), gs.n_observations.numpy().copy()

    # add_densification_stats (Model.py:243-246)
    vsp = torch.zeros(P, 3, requires_grad=True)
    vsp.grad = rnd(P, 3) * 3e-4
    radii = torch.randint(-1, 30, (P,), generator=g, dtype=torch.int32).clamp_min(0) * (torch.rand(P, generator=g) > 0.3)
    gs.add_densification_stats(vsp, radii > 0)
    blob.update(vsp_grad=vsp.grad.numpy().copy(), radii=radii.to(torch.int32).numpy().copy(),
                stats_accum=gs.densification_gradient_accum.numpy().copy(), stats_n_obs=gs.n_observations.numpy().copy())

    # ply export of the untouched model (Model.py:275-318)
    ply = gs.as_ply_dict()['vertex']
    blob['ply_names'] = np.array(ply.dtype.names)
    blob['ply_rows'] = np.stack([ply[n] for n in ply.dtype.names], axis=1)

    # densify_and_prune (Model.py:226-241) on the CPU
    noise_log = []
    real_zeros, real_normal = torch.zeros, torch.normal

    def zeros_cpu(*a, **kw):
        kw.pop('device', None)
        return real_zeros(*a, **kw)

    def normal_recorded(mean, std):
        z = torch.randn(std.shape, generator=g)
        noise_log.append(z)
        return mean + z * std
    torch.zeros, torch.normal = zeros_cpu, normal_recorded
    try:
        gs.densify_and_prune(grad_threshold=0.0002, min_opacity=0.005, prune_large_gaussians=True)
    finally:
        torch.zeros, torch.normal = real_zeros, real_normal
    blob['noise'] = noise_log[0].numpy()
    blob.update(grad_threshold=np.float64(0.0002), min_opacity=np.float64(0.005), extent=np.float64(extent), percent_dense=np.float64(percent_dense))
    for n, a in names.items():
        blob[f'out_{n}'] = getattr(gs, a).detach().numpy().copy()
        st = gs.optimizer.state[getattr(gs, a)]
        blob[f'out_{n}_exp_avg'], blob[f'out_{n}_exp_avg_sq'] = st['exp_avg'].numpy().copy(), st['exp_avg_sq'].numpy().copy()

    # adam_utils: sort + indexed state reset + opacity reset on the densified model (adam_utils.py:6-18,64-98; Model.py:152-155)
    n_now = gs._positions.shape[0]
    order = torch.randperm(n_now, generator=g)
    sorted_params = adam_utils.sort_param_groups(gs.optimizer, order, ['positions', 'rotations'])
    idx = torch.randint(0, n_now, (50,), generator=g)
    adam_utils.reset_state(gs.optimizer, ['positions'], idx)
    blob.update(sort_order=order.numpy(), reset_idx=idx.numpy(), sorted_positions=sorted_params['positions'].detach().numpy().copy(),
                sorted_positions_exp_avg=gs.optimizer.state[sorted_params['positions']]['exp_avg'].numpy().copy(),
                sorted_rotations_exp_avg_sq=gs.optimizer.state[sorted_params['rotations']]['exp_avg_sq'].numpy().copy())
    gs.reset_opacities()
    blob['reset_opacities'] = gs.optimizer.param_groups[3]['params'][0].detach().numpy().copy()
    # ---------------------------------------------------------------- checkpoint written by the reference itself (Base/Model.py:103-111)
    # a small trained-and-baked 3DGS model: bake_activations (Model.py:248-273) needs the Morton extension for the final reordering, which is
    # absent here -> identity order stand-in (the ORDER is not what this fixture pins; keys, activations and baked covariances are)
    from Methods.GaussianSplatting import Model as gs_model_module
    from Methods.GaussianSplatting.Model import GaussianSplattingModel
    Framework.config.MODEL = Framework.ConfigWrapper.fromDict({})
    gs_model_module.morton_encode = lambda positions: torch.arange(positions.shape[0])
    model = GaussianSplattingModel('golden').build()
    n_small = 48
    for a in names.values():
        setattr(model.gaussians, a, torch.nn.Parameter(getattr(gs, a).detach()[:n_small].clone()))
    model.gaussians.optimizer = torch.optim.Adam([{'params': [getattr(model.gaussians, a)], 'name': n} for n, a in names.items()], lr=0.0)
    model.gaussians.densification_gradient_accum = torch.zeros(n_small, 1)
    model.gaussians.n_observations = torch.zeros(n_small, 1, dtype=torch.int32)
    blob['ckpt_raw_opacities'] = model.gaussians._opacities.detach().numpy().copy()
    model.gaussians.bake_activations()
    model.num_iterations_trained = 30000
    model.creation_date = '2026-01-01-00-00-00'
    model.output_directory = Path('output/GaussianSplatting/golden_2026-01-01-00-00-00')
    model.save(OUT / 'gs_reference_checkpoint.pt')
    print('gs_reference_checkpoint.pt:', (OUT / 'gs_reference_checkpoint.pt').stat().st_size, 'B;', model.gaussians._positions.shape[0], 'Gaussians')
    np.savez_compressed(OUT / 'gs_densify.npz', **blob)
    print('gs_densify.npz:', (OUT / 'gs_densify.npz').stat().st_size, 'B; rows', P, '->', n_now, '; split noise rows', noise_log[0].shape[0])


def make_composite_bw():
    """Gradients of the reference's OWN differentiable compositing (integrate_samples, src/Methods/NeRF/utils.py:112-136) -> composite_bw.npz:
    the pin of composite_train_bw (volumerendering.cu:87-202) on reference arithmetic instead of finite differences of the builder's forward.
    The loss is linear in the four outputs the CUDA backward receives gradients for -- opacity, depth SUM (sum w t: taken from the returned
    blending weights, so the reference's normalisation by alpha stays out), colour without background, per-sample weights -- with seeded
    coefficients that are stored next to the gradients.  final_delta is integrate_samples' own parameter: 0.05 keeps T > 0 on every ray (no
    early-out in either implementation, T_threshold = 0 on the other side); a second case uses the default 1e10."""
    install_shims()
    sys.path.insert(0, str(REF))
    import Framework
    Framework.config = Framework.ConfigWrapper.fromDict({
        'GLOBAL': {'RANDOM_SEED': 1618033989, 'ANOMALY_DETECTION': False, 'GPU_INDICES': None, 'DEFAULT_DEVICE': torch.device('cpu'), 'METHOD_TYPE': 'NeRF'},
        'TRAINING': {'WANDB': {'ACTIVATE': False}},
    })
    from Methods.NeRF.utils import integrate_samples
    g = torch.Generator().manual_seed(4242)
    n, s = 29, 48
    depth = torch.sort(torch.rand(n, s, generator=g) * 4 + 2, dim=-1).values
    dirs = torch.randn(n, 3, generator=g)
    blob = dict(depth=depth.numpy(), dirs=dirs.numpy())
    for tag, final_delta, scale in (('open', 0.05, 1.5), ('closed', 1.0e10, 1.5)):
        dens = (torch.rand(n, s, generator=g) * scale).requires_grad_(True)
        with torch.no_grad():
            dens[2] = 0.0          # an empty ray
        cols = torch.rand(n, s, 3, generator=g).requires_grad_(True)
        go, gd = torch.randn(n, generator=g), torch.randn(n, generator=g)
        gr, gw = torch.randn(n, 3, generator=g), torch.randn(n, s, generator=g)
        rgb, _, alpha, w = integrate_samples(depth, dirs, dens, cols, None, final_delta)
        loss = (go * alpha[:, 0]).sum() + (gd * (w * depth).sum(-1)).sum() + (gr * rgb).sum() + (gw * w).sum()
        d_dens, d_cols = torch.autograd.grad(loss, (dens, cols))
        blob.update({f'{tag}_final_delta': np.float32(final_delta), f'{tag}_dens': dens.detach().numpy(), f'{tag}_cols': cols.detach().numpy(),
                     f'{tag}_go': go.numpy(), f'{tag}_gd': gd.numpy(), f'{tag}_gr': gr.numpy(), f'{tag}_gw': gw.numpy(),
                     f'{tag}_d_dens': d_dens.numpy(), f'{tag}_d_cols': d_cols.numpy(), f'{tag}_weights': w.detach().numpy(),
                     f'{tag}_alpha': alpha.detach().numpy(), f'{tag}_rgb': rgb.detach().numpy()})
    np.savez_compressed(OUT / 'composite_bw.npz', **blob)
    print('composite_bw.npz:', (OUT / 'composite_bw.npz').stat().st_size, 'B')


def make_gs_projection():
    """The rasterizer's screen-space means and depths pinned on the reference's own projection code: for ONE camera with an off-centre principal
    point, (a) View.project_points (Datasets/utils.py:1040-1044 -> PerspectiveCamera.cam_to_screen, Cameras/Perspective.py:39-52) of 400 points
    and (b) the matrices GaussianSplatting/Renderer.py:60-74 hands to the rasterizer (w2c.T, w2c.T @ P.T with P = get_projection_matrix,
    Perspective.py:96-119; tanfov = W / (2 fx)).  The rasterizer's convention  pix = ((ndc + 1) * W - 1) / 2  on those matrices must land on
    cam_to_screen - 0.5 (pixel centres at integers), and its view-space z on project_points' depth."""
    install_shims()
    sys.path.insert(0, str(REF))
    import Framework
    Framework.config = Framework.ConfigWrapper.fromDict({
        'GLOBAL': {'RANDOM_SEED': 1618033989, 'ANOMALY_DETECTION': False, 'GPU_INDICES': None, 'DEFAULT_DEVICE': torch.device('cpu'), 'METHOD_TYPE': 'NeRF'},
        'TRAINING': {'WANDB': {'ACTIVATE': False}},
    })
    from Cameras.Perspective import PerspectiveCamera
    from Cameras.utils import SharedCameraSettings
    from Datasets.utils import View
    prev = np.load(OUT / 'raygen.npz')                      # the same camera, pose and points as the project_points vectors stored there
    w, h, fx, fy, cx, cy, near, far = prev['proj_intr']
    settings = SharedCameraSettings(background_color=torch.tensor([0.0, 0.0, 0.0]), near_plane=float(near), far_plane=float(far))
    cam = PerspectiveCamera(shared_settings=settings, width=int(w), height=int(h), focal_x=float(fx), focal_y=float(fy), center_x=float(cx), center_y=float(cy))
    view = View(camera=cam, camera_index=0, frame_idx=0, global_frame_idx=0, c2w=prev['proj_c2w'])
    pts = torch.from_numpy(prev['proj_pts'])
    xy, depth, inside = view.project_points(pts)
    assert np.array_equal(xy.numpy(), prev['proj_xy'])      # the stored vectors are reproduced: same reference code path
    P = cam.get_projection_matrix()
    w2cT = view.w2c.T
    np.savez_compressed(OUT / 'gs_projection.npz', intr=prev['proj_intr'], c2w=prev['proj_c2w'], pts=pts.numpy(), xy=xy.numpy(), depth=depth.numpy(),
                        in_frustum=inside.numpy(), viewmatrix=w2cT.numpy(), projmatrix=(w2cT @ P.T).numpy(),
                        tanfov=np.array([cam.width / cam.focal_x * 0.5, cam.height / cam.focal_y * 0.5], dtype=np.float64), campos=view.position.numpy())
    print('gs_projection.npz:', (OUT / 'gs_projection.npz').stat().st_size, 'B')


def run_ingp_orchestration():
    """The reference's OWN InstantNGP host code -- InstantNGPRenderer.render_rays -> InstantNGPRayRenderingComponent.forward / render_rays_training /
    render_rays_inference / query_model (src/Methods/InstantNGP/Renderer.py:30-180) through VolumeRenderingV2/custom_functions.py's autograd classes
    -- executed in THIS container on CPU, with the native ops it calls (the twelve VolumeRenderingV2 functions, the two tinycudann networks) patched
    to the CPU oracle behind their own signatures (tests/oracle_ops.py).  Returns the inputs, the outputs of a training batch and of the alive-ray
    inference loop, and the sequence of native calls: what tests/golden/ingp_orchestration.npz holds and what the GPU mirror
    (nerficg_amd/instant_ngp.py) is compared with on the same rays."""
    import json
    install_shims()
    shims = str(Path(__file__).resolve().parents[2] / 'nerficg_amd' / 'shims')
    root = str(Path(__file__).resolve().parents[2])
    for q in (str(REF), root, shims):
        if q not in sys.path:
            sys.path.insert(0, q)
    import Framework
    Framework.config = Framework.ConfigWrapper.fromDict({
        'GLOBAL': {'RANDOM_SEED': 1618033989, 'ANOMALY_DETECTION': False, 'GPU_INDICES': [], 'DEFAULT_DEVICE': torch.device('cpu'), 'METHOD_TYPE': 'InstantNGP'},
        'TRAINING': {'WANDB': {'ACTIVATE': False}, 'MODEL_NAME': 'probe'}, 'MODEL': {}, 'RENDERER': {}})
    import torchmetrics.functional.image as _tfi   # names of absent metric packages that the method packages import at module level
    for _n in ('structural_similarity_index_measure', 'multiscale_structural_similarity_index_measure', 'learned_perceptual_image_patch_similarity'):
        setattr(_tfi, _n, lambda *a, **k: None)
    from Cameras.Perspective import PerspectiveCamera
    from Cameras.utils import SharedCameraSettings
    from Datasets.utils import View
    from Methods.InstantNGP.Model import InstantNGPModel
    from Methods.InstantNGP.Renderer import InstantNGPRenderer
    import Methods.InstantNGP.VolumeRenderingV2 as package
    import VolumeRenderingV2 as shim
    import nerficg_amd.tinycudann as tcnn
    from tests import oracle_ops, scenes
    oracle_ops.install(shim, package, tcnn.NetworkWithInputEncoding)
    w = h = 40
    table_gain, radius = 2.0e4, 0.35
    model = InstantNGPModel('probe').build()
    with torch.no_grad():     # U(-1e-4, 1e-4) renders a constant: the table of the fixture is the seeded initialisation times a constant
        model.encoding_xyz.params[model.n_params_encoding_mlp:] *= table_gain
        model.occupancy_bitfield.copy_(torch.from_numpy(scenes.sphere_bitfield(model.RESOLUTION, model.SCALE, radius, model.cascades)))
    renderer = InstantNGPRenderer(model)
    fx, fy, cx, cy = scenes.lego_intrinsics(w, h)
    settings = SharedCameraSettings(background_color=torch.tensor([1.0, 1.0, 1.0]), near_plane=0.2, far_plane=1000.0)
    cam = PerspectiveCamera(shared_settings=settings, width=w, height=h, focal_x=fx, focal_y=fy, center_x=cx, center_y=cy)
    c2w = scenes.orbit_pose(0.5, 0.3, scenes.LEGO_RADIUS)
    rays = View(camera=cam, camera_index=0, frame_idx=0, global_frame_idx=0, c2w=c2w).get_rays()
    bg = torch.tensor([0.2, 0.7, 0.4])
    blob = dict(size=np.array([w, h]), intr=np.array([fx, fy, cx, cy], dtype=np.float64), c2w=c2w, table_gain=np.float64(table_gain), radius=np.float64(radius),
                seed=np.int64(1618033989), origin=rays.origin.numpy(), view_direction=rays.view_direction.numpy(), bg=bg.numpy())
    # ---- a training batch (the march jitter is the reference's own torch.rand_like draw: recorded from the patched op's argument)
    oracle_ops.TRACE.clear()
    noise_seen = []
    inner = shim.raymarching_train
    def spy(*args):
        noise_seen.append(args[7].clone())
        return inner(*args)
    shim.raymarching_train = spy
    torch.manual_seed(7)
    out = renderer.render_rays(rays, cam, train_mode=True, custom_bg_color=bg)
    shim.raymarching_train = inner
    trace_train = list(oracle_ops.TRACE)
    blob.update(noise=noise_seen[0].numpy(), train_rgb=out['rgb'].detach().float().numpy(), train_alpha=out['alpha'].detach().float().numpy(),
                train_depth=out['depth'].detach().float().numpy(), train_rm_samples=np.int64(int(out['rm_samples'])))
    # ---- the same rays through the alive-ray inference loop
    oracle_ops.TRACE.clear()
    with torch.no_grad():
        out = renderer.render_rays(rays, cam, train_mode=False)
    trace_eval = list(oracle_ops.TRACE)
    blob.update(eval_rgb=out['rgb'].float().numpy(), eval_alpha=out['alpha'].float().numpy(), eval_depth=out['depth'].float().numpy())
    blob['trace_train'] = np.array(json.dumps(trace_train))
    blob['trace_eval'] = np.array(json.dumps(trace_eval))
    return blob


def make_ingp_orchestration():
    blob = run_ingp_orchestration()
    np.savez_compressed(OUT / 'ingp_orchestration.npz', **blob)
    print('ingp_orchestration.npz:', (OUT / 'ingp_orchestration.npz').stat().st_size, 'B;', int(blob['train_rm_samples']), 'training samples')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'gs_densify':
        make_gs_densify()   # only this fixture (the others keep their RNG streams)
    elif len(sys.argv) > 1 and sys.argv[1] == 'composite_bw':
        make_composite_bw()
    elif len(sys.argv) > 1 and sys.argv[1] == 'gs_projection':
        make_gs_projection()
    elif len(sys.argv) > 1 and sys.argv[1] == 'ingp_orchestration':
        make_ingp_orchestration()
    else:
        main()
        make_gs_densify()
        make_composite_bw()
        make_gs_projection()
