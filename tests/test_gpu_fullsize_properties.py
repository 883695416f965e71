"""GPU: size-independent properties at the FULL sizes of BASELINE.json's configurations -- the 800x800 InstantNGP image of bench.py
(640 000 rays, ~77 M samples; configs[1]), the 1 M-Gaussian 1297x840 3DGS frame (configs[2]), one of eight tile shards of a 1600x1060
InstantNGP image plus the data-parallel gradient split of a training batch drawn from it (configs[3]) and the 6 M-Gaussian frame (configs[4]) --
where no CPU oracle finishes in seconds: shards compose, alternative execution orders give the same picture, the background enters linearly,
reruns are deterministic, per-rank gradients add up to the single-GPU gradient, culled Gaussians get exactly zero gradient."""
import numpy as np
import pytest
import torch

import bench

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


@pytest.fixture(scope='module')
def ngp():
    return bench.build_scene(DEV)


def _img(out):
    return {k: out[k].clone() for k in ('rgb', 'alpha', 'depth')}


def test_ingp_full_image_properties(ngp):
    model, renderer, cam, poses = ngp
    pose = poses[7]
    ref = renderer.render_image_fused(cam, pose, return_stats=True, early_termination=False)
    full = _img(ref)
    n_samples, n_rows = ref['n_samples'], ref['n_rows']
    assert full['rgb'].shape == (800 * 800, 3) and bool(torch.isfinite(full['rgb']).all())
    assert 0.0 <= float(full['alpha'].min()) and float(full['alpha'].max()) <= 1.0 and float(full['rgb'].min()) >= 0.0 and float(full['rgb'].max()) <= 1.0
    assert 60e6 < n_samples < 100e6 and n_rows * 64 >= n_samples  # the workload bench.py quotes (120 samples per ray)
    # determinism: a second run is bit-identical
    again = _img(renderer.render_image_fused(cam, pose, early_termination=False))
    for k in full:
        assert torch.equal(full[k], again[k]), k
    # the count pass parking the samples vs marching every ray twice: same samples, same image, bit for bit
    renderer.provisional_march = False
    renderer._fused_ws = {}
    twice = _img(renderer.render_image_fused(cam, pose, early_termination=False))
    renderer.provisional_march = True
    renderer._fused_ws = {}
    for k in full:
        assert torch.equal(full[k], twice[k]), k
    # depth-slab order with early termination composites the same picture (the order of the MLP batches differs, the per-ray sums do not)
    slabs = _img(renderer.render_image_fused(cam, pose, early_termination=True))
    for k in full:
        assert float((full[k] - slabs[k]).abs().max()) <= 2e-6, k
    # five contiguous tile shards (the data-parallel inference split) compose to the same buffers, bit for bit
    nt = renderer.n_image_tiles(cam)
    out = {k: torch.zeros_like(v) for k, v in full.items()}
    bounds = np.linspace(0, nt, 6).astype(int)
    for b, e in zip(bounds[:-1], bounds[1:]):
        renderer.render_image_fused(cam, pose, tile_begin=int(b), n_tiles=int(e - b), out=out, early_termination=False)
    for k in full:
        assert torch.equal(full[k], out[k]), k


def test_ingp_background_enters_linearly(ngp):
    """rgb = sum_i w_i c_i + (1 - alpha) * bg (Renderer.py:134-136): two backgrounds differ by (1 - alpha) * (bg1 - bg0)."""
    from nerficg_amd.instant_ngp import Camera
    model, renderer, cam, poses = ngp
    cams = [Camera(width=cam.width, height=cam.height, focal_x=cam.focal_x, focal_y=cam.focal_y, center_x=cam.center_x, center_y=cam.center_y,
                   near_plane=cam.near_plane, far_plane=cam.far_plane, background_color=torch.tensor(bg)) for bg in ([0.0, 0.0, 0.0], [0.25, 0.5, 0.125])]
    a = _img(renderer.render_image_fused(cams[0], poses[3], early_termination=False))
    b = _img(renderer.render_image_fused(cams[1], poses[3], early_termination=False))
    assert torch.equal(a['alpha'], b['alpha']) and torch.equal(a['depth'], b['depth'])
    want = (1 - a['alpha'])[:, None] * torch.tensor([0.25, 0.5, 0.125], device=DEV)
    unclamped = (b['rgb'] < 1.0).all(dim=1)  # the final clamp to [0, 1] is not linear
    assert float(unclamped.float().mean()) > 0.9
    assert float((b['rgb'] - a['rgb'] - want)[unclamped].abs().max()) <= 2e-6


# ------------------------------------------------------------------------------------------------ configs[3]: 1600x1060 over 8 ranks
def _c4_camera(cam):
    from nerficg_amd.instant_ngp import Camera
    W, H = 1600, 1060
    return Camera(width=W, height=H, focal_x=cam.focal_x * W / cam.width, focal_y=cam.focal_x * W / cam.width, center_x=W / 2, center_y=H / 2,
                  near_plane=cam.near_plane, far_plane=cam.far_plane, background_color=cam.background_color)


def test_ingp_c4_tile_shard_equals_its_part_of_the_whole_image(ngp):
    """BASELINE configs[3]: a 1600x1060 image split into 8 contiguous tile ranges (parallel.shard_range).  Rank 3's shard, rendered alone,
    is bit-identical to the same pixels of the whole image, leaves every other pixel untouched, and is deterministic."""
    from nerficg_amd import parallel
    model, renderer, cam, poses = ngp
    big = _c4_camera(cam)
    pose = poses[11]
    whole = _img(renderer.render_image_fused(big, pose, early_termination=False))
    assert whole['rgb'].shape == (1600 * 1060, 3) and bool(torch.isfinite(whole['rgb']).all())
    nt = renderer.n_image_tiles(big)
    assert nt == 200 * 133
    b, e = parallel.shard_range(nt, 3, 8)
    sentinel = {'rgb': torch.full_like(whole['rgb'], -7.0), 'alpha': torch.full_like(whole['alpha'], -7.0), 'depth': torch.full_like(whole['depth'], -7.0)}
    res = renderer.render_image_fused(big, pose, tile_begin=b, n_tiles=e - b, out=sentinel, return_stats=True, early_termination=False)
    assert res['n_samples'] > 5e6
    tiles = torch.arange(b, e, device=DEV)
    ty, tx = tiles // 200, tiles % 200
    py = (ty[:, None] * 8 + torch.arange(64, device=DEV)[None] // 8).reshape(-1)
    px = (tx[:, None] * 8 + torch.arange(64, device=DEV)[None] % 8).reshape(-1)
    ok = py < 1060
    pix = (py * 1600 + px)[ok]
    mine = torch.zeros(1600 * 1060, dtype=torch.bool, device=DEV)
    mine[pix] = True
    for k in whole:
        assert torch.equal(sentinel[k][mine], whole[k][mine]), k
        assert bool((sentinel[k][~mine] == -7.0).all()), k
    again = {k: torch.full_like(v, -7.0) for k, v in whole.items()}
    renderer.render_image_fused(big, pose, tile_begin=b, n_tiles=e - b, out=again, early_termination=False)
    for k in whole:
        assert torch.equal(again[k], sentinel[k]), k
    # all eight shards together: the whole image, bit for bit
    out = {k: torch.zeros_like(v) for k, v in whole.items()}
    for r in range(8):
        b, e = parallel.shard_range(nt, r, 8)
        renderer.render_image_fused(big, pose, tile_begin=b, n_tiles=e - b, out=out, early_termination=False)
    for k in whole:
        assert torch.equal(out[k], whole[k]), k


def test_ingp_c4_rank_gradients_add_up_to_the_single_gpu_gradient(ngp):
    """BASELINE configs[3], training side: a batch of rays of the 1600x1060 camera, rank r takes ray_ids[r::8] (parallel.shard_ray_ids); the sum
    of the eight per-rank gradients (what the RCCL reduction produces) equals the gradient of the whole batch on one GPU up to the order of the
    float additions (fp16 backward activations, f32 / fixed-point accumulation): <= 2e-3 of the gradient scale per parameter tensor."""
    from nerficg_amd import parallel
    from nerficg_amd.raygen import generate_rays
    model, renderer, cam, poses = ngp
    big = _c4_camera(cam)
    rays = generate_rays(big.width, big.height, big.focal_x, big.focal_y, big.center_x, big.center_y, poses[5], device=DEV, want_direction=False)
    ids = torch.randperm(big.width * big.height, generator=torch.Generator().manual_seed(0))[:4096].to(DEV)
    target = torch.rand(big.width * big.height, 3, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    bg = torch.tensor([0.3, 0.6, 0.9], device=DEV)
    jitter = torch.rand(ids.shape[0], device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))  # ONE seeded vector for the global batch
    params = list(model.parameters())

    def grads_of(sel, noise):
        for p in params:
            p.grad = None
        with torch.amp.autocast('cuda'):
            out = renderer.render_rays(rays['origin'][sel], rays['view_direction'][sel], big, train_mode=True, custom_bg_color=bg, noise=noise)
            loss = (out['rgb'].float() - target[sel]).square().sum()  # a SUM over rays: per-rank losses add up
        (loss * 128.0).backward()
        return [p.grad.detach().clone() / 128.0 for p in params], int(out['rm_samples'].item())

    full, n_full = grads_of(ids, jitter)
    acc, n_sum = [torch.zeros_like(g) for g in full], 0
    for r in range(8):
        g, n = grads_of(parallel.shard_ray_ids(ids, r, 8), parallel.shard_ray_ids(jitter, r, 8))
        n_sum += n
        for a, b in zip(acc, g):
            a += b
    assert n_sum == n_full > 100_000  # the global sample set is the single-GPU one
    for a, f in zip(acc, full):
        scale = float(f.abs().max())
        assert scale > 0 and float((a - f).abs().max()) <= 2e-3 * scale, float((a - f).abs().max()) / scale
    for p in params:
        p.grad = None


@pytest.fixture(scope='module', params=[(1_000_000, None, None), (1_000_000, 1600, 1060), (6_000_000, None, None)], ids=['1M', '1M@1600x1060', '6M'])
def gs(request):
    """configs[2] (1 M Gaussians; at the reference yaml's 1297x840 and at the 1600x1060 BASELINE.json names -- SURVEY 8(d) lists both) and configs[4]
    (6 M Gaussians, the per-rank frame of the 8-GPU data-parallel run)."""
    n, w, h = request.param
    return bench.build_gs_scene(DEV, n, w=w, h=h)


def _gs_render(gs, tensors, bg=None, grad=False):
    from nerficg_amd.diff_gaussian_rasterization import GaussianRasterizer
    rast = gs['rast']
    if bg is not None:
        settings = rast.raster_settings._replace(bg=torch.tensor(bg, device=DEV)) if hasattr(rast.raster_settings, '_replace') else None
        if settings is None:
            import dataclasses
            settings = dataclasses.replace(rast.raster_settings, bg=torch.tensor(bg, device=DEV))
        rast = GaussianRasterizer(settings)
    t = {k: v.detach().requires_grad_(grad) for k, v in tensors.items()}
    m2d = torch.zeros_like(t['means3D'], requires_grad=grad)
    color, radii = rast(means3D=t['means3D'], means2D=m2d, opacities=t['opacities'], shs=t['shs'], scales=t['scales'], rotations=t['rotations'])
    return color, radii, t, m2d


def test_gs_million_gaussians_properties(gs):
    t = gs['tensors']
    color, radii, _, _ = _gs_render(gs, t)
    assert color.shape == (3, gs['h'], gs['w']) and bool(torch.isfinite(color).all())
    assert 0.8 * gs['n'] < int((radii > 0).sum()) < 0.9 * gs['n']
    # determinism of the forward: binning, depth order and blending are free of atomics-order effects
    color2, radii2, _, _ = _gs_render(gs, t)
    assert torch.equal(color, color2) and torch.equal(radii, radii2)
    # the order of the Gaussians in memory does not matter: the per-tile lists are ordered by depth.  Equal f32 depths are ordered by index
    # (1 M depths share ~8 M float values: tens of thousands of tied pairs, a few of them overlapping on screen), so a handful of pixels may differ
    perm = torch.randperm(t['means3D'].shape[0], device=DEV, generator=torch.Generator(device=DEV).manual_seed(0))
    shuffled = {k: v[perm].contiguous() for k, v in t.items()}
    color3, radii3, _, _ = _gs_render(gs, shuffled)
    assert torch.equal(radii3, radii[perm])
    diff = (color3 - color).abs()
    assert float((diff > 1e-6).float().mean()) < (1e-2 if gs['n'] <= 1_000_000 else 6e-2) and float(diff.max()) < 5e-2 and float(diff.mean()) < 3e-5
    # background enters through the final transmittance only: C(bg) = C(0) + T_final * bg, the same factor for the three channels
    color_bg, _, _, _ = _gs_render(gs, t, bg=[0.5, 0.25, 1.0])
    tfin = (color_bg - color) / torch.tensor([0.5, 0.25, 1.0], device=DEV)[:, None, None]
    assert float((tfin[0] - tfin[1]).abs().max()) <= 1e-6 and float((tfin[0] - tfin[2]).abs().max()) <= 1e-6
    assert float(tfin.min()) >= -1e-6 and float(tfin.max()) <= 1 + 1e-6


def test_gs_million_gaussians_gradients_are_stable_and_local(gs):
    t = gs['tensors']
    g = torch.rand(3, gs['h'], gs['w'], device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    grads = []
    for _ in range(2):
        color, radii, tt, m2d = _gs_render(gs, t, grad=True)
        color.backward(g)
        grads.append({k: v.grad.clone() for k, v in tt.items()} | {'means2D': m2d.grad.clone()})
    invisible = radii <= 0
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        scale = float(a.abs().max())
        assert bool(torch.isfinite(a).all()) and scale > 0, k
        # only the order of the f32 global atomics (one per tile and Gaussian) differs between two runs: a sum of n terms moves by at most
        # n * 2^-24 * sum|terms|; 2.1e-5 of the scale was the largest seen at 6 M, the bound leaves 50x
        assert float((a - b).abs().max()) <= 1e-3 * scale, k
        assert float(a[invisible].abs().max()) == 0.0, k        # culled Gaussians get exactly zero gradient
    # dL/dopacity of a Gaussian is the sum over its pixels: doubling the upstream gradient doubles it (linearity of the backward)
    color, _, tt, _ = _gs_render(gs, t, grad=True)
    color.backward(2.0 * g)
    assert float((tt['opacities'].grad - 2.0 * grads[0]['opacities']).abs().max()) <= 2e-3 * float(grads[0]['opacities'].abs().max())


@pytest.mark.parametrize('switch', ['NRC_QUERY_FUSED', 'NRC_ENC_FINE_SPLIT'])
def test_experiment_switches_of_the_query_paint_the_same_image(switch):
    """NRC_QUERY_FUSED=1 (one kernel: encode into registers, 8 v_permlane32_swap, MLP chain) and NRC_ENC_FINE_SPLIT=1 (levels 12-15 in a launch of their
    own, one level per XCD) against the default two kernels through the feature buffer, in a subprocess because the switches are read once per process:
    same fp16 features, same MFMA chain -> the same picture bit for bit."""
    import os
    import subprocess
    import sys
    code = (
        'import sys, hashlib, torch; sys.path.insert(0, "."); import bench\n'
        'm, r, cam, poses = bench.build_scene(torch.device("cuda", 0))\n'
        'o = r.render_image_fused(cam, poses[4], early_termination=False)\n'
        'print(hashlib.sha1(o["rgb"].cpu().numpy().tobytes() + o["alpha"].cpu().numpy().tobytes()).hexdigest())\n')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for flag in ('0', '1'):
        env = dict(os.environ, **{switch: flag})
        res = subprocess.run([sys.executable, '-c', code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append(res.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]
