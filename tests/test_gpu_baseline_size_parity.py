"""GPU parity at the sizes bench.py's CPU-baseline legs run the oracle on (seconds of CPU work), on the bench scenes themselves:

  * 3DGS: 250 000 Gaussians of the bench distribution on a 648x420 image (bench.gs_cpu_baseline) -- forward AND backward against
    oracle/gs_oracle_impl.h: radii, tile ranges and the depth-ordered id lists bit-exact, image <= 2e-5, all six gradients <= 2e-3 of the
    tensor scale;
  * InstantNGP: the bench scene at the 800x800 bench intrinsics and a bench pose; the oracle marches, queries and composites the 96x96
    central crop (bench.cpu_baseline), the HIP pipeline renders the whole image: per-ray sample counts bit-exact, rgb <= 2e-3, mean L1 <= 2e-4;
  * raymarching_test + composite_test_fw with cascades in {2, 3} and exponential steps (1/256): the reference passes `cascades` where
    calc_dt expects `scale` (raymarching.cu:370,399 vs :11) -- the upper dt clamp of this quirk is only reachable with esf > 0.
"""
import numpy as np
import pytest
import torch

import bench
import oracle
from tests import scenes

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


# ------------------------------------------------------------------------------------------------ 3DGS, 250 000 Gaussians
def test_gs_quarter_million_forward_and_backward_match_the_oracle():
    from nerficg_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    n, w, h = 250_000, 648, 420  # = bench.gs_cpu_baseline defaults
    sc = scenes.gs_random_scene(n, seed=0)
    cam = scenes.gs_camera(w, h, scenes.orbit_pose(0.8, 0.35, 4.5))
    bg = np.array([0.0, 0.0, 0.0], np.float32)
    settings = GaussianRasterizationSettings(
        image_height=h, image_width=w, tanfovx=cam['tanfovx'], tanfovy=cam['tanfovy'], bg=T(bg), scale_modifier=1.0, viewmatrix=T(cam['viewmatrix']),
        projmatrix=T(cam['projmatrix']), sh_degree=3, campos=T(cam['campos']), prefiltered=False, debug=False)
    t = {k: T(v).requires_grad_(True) for k, v in sc.items() if k != 'sh_degree'}
    m2d = torch.zeros_like(t['means3D'], requires_grad=True)
    color, radii = GaussianRasterizer(settings)(means3D=t['means3D'], means2D=m2d, opacities=t['opacities'][:, None], shs=t['shs'], scales=t['scales'],
                                                rotations=t['rotations'])
    o_color, o_radii, st = oracle.gs_forward(sc['means3D'], sc['opacities'], cam['viewmatrix'], cam['projmatrix'], cam['campos'], cam['tanfovx'],
                                             cam['tanfovy'], w, h, bg, sh_degree=3, shs=sc['shs'], scales=sc['scales'], rotations=sc['rotations'])
    assert st.num_rendered > 500_000 and (o_radii > 0).sum() > 100_000
    np.testing.assert_array_equal(radii.cpu().numpy(), o_radii)
    fn = color.grad_fn
    names = ['means3D', 'sh', 'col', 'sc', 'rot', 'cov', 'radii', 'points_xy', 'conic_opacity', 'rgb', 'clamped', 'cov3D', 'point_list', 'ranges',
             'n_contrib', 'final_T']
    sv = dict(zip(names, fn.saved_tensors))
    assert fn.num_rendered == st.num_rendered
    np.testing.assert_array_equal(sv['ranges'].cpu().numpy().astype(np.uint32), st.ranges)
    np.testing.assert_array_equal(sv['point_list'].cpu().numpy()[:st.num_rendered], st.point_list[:st.num_rendered])
    np.testing.assert_array_equal(sv['n_contrib'].cpu().numpy().astype(np.uint32), st.n_contrib)
    np.testing.assert_allclose(color.detach().cpu().numpy(), o_color, rtol=0, atol=2e-5)
    gpix = np.random.default_rng(0).normal(size=(3, h, w)).astype(np.float32)
    color.backward(T(gpix))
    ref = oracle.gs_backward(st, gpix)
    for name, got in (('mean3D', t['means3D'].grad), ('mean2D', m2d.grad), ('opacity', t['opacities'].grad), ('scale', t['scales'].grad),
                      ('rot', t['rotations'].grad), ('sh', t['shs'].grad)):
        want = ref[name]
        got = got.cpu().numpy().reshape(want.shape) if name != 'mean2D' else got.cpu().numpy()[:, :want.shape[1]]
        scale = np.abs(want).max()
        assert scale > 0, name
        assert np.abs(got - want).max() <= 2e-3 * scale, (name, np.abs(got - want).max() / scale)


# ------------------------------------------------------------------------------------------------ InstantNGP, bench scene, 800x800 intrinsics
def test_ingp_bench_pose_central_crop_matches_the_oracle():
    crop = 96
    model, renderer, cam, poses = bench.build_scene(DEV)
    with torch.no_grad():  # the bench table U(-1e-4, 1e-4) renders a constant; same scene geometry with a table that makes densities / colours vary
        g = torch.Generator().manual_seed(5)
        n = model.encoding_xyz.params.numel() - 3072
        model.encoding_xyz.params[3072:] = ((torch.rand(n, generator=g) * 2 - 1) * 2.0).to(DEV)
    pose = poses[0]
    out = renderer.render_image_fused(cam, pose, return_stats=True, early_termination=False)
    W, H = cam.width, cam.height
    ws = next(iter(renderer._fused_ws.values()))
    tw, th = 8, 8
    ys, xs = np.meshgrid(np.arange((H - crop) // 2, (H + crop) // 2), np.arange((W - crop) // 2, (W + crop) // 2), indexing='ij')
    slot = ((ys // th) * ((W + tw - 1) // tw) + xs // tw) * 64 + (ys % th) * tw + xs % tw   # ray_cnt is indexed tile-major, lane = pixel in tile
    cnt_gpu = ws['ray_cnt'].cpu().numpy()[slot.reshape(-1)]
    pix = (ys * W + xs).reshape(-1)
    # oracle: same rays as the image's centre window (bench.cpu_baseline)
    fx, fy, cx, cy = scenes.lego_intrinsics(W, H)
    o, _, d = scenes.numpy_rays(crop, crop, pose, fx, fy, cx - (W - crop) / 2, cy - (H - crop) / 2)
    o = o - model.center.cpu().numpy()
    _, ht, _ = oracle.ray_aabb_intersect(o, d, np.zeros((1, 3), np.float32), np.full((1, 3), 0.5, np.float32), 1)
    hits = ht[:, 0].copy()
    hits[:, 0] = np.maximum(hits[:, 0], np.float32(cam.near_plane))
    hits[:, 1] = np.minimum(hits[:, 1], np.float32(cam.far_plane))
    bf = model.occupancy_bitfield.cpu().numpy()
    rays_a, xyzs, dirs, deltas, ts, counter = oracle.raymarching_train(o, d, hits, bf, 1, 0.5, 0.0, np.zeros(len(o), np.float32), 128, 1024)
    assert int(counter[0]) > 1_000_000
    cnt_ref = np.zeros(len(o), np.int64)
    cnt_ref[rays_a[:, 0]] = rays_a[:, 2]
    np.testing.assert_array_equal(cnt_gpu, cnt_ref)  # per-ray sample counts, bit-exact
    pd = model.encoding_xyz.params.detach().half().float().cpu().numpy()
    pc = model.color_mlp_with_encoding.params.detach().half().float().cpu().numpy()
    grid_kw = {k: model.encoding_xyz.grid_cfg[k] for k in ('n_levels', 'log2_hashmap_size', 'base_resolution', 'per_level_scale')}
    sig, rgb, _ = oracle.ngp_query((xyzs + np.float32(0.5)) / np.float32(1.0), dirs, pd[:3072], pc, pd[3072:].reshape(-1, 2), **grid_kw)
    _, alpha, depth, col, _ = oracle.composite_train_fw(sig, rgb, deltas, ts, rays_a, 1e-4)
    alpha = np.clip(alpha, 0, 1)
    Tr = 1 - alpha
    col = np.clip(col + Tr[:, None] * cam.background_color.numpy()[None], 0, 1)
    got_rgb = out['rgb'].cpu().numpy()[pix]
    got_alpha = out['alpha'].cpu().numpy()[pix]
    assert alpha.max() > 0.3 and alpha.std() > 0.005 and col.std() > 0.005  # the crop is not a flat picture
    assert np.abs(got_rgb - col).max() <= 2e-3 and np.abs(got_rgb - col).mean() <= 2e-4
    assert np.abs(got_alpha - alpha).max() <= 2e-3


# ------------------------------------------------------------------------------------------------ raymarching_test quirk path
def _march_inputs(width, height, cascades, scale, pose=(0.7, 0.4)):
    c2w = scenes.orbit_pose(pose[0], pose[1], scenes.LEGO_RADIUS * (scale / 0.5))
    o, _, vd = scenes.numpy_rays(width, height, c2w)
    _, ht, _ = oracle.ray_aabb_intersect(o, vd, np.zeros((1, 3), np.float32), np.full((1, 3), scale, np.float32), 1)
    hits = ht[:, 0].copy()
    hits[:, 0] = np.maximum(hits[:, 0], 0.2)
    hits[:, 1] = np.minimum(hits[:, 1], 1000.0)
    # a thin shell far out: long empty stretches at large t, where the upper dt clamp (sqrt3 * 2 * "scale" / max_samples) is the active one
    return o, vd, hits, scenes.sphere_bitfield(128, scale, 0.9 * scale, cascades, shell=0.25 * scale)


@pytest.mark.parametrize('cascades,scale,esf', [(2, 1.0, 1 / 256), (3, 2.0, 1 / 256), (3, 4.0, 1 / 256), (2, 1.0, 1 / 32), (3, 2.0, 1 / 32), (3, 4.0, 1 / 32)])
@pytest.mark.parametrize('n_samples', [4, 64])
def test_raymarching_test_and_composite_with_cascades_and_exponential_steps(cascades, scale, esf, n_samples):
    """esf = 1/256 is what the reference's EXPONENTIAL_STEPS sets (Renderer.py:34-35); dt = clamp(t * esf, sqrt3/max_samples,
    sqrt3 * 2 * cascades / grid_size) then reaches its upper clamp only beyond t = 13.9 * cascades, outside these scenes -- so the same cases run
    with esf = 1/32 as well, where the `cascades`-for-`scale` clamp (raymarching.cu:370,399) is the active bound on most of the ray."""
    import nerficg_amd.VolumeRenderingV2 as vr
    max_samples = 1024
    o, d, hits, bitfield = _march_inputs(72, 56, cascades, scale)
    n = o.shape[0]
    rng = np.random.default_rng(cascades * 100 + n_samples)
    alive0 = np.sort(rng.choice(n, size=(2 * n) // 3, replace=False)).astype(np.int64)
    h_gpu, h_ref = T(hits), hits.copy()
    g_alive, r_alive = T(alive0), alive0.copy()
    g_op, g_dp, g_c = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, 3, device=DEV)
    r_op, r_dp, r_c = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    quirk = np.float32(1.7320508) * 2 * np.float32(cascades) / np.float32(128)
    fixed = np.float32(1.7320508) * 2 * np.float32(scale) / np.float32(128)
    n_on_clamp = n_exponential = total = 0
    for it in range(4):  # the reference's inference loop (Renderer.py:104-132): march, composite, drop dead rays, repeat
        got = vr.raymarching_test(T(o), T(d), h_gpu, g_alive, T(bitfield), cascades, scale, esf, 128, max_samples, n_samples)
        ref = oracle.raymarching_test(o, d, h_ref, r_alive, bitfield, cascades, scale, esf, 128, max_samples, n_samples)
        for name, a, b in zip(('xyzs', 'dirs', 'deltas', 'ts', 'n_eff'), got, ref):
            np.testing.assert_array_equal(a.cpu().numpy(), b, err_msg=f'{name} (iteration {it})')
        np.testing.assert_array_equal(h_gpu.cpu().numpy(), h_ref)
        xyzs, dirs, deltas, ts, n_eff = ref
        total += int(n_eff.sum())
        live = deltas[deltas > 0]
        n_on_clamp += int((live == quirk).sum())
        n_exponential += int(((live > np.float32(1.7320508) / max_samples) & (live < quirk)).sum())
        assert float(live.max(initial=0)) <= quirk
        a_, s_ = xyzs.shape[:2]
        sig = (rng.random((a_, s_)) * 6).astype(np.float32)
        rgbs = rng.random((a_, s_, 3)).astype(np.float32)
        vr.composite_test_fw(T(sig), T(rgbs), T(deltas), T(ts), h_gpu, g_alive, 1e-4, T(n_eff), g_op, g_dp, g_c)
        oracle.composite_test_fw(sig, rgbs, deltas, ts, r_alive, 1e-4, n_eff, r_op, r_dp, r_c)
        assert (g_alive.cpu().numpy() != r_alive).mean() < 5e-3  # which rays died: flips only on ties at the transmittance threshold
        np.testing.assert_allclose(g_op.cpu().numpy(), r_op, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(g_dp.cpu().numpy(), r_dp, rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(g_c.cpu().numpy(), r_c, rtol=2e-5, atol=2e-6)
        r_alive = r_alive[r_alive >= 0]  # both sides continue with the oracle's survivors
        g_alive = T(r_alive)
        if len(r_alive) == 0:
            break
    assert total > 1000
    if esf < 1 / 64:
        assert n_exponential > 100  # dt = t / 256 between the two clamps
    else:
        assert n_on_clamp > 100 and quirk != fixed, 'no delta sits on the `cascades` clamp: the case does not exercise raymarching.cu:370'
