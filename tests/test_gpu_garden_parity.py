"""GPU parity in the configuration an unbounded scene uses (BASELINE configs[3]: InstantNGP on Mip-NeRF360 garden): SCALE 2 -> three occupancy
cascades (src/Methods/InstantNGP/Model.py:46-52), EXPONENTIAL_STEPS -> exp_step_factor 1/256 (src/Methods/InstantNGP/Renderer.py:44), a
bitfield that is populated in EVERY cascade, rays that start inside the scene box.

What is compared (round-2 review: the fused image kernels k_render_count / k_render_write(_layers) had only ever run with one cascade and
constant steps, where ngp_march.hip takes its single-cascade specialisation):
  * raymarching_train with 3, 4 and 5 cascades and exponential steps against oracle.raymarching_train (raymarching.cu:19-32 mip from
    position / step, :166-280) -- bit-exact;
  * render_image_fused, single pass AND depth-slab order with early termination, against the CPU composition
    oracle.raymarching_train + oracle.ngp_query + oracle.composite_train_fw: per-ray sample counts bit-exact on the same rays; on INDEPENDENT
    numpy rays (scenes.numpy_rays, not the HIP ray generator: an ulp apart) all but <= 0.1 % of the rays bit-exact and those within one
    sample, rgb <= 2e-3 per pixel / 2e-4 mean L1, alpha <= 2e-3.
The inference march of the reference passes `cascades` where calc_dt expects `scale` (raymarching.cu:370,399): that changes only the UPPER step
clamp, sqrt3 * 2 * {scale | cascades} / grid; with 1/256 steps it would bind beyond t = 13.9 (scale 2), outside these scenes -- asserted below on
the oracle's deltas, so the train-rule oracle march IS the inference sample set here (the quirk path itself: test_gpu_baseline_size_parity.py).
"""
import math

import numpy as np
import pytest
import torch

import oracle
from tests import scenes

pytestmark = pytest.mark.gpu
DEV = 'cuda'
GRID = 128


layered_bitfield = scenes.layered_bitfield   # shared with bench.py's strong-scaling leg (BASELINE configs[3] shape)


def garden_model(seed=7, table_amp=2.0, scale=2.0):
    from nerficg_amd.instant_ngp import InstantNGPModel
    model = InstantNGPModel(SCALE=scale, RANDOM_SEED=seed, device=DEV)
    assert model.cascades == 1 + math.ceil(math.log2(2 * scale))
    with torch.no_grad():
        g = torch.Generator().manual_seed(seed)
        n = model.encoding_xyz.params.numel() - 3072
        model.encoding_xyz.params[3072:] = ((torch.rand(n, generator=g) * 2 - 1) * table_amp).to(DEV)
        model.occupancy_bitfield.copy_(torch.from_numpy(layered_bitfield(scale, model.cascades)).to(DEV))
    return model


def garden_camera(w, h, bg=(0.1, 0.6, 0.9)):
    from nerficg_amd.instant_ngp import Camera
    return Camera(width=w, height=h, focal_x=0.9 * w, focal_y=0.9 * w, center_x=w / 2, center_y=h / 2, near_plane=0.2, far_plane=1000.0,
                  background_color=torch.tensor(bg))


def inside_pose(theta, phi, radius=1.15):
    """a camera INSIDE the scale-2 box (like the garden capture: the scene surrounds the cameras), between the two shells, looking at the centre"""
    return scenes.orbit_pose(theta, phi, radius)


def oracle_image(model, cam, c2w, esf, max_samples=1024, rays=None):
    """CPU composition: box test -> near / far clamp -> march -> query -> composite with the early-out -> finalise.  Rays: independent numpy
    rays (scenes.numpy_rays) unless `rays` = (origin, view_direction) is given."""
    w, h = cam.width, cam.height
    if rays is None:
        o, _, d = scenes.numpy_rays(w, h, c2w, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y)
    else:
        o, d = rays
    o = o - model.center.cpu().numpy()
    s = np.float32(model.SCALE)
    _, ht, _ = oracle.ray_aabb_intersect(o, d, np.zeros((1, 3), np.float32), np.full((1, 3), s, np.float32), 1)
    hits = ht[:, 0].copy()
    hits[:, 0] = np.maximum(hits[:, 0], np.float32(cam.near_plane))
    hits[:, 1] = np.minimum(hits[:, 1], np.float32(cam.far_plane))
    bf = model.occupancy_bitfield.cpu().numpy()
    rays_a, xyzs, dirs, deltas, ts, counter = oracle.raymarching_train(o, d, hits, bf, model.cascades, float(s), esf, np.zeros(len(o), np.float32), GRID, max_samples)
    # neither upper step clamp (train: scale, inference quirk: cascades) binds -> both rules give these samples
    assert deltas.max() < np.float32(math.sqrt(3) * 2 * float(s) / GRID) * np.float32(0.999)
    pd = model.encoding_xyz.params.detach().half().float().cpu().numpy()
    pc = model.color_mlp_with_encoding.params.detach().half().float().cpu().numpy()
    x01 = (xyzs - (-s)) / (np.float32(2) * s)
    grid_kw = {k: model.encoding_xyz.grid_cfg[k] for k in ('n_levels', 'log2_hashmap_size', 'base_resolution', 'per_level_scale')}
    sig, rgb, _ = oracle.ngp_query(x01, dirs, pd[:3072], pc, pd[3072:].reshape(-1, 2), **grid_kw)
    _, alpha, depth, col, _ = oracle.composite_train_fw(sig, rgb, deltas, ts, rays_a, 1e-4)
    alpha = np.clip(alpha, 0, 1)
    Tr = 1 - alpha
    col = np.clip(col + Tr[:, None] * cam.background_color.numpy()[None], 0, 1)
    cnt = np.zeros(len(o), np.int64)
    cnt[rays_a[:, 0]] = rays_a[:, 2]
    return col, alpha, cnt, xyzs, deltas


def _ray_counts(renderer, cam):
    """per-pixel sample counts of the last fused frame (ray_cnt is tile-major, lane = pixel inside the 8x8 tile)"""
    ws = next(iter(renderer._fused_ws.values()))
    W, H = cam.width, cam.height
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    slot = ((ys // 8) * ((W + 7) // 8) + xs // 8) * 64 + (ys % 8) * 8 + xs % 8
    return ws['ray_cnt'].cpu().numpy()[slot.reshape(-1)]


# ------------------------------------------------------------------------------------------------ the drop-in march op, 3 to 5 cascades
@pytest.mark.parametrize('scale,esf', [(2.0, 1 / 256), (2.0, 0.0), (4.0, 1 / 256), (8.0, 1 / 256)])
def test_raymarching_train_many_cascades_exponential_steps_bit_exact(scale, esf):
    import nerficg_amd.VolumeRenderingV2 as vr
    cascades = 1 + math.ceil(math.log2(2 * scale))
    assert cascades in (3, 4, 5)
    bf = layered_bitfield(scale, cascades)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    for w, h in ((72, 56), (256, 168)):   # 4 032 rays: the wave-per-ray march of small batches; 43 008 (> 32 768): one thread per ray
        o, _, d = scenes.numpy_rays(w, h, inside_pose(0.9, 0.25), 0.9 * w, 0.9 * w, w / 2, h / 2)
        _, ht, _ = oracle.ray_aabb_intersect(o, d, np.zeros((1, 3), np.float32), np.full((1, 3), scale, np.float32), 1)
        hits = ht[:, 0].copy()
        hits[:, 0] = np.maximum(hits[:, 0], np.float32(0.2))
        noise = np.random.default_rng(3).random(len(o)).astype(np.float32)
        ref = oracle.raymarching_train(o, d, hits, bf, cascades, scale, esf, noise, GRID, 1024)
        got = vr.raymarching_train(T(o), T(d), T(hits), T(bf), cascades, scale, esf, T(noise), GRID, 1024)
        assert int(got[5][0]) == int(ref[5][0]) > 10 * len(o)
        np.testing.assert_array_equal(got[0].cpu().numpy(), ref[0])                      # rays_a: (ray, first sample, count)
        for k in (1, 2, 3, 4):                                                           # xyzs, dirs, deltas, ts
            np.testing.assert_array_equal(got[k].cpu().numpy(), ref[k])
        # the samples do come from several cascades and (with exponential steps) from several step sizes
        r = np.abs(ref[1]).max(axis=1)
        assert (r < 0.5).any() and (r > 1.0).any()
        if esf > 0:
            assert ref[3].max() > 4 * ref[3].min()


# ------------------------------------------------------------------------------------------------ the fused image pipeline
@pytest.mark.parametrize('exponential', [True, False])
@pytest.mark.parametrize('table_amp', [2.0, 40.0])   # 40: densities up to e^several -> rays saturate, tiles drop out of the later depth slabs
def test_fused_image_three_cascades_against_the_oracle(exponential, table_amp):
    from nerficg_amd.instant_ngp import InstantNGPRenderer
    from nerficg_amd.raygen import generate_rays
    model = garden_model(table_amp=table_amp)
    renderer = InstantNGPRenderer(model, EXPONENTIAL_STEPS=exponential)
    esf = 1 / 256 if exponential else 0.0
    w, h = 88, 64
    cam = garden_camera(w, h)
    for pose in (inside_pose(0.9, 0.25), inside_pose(3.6, -0.4, radius=1.05)):
        single = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in renderer.render_image_fused(cam, pose, return_stats=True, early_termination=False).items()}
        cnt_single = _ray_counts(renderer, cam)
        layered = renderer.render_image_fused(cam, pose, return_stats=True, early_termination=True)
        cnt_layered = _ray_counts(renderer, cam)
        col, alpha, cnt, xyzs, deltas = oracle_image(model, cam, pose, esf)
        assert np.array_equal(cnt_single, cnt_layered) and single['n_samples'] == layered['n_samples'] == int(cnt_single.sum())
        for k in ('rgb', 'alpha', 'depth'):                 # same per-ray arithmetic in both orders
            assert torch.equal(single[k], layered[k]), k
        # (1) index parity of the march itself: the oracle fed with the rays of the device generator (the same ray table the fused kernels
        #     build) marches EXACTLY the same number of samples on every ray
        hip = generate_rays(w, h, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, pose, want_direction=False)
        _, _, cnt_same_rays, _, _ = oracle_image(model, cam, pose, esf, rays=(hip['origin'].cpu().numpy(), hip['view_direction'].cpu().numpy()))
        np.testing.assert_array_equal(cnt_single, cnt_same_rays)
        # (2) the whole composition on INDEPENDENT rays: two ray generators agree to an ulp, not to the bit (the reference's CPU and device
        #     generators do not either, test_generate_rays_vs_reference_golden), and an ulp can move an isolated ray across a cell face: at most
        #     0.1 % of the rays may march one sample more or less; every other ray bit-exact, and the pixels within the stated tolerance
        off = cnt_single != cnt
        assert off.mean() <= 1e-3 and np.abs(cnt_single - cnt).max() <= 1, (int(off.sum()), int(np.abs(cnt_single - cnt).max()))
        got_rgb, got_alpha = single['rgb'].cpu().numpy(), single['alpha'].cpu().numpy()
        err = np.abs(got_rgb - col)
        # table amplitude 40: features of +-40 drive hidden activations into the hundreds, the f32 accumulation order of the MFMA chain then
        # moves a pre-activation by several fp16 ulps and ONE saturating sample carries the whole pixel -- 5e-3 there (measured 3.1e-3), the
        # stated 2e-3 on the natural-amplitude model; the mean L1 bound is the same for both
        tol = 2e-3 if table_amp <= 10 else 5e-3
        assert err[~off].max() <= tol and err.mean() <= 2e-4 and err.max() <= 2e-2, (err[~off].max(), err.mean(), err.max())
        assert np.abs(got_alpha - alpha)[~off].max() <= 2e-3
        # the frame exercises what it is meant to: samples in all three cascades' regions, a picture that is not flat
        r = np.abs(xyzs).max(axis=1)
        assert (r < 0.5).any() and ((r >= 0.5) & (r < 1.0)).any() and (r >= 1.0).any()
        assert col.std() > 0.02 and (table_amp > 10 or alpha.std() > 0.02)
        if exponential:
            assert deltas.max() > 3 * deltas.min()
    if table_amp > 10:
        assert (alpha > 0.999).mean() > 0.05                # rays do saturate: the slab order had tiles to drop
        assert int(next(iter(renderer._fused_ws.values()))['skipped'].item()) > 0


def test_fused_image_three_cascades_shards_compose():
    """config C4's decomposition (tile shards of one frame, parallel.shard_range) in the garden configuration: bit for bit the whole frame"""
    from nerficg_amd.instant_ngp import InstantNGPRenderer
    from nerficg_amd.parallel import shard_range
    model = garden_model()
    renderer = InstantNGPRenderer(model, EXPONENTIAL_STEPS=True)
    cam = garden_camera(100, 70)
    pose = inside_pose(2.0, 0.1)
    full = {k: v.clone() for k, v in renderer.render_image_fused(cam, pose, early_termination=False).items()}
    nt = renderer.n_image_tiles(cam)
    out = {k: torch.full_like(v, -1.0) for k, v in full.items()}
    for rank in range(8):
        lo, hi = shard_range(nt, rank, 8)
        renderer.render_image_fused(cam, pose, tile_begin=lo, n_tiles=hi - lo, out=out, early_termination=(rank % 2 == 0))
    for k in ('rgb', 'alpha', 'depth'):
        assert torch.equal(out[k], full[k]), k
