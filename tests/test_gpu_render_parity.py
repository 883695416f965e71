"""End-to-end InstantNGP rendering parity on the GPU.

  ray-list path  = InstantNGPRenderer.render_rays on explicit ray tensors (ray-major ops of the drop-in module: box test, march,
                   two-kernel query, wave-per-ray compositing)
  fused path     = InstantNGPRenderer.render_image_fused (no ray tensors, sample records, one host sync)
  oracle         = CPU composition of oracle/ngp_oracle.c + oracle/tcnn_oracle.c

Pixel tolerance (north_star: "pixel L1 within a stated fp tolerance"): rgb values come out of fp16 sigmoid outputs
(1 ulp = 4.9e-4 near 0.5) blended with f32 weights -> |delta rgb| <= 2e-3 per pixel, mean L1 <= 2e-4.
"""
import numpy as np
import pytest
import torch

import oracle
from tests import scenes

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def make_model(seed=5, table_amp=2.0):
    from nerficg_amd.instant_ngp import InstantNGPModel
    model = InstantNGPModel(RANDOM_SEED=seed, device=DEV)
    with torch.no_grad():
        # amplify the hash table so that density / colour vary over the scene (U(-1e-4,1e-4) would render a constant)
        g = torch.Generator().manual_seed(seed)
        n = model.encoding_xyz.params.numel() - 3072
        model.encoding_xyz.params[3072:] = ((torch.rand(n, generator=g) * 2 - 1) * table_amp).to(DEV)
        model.occupancy_bitfield.copy_(torch.from_numpy(scenes.sphere_bitfield(128, 0.5, 0.35, 1)).to(DEV))
    return model


def make_camera(w, h, bg=(1.0, 1.0, 1.0)):
    from nerficg_amd.instant_ngp import Camera
    fx, fy, cx, cy = scenes.lego_intrinsics(w, h)
    return Camera(width=w, height=h, focal_x=fx, focal_y=fy, center_x=cx, center_y=cy, near_plane=0.2, far_plane=1000.0,
                  background_color=torch.tensor(bg))


@pytest.fixture(scope='module')
def setup():
    from nerficg_amd.instant_ngp import InstantNGPRenderer
    model = make_model()
    return model, InstantNGPRenderer(model)


def _oracle_image(model, cam, c2w):
    """march all -> query all -> composite with early-out -> finalise, on the CPU oracle (same sample set as the chunked loop)."""
    w, h = cam.width, cam.height
    from nerficg_amd.raygen import generate_rays
    rays = generate_rays(w, h, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, c2w, want_direction=False)
    o = (rays['origin'] - model.center).cpu().numpy()
    d = rays['view_direction'].cpu().numpy()
    _, ht, _ = oracle.ray_aabb_intersect(o, d, np.zeros((1, 3), np.float32), np.full((1, 3), 0.5, np.float32), 1)
    hits = ht[:, 0].copy()
    hits[:, 0] = np.maximum(hits[:, 0], np.float32(cam.near_plane))
    hits[:, 1] = np.minimum(hits[:, 1], np.float32(cam.far_plane))
    bf = model.occupancy_bitfield.cpu().numpy()
    rays_a, xyzs, dirs, deltas, ts, counter = oracle.raymarching_train(o, d, hits, bf, 1, 0.5, 0.0, np.zeros(len(o), np.float32), 128, 1024)
    pd = model.encoding_xyz.params.detach().half().float().cpu().numpy()
    pc = model.color_mlp_with_encoding.params.detach().half().float().cpu().numpy()
    x01 = (xyzs - np.float32(-0.5)) / np.float32(1.0)
    grid_kw = {k: model.encoding_xyz.grid_cfg[k] for k in ('n_levels', 'log2_hashmap_size', 'base_resolution', 'per_level_scale')}
    sig, rgb, _ = oracle.ngp_query(x01, dirs, pd[:3072], pc, pd[3072:].reshape(-1, 2), **grid_kw)
    total, alpha, depth, col, ws = oracle.composite_train_fw(sig, rgb, deltas, ts, rays_a, 1e-4)
    alpha = np.clip(alpha, 0, 1)
    T = 1 - alpha
    col = np.clip(col + T[:, None] * cam.background_color.numpy()[None], 0, 1)
    depth = np.where(T < 1, depth / np.where(alpha > 0, alpha, 1), 0)
    return col, alpha, depth, int(counter[0])


@pytest.mark.parametrize('w,h,pose', [(96, 80, (0.7, 0.4)), (64, 64, (2.9, -0.6))])
def test_fused_image_equals_ray_list_image_and_oracle(setup, w, h, pose):
    from nerficg_amd.raygen import generate_rays
    model, renderer = setup
    cam = make_camera(w, h, bg=(1.0, 0.5, 0.25))
    c2w = scenes.orbit_pose(pose[0], pose[1], scenes.LEGO_RADIUS)
    rays = generate_rays(w, h, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, c2w, want_direction=False)
    ref = renderer.render_rays(rays['origin'], rays['view_direction'], cam)  # ray-major ops on explicit ray tensors
    fused = renderer.render_image_fused(cam, c2w, return_stats=True)
    o_rgb, o_alpha, o_depth, o_total = _oracle_image(model, cam, c2w)
    assert fused['n_samples'] == o_total  # bit-exact sample set (index parity)
    for name, a, b, tol in (('rgb', fused['rgb'], ref['rgb'].reshape(-1, 3), 2e-3), ('alpha', fused['alpha'], ref['alpha'].reshape(-1), 1e-3)):
        diff = (a - b).abs()
        assert diff.max().item() <= tol and diff.mean().item() <= tol / 10, (name, diff.max().item(), diff.mean().item())
    hit = ref['alpha'].reshape(-1) > 1e-3
    dd = (fused['depth'] - ref['depth'].reshape(-1)).abs()[hit]
    assert dd.max().item() <= 5e-3
    # against the CPU oracle
    np.testing.assert_allclose(fused['rgb'].cpu().numpy(), o_rgb, rtol=0, atol=2e-3)
    np.testing.assert_allclose(fused['alpha'].cpu().numpy(), o_alpha, rtol=0, atol=1e-3)
    assert np.abs(fused['rgb'].cpu().numpy() - o_rgb).mean() <= 2e-4
    # the scene is not trivial: some pixels hit, some miss, alpha varies
    assert 0.2 < hit.float().mean().item() < 0.8 and fused['alpha'].std().item() > 0.05


def test_fused_image_shards_compose(setup):
    """Multi-GPU sharding unit: disjoint ranges of 8x8 tiles rendered separately reproduce the full image bit for bit
    (image size not a multiple of 8: edge tiles are partially outside)."""
    _, renderer = setup
    cam = make_camera(70, 50)
    c2w = scenes.orbit_pose(1.3, 0.2, scenes.LEGO_RADIUS)
    full = {k: v.clone() for k, v in renderer.render_image_fused(cam, c2w).items()}
    nt = renderer.n_image_tiles(cam)
    assert nt == 9 * 7
    out = {'rgb': torch.full((70 * 50, 3), -1.0, device=DEV), 'alpha': torch.full((70 * 50,), -1.0, device=DEV), 'depth': torch.full((70 * 50,), -1.0, device=DEV)}
    cuts = [0, 11, 12, 40, nt]
    for a, b in zip(cuts[:-1], cuts[1:]):
        renderer.render_image_fused(cam, c2w, tile_begin=a, n_tiles=b - a, out=out)
    for k in ('rgb', 'alpha', 'depth'):
        assert torch.equal(out[k], full[k]), k


def test_render_image_is_the_fused_picture_reshaped(setup):
    _, renderer = setup
    cam = make_camera(48, 40)
    c2w = scenes.orbit_pose(0.2, 0.1, scenes.LEGO_RADIUS)
    flat = {k: v.clone() for k, v in renderer.render_image_fused(cam, c2w).items()}
    img = renderer.render_image(cam, c2w, to_chw=True)
    assert img['rgb'].shape == (3, 40, 48) and img['alpha'].shape == (1, 40, 48)
    assert torch.equal(img['rgb'].permute(1, 2, 0).reshape(-1, 3), flat['rgb']) and torch.equal(img['depth'].reshape(-1), flat['depth'])


def test_ray_list_chunks_do_not_change_the_picture(setup):
    from nerficg_amd.raygen import generate_rays
    _, renderer = setup
    cam = make_camera(40, 36)
    rays = generate_rays(40, 36, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, scenes.orbit_pose(1.1, 0.3, scenes.LEGO_RADIUS), want_direction=False)
    whole = renderer.render_rays(rays['origin'], rays['view_direction'], cam)
    renderer.RAY_CHUNK = 500
    try:
        parts = renderer.render_rays(rays['origin'], rays['view_direction'], cam)
    finally:
        del renderer.RAY_CHUNK
    for k in whole:
        assert torch.equal(whole[k], parts[k]), k


@pytest.mark.parametrize('optimizer', ['torch_adam', 'fused_adam'])
def test_training_iterations_reduce_loss(setup, optimizer):
    """InstantNGP/Trainer.py:79-94 call sequence (autocast, random bg, GradScaler 128, Adam eps 1e-15) on a constant-colour target, with
    torch.optim.Adam and with the shipped apex replacement (Trainer.py:33-38) -- whose kernel writes the parameters through a raw pointer
    and therefore has to keep the fp16 compute copy of the tinycudann modules fresh itself."""
    from nerficg_amd.instant_ngp import InstantNGPRenderer
    from nerficg_amd.apex_optimizers import FusedAdam
    model = make_model(seed=9, table_amp=1e-4)
    renderer = InstantNGPRenderer(model)
    cam = make_camera(64, 64)
    from nerficg_amd.raygen import generate_rays
    rays = generate_rays(64, 64, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, scenes.orbit_pose(0.5, 0.3, scenes.LEGO_RADIUS), want_direction=False)
    if optimizer == 'fused_adam':
        opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)
    else:
        opt = torch.optim.Adam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99))
    scaler = torch.amp.GradScaler(init_scale=128.0, growth_interval=10 ** 6)
    target = torch.tensor([0.8, 0.3, 0.1], device=DEV)
    losses = []
    torch.manual_seed(0)
    probe = torch.rand(512, 3, device=DEV)
    first_forward = model.encoding_xyz(probe).float().clone()
    for it in range(30):
        with torch.amp.autocast('cuda'):
            bg = torch.rand(3, device=DEV)
            out = renderer.render_rays(rays['origin'], rays['view_direction'], cam, train_mode=True, custom_bg_color=bg)
            gt = target * out['alpha'].detach()[:, None] + bg * (1 - out['alpha'].detach()[:, None])
            loss = torch.nn.functional.mse_loss(out['rgb'].float(), gt) + 0.5e-6 * model.weight_decay_mlp()
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        opt.zero_grad()
        losses.append(loss.item())
        assert int(out['rm_samples'].item()) > 0
        if it == 0:  # ONE optimizer step must change what the network computes
            assert not torch.equal(model.encoding_xyz(probe).float(), first_forward)
    assert np.isfinite(losses).all() and losses[-1] < 0.5 * losses[0], losses
    # the fp16 compute copy equals a fresh conversion of the trained fp32 master parameters
    for net in (model.encoding_xyz, model.color_mlp_with_encoding):
        assert torch.equal(net._half_params(), net.params.detach().half())


def test_update_occupancy_grid_runs_and_packs(setup):
    from nerficg_amd.instant_ngp import InstantNGPRenderer
    model = make_model(seed=3)
    renderer = InstantNGPRenderer(model)
    torch.manual_seed(0)
    renderer.update_occupancy_grid(warmup=True)
    g = model.occupancy_grid
    assert (g > 0).any()
    used, mean = renderer.occupancy_threshold.tolist()  # stays on the device in production; read here to check it
    np.testing.assert_allclose(mean, g[g > 0].double().mean().item(), rtol=1e-6)
    assert used == np.float32(min(mean, renderer.density_threshold))
    np.testing.assert_array_equal(model.occupancy_bitfield.cpu().numpy(), oracle.packbits(g.float().cpu().numpy(), used))
    before = g.clone()
    renderer.update_occupancy_grid(warmup=False)
    assert bool((model.occupancy_grid >= before * 0.95 - 1e-12).all())  # EMA: never below the decayed value


def test_draw_update_cells_positions_and_choices():
    """nrc_occupancy_draw_cells (Renderer.py:183-206 cell choice, :251-258 positions): Morton indices in range, every position inside its own
    cell (lattice centre +- s/G, numpy restatement of the lattice), the second half of a cascade drawn only from cells above the threshold,
    both halves spread over their populations, ignored entries where a cascade has no occupied cell, same seed -> same draws."""
    from nerficg_amd.instant_ngp import InstantNGPModel, InstantNGPRenderer
    G = 32
    model = InstantNGPModel(RANDOM_SEED=2, SCALE=1.0, RESOLUTION=G, device=DEV)
    renderer = InstantNGPRenderer(model)
    occupied = np.sort(np.random.default_rng(0).choice(G ** 3, size=700, replace=False))
    with torch.no_grad():
        model.occupancy_grid[0, torch.from_numpy(occupied).to(DEV)] = 2 * renderer.density_threshold
    seed = torch.tensor([1234567], dtype=torch.int64, device=DEV)
    idx, pts = renderer.draw_update_cells(warmup=False, seed=seed)
    n = G ** 3 // 4
    assert idx.shape == (2, 2 * n) and pts.shape == (2 * 2 * n, 3)
    idx_h, pts_h = idx.cpu().numpy(), pts.cpu().numpy().reshape(2, 2 * n, 3)
    assert (idx_h[1, n:] == -1).all() and (idx_h[0] >= 0).all() and (idx_h[1, :n] >= 0).all() and idx_h.max() < G ** 3
    assert np.isin(idx_h[0, n:], occupied).all() and len(np.unique(idx_h[0, n:])) > 600   # only occupied cells, nearly all of them drawn
    assert len(np.unique(idx_h[0, :n])) > 0.2 * G ** 3                                      # uniform half covers the grid
    for c in range(2):
        s_c = min(2.0 ** (c - 1), 1.0)
        live = idx_h[c] >= 0
        coords = oracle.morton3D_invert(idx_h[c][live].astype(np.int32)).astype(np.float64)
        centre = (coords / (G - 1) * 2 - 1) * (s_c - s_c / G)
        off = np.abs(pts_h[c][live] - centre)
        assert off.max() <= s_c / G * (1 + 1e-5) and off.mean() > 0.3 * s_c / G  # jitter fills the cell
    idx2, pts2 = renderer.draw_update_cells(warmup=False, seed=seed)
    assert torch.equal(idx, idx2) and torch.equal(pts, pts2)
    idx3, _ = renderer.draw_update_cells(warmup=False, seed=seed + 1)
    assert not torch.equal(idx, idx3)
    all_idx, all_pts = renderer.draw_update_cells(warmup=True, seed=seed)
    assert torch.equal(all_idx, torch.arange(G ** 3, device=DEV).expand(2, -1))


def test_update_occupancy_grid_with_a_cascade_without_occupied_cells():
    """SCALE 1.0 -> two cascades; only cascade 0 has occupied cells: cascade 1's second half consists of ignored (-1) entries (the reference
    draws fewer samples there, Renderer.py:190-198).  Carved (-1) cells must survive."""
    from nerficg_amd.instant_ngp import InstantNGPModel, InstantNGPRenderer
    model = InstantNGPModel(RANDOM_SEED=2, SCALE=1.0, RESOLUTION=64, device=DEV)
    assert model.cascades == 2
    renderer = InstantNGPRenderer(model)
    with torch.no_grad():
        model.occupancy_grid[0, :5000] = 1.0
        model.occupancy_grid[1, 100:200] = -1.0
    torch.manual_seed(1)
    renderer.update_occupancy_grid(warmup=False)
    g = model.occupancy_grid
    assert bool((g[1, 100:200] == -1.0).all()) and bool((g[0, :5000] >= 0.95 - 1e-6).all())
    assert int((g[1] > 0).sum()) > 1000  # cascade 1 received its n uniform samples (densities of a random-init net are positive)
    used, mean = renderer.occupancy_threshold.tolist()
    np.testing.assert_allclose(mean, g[g > 0].double().mean().item(), rtol=1e-6)
    np.testing.assert_array_equal(model.occupancy_bitfield.cpu().numpy(), oracle.packbits(g.float().cpu().numpy(), used))


@pytest.mark.parametrize('cascades,dtype,n_samples', [(1, torch.float32, 20000), (2, torch.float16, 70001), (3, torch.float32, 0)])
def test_occupancy_update_call_matches_oracle(cascades, dtype, n_samples):
    """C ABI group 11 against oracle.occupancy_update (Renderer.py:258-272): grid bit-exact, threshold within f32 rounding of the f64 mean,
    bitfield bit-exact at the threshold the device used; carved (negative) cells untouched; duplicate and padded (-1) indices."""
    from nerficg_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(cascades * 1000 + n_samples)
    N = 64 ** 3
    grid = (rng.random((cascades, N)) * 0.5).astype(np.float32)
    grid[rng.random((cascades, N)) < 0.2] = -1.0
    grid[rng.random((cascades, N)) < 0.3] = 0.0
    idx = rng.integers(0, N, size=(cascades, n_samples)).astype(np.int64)  # duplicates are certain at these counts
    if n_samples:
        idx[:, -5:] = -1
    den = (rng.random((cascades, n_samples)) * 2.0).astype(np.float16 if dtype == torch.float16 else np.float32)
    tg = torch.from_numpy(grid.copy()).to(DEV)
    ti, td = torch.from_numpy(idx).to(DEV), torch.from_numpy(den).to(DEV)
    bits = torch.full((cascades * N // 8,), 0xAA, dtype=torch.uint8, device=DEV)
    thr = torch.zeros(2, device=DEV)
    ws = torch.empty(int(lib.nrc_occupancy_update_ws_bytes(cascades * N)), dtype=torch.uint8, device=DEV)
    _lib.check(lib.nrc_occupancy_update(_lib.ptr(tg), _lib.ptr(ti), _lib.ptr(td), 0 if dtype == torch.float32 else 1, cascades, N, n_samples, 0.95, 0.01,
                                        _lib.ptr(bits), _lib.ptr(thr), _lib.ptr(ws), _lib.stream_of(tg)), 'occupancy_update')
    want_grid, _, want_thr, want_mean = oracle.occupancy_update(grid, idx, den, 0.95, 0.01)
    np.testing.assert_array_equal(tg.cpu().numpy(), want_grid)
    used, mean = thr.tolist()
    np.testing.assert_allclose(mean, want_mean, rtol=1e-6)
    np.testing.assert_allclose(used, want_thr, rtol=1e-6)
    np.testing.assert_array_equal(bits.cpu().numpy(), oracle.packbits(want_grid, used))


def test_occupancy_update_without_positive_cells_clears_the_bitfield():
    from nerficg_amd import _lib
    lib = _lib.load()
    N = 32 ** 3
    tg = torch.full((1, N), -1.0, device=DEV)
    tg[0, ::3] = 0.0
    bits = torch.full((N // 8,), 0xFF, dtype=torch.uint8, device=DEV)
    thr = torch.zeros(2, device=DEV)
    ws = torch.empty(int(lib.nrc_occupancy_update_ws_bytes(N)), dtype=torch.uint8, device=DEV)
    _lib.check(lib.nrc_occupancy_update(_lib.ptr(tg), None, None, 0, 1, N, 0, 0.95, 0.01, _lib.ptr(bits), _lib.ptr(thr), _lib.ptr(ws), _lib.stream_of(tg)),
               'occupancy_update')
    assert bool(torch.isnan(thr).all()) and int(bits.max()) == 0  # the reference's empty mean is NaN: `cell > NaN` is never true
    assert lib.nrc_occupancy_update(_lib.ptr(tg), None, None, 0, 1, N + 4, 0, 0.95, 0.01, _lib.ptr(bits), _lib.ptr(thr), _lib.ptr(ws), None) == -1


def test_carve_occupancy_grid_matches_an_independent_numpy_statement():
    """Renderer.py:208-245: frustum carving (+ alpha masks) and the 3x3x3 dilation, against numpy/scipy on the same cell positions."""
    from scipy import ndimage
    from nerficg_amd.instant_ngp import InstantNGPModel, InstantNGPRenderer
    model = InstantNGPModel(RANDOM_SEED=1, RESOLUTION=32, device=DEV)
    renderer = InstantNGPRenderer(model)
    cam = make_camera(40, 30)
    views = []
    for k, (az, el, rad) in enumerate([(0.3, 0.2, 1.1), (2.0, -0.4, 0.9)]):
        alpha = torch.zeros(1, 30, 40)
        alpha[:, 5:22, 8 + 6 * k:30] = 1.0
        views.append((cam, scenes.orbit_pose(az, el, rad), alpha))
    R = 32
    coords = model.grid_coords.cpu().numpy().astype(np.float64)
    morton = oracle.morton3D(model.grid_coords.cpu().numpy())
    half = 0.5 / R
    pos = (coords / (R - 1) * 2 - 1) * (0.5 - half)
    for subtractive, use_alpha in ((False, False), (True, False), (False, True)):
        model.occupancy_grid.zero_()
        renderer.carve_occupancy_grid(views, subtractive=subtractive, use_alpha=use_alpha)
        keep = np.full(R ** 3, subtractive)
        for cam_, c2w, alpha in views:
            c2w = np.asarray(c2w, np.float64)
            pc = (pos - c2w[:3, 3]) @ c2w[:3, :3]
            z = pc[:, 2]
            xy = pc[:, :2] / np.maximum(z, 1e-8)[:, None] * np.array([cam_.focal_x, cam_.focal_y]) + np.array([cam_.center_x, cam_.center_y])
            ins = (xy >= 0).all(1) & (xy[:, 0] < cam_.width) & (xy[:, 1] < cam_.height) & (z > cam_.near_plane) & (z < cam_.far_plane)
            if use_alpha:
                mask = ndimage.binary_dilation(alpha[0].numpy() > 0, structure=np.ones((3, 3)))
                px = np.floor(xy[ins]).astype(int)
                ins[np.nonzero(ins)[0]] = mask[px[:, 1], px[:, 0]]
            keep = keep & ins if subtractive else keep | ins
        dil = ndimage.binary_dilation(keep.reshape(R, R, R), structure=np.ones((3, 3, 3))).reshape(-1)
        expect = np.zeros(R ** 3, np.float32)
        expect[morton] = np.where(dil, 0.0, -1.0)
        got = model.occupancy_grid[0].cpu().numpy()
        assert 0 < (expect < 0).sum() < R ** 3
        # cells whose projection lies within float rounding of a frustum edge may differ (f32 on the device vs f64 here)
        assert np.mean(got != expect) < 2e-3, (subtractive, use_alpha, np.mean(got != expect))


@pytest.mark.parametrize('table_amp', [2.0, 60.0])  # 60: densities up to e^several -> most rays saturate within a few samples
def test_layer_ordered_early_termination_gives_the_same_image(table_amp):
    """Depth-slab processing with finished tiles dropping out (Renderer.py:118-132 semantics at slab granularity) against the
    single pass over all samples: the per-ray arithmetic is the same sequence, so the images are bit-identical."""
    from nerficg_amd.instant_ngp import InstantNGPRenderer
    model = make_model(seed=9, table_amp=table_amp)
    renderer = InstantNGPRenderer(model)
    cam = make_camera(200, 152, bg=(0.2, 0.5, 0.9))
    for pose in ((0.4, 0.3), (2.2, -0.2)):
        c2w = scenes.orbit_pose(pose[0], pose[1], scenes.LEGO_RADIUS)
        a = {k: v.clone() for k, v in renderer.render_image_fused(cam, c2w, early_termination=False).items()}
        b = renderer.render_image_fused(cam, c2w, early_termination=True)
        for k in ('rgb', 'alpha', 'depth'):
            assert torch.equal(a[k], b[k]), k
    if table_amp > 10:
        assert (a['alpha'] > 0.999).float().mean() > 0.05  # the dense variant does saturate
    # shards compose in layer order too
    nt = renderer.n_image_tiles(cam)
    out = {k: torch.zeros_like(v) for k, v in a.items()}
    for lo, hi in ((0, nt // 3), (nt // 3, nt)):
        renderer.render_image_fused(cam, c2w, tile_begin=lo, n_tiles=hi - lo, out=out, early_termination=True)
    for k in ('rgb', 'alpha', 'depth'):
        assert torch.equal(a[k], out[k]), k


@pytest.mark.parametrize('m', [777, 20011])  # >= 16384: the hashed levels take the ownership backward
def test_fused_training_query_matches_the_op_by_op_modules(m):  # noqa: E302
    """nerficg_amd.ngp.query_train (one autograd node) against the statement sequence of query_model (Renderer.py:48-53) through the
    drop-in modules: same forward bits, gradients within the fp16 rounding of the intermediate d_out tensors."""
    from nerficg_amd.instant_ngp import InstantNGPRenderer
    import nerficg_amd.VolumeRenderingV2 as vr
    model = make_model(seed=4)
    renderer = InstantNGPRenderer(model)

    def op_by_op(x, d):  # the module-level statement sequence of the reference's query (Renderer.py:48-53) on the drop-in modules
        h = model.encoding_xyz((x - model.xyz_min) / model.xyz_size)
        return vr.TruncExp.apply(h[:, 0]), model.color_mlp_with_encoding(torch.cat([(d * 0.5 + 0.5).to(h.dtype), h], dim=-1))
    rng = np.random.default_rng(m)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    x = T((rng.random((m, 3)) - 0.5).astype(np.float32) * 0.98)
    d = rng.normal(size=(m, 3)).astype(np.float32)
    d = T(d / np.linalg.norm(d, axis=1, keepdims=True))
    gs, gr = T(rng.normal(size=m).astype(np.float32) * 1e-2), T(rng.normal(size=(m, 3)).astype(np.float32) * 1e-2)  # fp16-safe magnitudes
    res = {}
    for fused in (False, True):
        model.zero_grad()
        with torch.amp.autocast('cuda'):
            sig, rgb = renderer.query(x, d) if fused else op_by_op(x, d)
        (sig.float() * gs).sum().add((rgb.float() * gr).sum()).backward()
        res[fused] = (sig.detach().float().clone(), rgb.detach().float().clone(), model.encoding_xyz.params.grad.clone(),
                      model.color_mlp_with_encoding.params.grad.clone())
    assert torch.equal(res[False][0], res[True][0]) and torch.equal(res[False][1], res[True][1])
    for k, name in ((2, 'grid net'), (3, 'colour net')):
        a, b = res[False][k].cpu().numpy(), res[True][k].cpu().numpy()
        assert np.isfinite(a).all() and np.isfinite(b).all()
        mlp = slice(0, 3072) if k == 2 else slice(0, a.size)
        np.testing.assert_allclose(b[mlp], a[mlp], rtol=0, atol=2e-2 * np.abs(a[mlp]).max(), err_msg=name)
        if k == 2:
            ta, tb = a[3072:], b[3072:]
            assert np.abs(ta).max() > 0
            np.testing.assert_allclose(tb, ta, rtol=0, atol=2e-2 * np.abs(ta).max(), err_msg='hash table')
            assert np.abs(tb - ta).mean() < 2e-3 * np.abs(ta).mean() + 1e-12


def test_pipelined_image_equals_the_one_pass_image():
    """render_image_pipelined (the march of tile range k + 1 on a side stream next to the encode / MLP kernels of range k) paints the picture of
    render_image_fused bit for bit, reports the same sample count, and can be called back to back (buffers of a range are reused every frame)."""
    from nerficg_amd.instant_ngp import InstantNGPRenderer
    model = make_model()
    renderer = InstantNGPRenderer(model)
    cam = make_camera(96, 80, bg=(0.2, 0.4, 0.6))
    poses = [scenes.orbit_pose(0.3 + 0.8 * k, 0.2, scenes.LEGO_RADIUS) for k in range(3)]
    for shards in (1, 3, 5):
        for pose in poses:
            ref = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in renderer.render_image_fused(cam, pose, return_stats=True, early_termination=False).items()}
            got = renderer.render_image_pipelined(cam, pose, shards=shards, return_stats=True)
            assert got['n_samples'] == ref['n_samples'] and got['n_rows'] >= ref['n_rows']   # every range pads its own last row
            for k in ('rgb', 'alpha', 'depth'):
                assert torch.equal(got[k], ref[k]), (shards, k)


@pytest.mark.parametrize('cascades_esf', [(1, 0.0), (3, 1.0 / 256)])
def test_arena_in_place_and_mailbox_change_nothing(cascades_esf):
    """The single-pass frame queries the samples where the count pass parked them (ARENA_IN_PLACE: no copy into compact rows, row_tile from the
    SH launch) and takes its row count from a host mailbox (COUNT_MAILBOX): pictures, row and sample counts equal those of the copied / read-back
    frame bit for bit -- also on a shard of the image, with several occupancy cascades and exponential steps, and frame after frame (the arena of
    a frame is overwritten by the next one)."""
    from nerficg_amd.instant_ngp import InstantNGPRenderer
    cascades, esf = cascades_esf
    if cascades > 1:
        from tests.test_gpu_garden_parity import garden_model
        model = garden_model()
        assert model.cascades == cascades
    else:
        model = make_model()
    cam = make_camera(104, 72, bg=(0.1, 0.3, 0.5))
    a, b = InstantNGPRenderer(model, EXPONENTIAL_STEPS=esf > 0), InstantNGPRenderer(model, EXPONENTIAL_STEPS=esf > 0)
    b.ARENA_IN_PLACE = False
    b.COUNT_MAILBOX = False
    n_tiles = a.n_image_tiles(cam)
    for k in range(4):
        pose = scenes.orbit_pose(0.4 + 0.9 * k, 0.15 + 0.1 * k, scenes.LEGO_RADIUS if cascades == 1 else 1.15)
        for (t0, nt) in ((0, None), (n_tiles // 3, n_tiles // 2)):
            for slabs in (False, True):   # the single pass and the slab order (which has its own arena form: row_k)
                ra = {kk: (v.clone() if torch.is_tensor(v) else v) for kk, v in
                      a.render_image_fused(cam, pose, tile_begin=t0, n_tiles=nt, return_stats=True, early_termination=slabs).items()}
                rb = b.render_image_fused(cam, pose, tile_begin=t0, n_tiles=nt, return_stats=True, early_termination=slabs)
                assert ra['n_rows'] == rb['n_rows'] and ra['n_samples'] == rb['n_samples'] and ra['n_samples'] > 0
                for key in ('rgb', 'alpha', 'depth'):
                    assert torch.equal(ra[key], rb[key]), (k, t0, slabs, key)


def test_fixed_row_capacity_frame_needs_no_host_read_and_can_be_recorded():
    """render_image_fused(row_capacity=N) (round 4): the frame is only ENQUEUED -- sample buffers for N rows, the kernels take the number of rows that exist
    from the device counter -- so it paints the picture of the sized call, reports an overflow through the counter without touching memory it does not
    own, and can sit in a HIP graph (the review's item 8: no D2H copy between nrc_ngp_render_count and nrc_ngp_composite_image)."""
    from nerficg_amd.instant_ngp import InstantNGPRenderer
    model = make_model()
    cam = make_camera(96, 80, bg=(0.2, 0.4, 0.6))
    poses = [scenes.orbit_pose(0.3 + 0.8 * k, 0.2, scenes.LEGO_RADIUS) for k in range(3)]
    sized = InstantNGPRenderer(model)
    fixed = InstantNGPRenderer(model)
    refs = []
    for pose in poses:
        ref = sized.render_image_fused(cam, pose, return_stats=True, early_termination=False)
        refs.append(({k: ref[k].clone() for k in ('rgb', 'alpha', 'depth')}, ref['n_rows'], ref['n_samples']))
    cap = int(1.2 * max(r[1] for r in refs)) + 8
    for pose, (ref, rows, samples) in zip(poses, refs):
        got = fixed.render_image_fused(cam, pose, row_capacity=cap)
        assert got['counter'].is_cuda and got['counter'].tolist() == [rows, samples]
        for k in ('rgb', 'alpha', 'depth'):
            assert torch.equal(got[k], ref[k]), k
    # far too small a capacity: nothing is written or read outside the buffers (the run survives), and the counter says so
    small = InstantNGPRenderer(model)
    got = small.render_image_fused(cam, poses[0], row_capacity=max(refs[0][1] // 3, 1))
    torch.cuda.synchronize()
    assert got['counter'].tolist()[0] == refs[0][1] > refs[0][1] // 3 and bool(torch.isfinite(got['rgb']).all())
    # recorded: a replay repaints the frame of the recorded pose (the pose is a kernel argument) into the same buffers
    fixed.render_image_fused(cam, poses[1], row_capacity=cap)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fixed.render_image_fused(cam, poses[1], row_capacity=cap)
    out['rgb'].zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out['rgb'], refs[1][0]['rgb']) and out['counter'].tolist() == [refs[1][1], refs[1][2]]


def test_mirror_follows_the_reference_orchestration_fixture(golden_dir):
    """tests/golden/ingp_orchestration.npz = the reference's OWN host code (InstantNGPRenderer.render_rays -> InstantNGPRayRenderingComponent, Renderer.py:30-180,
    through custom_functions.py) run in the build container on the oracle's ops (tests/test_shims.py reproduces it from the reference there).  The GPU
    mirror -- nerficg_amd.instant_ngp.InstantNGPRenderer on the HIP ops, own structure -- must paint the same training batch (same jitter, custom
    background, training depth) and the same inference picture (camera background, the alive-ray loop's early-out, inference depth) on those rays:
    the reference's orchestration is pinned by execution.  Tolerances: the fp16 network outputs differ from the oracle's by an fp16 ulp at most."""
    from nerficg_amd.instant_ngp import Camera, InstantNGPModel, InstantNGPRenderer
    fx = np.load(golden_dir / 'ingp_orchestration.npz')
    w, h = (int(v) for v in fx['size'])
    model = InstantNGPModel(RANDOM_SEED=int(fx['seed']), device=DEV)
    with torch.no_grad():
        model.encoding_xyz.params[model.n_params_encoding_mlp:] *= float(fx['table_gain'])
        model.occupancy_bitfield.copy_(torch.from_numpy(scenes.sphere_bitfield(model.RESOLUTION, model.SCALE, float(fx['radius']), model.cascades)).to(DEV))
    renderer = InstantNGPRenderer(model)
    f = fx['intr']
    cam = Camera(width=w, height=h, focal_x=float(f[0]), focal_y=float(f[1]), center_x=float(f[2]), center_y=float(f[3]), near_plane=0.2, far_plane=1000.0,
                 background_color=torch.tensor([1.0, 1.0, 1.0]))
    # the rays themselves: the reference's View.get_rays against nrc_generate_rays
    from nerficg_amd.raygen import generate_rays
    rays = generate_rays(w, h, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, fx['c2w'], device=DEV, want_direction=False)
    np.testing.assert_allclose(rays['origin'].cpu().numpy(), fx['origin'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(rays['view_direction'].cpu().numpy(), fx['view_direction'], rtol=0, atol=2e-6)
    o, d = torch.from_numpy(fx['origin']).to(DEV), torch.from_numpy(fx['view_direction']).to(DEV)
    out = renderer.render_rays(o, d, cam, train_mode=True, custom_bg_color=torch.from_numpy(fx['bg']).to(DEV), noise=torch.from_numpy(fx['noise']).to(DEV))
    assert int(out['rm_samples']) == int(fx['train_rm_samples'])                      # the march is integer-exact
    for key, tol in (('rgb', 4e-3), ('alpha', 4e-3), ('depth', 2e-2)):
        got, want = out[key].detach().float().cpu().numpy(), fx['train_' + key]
        assert np.abs(got - want).max() <= tol and np.abs(got - want).mean() <= tol / 20, (key, np.abs(got - want).max(), np.abs(got - want).mean())
    with torch.no_grad():
        ev = renderer.render_rays(o, d, cam, train_mode=False)
    for key, tol in (('rgb', 4e-3), ('alpha', 4e-3), ('depth', 2e-2)):
        got, want = ev[key].float().cpu().numpy(), fx['eval_' + key]
        assert np.abs(got - want).max() <= tol and np.abs(got - want).mean() <= tol / 20, (key, np.abs(got - want).max(), np.abs(got - want).mean())
    assert float(fx['train_alpha'].max()) > 0.3 and float((fx['eval_alpha'] > 0).mean()) > 0.3    # the fixture is not an empty picture


def test_pose_dependent_encoder_shape_paints_the_same_picture():
    """InstantNGPRenderer.POSE_ENCODER_SHAPE: the brick of samples a wave of the encoder gathers for follows the camera's axes (nrc_ngp_set_encoder_shape).
    Every shape computes the same features: the picture of a pose whose image columns run along world x (shape 4 x 4 x 1) and of one looking down x
    (4 x 2 x 8) is bit-identical to the picture with the fixed default brick."""
    from nerficg_amd.instant_ngp import InstantNGPRenderer
    model = make_model()
    cam = make_camera(96, 80)
    def look(fwd, down):
        fwd, down = np.asarray(fwd, np.float64), np.asarray(down, np.float64)
        fwd /= np.linalg.norm(fwd)
        right = np.cross(down, fwd); right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        c2w = np.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, -1.3 * fwd
        return c2w
    poses = {'columns along x': look((0.1, 0.2, 1.0), (1.0, 0.1, 0.0)), 'looking down x': look((1.0, 0.05, 0.1), (0.0, 1.0, 0.0)), 'rows along x': look((0.0, 0.3, 1.0), (0.0, 1.0, 0.0))}
    shapes = set()
    for name, c2w in poses.items():
        pics = {}
        for on in (True, False):
            r = InstantNGPRenderer(model)
            r.POSE_ENCODER_SHAPE = on
            out = r.render_image_fused(cam, c2w, early_termination=False)
            pics[on] = {k: out[k].clone() for k in ('rgb', 'alpha', 'depth')}
            if on:
                shapes.add(r._frame_constants(cam, c2w)['enc_shape'])
        for k in pics[True]:
            assert torch.equal(pics[True][k], pics[False][k]), (name, k)
        assert float(pics[True]['alpha'].max()) > 0.5, name
    assert shapes == {(3, 1), (2, 2), (2, 1)}, shapes
    from nerficg_amd import _lib
    _lib.load().nrc_ngp_set_encoder_shape(-1, -1)
