"""GPU: the single-launch forms of the renderer's training batch against the op-by-op expressions they replace (Renderer.py:55-84)."""
import numpy as np
import pytest
import torch

from tests import scenes
from tests.test_gpu_render_parity import make_camera

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _rays(n=5000, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    o = (torch.rand(n, 3, device=DEV, generator=g) - 0.5) * 3.0
    d = torch.nn.functional.normalize(torch.randn(n, 3, device=DEV, generator=g), dim=-1)
    d[::97, 0] = 0.0   # axis-parallel components: the slab test divides by them
    return o.contiguous(), d.contiguous()


def test_clip_rays_equals_intersect_plus_clamps():
    from nerficg_amd import VolumeRenderingV2 as vr
    from nerficg_amd.instant_ngp import InstantNGPModel, InstantNGPRenderer
    model = InstantNGPModel(RANDOM_SEED=0, device=DEV)
    renderer = InstantNGPRenderer(model)
    cam = make_camera(8, 8)
    o, d = _rays()
    o2, d2, span = renderer.clip_rays(o, d, cam)
    ref_o = (o - model.center).contiguous()
    ref = vr.ray_aabb_intersect(ref_o, d, torch.zeros(1, 3, device=DEV), model.half_size, 1)[1][:, 0].clone()
    ref[:, 0].clamp_(min=cam.near_plane)
    ref[:, 1].clamp_(max=cam.far_plane)
    assert torch.equal(o2, ref_o) and torch.equal(d2, d)
    assert torch.equal(span, ref)
    assert int((span[:, 1] > span[:, 0]).sum()) > 100 and int((span[:, 1] < 0).sum()) > 100   # hits and misses both present


def test_composite_over_background_matches_the_torch_expressions():
    from nerficg_amd import VolumeRenderingV2 as vr
    from nerficg_amd.ngp import composite_over_background
    g = torch.Generator(device=DEV).manual_seed(1)
    n_rays, per_ray = 700, 37
    m = n_rays * per_ray
    rays_a = torch.stack([torch.arange(n_rays, device=DEV), torch.arange(n_rays, device=DEV) * per_ray,
                          torch.full((n_rays,), per_ray, device=DEV)], dim=1).to(torch.int64)
    rays_a[::11, 2] = 0   # rays without samples
    deltas = torch.full((m,), 0.02, device=DEV)
    ts = (torch.arange(per_ray, device=DEV, dtype=torch.float32) * 0.02 + 0.3).repeat(n_rays)
    bg = torch.rand(3, device=DEV, generator=g)
    base_s = torch.rand(m, device=DEV, generator=g) * 8.0
    base_c = torch.rand(m, 3, device=DEV, generator=g)
    w_rgb, w_a, w_d = torch.rand(n_rays, 3, device=DEV, generator=g), torch.rand(n_rays, device=DEV, generator=g), torch.rand(n_rays, device=DEV, generator=g)
    results = []
    for fused in (False, True):
        s, c = base_s.clone().requires_grad_(True), base_c.clone().requires_grad_(True)
        if fused:
            rgb, alpha, depth = composite_over_background(s, c, deltas, ts, rays_a, bg, 1e-4)
        else:
            _, alpha, depth_sum, radiance, _ = vr.VolumeRenderer.apply(s, c, deltas, ts, rays_a, 1e-4)
            rgb = radiance + (1 - alpha)[:, None] * bg
            depth = depth_sum / (alpha + 1e-6)
        ((rgb * w_rgb).sum() + (alpha * w_a).sum() + (depth * w_d).sum()).backward()
        results.append((rgb.detach(), alpha.detach(), depth.detach(), s.grad, c.grad))
    for a, b in zip(results[0][:3], results[1][:3]):
        assert torch.equal(a, b)                      # same operations in the same order: bit-equal pixels
    torch.testing.assert_close(results[1][3], results[0][3], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(results[1][4], results[0][4], rtol=1e-5, atol=1e-6)
    # only the colour receives a gradient (the training loss): the NULL-gradient path of the kernels
    s, c = base_s.clone().requires_grad_(True), base_c.clone().requires_grad_(True)
    rgb, _, _ = composite_over_background(s, c, deltas, ts, rays_a, bg, 1e-4)
    (rgb * w_rgb).sum().backward()
    s2, c2 = base_s.clone().requires_grad_(True), base_c.clone().requires_grad_(True)
    _, alpha, _, radiance, _ = vr.VolumeRenderer.apply(s2, c2, deltas, ts, rays_a, 1e-4)
    ((radiance + (1 - alpha)[:, None] * bg) * w_rgb).sum().backward()
    torch.testing.assert_close(s.grad, s2.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(c.grad, c2.grad, rtol=1e-5, atol=1e-6)


def test_rasterizer_abi_host_camera_equals_device_camera_block():
    """include/nerficg_hip.h: the pose either as HOST arrays or as the DEVICE block `camera_dev`; the Python wrapper only uses the second."""
    import ctypes
    from nerficg_amd import _lib
    lib = _lib.load()
    sc = scenes.gs_random_scene(20000, seed=3)
    W, H = 320, 200
    cam = scenes.gs_camera(W, H, scenes.orbit_pose(0.8, 0.35, 4.5))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    means, shs, ops, scl, rot = T(sc['means3D']), T(sc['shs']), T(sc['opacities']), T(sc['scales']), T(sc['rotations'])
    P, M = means.shape[0], shs.shape[1]
    nt = ((W + 15) // 16) * ((H + 15) // 16)
    f = lambda a: (ctypes.c_float * len(a))(*[float(v) for v in a])
    vm, pm, cp = f(cam['viewmatrix'].reshape(-1)), f(cam['projmatrix'].reshape(-1)), f(cam['campos'].reshape(-1))
    block = torch.cat([T(cam['viewmatrix']).reshape(-1), T(cam['projmatrix']).reshape(-1), T(cam['campos']).reshape(-1), torch.zeros(3, device=DEV)]).float()
    outs = []
    mailbox = ctypes.c_void_p()
    _lib.check(lib.nrc_host_mailbox_alloc(ctypes.byref(mailbox)), 'host_mailbox_alloc')
    for use_block in (False, True):
        i32, f32 = torch.int32, torch.float32
        bufs = dict(radii=torch.empty(P, dtype=i32, device=DEV), depths=torch.empty(P, device=DEV), xy=torch.empty(P, 2, device=DEV),
                    co=torch.empty(P, 4, device=DEV), rgb=torch.empty(P, 3, device=DEV), clamped=torch.empty(P, dtype=torch.uint8, device=DEV),
                    cov=torch.empty(P, 6, device=DEV), tt=torch.empty(P, dtype=i32, device=DEV), splat=torch.empty(P, 16, device=DEV),
                    tc=torch.empty(nt, dtype=i32, device=DEV), tf=torch.empty(nt, dtype=i32, device=DEV), ranges=torch.empty(nt, 2, dtype=i32, device=DEV),
                    num=torch.empty(2, dtype=torch.int64, device=DEV))
        hist = torch.empty(int(lib.nrc_gs_bin_hist_bytes(P, W, H, 0)) // 4, dtype=i32, device=DEV)
        cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
        _lib.check(lib.nrc_gs_preprocess(
            P, 3, M, W, H, _lib.ptr(means), _lib.ptr(shs), None, 0, None, _lib.ptr(ops), _lib.ptr(scl), 1.0, _lib.ptr(rot), None,
            None if use_block else cast(vm), None if use_block else cast(pm), None if use_block else cast(cp), _lib.ptr(block) if use_block else None,
            float(cam['tanfovx']), float(cam['tanfovy']), _lib.ptr(bufs['radii']), _lib.ptr(bufs['depths']), _lib.ptr(bufs['xy']), _lib.ptr(bufs['co']),
            _lib.ptr(bufs['rgb']), _lib.ptr(bufs['clamped']), _lib.ptr(bufs['cov']), _lib.ptr(bufs['tt']), _lib.ptr(bufs['tc']), _lib.ptr(bufs['ranges']),
            _lib.ptr(bufs['tf']), _lib.ptr(hist), 0, 0, _lib.ptr(bufs['splat']), _lib.ptr(bufs['num']), mailbox, 41 + int(use_block), _lib.stream_of(means)), 'gs_preprocess')
        torch.cuda.synchronize()
        # ABI 4: the counts also arrive in the host mailbox, the ticket of THIS call behind them
        seen = (ctypes.c_int64 * 3).from_address(mailbox.value)
        assert list(seen) == bufs['num'].tolist() + [41 + int(use_block)]
        outs.append({k: v.clone() for k, v in bufs.items() if k in ('radii', 'depths', 'xy', 'co', 'rgb', 'clamped', 'tt', 'ranges', 'num')})
    assert int(outs[0]['num'][0]) > 10000
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    # neither form given: refused
    rc = lib.nrc_gs_preprocess(P, 3, M, W, H, _lib.ptr(means), _lib.ptr(shs), None, 0, None, _lib.ptr(ops), _lib.ptr(scl), 1.0, _lib.ptr(rot), None,
                               None, None, None, None, float(cam['tanfovx']), float(cam['tanfovy']), _lib.ptr(bufs['radii']), _lib.ptr(bufs['depths']),
                               _lib.ptr(bufs['xy']), _lib.ptr(bufs['co']), _lib.ptr(bufs['rgb']), _lib.ptr(bufs['clamped']), _lib.ptr(bufs['cov']),
                               _lib.ptr(bufs['tt']), _lib.ptr(bufs['tc']), _lib.ptr(bufs['ranges']), _lib.ptr(bufs['tf']), _lib.ptr(hist), 0, 0,
                               _lib.ptr(bufs['splat']), _lib.ptr(bufs['num']), None, 0, _lib.stream_of(means))
    assert rc != 0
    _lib.check(lib.nrc_host_mailbox_free(mailbox), 'host_mailbox_free')


def test_stage_timer_names_and_times_the_kernels_of_an_entry_point():
    """include/nerficg_hip.h group 12: armed, the library records a HIP event behind every kernel of the rasterizer's three entry points; the stages
    come back in launch order with positive times that add up to no more than the wall time of the calls; disarmed, nothing is recorded."""
    import time
    from nerficg_amd import _lib
    from nerficg_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from tests import scenes
    sc = scenes.gs_random_scene(50_000, seed=2, extent=1.0, log_scale_mean=np.log(0.03))
    cam = scenes.gs_camera(320, 240, scenes.orbit_pose(0.5, 0.3, 3.0))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    rs = GaussianRasterizationSettings(image_height=240, image_width=320, tanfovx=cam['tanfovx'], tanfovy=cam['tanfovy'], bg=torch.zeros(3, device=DEV), scale_modifier=1.0,
                                       viewmatrix=T(cam['viewmatrix']), projmatrix=T(cam['projmatrix']), sh_degree=3, campos=T(cam['campos']), prefiltered=False, debug=False)

    def frame():
        means = T(sc['means3D']).requires_grad_(True)
        color, _ = GaussianRasterizer(rs)(means3D=means, means2D=torch.zeros_like(means), opacities=T(sc['opacities'])[:, None], shs=T(sc['shs']),
                                          scales=T(sc['scales']), rotations=T(sc['rotations']))
        color.sum().backward()

    frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with _lib.stage_timer() as st:
        frame(); frame()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3
    names = [n for n, _ in st.stages]
    per = st.by_name()
    for k in ('k_preprocess', 'k_depth_keys', 'k_radix_pass', 'k_span_sweep', 'k_item_scan', 'k_item_scatter', 'k_render', 'k_render_bw', 'k_preprocess_bw'):
        assert k in per, (k, sorted(per))
    assert per['k_radix_pass'][1] == 8 and per['k_render'][1] == 2                         # four radix passes (one launch each) per frame, two frames
    assert 'k_zero_grads' not in per or per['k_zero_grads'][1] <= 1                         # the accumulator records come back cleared: no clearing launch after the first backward
    assert names.index('k_preprocess') < names.index('k_render') < names.index('k_render_bw') < names.index('k_preprocess_bw')
    assert all(t > 0 for _, t in st.stages) and sum(t for _, t in st.stages) <= wall_ms
    with _lib.stage_timer() as empty:
        pass
    frame()                                                                                  # disarmed again: nothing is recorded, nothing leaks into the next use
    with _lib.stage_timer() as again:
        pass
    assert empty.stages == [] and again.stages == []


def test_scaled_mse_loss_equals_mse_loss_times_the_scale():
    """nrc_mse_scaled_forward / _backward against torch.nn.functional.mse_loss + multiplication (Trainer.py:87-89): both outputs, and the
    gradient through either one (the recorded iteration backpropagates the scaled output from a resident one)."""
    from nerficg_amd.ngp import scaled_mse_loss
    g = torch.Generator(device=DEV).manual_seed(3)
    for n in (1, 7, 2200, 40_000):
        pred = torch.rand(n, 3, device=DEV, generator=g, requires_grad=True)
        target = torch.rand(n, 3, device=DEV, generator=g)
        scale = torch.tensor([128.0], device=DEV)
        loss, scaled = scaled_mse_loss(pred, target, scale)
        ref = torch.nn.functional.mse_loss(pred.detach().double(), target.double())
        assert abs(float(loss.detach()) - float(ref)) <= 1e-6 * float(ref) + 1e-12
        assert float(scaled.detach()) == float(loss.detach()) * 128.0
        ref_grad = 2.0 / (3 * n) * (pred.detach() - target)
        (g_scaled,) = torch.autograd.grad(scaled, pred, grad_outputs=torch.ones((), device=DEV), retain_graph=True)
        (g_plain,) = torch.autograd.grad(loss, pred, retain_graph=True)
        (g_both,) = torch.autograd.grad(loss + 2.0 * scaled, pred)
        torch.testing.assert_close(g_plain, ref_grad, rtol=1e-6, atol=1e-12)
        torch.testing.assert_close(g_scaled, 128.0 * ref_grad, rtol=1e-6, atol=1e-12)
        torch.testing.assert_close(g_both, 257.0 * ref_grad, rtol=1e-6, atol=1e-12)
    # the same value on every run (one workgroup, fixed order)
    pred = torch.rand(9000, 3, device=DEV, generator=g); target = torch.rand(9000, 3, device=DEV, generator=g)
    a = [float(scaled_mse_loss(pred, target, torch.ones(1, device=DEV))[0]) for _ in range(3)]
    assert a[0] == a[1] == a[2]
    with pytest.raises(RuntimeError):
        scaled_mse_loss(pred, target[:10], torch.ones(1, device=DEV))


def test_gather_ray_batch_equals_fancy_indexing():
    from nerficg_amd.ngp import gather_ray_batch
    g = torch.Generator(device=DEV).manual_seed(4)
    pool = {k: torch.rand(50_000, 3, device=DEV, generator=g) for k in ('origin', 'view_direction', 'rgb')}
    pool['alpha'] = torch.rand(50_000, device=DEV, generator=g)
    ids = torch.randint(0, 50_000, (2200,), device=DEV, generator=g)
    got = gather_ray_batch(ids, **pool)
    for k, v in pool.items():
        assert torch.equal(got[k], v[ids]), k
    got = gather_ray_batch(ids, pool['origin'], pool['view_direction'])
    assert set(got) == {'origin', 'view_direction'} and torch.equal(got['origin'], pool['origin'][ids])
    # torch's indexing rules (round-3 advisor finding): a negative id counts from the end; an id out of range -- torch raises a device assert --
    # gives a NaN row (never a read outside the pool, never a silent black ray)
    bad = ids.clone(); bad[5] = 50_000; bad[6] = -1; bad[7] = -50_000; bad[8] = -50_001
    got = gather_ray_batch(bad, **pool)
    assert torch.isnan(got['rgb'][5]).all() and torch.isnan(got['alpha'][8]) and torch.isnan(got['origin'][8]).all()
    assert torch.equal(got['origin'][6], pool['origin'][-1]) and float(got['alpha'][7]) == float(pool['alpha'][0])
    assert torch.equal(got['origin'][9:], pool['origin'][ids][9:])
    with pytest.raises(RuntimeError):
        gather_ray_batch(ids, pool['origin'], pool['view_direction'][:100])


def test_train_query_backward_set_equals_zero_fill_plus_accumulate():
    """nrc_ngp_train_query_backward_set on uninitialised (here: poisoned) gradient buffers against the accumulating call on zeroed ones: the hashed
    levels bit for bit (fixed-point sums, written instead of added to zeros), the dense levels / MLP parts up to the order of their float atomics;
    batches above and below the bucketed path's threshold, and an empty one."""
    from nerficg_amd import _lib
    from nerficg_amd.instant_ngp import InstantNGPModel
    from nerficg_amd import ngp
    lib = _lib.load()
    model = InstantNGPModel(RANDOM_SEED=0, device=DEV)
    dn, cn = model.encoding_xyz, model.color_mlp_with_encoding
    gcfg = dn.grid_cfg
    g = torch.Generator(device=DEV).manual_seed(5)
    for m in (40_000, 3000, 0):
        xyz = (torch.rand(m, 3, device=DEV, generator=g) - 0.5) * 0.9
        dirs = torch.nn.functional.normalize(torch.randn(m, 3, device=DEV, generator=g), dim=-1) if m else torch.zeros(0, 3, device=DEV)
        with torch.amp.autocast('cuda'):
            sig, rgb = ngp.query_train(dn, cn, xyz, dirs, model.xyz_min.detach().float().cpu().reshape(3).contiguous(),
                                       model.xyz_size.detach().float().cpu().reshape(3).contiguous())
        d_sig = torch.randn(m, device=DEV, generator=g) * 1e-3
        d_rgb = torch.randn(m, 3, device=DEV, generator=g) * 1e-3
        fn = sig.grad_fn
        x01, h, rgb16, sd_in, sd_acts, sc_in, sc_acts, wd, wc = fn.saved_tensors
        scratch = torch.empty(max(int(lib.nrc_ngp_train_query_scratch_bytes(m)), 16), dtype=torch.uint8, device=DEV)
        common = (_lib.ptr(d_sig), _lib.ptr(d_rgb), m, _lib.ptr(x01), _lib.ptr(wd), _lib.ptr(wc), gcfg['n_levels'], gcfg['log2_hashmap_size'], gcfg['base_resolution'],
                  float(gcfg['per_level_scale']), _lib.ptr(h), _lib.ptr(rgb16), _lib.ptr(sd_in), _lib.ptr(sd_acts), _lib.ptr(sc_in), _lib.ptr(sc_acts), float(dn.loss_scale))
        gd0 = torch.zeros(dn.params.numel(), device=DEV); gc0 = torch.zeros(cn.params.numel(), device=DEV)
        _lib.check(lib.nrc_ngp_train_query_backward(*common, _lib.ptr(gd0), _lib.ptr(gc0), dn.n_mlp_params, _lib.ptr(scratch), _lib.stream_of(gd0)), 'bw')
        gd1 = torch.full((dn.params.numel(),), float('nan'), device=DEV); gc1 = torch.full((cn.params.numel(),), float('nan'), device=DEV)
        _lib.check(lib.nrc_ngp_train_query_backward_set(*common, _lib.ptr(gd1), _lib.ptr(gc1), dn.n_mlp_params, gd1.numel(), gc1.numel(), _lib.ptr(scratch),
                                                        _lib.stream_of(gd1)), 'bw_set')
        assert bool(torch.isfinite(gd1).all()) and bool(torch.isfinite(gc1).all()), m
        if m == 0:
            assert float(gd1.abs().sum()) == 0.0 and float(gc1.abs().sum()) == 0.0
            continue
        scale = float(gd0.abs().max())
        assert scale > 0 and float((gd1 - gd0).abs().max()) <= 2e-3 * scale, (m, float((gd1 - gd0).abs().max()) / scale)
        assert float((gc1 - gc0).abs().max()) <= 2e-3 * float(gc0.abs().max())
        if m >= 16384:   # bucketed path: the hashed levels' sums do not depend on the order of anything
            first_hashed = dn.n_mlp_params + 2 * 351_000   # behind the five dense levels (16^3 .. 60^3 -> < 351 K entries): level 5 starts before this
            assert torch.equal(gd1[first_hashed + 2 * (1 << 19):], gd0[first_hashed + 2 * (1 << 19):])


def _wd_reference(model):
    """Model.py:38-44 as torch expressions (what weight_decay_mlp computed before it became one node)."""
    squares = lambda t: t.square().sum()
    return (squares(model.encoding_xyz.params[:model.n_params_encoding_mlp]) + squares(model.color_mlp_with_encoding.params)) / model.n_mlp_params


def test_weight_decay_node_alone_leaves_the_gradient_of_the_torch_expression():
    """No training query in the graph: nobody picks the gradient seeds up, the end-of-pass callback turns them into ordinary gradients."""
    from nerficg_amd.instant_ngp import InstantNGPModel
    from nerficg_amd import ngp
    model = InstantNGPModel(RANDOM_SEED=3, device=DEV)
    value = model.weight_decay_mlp()
    ref = _wd_reference(model)
    np.testing.assert_allclose(float(value), float(ref), rtol=2e-6)
    (128.0 * 0.5e-6 * value).backward()
    got = [p.grad.clone() for p in model.parameters()]
    assert not ngp._GRAD_SEEDS
    model.zero_grad()
    (128.0 * 0.5e-6 * ref).backward()
    for a, b in zip(got, (p.grad for p in model.parameters())):
        assert torch.allclose(a, b, rtol=1e-6, atol=0) and float(a.abs().max()) > 0
    # under no_grad / for frozen parameters the torch expression answers
    with torch.no_grad():
        np.testing.assert_allclose(float(model.weight_decay_mlp()), float(ref), rtol=2e-6)


def test_weight_decay_seeds_join_the_training_query_gradient():
    """The trainer's loss (Loss.py:15-22) through render_rays: the seeds are consumed by the query's backward (nrc_clear_seed_two) and the parameters'
    gradients equal those of the torch-expression term within the run-to-run spread of the atomics; with a weight decay 1e6 times the reference's so
    that the term shows."""
    from nerficg_amd import ngp
    from nerficg_amd.instant_ngp import InstantNGPRenderer
    from tests.noise import assert_within_run_to_run_noise
    from tests.test_gpu_graphs import _rays as image_rays
    from tests.test_gpu_render_parity import make_model
    cam, o, d = image_rays()
    gen = torch.Generator(device=DEV).manual_seed(2)
    ids = torch.randint(0, o.shape[0], (1024,), device=DEV, generator=gen)
    rgb, bg, noise = torch.rand(1024, 3, device=DEV, generator=gen), torch.rand(3, device=DEV, generator=gen), torch.rand(1024, device=DEV, generator=gen)
    results = []
    for form in ('torch', 'torch', 'node'):
        model = make_model(seed=6, table_amp=1e-4)
        renderer = InstantNGPRenderer(model)
        with torch.amp.autocast('cuda'):
            out = renderer.render_rays(o[ids], d[ids], cam, train_mode=True, custom_bg_color=bg, noise=noise)
            term = model.weight_decay_mlp() if form == 'node' else _wd_reference(model)
            loss = torch.nn.functional.mse_loss(out['rgb'].float(), rgb) + 0.5 * term
        (128.0 * loss).backward()
        assert not ngp._GRAD_SEEDS
        assert out['rm_samples'].device.type == 'cpu' and int(out['rm_samples']) > 0      # the host already knew the count
        results.append([p.grad.detach().clone() for p in model.parameters()])
    assert_within_run_to_run_noise(results[2], results[0], results[1], atol=1e-6, rtol=1e-3, what='gradients with the weight-decay seeds')
    n = results[0][0].numel()
    mlp = slice(0, 3072)
    plain = make_model(seed=6, table_amp=1e-4)     # the term matters at this strength: the MLP gradient without it is elsewhere
    renderer = InstantNGPRenderer(plain)
    with torch.amp.autocast('cuda'):
        out = renderer.render_rays(o[ids], d[ids], cam, train_mode=True, custom_bg_color=bg, noise=noise)
        loss = torch.nn.functional.mse_loss(out['rgb'].float(), rgb)
    (128.0 * loss).backward()
    without = plain.encoding_xyz.params.grad
    assert float((without[mlp] - results[0][0][mlp]).abs().mean()) > 10 * float((results[2][0][mlp] - results[0][0][mlp]).abs().mean())


def test_instant_ngp_loss_module_equals_the_reference_expression_and_the_fused_scaler_step_the_plain_one():
    """InstantNGPLoss (Loss.py:11-26) as one node + GradScaler.step / update as one library call (nrc_amp_adam_step) against the statement-by-statement
    forms: mse_loss + 0.5e-6 * (torch expression of the weight decay), torch.amp.GradScaler around the same FusedAdam.  Three iterations each; the
    weight decay 1e6 times the reference's so that it shows."""
    from nerficg_amd.amp import GradScaler
    from nerficg_amd.apex_optimizers import FusedAdam
    from nerficg_amd.instant_ngp import InstantNGPLoss, InstantNGPRenderer
    from tests.noise import assert_within_run_to_run_noise
    from tests.test_gpu_graphs import _rays as image_rays
    from tests.test_gpu_render_parity import make_model
    cam, o, d = image_rays()
    gen = torch.Generator(device=DEV).manual_seed(8)
    batches = [dict(ids=torch.randint(0, o.shape[0], (1024,), device=DEV, generator=gen), rgb=torch.rand(1024, 3, device=DEV, generator=gen),
                    alpha=torch.rand(1024, device=DEV, generator=gen), bg=torch.rand(3, device=DEV, generator=gen), noise=torch.rand(1024, device=DEV, generator=gen))
               for _ in range(3)]
    results, losses = [], []
    for form in ('plain', 'plain', 'fused'):
        model = make_model(seed=6, table_amp=1e-4)
        renderer = InstantNGPRenderer(model)
        opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)
        scaler = GradScaler(init_scale=128.0, growth_interval=2) if form == 'fused' else torch.amp.GradScaler(init_scale=128.0, growth_interval=2)
        criterion = InstantNGPLoss(model, weight_decay_weight=0.5)
        ls = []
        for b in batches:
            ids = b['ids']
            with torch.amp.autocast('cuda'):
                out = renderer.render_rays(o[ids], d[ids], cam, train_mode=True, custom_bg_color=b['bg'], noise=b['noise'])
                if form == 'fused':
                    loss = criterion(out, {'rgb': b['rgb'], 'alpha': b['alpha']}, b['bg'])
                else:
                    gt = torch.lerp(b['bg'], b['rgb'], b['alpha'][:, None]).clamp(0, 1)
                    loss = torch.nn.functional.mse_loss(out['rgb'].float(), gt) + 0.5 * _wd_reference(model)
            scaler.scale(loss).backward()
            scaler.step(opt); scaler.update(); opt.zero_grad()
            ls.append(float(loss))
        results.append([p.detach().clone() for p in model.parameters()])
        losses.append(ls)
        assert float(scaler.get_scale()) == 256.0 and opt.effective_step(opt.param_groups[0]) == 3      # one growth after two clean steps
        if form == 'fused':
            np.testing.assert_allclose(float(criterion.last[0]), ls[-1], rtol=1e-6)
            np.testing.assert_allclose(float(criterion.psnr()), -10 * np.log10(float(criterion.last[1])), rtol=1e-5)
    np.testing.assert_allclose(losses[2], losses[0], rtol=2e-3)
    assert_within_run_to_run_noise(results[2], results[0], results[1], atol=1e-5, rtol=1e-3, what='three iterations, fused loss node and fused scaler step')


def test_renderer_follows_a_box_that_changes_after_the_first_frame():
    """The renderer keeps HOST copies of the model's box (no read-back per frame); they must be re-read when the buffers are written or replaced --
    a checkpoint with another CENTER loaded into an existing model (advisor finding of round 4)."""
    from nerficg_amd.instant_ngp import InstantNGPModel, InstantNGPRenderer
    model = InstantNGPModel(RANDOM_SEED=0, device=DEV)
    renderer = InstantNGPRenderer(model)
    c0, h0 = renderer._scene_box()
    assert renderer._scene_box()[0] is c0                       # cached while nothing changes
    model.center.add_(0.25)                                     # written in place
    c1, _ = renderer._scene_box()
    assert torch.allclose(c1, c0 + 0.25)
    model.xyz_min = model.xyz_min * 2                           # replaced by a new tensor
    assert torch.allclose(renderer._box()[0], torch.full((3,), -1.0))
    o, d = _rays(2000)
    o2, _, _ = renderer.clip_rays(o, d, make_camera(8, 8))
    assert torch.equal(o2, o - model.center)
