"""GPU: the fused InstantNGP training iteration (nerficg_amd.ngp_trainer, include/nerficg_hip.h group 13) against the same iteration through the
drop-in modules (src/Methods/InstantNGP/Trainer.py:79-94 sequence).

Pinned here: (1) every piece of the fused batch preparation equals the op it replaces (gather, clip, apply_background_color, capped march) bit for
bit; (2) the fused compositing + loss kernel equals the five-op chain it replaces (pixels, loss, sample gradients); (3) whole iterations -- eager,
recorded, with the next batch marched ahead on a side stream -- leave the parameters the op-by-op loop leaves, within that loop's own run-to-run
spread (tests/noise.py); (4) the device-side GradScaler / step-counter rule on an overflow; (5) the resident sampling order, the device batch size
and the generator's reproducibility.
"""
import numpy as np
import pytest
import torch

from tests import scenes
from tests.test_gpu_graphs import _rays, _train_pair

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _pool(n_extra_alpha=False, seed=3, size=64):
    cam, o, d = _rays(size, size)
    g = torch.Generator(device=DEV).manual_seed(seed)
    pool = {'origin': o, 'view_direction': d, 'rgb': torch.rand(o.shape[0], 3, device=DEV, generator=g)}
    if n_extra_alpha:
        pool['alpha'] = torch.rand(o.shape[0], device=DEV, generator=g)
    return cam, pool


def _fused(model, renderer, cam, pool, n, cap, **kw):
    from nerficg_amd.amp import GradScaler
    from nerficg_amd.apex_optimizers import FusedAdam
    from nerficg_amd.ngp_trainer import FusedTrainingIteration
    opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False, capturable=True)
    scaler = GradScaler(init_scale=128.0, growth_interval=10 ** 6)
    return FusedTrainingIteration(model, renderer, opt, scaler, cam, pool, n, cap, **kw), opt, scaler


@pytest.mark.parametrize('with_alpha', [False, True])
def test_batch_preparation_equals_the_ops_it_replaces(with_alpha):
    """rays_o / rays_d / hits_t / target / rays_a / samples of nrc_ngp_train_march against gather + clip_rays + lerp + the capped march."""
    from nerficg_amd import VolumeRenderingV2 as vr
    cam, pool = _pool(with_alpha)
    model, renderer, _ = _train_pair(seed=2)
    n, cap = 1500, 400_000
    it, _, _ = _fused(model, renderer, cam, pool, 2048, cap, prefetch=False, graph=False)
    it.set_batch_size(n)
    g = torch.Generator(device=DEV).manual_seed(1)
    ids = torch.randint(0, pool['origin'].shape[0], (2048,), device=DEV, generator=g)
    ids[7] = -3      # torch's negative indexing
    bg, noise = torch.rand(3, device=DEV, generator=g), torch.rand(2048, device=DEV, generator=g)
    it.ids.copy_(ids); it.bg_in.copy_(bg); it.noise_in.copy_(noise)
    it._march(0, it._state_tensors(), (True, True, True))
    b = it.sets[0]
    o, d, span = renderer.clip_rays(pool['origin'][ids[:n]], pool['view_direction'][ids[:n]], cam)
    assert torch.equal(b.rays_o[:n], o) and torch.equal(b.rays_d[:n], d) and torch.equal(b.hits_t[:n], span)
    assert torch.equal(b.bg, bg)
    want = pool['rgb'][ids[:n]]
    if with_alpha:
        want = torch.lerp(bg, want, pool['alpha'][ids[:n], None]).clamp(0, 1)
    assert torch.equal(b.target[:n], want)
    assert int(b.counter[1]) == n and bool((b.hits_t[n:, 1] == -1).all())       # rows behind the live rays miss everything
    m = model
    rays_a, xyzs, dirs, deltas, ts, counter, overflow = vr.raymarching_train(o, d, span, m.occupancy_bitfield, m.cascades, m.SCALE, 0.0, noise[:n].contiguous(),
                                                                             m.RESOLUTION, renderer.MAX_SAMPLES, sample_capacity=cap, return_overflow=True)
    total = int(counter[0])
    assert 0 < total < cap and int(b.counter[0]) == total and int(b.overflow[0]) == 0
    assert torch.equal(b.rays_a[:n], rays_a) and bool((b.rays_a[n:, 2] == 0).all())
    for mine, ref in ((b.xyzs, xyzs), (b.dirs, dirs), (b.deltas, deltas), (b.ts, ts)):
        assert torch.equal(mine, ref)


def test_generator_draws_are_uniform_reproducible_and_independent_of_the_cut():
    """Jitter / background of the device generator: in [0, 1), flat, a pure function of (seed, iteration, global ray index) -- two halves of a
    batch marched by two 'ranks' (ray_offset) draw what the whole batch draws."""
    cam, pool = _pool()
    model, renderer, _ = _train_pair(seed=2)
    n = 2048
    order = torch.arange(pool['origin'].shape[0], device=DEV)
    whole, _, _ = _fused(model, renderer, cam, pool, n, 500_000, prefetch=False, graph=False, order=order, seed=77)
    st = whole._state_tensors()
    whole._march(0, st, (False, False, False))
    bg0 = whole.sets[0].bg.clone()
    assert bool(((bg0 >= 0) & (bg0 < 1)).all())
    halves = []
    for r in range(2):
        half, _, _ = _fused(model, renderer, cam, pool, n // 2, 300_000, prefetch=False, graph=False, order=order[r * n // 2:], seed=77, ray_offset=r * n // 2)
        half._march(0, half._state_tensors(), (False, False, False))
        assert torch.equal(half.sets[0].bg, bg0)
        halves.append(half.sets[0])
    w = whole.sets[0]
    w_ts = w.ts.clone()
    assert torch.equal(torch.cat([h.rays_a[:, 2] for h in halves]), w.rays_a[:, 2])          # same jitter -> same sample counts per ray
    assert torch.equal(torch.cat([h.ts[:int(h.counter[0])] for h in halves]), w.ts[:int(w.counter[0])])
    # a second iteration draws other numbers, the same seed replays the first
    whole._march(0, st, (False, False, False))
    assert not torch.equal(whole.sets[0].bg, bg0) and int(whole.cursor) == 2 * n and int(whole.rng[1]) == 2
    again, _, _ = _fused(model, renderer, cam, pool, n, 500_000, prefetch=False, graph=False, order=order, seed=77)
    again._march(0, again._state_tensors(), (False, False, False))
    assert torch.equal(again.sets[0].bg, bg0) and torch.equal(again.sets[0].ts, w_ts)
    # uniformity of the jitter: recover it from the first sample of rays that start inside an occupied cell is indirect -- draw directly instead
    big, _, _ = _fused(model, renderer, cam, pool, 4096, 1_000_000, prefetch=False, graph=False, order=order, seed=5)
    t_first = []
    for _ in range(3):
        big._march(0, big._state_tensors(), (False, False, False))
        b = big.sets[0]
        hit = b.rays_a[:, 2] > 0
        dt = 3 ** 0.5 / renderer.MAX_SAMPLES
        t_first.append(((b.ts[b.rays_a[hit, 1]] - b.hits_t[hit, 0]) / dt) % 1.0)
    u = torch.cat(t_first)
    assert u.numel() > 1500 and abs(float(u.mean()) - 0.5) < 0.03 and abs(float(u.var()) - 1 / 12) < 0.01


def test_generator_is_philox4x32_10_bit_for_bit():
    """The in-kernel generator of the fused iteration against tests/philox_ref.py (numpy Philox4x32-10, pinned on Random123's known-answer vectors in the
    CPU suite): the background colour of iterations 0 .. 2 bit for bit, and every marched ray's jitter -- recovered from its first sample, which sits
    on the lattice t0 + (jitter + k) dt -- within the float error of the march's additions."""
    from tests import philox_ref
    cam, pool = _pool(size=160)      # 25 600 rays: three batches of 2 048 stay inside the order
    model, renderer, _ = _train_pair(seed=2)
    n, seed, offset = 2048, 0x1234_5678_9abc, 4096
    order = torch.randperm(pool['origin'].shape[0], generator=torch.Generator().manual_seed(3)).to(DEV)   # every batch sees the object
    it, _, _ = _fused(model, renderer, cam, pool, n, 600_000, prefetch=False, graph=False, order=order, seed=seed, ray_offset=offset)
    dt = 3 ** 0.5 / renderer.MAX_SAMPLES
    checked = 0
    for iteration in range(3):
        it._march(0, it._state_tensors(), (False, False, False))
        b = it.sets[0]
        np.testing.assert_array_equal(b.bg.cpu().numpy(), philox_ref.background(seed, iteration))
        want = philox_ref.u01(philox_ref.train_draw(seed, iteration, 0, np.arange(offset, offset + n, dtype=np.uint64))[0])
        hit = (b.rays_a[:, 2] > 0).cpu().numpy()
        first = ((b.ts[b.rays_a[:, 1].clamp(min=0)] - b.hits_t[:, 0]) / dt).cpu().numpy()
        resid = (first - want + 0.5) % 1.0 - 0.5          # distance to the lattice point, wrap-aware
        assert hit.sum() > 300, hit.sum()
        assert np.abs(resid[hit]).max() < 2e-3, np.abs(resid[hit]).max()   # measured 1.1e-3: ~600 additions of dt in f32 behind the first candidate
        checked += int(hit.sum())
    assert checked > 1000


def test_two_rank_slices_of_a_batch_average_to_the_whole_batch_gradient():
    """Data parallelism of the fused iteration without a process group: 'rank' r marches rows [r n/2, (r + 1) n/2) of the global batch (its slice of the
    order, ray_offset = r n/2, the same seed -> the same background and, per GLOBAL ray index, the same jitter).  The loss is a mean over rays, so the
    average of the two ranks' gradient buffers -- what parallel.allreduce_flat(average=True) leaves on every rank -- must be the gradient of one
    rank marching the whole batch; the marched sample counts add up exactly."""
    cam, pool = _pool(size=96)
    n = 1024
    order = torch.randperm(pool['origin'].shape[0], generator=torch.Generator().manual_seed(8)).to(DEV)

    def grads(n_rays, order_r, offset):
        model, renderer, _ = _train_pair(seed=4)
        it, _, _ = _fused(model, renderer, cam, pool, n_rays, 300_000, prefetch=False, graph=False, fused_step=False, order=order_r, seed=31, ray_offset=offset)
        out = it()
        return it.grads.detach().clone(), int(out['rm_samples']), float(out['loss']), it.sets[0].bg.clone()

    g_all, m_all, loss_all, bg_all = grads(n, order, 0)
    g_again, _, _, _ = grads(n, order, 0)
    parts = [grads(n // 2, order[r * n // 2:], r * n // 2) for r in range(2)]
    assert parts[0][1] + parts[1][1] == m_all and m_all > 50_000
    assert torch.equal(parts[0][3], bg_all) and torch.equal(parts[1][3], bg_all)
    np.testing.assert_allclose(0.5 * (parts[0][2] + parts[1][2]), loss_all, rtol=1e-5)
    g_avg = 0.5 * (parts[0][0] + parts[1][0])
    scale = float(g_all.abs().max())
    noise = float((g_all - g_again).abs().max())                      # float atomics on the coarse levels: two runs of ONE code differ by this much
    err = float((g_avg - g_all).abs().max())
    assert err <= max(4 * noise, 2e-4 * scale), (err, noise, scale)
    assert abs(float(torch.dot(g_avg, g_all) / torch.dot(g_all, g_all)) - 1.0) < 1e-4


@pytest.mark.parametrize('case', ['empty_grid', 'one_ray', 'tiny_capacity'])
@pytest.mark.parametrize('fused_step', [True, False])
def test_edge_batches_run_clean(case, fused_step):
    """Batches a trainer meets at the edges: an occupancy grid without a set bit (no sample at all: the pixels are the background, the hash table gets
    no gradient, the MLP weights move by their weight decay alone), a batch of ONE ray, and a sample capacity a tenth of what the batch marches (the
    cut is reported, nothing is written out of bounds, the step is taken).  Four iterations each, marching ahead."""
    cam, pool = _pool(size=96)
    model, renderer, _ = _train_pair(seed=4)
    n, cap = (1 if case == 'one_ray' else 512), (4096 if case == 'tiny_capacity' else 100_000)
    if case == 'empty_grid':
        with torch.no_grad():
            model.occupancy_bitfield.zero_()
    order = torch.randperm(pool['origin'].shape[0], generator=torch.Generator().manual_seed(8)).to(DEV)
    it, opt, scaler = _fused(model, renderer, cam, pool, n, cap, prefetch=True, graph=False, fused_step=fused_step, order=order, seed=1)
    dn = model.encoding_xyz
    table0 = dn.params.detach()[dn.n_mlp_params:].clone()
    mlp0 = dn.params.detach()[:dn.n_mlp_params].clone()
    outs = [it() for _ in range(4)]
    torch.cuda.synchronize()
    o = outs[-1]
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters()) and bool(torch.isfinite(o['loss']))
    assert opt.effective_step(opt.param_groups[0]) == 4 and float(scaler.get_scale()) == 128.0
    if case == 'empty_grid':
        assert int(o['rm_samples']) == 0 and int(o['sample_overflow']) == 0
        assert torch.equal(dn.params.detach()[dn.n_mlp_params:], table0)                       # no sample, no table gradient, no table update
        assert float((dn.params.detach()[:dn.n_mlp_params] - mlp0).abs().max()) > 0             # weight decay alone moves the MLP weights
        ids = order[3 * n:4 * n]                                                                 # the last batch: its pixels are the background colour
        want = float(((o['bg'][None] - pool['rgb'][ids]) ** 2).mean())
        np.testing.assert_allclose(float(o['loss']), want, rtol=1e-5)
    elif case == 'one_ray':
        assert 0 <= int(o['rm_samples']) <= renderer.MAX_SAMPLES and int(o['sample_overflow']) == 0
    else:
        assert int(o['rm_samples']) > 10 * cap // 2 and int(o['sample_overflow']) == int(o['rm_samples']) - cap


def test_fused_loss_kernel_equals_the_chain_it_replaces():
    """nrc_ngp_train_loss against composite_over_background + scaled_mse_loss + their autograd backward on the same samples."""
    from nerficg_amd import _lib
    from nerficg_amd.ngp import composite_over_background, scaled_mse_loss
    cam, pool = _pool()
    model, renderer, _ = _train_pair(seed=5)
    n, cap = 2048, 450_000
    it, _, _ = _fused(model, renderer, cam, pool, n, cap, prefetch=False, graph=False, order=torch.arange(pool['origin'].shape[0], device=DEV), seed=1)
    it.set_batch_size(1900)
    st = it._state_tensors()
    it._march(0, st, (False, False, False))
    b = it.sets[0]
    g = torch.Generator(device=DEV).manual_seed(0)
    sig = (torch.rand(cap, device=DEV, generator=g) * 40).requires_grad_()          # dense enough that many rays saturate
    rgb = torch.rand(cap, 3, device=DEV, generator=g).requires_grad_()
    scale = torch.tensor([128.0], device=DEV)
    lib, p = _lib.load(), _lib.ptr
    gd, gc = torch.full((1000,), 7.0, device=DEV), torch.full((300,), 7.0, device=DEV)
    d_sig, d_rgb = torch.full((cap,), 9.0, device=DEV), torch.full((cap, 3), 9.0, device=DEV)
    _lib.check(lib.nrc_ngp_train_loss(p(sig.detach()), p(rgb.detach()), p(b.deltas), p(b.ts), p(b.rays_a), p(b.counter), n, cap, 1e-4, p(b.bg), p(b.target), p(scale),
                                      p(it.ray_rgb), p(it.ray_alpha), None, p(it.loss2), p(d_sig), p(d_rgb), p(gd), 600, p(gc), 300, p(it.loss_ws),
                                      _lib.stream_of(sig)), 'ngp_train_loss')
    live = 1900
    out_rgb, out_alpha, _ = composite_over_background(sig, rgb, b.deltas, b.ts, b.rays_a[:live].contiguous(), b.bg, 1e-4)
    loss, scaled = scaled_mse_loss(out_rgb, b.target[:live].contiguous(), scale)
    scaled.backward()
    assert torch.allclose(it.ray_rgb[:live], out_rgb.detach(), rtol=0, atol=1e-6) and torch.allclose(it.ray_alpha[:live], out_alpha.detach(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(float(it.loss2[0]), float(loss), rtol=2e-6)
    np.testing.assert_allclose(float(it.loss2[1]), float(scaled), rtol=2e-6)
    for mine, ref in ((d_sig, sig.grad), (d_rgb, rgb.grad)):
        tol = 1e-5 * float(ref.abs().max())
        assert float((mine - ref).abs().max()) <= tol, (float((mine - ref).abs().max()), tol)
    assert int((d_sig != 0).sum()) > 10_000 and int((d_sig[: int(b.counter[0])] == 0).sum()) > 1000     # saturated tails exist and are zero
    assert bool((gd[:600] == 0).all()) and bool((gd[600:] == 7).all()) and bool((gc == 0).all())          # the clearing ranges, nothing else
    assert int(it.loss_ws[:17 * 64].view(torch.int32).abs().sum()) == 0                                                     # the ticket is back at zero


def _op_by_op(batches, seed, cam, pool, weight_decay=0.5e-6):
    """the iteration through the drop-in modules, weight decay as FusedAdam's L2 slice (what the fused iteration does)"""
    from nerficg_amd.amp import GradScaler
    from nerficg_amd.apex_optimizers import FusedAdam
    model, renderer, _ = _train_pair(seed=seed)
    opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)
    coeff = 2 * weight_decay / model.n_mlp_params
    opt.set_l2_slice(model.encoding_xyz.params, model.n_params_encoding_mlp, coeff)
    opt.set_l2_slice(model.color_mlp_with_encoding.params, model.color_mlp_with_encoding.params.numel(), coeff)
    scaler = GradScaler(init_scale=128.0, growth_interval=10 ** 6)
    losses, marched = [], []
    for b in batches:
        ids = b['ids']
        with torch.amp.autocast('cuda'):
            out = renderer.render_rays(pool['origin'][ids], pool['view_direction'][ids], cam, train_mode=True, custom_bg_color=b['bg'], noise=b['noise'])
            loss = torch.nn.functional.mse_loss(out['rgb'].float(), pool['rgb'][ids])
        scaler.scale(loss).backward()
        scaler.step(opt); scaler.update(); opt.zero_grad()
        losses.append(float(loss)); marched.append(int(out['rm_samples']))
    return losses, marched, [p.detach().clone() for p in model.parameters()], model


def test_fused_iteration_equals_one_iteration_of_the_cpu_oracle():
    """ONE training iteration, end to end, restated on the CPU oracle (oracle/*.c): march -> hash-grid encode -> density MLP -> TruncExp -> [SH | h] ->
    colour MLP -> compositing -> pixel over background -> MSE -> compositing backward -> both MLP backwards -> hash-grid backward, on the batch, jitter
    and background the fused iteration used.  Compared: the marched samples (bit for bit), the loss, and the three gradient blocks the fused iteration
    hands to Adam (colour MLP, density MLP, hash table) -- tolerances ten to thirty times the measured differences (fp16 intermediates are restated by
    the oracle; what is left is MFMA summation order and float atomics on the coarse levels)."""
    import oracle
    from tests.test_gpu_render_parity import make_model
    from nerficg_amd.instant_ngp import InstantNGPRenderer
    cam, pool = _pool(size=96)
    model = make_model(seed=7, table_amp=1.0)        # a table with structure: densities and colours vary over the scene
    renderer = InstantNGPRenderer(model)
    n, S = 384, 128.0
    g = torch.Generator(device=DEV).manual_seed(2)
    ids = torch.randint(0, pool['origin'].shape[0], (n,), device=DEV, generator=g)
    bg, noise = torch.rand(3, device=DEV, generator=g), torch.rand(n, device=DEV, generator=g)
    dn, cn = model.encoding_xyz, model.color_mlp_with_encoding
    p_d = oracle.round_half(dn.params.detach().cpu().numpy())
    p_c = oracle.round_half(cn.params.detach().cpu().numpy())
    n_mlp_d = dn.n_mlp_params
    it, opt, scaler = _fused(model, renderer, cam, pool, n, 120_000, prefetch=False, graph=False, fused_step=False)
    out = it(ids=ids, bg=bg, noise=noise)
    b = it.sets[0]
    m = int(b.counter[0])
    assert 20_000 < m <= 120_000 and int(out['sample_overflow']) == 0
    # ---- march (the clipped rays are the kernel's: nrc_ngp_clip_rays is compared with the oracle in test_gpu_ngp_parity.py)
    cpu = lambda t: t.detach().cpu().numpy()
    r = renderer
    rays_a, xyzs, dirs, deltas, ts, counter = oracle.raymarching_train(cpu(b.rays_o), cpu(b.rays_d), cpu(b.hits_t), cpu(model.occupancy_bitfield), model.cascades,
                                                                       float(model.SCALE), 0.0, cpu(noise), model.RESOLUTION, r.MAX_SAMPLES)
    assert int(counter[0]) == m
    np.testing.assert_array_equal(cpu(b.rays_a), rays_a)
    for name, want in (('xyzs', xyzs), ('dirs', dirs), ('deltas', deltas), ('ts', ts)):
        np.testing.assert_array_equal(cpu(getattr(b, name))[:m], want[:m], err_msg=name)
    # ---- query
    mn, sz = (cpu(t) for t in r._box())
    x01 = (xyzs[:m] - mn) / sz          # (x - xyz_min) / xyz_size, Renderer.py:50
    grid = dict(n_levels=16, log2_hashmap_size=19, base_resolution=16, per_level_scale=float(dn.grid_cfg['per_level_scale']))
    table = p_d[n_mlp_d:].reshape(-1, 2)
    enc = oracle.grid_encode_fw(x01.astype(np.float32), table, **grid)
    h, acts_d = oracle.mlp_fw(enc, p_d[:n_mlp_d], n_hidden=1, out_act=0, want_acts=True)
    h16 = oracle.round_half(h)
    sigmas = np.exp(h16[:, 0])
    d01 = oracle.round_half(dirs[:m] * np.float32(0.5) + np.float32(0.5))
    cin = np.concatenate([oracle.sh4_encode(d01), h16], 1)
    rgb_out, acts_c = oracle.mlp_fw(cin, p_c, n_hidden=2, out_act=1, want_acts=True)
    rgbs = oracle.round_half(rgb_out[:, :3])
    # ---- compositing, pixel, loss
    T_thr = it.T_THRESHOLD
    _, opacity, depth, rgb, ws = oracle.composite_train_fw(sigmas, rgbs, deltas[:m], ts[:m], rays_a, T_thr)
    bg_np, target = cpu(bg), cpu(pool['rgb'][ids])
    pixel = rgb + (1.0 - opacity)[:, None] * bg_np[None]
    loss = float(((pixel - target) ** 2).mean())
    assert abs(float(out['loss']) - loss) <= 1e-5 * loss, (float(out['loss']), loss)      # measured 3e-7
    # ---- backward: d(S * loss) / d pixel, through the compositor, the two networks and the encoding
    d_pix = (2.0 * S / (3 * n)) * (pixel - target)
    ds, dr = oracle.composite_train_bw(-(d_pix * bg_np[None]).sum(1), np.zeros(n, np.float32), d_pix.astype(np.float32), np.zeros(m, np.float32), sigmas, rgbs, ws,
                                       deltas[:m], ts[:m], rays_a, opacity, depth, rgb, T_thr)
    g_pad = np.zeros((m, 16), np.float32)
    g_pad[:, :3] = dr * 128.0                                  # tiny-cuda-nn's internal loss scale, applied before the fp16 rounding of dZ
    dW_c, d_cin = oracle.mlp_bw(cin, p_c, rgb_out, acts_c, g_pad, n_hidden=2, out_act=1)
    d_h = d_cin[:, 16:] / 128.0
    d_h[:, 0] += ds * np.exp(np.clip(h16[:, 0], -15.0, 15.0))  # TruncExp backward (custom_functions.py:207-210)
    d_h16 = oracle.round_half(d_h)
    dW_d, d_enc = oracle.mlp_bw(enc, p_d[:n_mlp_d], h, acts_d, d_h16 * 128.0, n_hidden=1, out_act=0)
    g_table = oracle.grid_encode_bw(x01.astype(np.float32), d_enc / 128.0, table.shape[0], **grid)
    # ---- the fused iteration's gradient buffers (scaled by the GradScaler's S; cleared regions are exact zeros)
    got_c, got_d = cpu(it.gc) / S, cpu(it.gd) / S
    dW_c, dW_d = dW_c / (128.0 * S), dW_d / (128.0 * S)
    g_table = g_table / S
    # measured on MI355X: largest error 3e-5 (colour MLP), 7e-6 (density MLP), 6e-5 (table) of each block's largest entry
    np.testing.assert_allclose(got_c, dW_c, rtol=2e-2, atol=5e-4 * np.abs(dW_c).max())
    np.testing.assert_allclose(got_d[:n_mlp_d], dW_d, rtol=2e-2, atol=5e-4 * np.abs(dW_d).max())
    got_t = got_d[n_mlp_d:].reshape(-1, 2)
    np.testing.assert_allclose(got_t, g_table, rtol=2e-2, atol=1e-3 * np.abs(g_table).max())
    assert np.count_nonzero(g_table) > 10_000 and np.mean((got_t == 0) != (g_table == 0)) < 1e-3
    print('fused iteration vs oracle iteration: samples', m, 'loss', float(out['loss']), loss, 'max |dW_c err| / max', np.abs(got_c - dW_c).max() / np.abs(dW_c).max(),
          'dW_d', np.abs(got_d[:n_mlp_d] - dW_d).max() / np.abs(dW_d).max(), 'table', np.abs(got_t - g_table).max() / np.abs(g_table).max())
    # correlation: a scale error or a missing factor cannot hide behind the tolerances
    for a_, b_ in ((got_c, dW_c), (got_d[:n_mlp_d], dW_d), (got_t.ravel(), g_table.ravel())):
        assert abs(float(np.dot(a_, b_) / np.dot(b_, b_)) - 1.0) < 2e-3


@pytest.mark.parametrize('graph,fused_step', [(False, True), (True, True), (False, False), (True, False)])
def test_fused_iterations_follow_the_op_by_op_iterations(graph, fused_step):
    from tests.noise import assert_within_run_to_run_noise
    cam, pool = _pool()
    n = 2048
    g = torch.Generator(device=DEV).manual_seed(11)
    batches = [dict(ids=torch.randint(0, pool['origin'].shape[0], (n,), device=DEV, generator=g), bg=torch.rand(3, device=DEV, generator=g),
                    noise=torch.rand(n, device=DEV, generator=g)) for _ in range(6)]
    l0, m0, p0, _ = _op_by_op(batches, 9, cam, pool)
    l2, m2, p2, _ = _op_by_op(batches, 9, cam, pool)
    model, renderer, _ = _train_pair(seed=9)
    it, opt, scaler = _fused(model, renderer, cam, pool, n, 400_000, prefetch=False, graph=graph, fused_step=fused_step)
    l1, m1 = [], []
    for b in batches:
        out = it(ids=b['ids'], bg=b['bg'], noise=b['noise'])
        l1.append(float(out['loss'])); m1.append(int(out['rm_samples']))
        assert int(out['sample_overflow']) == 0
    assert m0 == m1 == m2, (m0, m1)
    spread = max(abs(a - b) / abs(a) for a, b in zip(l0, l2))
    np.testing.assert_allclose(l1, l0, rtol=max(2e-3, 4 * spread))
    p1 = [p.detach().clone() for p in model.parameters()]
    assert_within_run_to_run_noise(p1, p0, p2, atol=1e-4, rtol=1e-2, what='parameters after six fused iterations')
    assert opt.effective_step(opt.param_groups[0]) == 6 and float(scaler.get_scale()) == 128.0
    for net in (model.encoding_xyz, model.color_mlp_with_encoding):
        assert torch.equal(net._half_params(), net.params.detach().half())     # the fp16 compute copy was written by the fused Adam launch
    if graph:
        assert len(it._graphs) == 1


def test_marching_ahead_changes_nothing_but_the_schedule():
    """Resident order + device generator: prefetch on (recorded, two buffer sets, side stream) and off (eager) see the same batches and the same
    draws, so they must agree like two runs of one code; an occupancy update in between is honoured by the prefetch=False call in front of it."""
    from tests.noise import assert_within_run_to_run_noise
    cam, pool = _pool(size=160)
    n = 2048
    order = torch.randperm(pool['origin'].shape[0], generator=torch.Generator().manual_seed(4)).to(DEV)
    results = {}
    for mode in ('plain', 'plain_again', 'ahead'):
        model, renderer, _ = _train_pair(seed=3)
        it, opt, _ = _fused(model, renderer, cam, pool, n, 400_000, prefetch=(mode == 'ahead'), graph=(mode == 'ahead'), order=order, seed=21)
        losses, marched = [], []
        for i in range(9):
            before_update = i == 4
            out = it(prefetch=False) if (before_update or mode != 'ahead') else it()
            losses.append(float(out['loss'])); marched.append(int(out['rm_samples']))
            if before_update:      # the bitfield changes between iterations 4 and 5: batch 5 must be marched against the new one
                model.occupancy_bitfield.copy_(torch.from_numpy(scenes.sphere_bitfield(128, 0.5, 0.25, 1)).to(DEV))
        results[mode] = (losses, marched, [p.detach().clone() for p in model.parameters()])
        assert int(it.cursor) == (9 if mode != 'ahead' else 10) * n      # marching ahead has consumed one more batch
        if mode == 'ahead':
            assert len(it._graphs) >= 3 and it.remaining_batches() == (order.numel() - 10 * n) // n
    (l0, m0, p0), (l2, m2, p2), (l1, m1, p1) = results['plain'], results['plain_again'], results['ahead']
    assert m0 == m1 == m2 and m0[5] < m0[4], (m0, m1)
    spread = max(abs(a - b) / abs(a) for a, b in zip(l0, l2))
    np.testing.assert_allclose(l1, l0, rtol=max(2e-3, 4 * spread))
    assert_within_run_to_run_noise(p1, p0, p2, atol=1e-4, rtol=1e-2, what='parameters after nine iterations, batches marched ahead')


@pytest.mark.parametrize('fused_step', [True, False])
def test_overflow_skips_the_step_and_backs_the_scale_off(fused_step):
    """A loss scale that overflows the scaled loss (the case the GradScaler exists for): every gradient is inf / NaN, so parameters and moments
    stay, the step counter stands still, the scale is backed off and the growth tracker resets -- torch.amp.GradScaler's rule
    (Trainer.py:89-91), executed on the device; the next iteration is taken again."""
    cam, pool = _pool()
    model, renderer, _ = _train_pair(seed=1)
    n = 1024
    it, opt, scaler = _fused(model, renderer, cam, pool, n, 200_000, prefetch=False, graph=True, fused_step=fused_step)
    g = torch.Generator(device=DEV).manual_seed(3)
    ids = lambda: torch.randint(0, pool['origin'].shape[0], (n,), device=DEV, generator=g)
    it(ids=ids()); it(ids=ids())
    assert opt.effective_step(opt.param_groups[0]) == 2 and int(scaler._growth_tracker) == 2
    scaler._scale.fill_(3e38)
    before = [p.detach().clone() for p in model.parameters()]
    m_before = opt.state[model.encoding_xyz.params]['exp_avg'].clone()
    out = it(ids=ids())
    assert np.isfinite(float(out['loss'])) and float(it.loss2[1]) > 1e37      # the scaled loss still fits f32; its gradients do not fit the networks' fp16
    for a, b in zip(before, model.parameters()):
        assert torch.equal(a, b.detach())
    assert torch.equal(m_before, opt.state[model.encoding_xyz.params]['exp_avg'])
    assert opt.effective_step(opt.param_groups[0]) == 2 and float(scaler.get_scale()) == float(np.float32(3e38) * np.float32(0.5)) and int(scaler._growth_tracker) == 0
    scaler._scale.fill_(128.0)
    it(ids=ids())
    assert opt.effective_step(opt.param_groups[0]) == 3 and float(scaler.get_scale()) == 128.0 and int(scaler._growth_tracker) == 1
    assert not torch.equal(before[0], model.encoding_xyz.params.detach())


@pytest.mark.parametrize('poison', [None, 'a', 'b'])
def test_amp_adam_step_equals_check_prepare_adam_update(poison):
    """nrc_amp_adam_step (two launches) against nrc_nonfinite_check4 + nrc_adam_prepare + 2 x nrc_adam_step + torch's GradScaler.update on the
    same numbers: bit-identical parameters / moments / fp16 copies, the same step counter, scale and growth tracker -- also when a gradient
    holds an inf (the step is skipped) and when the growth interval is reached (the scale doubles)."""
    from nerficg_amd import _lib
    from nerficg_amd.amp import GradScaler
    from nerficg_amd.apex_optimizers import FusedAdam
    gen = torch.Generator(device=DEV).manual_seed(0)
    na, nb = 1_000_003, 7168
    mk = lambda n: torch.randn(n, device=DEV, generator=gen)
    pa, pb = mk(na), mk(nb)
    steps = []
    for k in range(4):
        ga, gb = mk(na) * 128, mk(nb) * 128
        if k == 2 and poison == 'a':
            ga[na - 2] = float('inf')
        if k == 2 and poison == 'b':
            gb[17] = float('nan')
        steps.append((ga, gb))
    # reference: the drop-in classes
    ra, rb = torch.nn.Parameter(pa.clone()), torch.nn.Parameter(pb.clone())
    opt = FusedAdam([ra, rb], lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False, capturable=True)
    opt.set_l2_slice(ra, 3000, 1e-3)
    scaler = GradScaler(init_scale=128.0, growth_interval=2)
    for ga, gb in steps:
        ra.grad, rb.grad = ga.clone(), gb.clone()
        scaler._lazy_init_scale_growth_tracker(torch.device(DEV)) if scaler._scale is None else None
        scaler.step(opt); scaler.update()
    # the two-launch form
    lib, p = _lib.load(), _lib.ptr
    ma, va, mb, vb = (torch.zeros_like(t) for t in (pa, pa, pb, pb))
    ha, hb = torch.zeros(na, dtype=torch.float16, device=DEV), torch.zeros(nb, dtype=torch.float16, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV); bc = torch.ones(2, device=DEV)
    scale = torch.full((1,), 128.0, device=DEV); tracker = torch.zeros(1, dtype=torch.int32, device=DEV)
    state, ticket = torch.zeros(4, device=DEV), torch.zeros(17 * 16, dtype=torch.int32, device=DEV)
    for ga, gb in steps:
        _lib.check(lib.nrc_amp_adam_step(p(pa), p(ga), p(ma), p(va), p(ha), na, 1e-3, 3000, p(pb), p(gb), p(mb), p(vb), p(hb), nb, 0.0, 0, 1e-2, None, 0.9, 0.99,
                                         1e-15, 0.0, 0, p(step), p(bc), p(scale), p(tracker), 2.0, 0.5, 2, p(state), p(ticket), None, _lib.stream_of(pa)), 'amp_adam_step')
    assert torch.equal(pa, ra.detach()) and torch.equal(pb, rb.detach())
    assert torch.equal(ma, opt.state[ra]['exp_avg']) and torch.equal(vb, opt.state[rb]['exp_avg_sq'])
    assert torch.equal(ha, pa.half()) and torch.equal(hb, pb.half())
    assert int(step) == opt.effective_step(opt.param_groups[0]) == (4 if poison is None else 3)
    assert float(scale) == float(scaler.get_scale()) and int(tracker) == int(scaler._growth_tracker)
    assert float(scale) == (512.0 if poison is None else 128.0)      # two growths / one growth and one back-off
    assert float(state[0]) == 0 and int(ticket.abs().sum()) == 0


def test_recordings_follow_replaced_state_and_the_order_runs_out_loudly():
    """load_state_dict replaces the moment tensors: the recorded iteration must not keep writing the old ones (advisor finding of round 4 for
    GraphedIteration); an exhausted order raises instead of reading behind the permutation."""
    cam, pool = _pool()
    model, renderer, _ = _train_pair(seed=1)
    n = 1024
    order = torch.arange(5 * n, device=DEV)
    it, opt, _ = _fused(model, renderer, cam, pool, n, 200_000, prefetch=False, graph=True, order=order)
    it(); it()
    sd = opt.state_dict()
    opt.load_state_dict(sd)
    live = opt.state[model.encoding_xyz.params]['exp_avg']
    snapshot = live.clone()
    it()
    assert not torch.equal(live, snapshot)         # the step after the load moved the LIVE moments
    it(); it()
    with pytest.raises(RuntimeError, match='used up'):
        it()
    it.rewind(order.flip(0))
    it()
    assert int(it.cursor) == n
