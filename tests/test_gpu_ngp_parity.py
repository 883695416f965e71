"""GPU parity: HIP kernels (through the C ABI, via nerficg_amd.VolumeRenderingV2 / MortonEncoding / raygen) against the
CPU oracle on identical seeded inputs.

Bars (north_star): bit-exact for integer / index outputs (ray & sample indices, counts, Morton codes, bitfields, the
sampled t's that define which cell a sample falls in); f32 radiometric outputs within the tolerance stated per test
(tree-ordered wave reductions and v_exp_f32 vs the oracle's serial sums and expf).
"""
import numpy as np
import pytest
import torch

import oracle
from tests import scenes

pytestmark = pytest.mark.gpu

DEV = 'cuda'
RTOL, ATOL = 2e-5, 2e-6  # f32 compositing tolerance (relative to O(1) radiance values)


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


@pytest.fixture(scope='module')
def vr():
    import nerficg_amd.VolumeRenderingV2 as m
    return m


def _rng(seed):
    return np.random.default_rng(seed)


# ------------------------------------------------------------------------------------------------ bit / index kernels
@pytest.mark.parametrize('n', [0, 1, 63, 4097])
def test_morton3d_bit_exact(vr, n):
    rng = _rng(n)
    coords = rng.integers(0, 1024, size=(n, 3)).astype(np.int32)
    idx = vr.morton3D(T(coords))
    np.testing.assert_array_equal(idx.cpu().numpy(), oracle.morton3D(coords))
    back = vr.morton3D_invert(idx)
    np.testing.assert_array_equal(back.cpu().numpy(), coords)
    np.testing.assert_array_equal(back.cpu().numpy(), oracle.morton3D_invert(oracle.morton3D(coords)))


def test_morton3d_full_grid_roundtrip(vr):
    g = torch.arange(128, dtype=torch.int32, device=DEV)
    coords = torch.stack(torch.meshgrid(g, g, g, indexing='xy'), -1).reshape(-1, 3).contiguous()
    idx = vr.morton3D(coords)
    assert torch.equal(torch.sort(idx.long())[0], torch.arange(128 ** 3, device=DEV))
    assert torch.equal(vr.morton3D_invert(idx), coords)


@pytest.mark.parametrize('n_bytes', [1, 3, 64, 1001, 128 ** 3 // 8])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_packbits_bit_exact(vr, n_bytes, dtype):
    rng = _rng(n_bytes)
    grid = rng.normal(size=n_bytes * 8).astype(np.float32)
    grid[rng.random(grid.size) < 0.1] = -1.0  # carved cells (negative densities)
    g = T(grid, dtype)
    out = torch.zeros(n_bytes, dtype=torch.uint8, device=DEV)
    vr.packbits(g, 0.01, out)
    np.testing.assert_array_equal(out.cpu().numpy(), oracle.packbits(g.float().cpu().numpy(), 0.01))
    # unaligned views take the scalar path
    if n_bytes > 8:
        out2 = torch.zeros(n_bytes + 1, dtype=torch.uint8, device=DEV)[1:]
        vr.packbits(g, 0.01, out2)
        assert torch.equal(out2, out)


def test_morton_encode_bit_exact():
    from nerficg_amd.MortonEncoding import morton_encode
    for n in (1, 1000, 300_001):
        pos = (_rng(n).normal(size=(n, 3)) * 3).astype(np.float32)
        codes = morton_encode(T(pos))
        assert codes.dtype == torch.int64
        np.testing.assert_array_equal(codes.cpu().numpy(), oracle.morton_encode(pos))
    with pytest.raises(RuntimeError):
        morton_encode(T(pos).double())
    with pytest.raises(RuntimeError):
        morton_encode(torch.zeros(4, 3))


# ------------------------------------------------------------------------------------------------ ray generation
@pytest.mark.parametrize('tag', ['64', '800', '53x31'])
def test_generate_rays_vs_reference_golden(golden_dir, tag):
    """Golden vectors come from the reference's View.get_rays on CPU; torch's CPU linspace/matmul differ from the device
    formulas in the last ulp, hence 2 ulp-level tolerances (rtol 3e-7 on O(1) values)."""
    from nerficg_amd.raygen import generate_rays
    g = np.load(golden_dir / 'raygen.npz')
    w, h, fx, fy, cx, cy = g[f'{tag}_intr']
    out = generate_rays(int(w), int(h), fx, fy, cx, cy, g[f'{tag}_c2w'])
    idx = torch.from_numpy(g[f'{tag}_idx']).to(DEV)
    np.testing.assert_array_equal(out['origin'][idx].cpu().numpy(), g[f'{tag}_origin'])
    np.testing.assert_allclose(out['direction'][idx].cpu().numpy(), g[f'{tag}_direction'], rtol=0, atol=4e-7)
    np.testing.assert_allclose(out['view_direction'][idx].cpu().numpy(), g[f'{tag}_view_direction'], rtol=0, atol=4e-7)


# ------------------------------------------------------------------------------------------------ intersections
def test_ray_aabb_bit_exact(vr):
    rng = _rng(5)
    n = 20_000
    o = (rng.normal(size=(n, 3)) * 1.5).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d[:7] = [[1, 0, 0], [0, -1, 0], [0, 0, 1], [-1, 0, 0], [0, 1, 0], [0, 0, -1], [0.70710677, 0.70710677, 0]]  # axis-parallel: inf / NaN slabs
    for centers, half, mh in ((np.zeros((1, 3)), np.full((1, 3), 0.5), 1), (rng.normal(size=(6, 3)), rng.random((6, 3)) * 0.6 + 0.05, 4),
                              (rng.normal(size=(3, 3)), rng.random((3, 3)) * 0.6 + 0.05, 5)):
        centers, half = centers.astype(np.float32), half.astype(np.float32)
        cnt, ht, hv = vr.ray_aabb_intersect(T(o), T(d), T(centers), T(half), mh)
        rc, rt, rv = oracle.ray_aabb_intersect(o, d, centers, half, mh)
        np.testing.assert_array_equal(cnt.cpu().numpy(), rc)
        np.testing.assert_array_equal(ht.cpu().numpy(), rt)
        np.testing.assert_array_equal(hv.cpu().numpy(), rv)
    with pytest.raises(RuntimeError):  # non-contiguous input -> RuntimeError, like CHECK_INPUT (csrc/include/utils.h:4-6)
        vr.ray_aabb_intersect(T(np.ascontiguousarray(o.T)).t(), T(d), T(centers), T(half), 1)


def test_ray_sphere_bit_exact(vr):
    rng = _rng(6)
    n = 10_000
    o = (rng.normal(size=(n, 3)) * 2).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    centers = rng.normal(size=(5, 3)).astype(np.float32)
    radii = (rng.random(5) * 0.8 + 0.1).astype(np.float32)
    cnt, ht, hv = vr.ray_sphere_intersect(T(o), T(d), T(centers), T(radii), 3)
    rc, rt, rv = oracle.ray_sphere_intersect(o, d, centers, radii, 3)
    np.testing.assert_array_equal(cnt.cpu().numpy(), rc)
    np.testing.assert_array_equal(hv.cpu().numpy(), rv)
    # sqrt/div are correctly rounded on both sides; dot products are contraction-free on both sides
    np.testing.assert_array_equal(ht.cpu().numpy(), rt)


# ------------------------------------------------------------------------------------------------ ray marching
def _march_inputs(width, height, cascades=1, scale=0.5, radius=0.35, pose=(0.7, 0.4)):
    c2w = scenes.orbit_pose(pose[0], pose[1], scenes.LEGO_RADIUS * (scale / 0.5))
    o, _, vd = scenes.numpy_rays(width, height, c2w)
    _, ht, _ = oracle.ray_aabb_intersect(o, vd, np.zeros((1, 3), np.float32), np.full((1, 3), scale, np.float32), 1)
    hits = ht[:, 0].copy()
    hits[:, 0] = np.maximum(hits[:, 0], 0.2)
    hits[:, 1] = np.minimum(hits[:, 1], 1000.0)
    return o, vd, hits, scenes.sphere_bitfield(128, scale, radius, cascades)


@pytest.mark.parametrize('cfg', [
    dict(w=96, h=64, cascades=1, scale=0.5, esf=0.0, max_samples=1024),
    dict(w=61, h=47, cascades=3, scale=2.0, esf=1.0 / 256, max_samples=1024),
    dict(w=40, h=40, cascades=1, scale=0.5, esf=0.0, max_samples=37),   # max_samples cap reached
    dict(w=1, h=1, cascades=1, scale=0.5, esf=0.0, max_samples=1024),
    dict(w=256, h=160, cascades=1, scale=0.5, esf=0.0, max_samples=1024),  # > 32768 rays: thread-per-ray kernels (below: wave-per-ray)
    dict(w=97, h=33, cascades=2, scale=1.0, esf=1.0 / 256, max_samples=64),
])
def test_raymarching_train_bit_exact(vr, cfg):
    o, d, hits, bitfield = _march_inputs(cfg['w'], cfg['h'], cfg['cascades'], cfg['scale'], radius=0.35 * cfg['scale'] / 0.5)
    n = o.shape[0]
    noise = _rng(n).random(n).astype(np.float32)
    got = vr.raymarching_train(T(o), T(d), T(hits), T(bitfield), cfg['cascades'], cfg['scale'], cfg['esf'], T(noise), 128, cfg['max_samples'])
    ref = oracle.raymarching_train(o, d, hits, bitfield, cfg['cascades'], cfg['scale'], cfg['esf'], noise, 128, cfg['max_samples'])
    assert int(ref[5][0]) > 0 or n == 1
    for name, a, b in zip(('rays_a', 'xyzs', 'dirs', 'deltas', 'ts', 'counter'), got, ref):
        np.testing.assert_array_equal(a.cpu().numpy(), b, err_msg=name)  # positions/ts bit-exact: same f32 op sequence, no FMA


def test_raymarching_train_empty_and_errors(vr):
    o, d, hits, bitfield = _march_inputs(8, 8)
    z = lambda *s: torch.zeros(*s, device=DEV)
    out = vr.raymarching_train(z(0, 3), z(0, 3), z(0, 2), T(bitfield), 1, 0.5, 0.0, z(0), 128, 1024)
    assert out[0].shape == (0, 3) and out[1].shape == (0, 3) and out[5].tolist() == [0, 0]
    # all rays miss -> zero samples
    miss = np.full_like(hits, -1.0)
    miss[:, 0] = 0.2
    out = vr.raymarching_train(T(o), T(d), T(miss), T(bitfield), 1, 0.5, 0.0, z(64), 128, 1024)
    assert out[5][0].item() == 0 and out[1].shape[0] == 0 and torch.all(out[0][:, 2] == 0)
    with pytest.raises(RuntimeError):
        vr.raymarching_train(T(o).cpu(), T(d), T(hits), T(bitfield), 1, 0.5, 0.0, z(64), 128, 1024)


@pytest.mark.parametrize('n_samples', [1, 2, 5, 16, 64])
def test_raymarching_test_bit_exact_and_inplace(vr, n_samples):
    o, d, hits, bitfield = _march_inputs(80, 60)
    n = o.shape[0]
    rng = _rng(n_samples)
    alive = np.sort(rng.choice(n, size=n // 2, replace=False)).astype(np.int64)
    h_gpu = T(hits)
    h_ref = hits.copy()
    for _ in range(3):  # consecutive marches continue from the advanced hits_t
        got = vr.raymarching_test(T(o), T(d), h_gpu, T(alive), T(bitfield), 1, 0.5, 0.0, 128, 1024, n_samples)
        ref = oracle.raymarching_test(o, d, h_ref, alive, bitfield, 1, 0.5, 0.0, 128, 1024, n_samples)
        for name, a, b in zip(('xyzs', 'dirs', 'deltas', 'ts', 'n_eff'), got, ref):
            np.testing.assert_array_equal(a.cpu().numpy(), b, err_msg=name)
        np.testing.assert_array_equal(h_gpu.cpu().numpy(), h_ref)


# ------------------------------------------------------------------------------------------------ compositing
def _composite_inputs(seed, n_rays, max_len, sat_frac=0.3):
    rng = _rng(seed)
    rays_a, m = scenes.random_ragged_rays(rng, n_rays, max_len)
    sig = (rng.random(m) * 4).astype(np.float32)
    for r, st, ln in rays_a:
        if rng.random() < sat_frac:
            sig[st:st + ln] *= 60  # rays that saturate (early-out path)
    rgbs = rng.random((m, 3)).astype(np.float32)
    dl = (rng.random(m) * 0.01 + 0.001).astype(np.float32)
    ts = np.zeros(m, np.float32)
    for _, st, ln in rays_a:
        ts[st:st + ln] = np.sort(rng.random(ln) * 4 + 2)
    # shuffle the order of rays_a rows (the reference's rows are in atomic-arrival order)
    perm = rng.permutation(n_rays)
    return rays_a[perm].copy(), m, sig, rgbs, dl, ts, rng


def _assert_total_samples(total_gpu, total_ref, sig, dl, rays_a, thr):
    """integer output decided by an f32 threshold test: must agree except where the oracle's T at its break sample is within
    1e-5 relative of the threshold (exp implementation / product association noise)."""
    bad = np.nonzero(total_gpu != total_ref)[0]
    for r in bad:
        row = rays_a[rays_a[:, 0] == r][0]
        st, ln = row[1], row[2]
        Tcum = np.cumprod(np.exp(-(sig[st:st + ln].astype(np.float64) * dl[st:st + ln])))
        k = min(total_gpu[r], total_ref[r])
        assert abs(int(total_gpu[r]) - int(total_ref[r])) == 1 and abs(Tcum[k] - thr) <= 1e-5 * thr * 10, (r, total_gpu[r], total_ref[r], Tcum[k])


@pytest.mark.parametrize('n_rays,max_len', [(1, 5), (257, 40), (300, 300), (33, 1024)])
def test_composite_train_fw_parity(vr, n_rays, max_len):
    rays_a, m, sig, rgbs, dl, ts, _ = _composite_inputs(n_rays, n_rays, max_len)
    got = [x.cpu().numpy() for x in vr.composite_train_fw(T(sig), T(rgbs), T(dl), T(ts), T(rays_a), 1e-4)]
    ref = oracle.composite_train_fw(sig, rgbs, dl, ts, rays_a, 1e-4)
    _assert_total_samples(got[0], ref[0], sig, dl, rays_a, 1e-4)
    for name, a, b in zip(('opacity', 'depth', 'rgb'), got[1:4], ref[1:4]):
        np.testing.assert_allclose(a, b, rtol=RTOL, atol=ATOL * 8, err_msg=name)
    # ws: same support (zeros beyond the early-out) except on threshold ties, values within tolerance
    same = got[0] == ref[0]
    ray_of = np.zeros(m, np.int64)
    for r, st, ln in rays_a:
        ray_of[st:st + ln] = r
    keep = same[ray_of]
    np.testing.assert_allclose(got[4][keep], ref[4][keep], rtol=RTOL, atol=ATOL)
    assert np.array_equal(got[4][keep] == 0, ref[4][keep] == 0)


def test_composite_train_fw_bw_against_the_reference_autograd_golden(vr, golden_dir):
    """The HIP compositing kernels against gradients the REFERENCE computed: torch.autograd through its own integrate_samples
    (src/Methods/NeRF/utils.py:112-136), stored by tests/golden/make_golden.py::make_composite_bw ('open' case: final delta 0.05, T > 0 on every
    ray, T_threshold = 0 here -- no early-out on either side)."""
    g = np.load(golden_dir / 'composite_bw.npz')
    depth, dirs = g['depth'], g['dirs']
    n, s = depth.shape
    dens, cols = g['open_dens'], g['open_cols']
    deltas = (np.concatenate([depth[:, 1:] - depth[:, :-1], np.full((n, 1), g['open_final_delta'], np.float32)], -1)
              * np.linalg.norm(dirs, axis=-1, keepdims=True)).astype(np.float32)
    rays_a = np.stack([np.arange(n), np.arange(n) * s, np.full(n, s)], -1).astype(np.int64)
    flat = lambda a: T(np.ascontiguousarray(a.reshape(-1, *a.shape[2:])))
    total, opacity, dsum, rgb, ws = vr.composite_train_fw(flat(dens), flat(cols), flat(deltas), flat(depth), T(rays_a), 0.0)
    np.testing.assert_allclose(ws.cpu().numpy().reshape(n, s), g['open_weights'], rtol=3e-5, atol=1e-7)
    np.testing.assert_allclose(rgb.cpu().numpy(), g['open_rgb'], rtol=3e-5, atol=2e-6)
    np.testing.assert_allclose(opacity.cpu().numpy(), g['open_alpha'][:, 0], rtol=3e-5, atol=2e-6)
    ds, dr = vr.composite_train_bw(T(g['open_go']), T(g['open_gd']), T(g['open_gr']), flat(g['open_gw']), flat(dens), flat(cols), ws, flat(deltas),
                                   flat(depth), T(rays_a), opacity, dsum, rgb, 0.0)
    want_s, want_c = g['open_d_dens'], g['open_d_cols']
    np.testing.assert_allclose(dr.cpu().numpy().reshape(n, s, 3), want_c, rtol=1e-4, atol=2e-6 * np.abs(want_c).max())
    np.testing.assert_allclose(ds.cpu().numpy().reshape(n, s), want_s, rtol=3e-4, atol=5e-5 * np.abs(want_s).max())   # suffix sums: relative to the scale


@pytest.mark.parametrize('n_rays,max_len', [(64, 30), (200, 300)])
def test_composite_train_bw_parity(vr, n_rays, max_len):
    rays_a, m, sig, rgbs, dl, ts, rng = _composite_inputs(100 + n_rays, n_rays, max_len)
    total, opacity, depth, rgb, ws = oracle.composite_train_fw(sig, rgbs, dl, ts, rays_a, 1e-4)
    go, gd = rng.normal(size=n_rays).astype(np.float32), rng.normal(size=n_rays).astype(np.float32)
    gr, gw = rng.normal(size=(n_rays, 3)).astype(np.float32), rng.normal(size=m).astype(np.float32)
    ds, dr = vr.composite_train_bw(T(go), T(gd), T(gr), T(gw), T(sig), T(rgbs), T(ws), T(dl), T(ts), T(rays_a), T(opacity), T(depth), T(rgb), 1e-4)
    rs, rr = oracle.composite_train_bw(go, gd, gr, gw, sig, rgbs, ws, dl, ts, rays_a, opacity, depth, rgb, 1e-4)
    # gradients carry cancellation ((R - r) differences of O(1) sums): tolerance relative to the per-tensor scale
    for a, b in ((ds.cpu().numpy(), rs), (dr.cpu().numpy(), rr)):
        scale = np.abs(b).max()
        mism = np.abs(a - b) > 5e-5 * scale + 1e-4 * np.abs(b)
        # threshold ties move the support by one sample on isolated rays
        assert mism.mean() < 2e-3, mism.mean()


def test_volume_renderer_autograd_matches_oracle(vr):
    rays_a, m, sig, rgbs, dl, ts, rng = _composite_inputs(7, 50, 60, sat_frac=0.0)
    s = T(sig).requires_grad_(True)
    c = T(rgbs).requires_grad_(True)
    total, opacity, depth, rgb, ws = vr.VolumeRenderer.apply(s, c, T(dl), T(ts), T(rays_a), 1e-4)
    go, gd = rng.normal(size=50).astype(np.float32), rng.normal(size=50).astype(np.float32)
    gr, gw = rng.normal(size=(50, 3)).astype(np.float32), rng.normal(size=m).astype(np.float32)
    loss = (opacity * T(go)).sum() + (depth * T(gd)).sum() + (rgb * T(gr)).sum() + (ws * T(gw)).sum()
    loss.backward()
    f = oracle.composite_train_fw(sig, rgbs, dl, ts, rays_a, 1e-4)
    rs, rr = oracle.composite_train_bw(go, gd, gr, gw, sig, rgbs, f[4], dl, ts, rays_a, f[1], f[2], f[3], 1e-4)
    np.testing.assert_allclose(s.grad.cpu().numpy(), rs, rtol=2e-4, atol=5e-5 * np.abs(rs).max())
    np.testing.assert_allclose(c.grad.cpu().numpy(), rr, rtol=2e-4, atol=5e-5 * np.abs(rr).max())
    assert int(total.item()) == int(f[0].sum())


@pytest.mark.parametrize('n_samples', [1, 2, 3, 8, 13, 64])
def test_composite_test_fw_parity_and_inplace(vr, n_samples):
    rng = _rng(n_samples)
    n_total, a = 500, 321
    alive = np.sort(rng.choice(n_total, size=a, replace=False)).astype(np.int64)
    n_eff = rng.integers(0, n_samples + 1, size=a).astype(np.int32)
    sig = (rng.random((a, n_samples)) * 400).astype(np.float32)
    sig[rng.random(a) < 0.5] *= 0.01
    rgbs = rng.random((a, n_samples, 3)).astype(np.float32)
    dl = np.full((a, n_samples), 0.005, np.float32)
    ts = np.sort(rng.random((a, n_samples)) * 3 + 1, -1).astype(np.float32)
    op0 = (rng.random(n_total) * 0.9).astype(np.float32)
    dp0, c0 = rng.random(n_total).astype(np.float32), rng.random((n_total, 3)).astype(np.float32)
    g_alive, g_op, g_dp, g_c = T(alive), T(op0), T(dp0), T(c0)
    vr.composite_test_fw(T(sig), T(rgbs), T(dl), T(ts), torch.zeros(n_total, 2, device=DEV), g_alive, 1e-4, T(n_eff), g_op, g_dp, g_c)
    r_alive, r_op, r_dp, r_c = alive.copy(), op0.copy(), dp0.copy(), c0.copy()
    oracle.composite_test_fw(sig, rgbs, dl, ts, r_alive, 1e-4, n_eff, r_op, r_dp, r_c)
    np.testing.assert_allclose(g_op.cpu().numpy(), r_op, rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(g_dp.cpu().numpy(), r_dp, rtol=RTOL, atol=ATOL * 4)
    np.testing.assert_allclose(g_c.cpu().numpy(), r_c, rtol=RTOL, atol=ATOL)
    diff = g_alive.cpu().numpy() != r_alive
    assert diff.mean() < 5e-3  # alive flags flip only on threshold ties
    assert np.array_equal(g_alive.cpu().numpy()[n_eff == 0], np.full((n_eff == 0).sum(), -1))


# ------------------------------------------------------------------------------------------------ distortion loss
def test_distortion_loss_parity_and_autograd(vr):
    rng = _rng(11)
    rays_a, m = scenes.random_ragged_rays(rng, 150, 200)
    ws = rng.random(m).astype(np.float32) * 0.05
    dl = (rng.random(m) * 0.01).astype(np.float32)
    ts = np.zeros(m, np.float32)
    for _, st, ln in rays_a:
        ts[st:st + ln] = np.sort(rng.random(ln))
    w = T(ws).requires_grad_(True)
    loss = vr.DistortionLoss.apply(w, T(dl), T(ts), T(rays_a))
    r_loss, r_wi, r_wti = oracle.distortion_loss_fw(ws, dl, ts, rays_a)
    np.testing.assert_allclose(loss.detach().cpu().numpy(), r_loss, rtol=5e-4, atol=1e-6)
    g = rng.normal(size=150).astype(np.float32)
    (loss * T(g)).sum().backward()
    r_g = oracle.distortion_loss_bw(g, r_wi, r_wti, ws, dl, ts, rays_a)
    np.testing.assert_allclose(w.grad.cpu().numpy(), r_g, rtol=2e-3, atol=2e-5 * np.abs(r_g).max())


def test_trunc_exp(vr):
    x = torch.tensor([-20.0, -1.0, 0.0, 3.0, 20.0], device=DEV, requires_grad=True)
    y = vr.TruncExp.apply(x)
    y.sum().backward()
    assert torch.allclose(y, torch.exp(x.detach()))
    assert torch.allclose(x.grad, torch.exp(x.detach().clamp(-15, 15)))
