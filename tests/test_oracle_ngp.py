"""Pins for oracle/ngp_oracle.c (CPU, no GPU needed).

The reference has no tests for its CUDA kernels, so the oracle is pinned against:
  * golden vectors generated from the reference's OWN PyTorch code (tests/golden/make_golden.py): integrate_samples is
    an independent statement of the compositing math of volumerendering.cu;
  * numpy.packbits, Morton round trips / monotonicity, geometric properties of the DDA march;
  * finite differences for the analytic backward kernels.
"""
import numpy as np
import pytest

import oracle
from tests import scenes

RNG = np.random.default_rng(1234)


# ------------------------------------------------------------------------------------------------ compositing
def test_composite_train_fw_matches_reference_integrate_samples(golden_dir):
    """integrate_samples (src/Methods/NeRF/utils.py:112-136) == composite_train_fw when deltas are supplied explicitly
    and no ray saturates (T_threshold = 0 disables the early-out; the oracle's early-out is tested separately)."""
    g = np.load(golden_dir / 'nerf_sampling.npz')
    depth, dirs, dens, cols = g['depth'], g['dirs'], g['dens'], g['cols']
    n, s = depth.shape
    deltas = np.concatenate([depth[:, 1:] - depth[:, :-1], np.full((n, 1), 1e10, np.float32)], -1) * np.linalg.norm(dirs, axis=-1, keepdims=True)
    rays_a = np.stack([np.arange(n), np.arange(n) * s, np.full(n, s)], -1).astype(np.int64)
    total, opacity, dsum, rgb, ws = oracle.composite_train_fw(dens.reshape(-1), cols.reshape(-1, 3), deltas.reshape(-1).astype(np.float32),
                                                              depth.reshape(-1), rays_a, 0.0)
    np.testing.assert_allclose(ws.reshape(n, s), g['weights'], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(opacity, g['alpha'][:, 0], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(rgb, g['rgb_nobg'], rtol=2e-5, atol=1e-6)
    # reference depth = sum(w t)/alpha where T_final < 1 (utils.py:131)
    T_final = 1.0 - g['alpha'][:, 0]
    ref_depth = g['depth_out'][:, 0]
    mask = T_final < 1.0
    np.testing.assert_allclose(dsum[mask] / opacity[mask], ref_depth[mask], rtol=5e-5)
    # the 1e10 final delta drives T to exactly 0 at the last sample (or earlier on the saturating ray): with T_threshold = 0 the
    # early-out then fires there and that sample is composited but not counted (volumerendering.cu:41-44)
    assert total[3] == s and np.all(np.delete(total, [3, 5]) == s - 1) and total[5] < s - 1


def test_composite_train_fw_early_out_semantics():
    """volumerendering.cu:41-44: the saturating sample is composited, not counted; later ws stay 0."""
    sig = np.array([1.0, 50.0, 50.0, 3.0, 2.0], np.float32)
    dl = np.full(5, 0.2, np.float32)
    ts = np.linspace(1, 2, 5).astype(np.float32)
    rgbs = RNG.random((5, 3)).astype(np.float32)
    rays_a = np.array([[0, 0, 5]], np.int64)
    total, opacity, depth, rgb, ws = oracle.composite_train_fw(sig, rgbs, dl, ts, rays_a, 1e-4)
    a = 1 - np.exp(-sig * dl)
    T = np.cumprod(np.concatenate([[1.0], 1 - a]))
    k = int(np.argmax(T[1:] <= 1e-4))  # index of the saturating sample
    assert total[0] == k
    assert np.all(ws[k + 1:] == 0) and ws[k] > 0
    np.testing.assert_allclose(opacity[0], (a * T[:-1])[:k + 1].sum(), rtol=1e-5)


def _composite_loss(sig, rgbs, dl, ts, rays_a, go, gd, gr, gw, thr):
    total, opacity, depth, rgb, ws = oracle.composite_train_fw(sig, rgbs, dl, ts, rays_a, thr)
    return float((go * opacity).sum() + (gd * depth).sum() + (gr * rgb).sum() + (gw * ws).sum())


def test_composite_train_bw_matches_finite_differences():
    rays_a, m = scenes.random_ragged_rays(RNG, 9, 12)
    n = rays_a.shape[0]
    sig = (RNG.random(m) * 4).astype(np.float64)
    rgbs = RNG.random((m, 3)).astype(np.float64)
    dl = (RNG.random(m) * 0.05 + 0.01).astype(np.float32)
    ts = np.sort(RNG.random(m)).astype(np.float32)
    go, gd, gr, gw = RNG.normal(size=n), RNG.normal(size=n), RNG.normal(size=(n, 3)), RNG.normal(size=m)
    f = lambda s, c: _composite_loss(s.astype(np.float32), c.astype(np.float32), dl, ts, rays_a, go, gd, gr, gw, 0.0)
    total, opacity, depth, rgb, ws = oracle.composite_train_fw(sig, rgbs, dl, ts, rays_a, 0.0)
    ds, dr = oracle.composite_train_bw(go, gd, gr, gw, sig, rgbs, ws, dl, ts, rays_a, opacity, depth, rgb, 0.0)
    eps = 1e-2
    for k in RNG.choice(m, size=min(m, 12), replace=False):
        sp, sm = sig.copy(), sig.copy()
        sp[k] += eps; sm[k] -= eps
        fd = (f(sp, rgbs) - f(sm, rgbs)) / (2 * eps)
        assert abs(fd - ds[k]) <= 2e-2 * max(1.0, abs(fd)), (k, fd, ds[k])
        cp, cm = rgbs.copy(), rgbs.copy()
        cp[k, 1] += eps; cm[k, 1] -= eps
        fd = (f(sig, cp) - f(sig, cm)) / (2 * eps)
        assert abs(fd - dr[k, 1]) <= 2e-2 * max(1.0, abs(fd)), (k, fd, dr[k, 1])


@pytest.mark.parametrize('tag', ['open', 'closed'])
def test_composite_train_bw_matches_reference_autograd(golden_dir, tag):
    """composite_train_bw (volumerendering.cu:87-202) against torch.autograd through the reference's own integrate_samples
    (src/Methods/NeRF/utils.py:112-136; tests/golden/make_golden.py::make_composite_bw): same deltas supplied explicitly, T_threshold = 0.
    'open': final delta 0.05, T stays > 0 on every ray -- no early-out on either side.  'closed': the reference's default 1e10, where the last
    sample drives T to exactly 0 (its own gradient d alpha / d sigma = delta * exp(-sigma delta) is 0 on both sides)."""
    g = np.load(golden_dir / 'composite_bw.npz')
    depth, dirs = g['depth'], g['dirs']
    n, s = depth.shape
    dens, cols = g[f'{tag}_dens'], g[f'{tag}_cols']
    final = np.full((n, 1), g[f'{tag}_final_delta'], np.float32)
    deltas = (np.concatenate([depth[:, 1:] - depth[:, :-1], final], -1) * np.linalg.norm(dirs, axis=-1, keepdims=True)).astype(np.float32)
    rays_a = np.stack([np.arange(n), np.arange(n) * s, np.full(n, s)], -1).astype(np.int64)
    flat = lambda a: np.ascontiguousarray(a.reshape(-1, *a.shape[2:]))
    total, opacity, dsum, rgb, ws = oracle.composite_train_fw(flat(dens), flat(cols), flat(deltas), flat(depth), rays_a, 0.0)
    np.testing.assert_allclose(ws.reshape(n, s), g[f'{tag}_weights'], rtol=3e-5, atol=1e-7)
    np.testing.assert_allclose(rgb, g[f'{tag}_rgb'], rtol=3e-5, atol=1e-6)
    ds, dr = oracle.composite_train_bw(g[f'{tag}_go'], g[f'{tag}_gd'], g[f'{tag}_gr'], flat(g[f'{tag}_gw']), flat(dens), flat(cols), ws, flat(deltas),
                                       flat(depth), rays_a, opacity, dsum, rgb, 0.0)
    want_s, want_c = g[f'{tag}_d_dens'], g[f'{tag}_d_cols']
    np.testing.assert_allclose(dr.reshape(n, s, 3), want_c, rtol=1e-4, atol=1e-6 * np.abs(want_c).max())
    # d/d sigma is a suffix sum over the ray (cancellation): tolerance relative to the tensor's scale
    np.testing.assert_allclose(ds.reshape(n, s), want_s, rtol=2e-4, atol=2e-5 * np.abs(want_s).max())
    assert np.abs(want_s).max() > 0.1 and np.abs(want_s[2]).max() > 0   # also on the empty ray (sigma = 0): d alpha / d sigma = delta there


def test_composite_test_fw_chunked_equals_train_fw():
    """Compositing a ray in chunks through composite_test_fw (volumerendering.cu:205-249) reproduces composite_train_fw."""
    n, s_total, chunk = 6, 24, 8
    sig = (RNG.random((n, s_total)) * 30).astype(np.float32)
    sig[2] *= 0.01
    rgbs = RNG.random((n, s_total, 3)).astype(np.float32)
    dl = np.full((n, s_total), 0.02, np.float32)
    ts = np.tile(np.linspace(0.5, 1.5, s_total, dtype=np.float32), (n, 1))
    rays_a = np.stack([np.arange(n), np.arange(n) * s_total, np.full(n, s_total)], -1).astype(np.int64)
    _, o_ref, d_ref, c_ref, _ = oracle.composite_train_fw(sig.reshape(-1), rgbs.reshape(-1, 3), dl.reshape(-1), ts.reshape(-1), rays_a, 1e-4)
    opacity, depth, rgb = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros((n, 3), np.float32)
    alive = np.arange(n, dtype=np.int64)
    for c in range(0, s_total, chunk):
        idx = alive.copy()
        n_eff = np.full(len(idx), chunk, np.int32)
        oracle.composite_test_fw(sig[idx, c:c + chunk], rgbs[idx, c:c + chunk], dl[idx, c:c + chunk], ts[idx, c:c + chunk], alive, 1e-4,
                                 n_eff, opacity, depth, rgb)
        alive = alive[alive >= 0]
    np.testing.assert_allclose(opacity, o_ref, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(depth, d_ref, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(rgb, c_ref, rtol=1e-5, atol=1e-7)


# ------------------------------------------------------------------------------------------------ distortion loss
def test_distortion_loss_matches_quadratic_definition_and_fd():
    """losses.cu follows DVGO-v2's O(N) form of  sum_ij w_i w_j |t_i - t_j| + 1/3 sum_i w_i^2 delta_i."""
    rays_a, m = scenes.random_ragged_rays(RNG, 7, 10)
    ws = RNG.random(m)
    dl = (RNG.random(m) * 0.1).astype(np.float32)
    ts = np.zeros(m, np.float32)
    for _, st, ln in rays_a:
        ts[st:st + ln] = np.sort(RNG.random(ln))
    loss, wi, wti = oracle.distortion_loss_fw(ws, dl, ts, rays_a)
    for r, st, ln in rays_a:
        w, t, d = ws[st:st + ln], ts[st:st + ln].astype(np.float64), dl[st:st + ln]
        ref = (w[:, None] * w[None, :] * np.abs(t[:, None] - t[None, :])).sum() + (w * w * d).sum() / 3
        np.testing.assert_allclose(loss[r], ref, rtol=2e-4, atol=1e-6)
    g = RNG.normal(size=rays_a.shape[0]).astype(np.float32)
    dws = oracle.distortion_loss_bw(g, wi, wti, ws, dl, ts, rays_a)
    eps = 1e-3
    for k in range(0, m, max(1, m // 10)):
        wp, wm = ws.copy(), ws.copy()
        wp[k] += eps; wm[k] -= eps
        fd = ((oracle.distortion_loss_fw(wp, dl, ts, rays_a)[0] * g).sum() - (oracle.distortion_loss_fw(wm, dl, ts, rays_a)[0] * g).sum()) / (2 * eps)
        assert abs(fd - dws[k]) <= 5e-2 * max(0.1, abs(fd)), (k, fd, dws[k])


# ------------------------------------------------------------------------------------------------ bit / index kernels
def test_packbits_matches_numpy_little_endian():
    grid = RNG.normal(size=128 * 64).astype(np.float32)
    thr = 0.1
    np.testing.assert_array_equal(oracle.packbits(grid, thr), np.packbits(grid > thr, bitorder='little'))


def test_morton3d_roundtrip_and_interleave():
    coords = RNG.integers(0, 1024, size=(5000, 3)).astype(np.int32)
    coords[:4] = [[0, 0, 0], [1023, 1023, 1023], [1, 0, 0], [0, 0, 1]]
    idx = oracle.morton3D(coords)
    np.testing.assert_array_equal(oracle.morton3D_invert(idx), coords)
    np.testing.assert_array_equal(idx.astype(np.int64) & 0xFFFFFFFF, scenes.morton3d_np(coords[:, 0], coords[:, 1], coords[:, 2]) & 0xFFFFFFFF)
    assert idx[2] == 1 and idx[3] == 4 and idx[1] == 0x3FFFFFFF


def test_morton_encode_63bit_properties():
    pos = RNG.normal(size=(4096, 3)).astype(np.float32)
    codes = oracle.morton_encode(pos)
    assert codes.min() >= 0
    mn = pos.min(0)
    cube = (pos.max(0) - mn).max()
    q = (np.clip((pos - mn) * np.float32(1.0 / cube), 0, 1) * np.float32(2097151.0)).astype(np.uint64)
    # de-interleave and compare with the quantised coordinates
    def compact(v):
        out = np.zeros_like(v)
        for b in range(21):
            out |= ((v >> np.uint64(3 * b)) & np.uint64(1)) << np.uint64(b)
        return out
    c = codes.astype(np.uint64)
    np.testing.assert_array_equal(compact(c), q[:, 0])
    np.testing.assert_array_equal(compact(c >> np.uint64(1)), q[:, 1])
    np.testing.assert_array_equal(compact(c >> np.uint64(2)), q[:, 2])
    # the point attaining the minimum on every axis would be code 0; the largest axis reaches 2^21-1
    assert q.max() == 2097151


# ------------------------------------------------------------------------------------------------ intersections
def test_ray_aabb_matches_analytic_slab():
    n = 2000
    o = (RNG.normal(size=(n, 3)) * 1.5).astype(np.float32)
    d = RNG.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    cnt, ht, hv = oracle.ray_aabb_intersect(o, d, np.zeros((1, 3), np.float32), np.full((1, 3), 0.5, np.float32), 1)
    inv = 1.0 / d.astype(np.float64)
    t0, t1 = (-0.5 - o) * inv, (0.5 - o) * inv
    tn, tf = np.minimum(t0, t1).max(-1), np.maximum(t0, t1).min(-1)
    hit = (tn <= tf) & (tf > 0)
    assert np.array_equal(cnt == 1, hit) or np.mean((cnt == 1) != hit) < 2e-3
    both = hit & (cnt == 1)
    np.testing.assert_allclose(ht[both, 0, 0], np.maximum(tn[both], 0), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(ht[both, 0, 1], tf[both], rtol=1e-4, atol=1e-5)
    assert np.all(ht[cnt == 0] == -1) and np.all(hv[cnt == 0] == -1)


def test_ray_aabb_multi_voxel_sorted_like_torch_sort():
    o = np.array([[-3.0, 0.01, 0.02]], np.float32)
    d = np.array([[1.0, 0.0, 0.0]], np.float32) + 1e-6
    d /= np.linalg.norm(d)
    centers = np.array([[1.0, 0, 0], [-1.0, 0, 0], [0.0, 0, 0], [0, 5.0, 0]], np.float32)
    half = np.full((4, 3), 0.4, np.float32)
    cnt, ht, hv = oracle.ray_aabb_intersect(o, d, centers, half, 4)
    assert cnt[0] == 3
    # unused slot (-1) first, then near-to-far: voxel 1 (x=-1), 2 (x=0), 0 (x=1)  -- torch::sort ascending on t1
    assert list(hv[0]) == [-1, 1, 2, 0]
    assert np.all(np.diff(ht[0, :, 0]) >= 0)


def test_ray_sphere_matches_quadratic():
    n = 500
    o = (RNG.normal(size=(n, 3)) * 2).astype(np.float32)
    d = RNG.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    cnt, ht, hv = oracle.ray_sphere_intersect(o, d, np.zeros((1, 3), np.float32), np.array([0.8], np.float32), 1)
    b = (o.astype(np.float64) * d).sum(-1)
    c = (o.astype(np.float64) ** 2).sum(-1) - 0.64
    disc = b * b - c
    hit = (disc >= 0) & ((-b + np.sqrt(np.maximum(disc, 0))) > 0)
    assert np.mean((cnt == 1) != hit) < 5e-3
    ok = hit & (cnt == 1)
    np.testing.assert_allclose(ht[ok, 0, 1], (-b + np.sqrt(disc))[ok], rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------------------------ ray marching
def _march_inputs(width=48, height=40, cascades=1, scale=0.5):
    c2w = scenes.orbit_pose(0.7, 0.4, scenes.LEGO_RADIUS)
    o, _, vd = scenes.numpy_rays(width, height, c2w)
    half = np.full((1, 3), scale, np.float32)
    _, ht, _ = oracle.ray_aabb_intersect(o, vd, np.zeros((1, 3), np.float32), half, 1)
    hits = ht[:, 0].copy()
    hits[:, 0] = np.maximum(hits[:, 0], 0.2)  # camera near plane clamp (InstantNGP/Renderer.py:42-43)
    hits[:, 1] = np.minimum(hits[:, 1], 1000.0)
    bitfield = scenes.sphere_bitfield(128, scale, 0.35, cascades)
    return o, vd, hits, bitfield


def test_raymarching_train_geometry_and_layout():
    o, d, hits, bitfield = _march_inputs()
    n = o.shape[0]
    noise = RNG.random(n).astype(np.float32)
    rays_a, xyzs, dirs, deltas, ts, counter = oracle.raymarching_train(o, d, hits, bitfield, 1, 0.5, 0.0, noise, 128, 1024)
    total = int(counter[0])
    assert counter[1] == n and total == rays_a[:, 2].sum() == xyzs.shape[0] > 0
    np.testing.assert_array_equal(rays_a[:, 0], np.arange(n))
    np.testing.assert_array_equal(rays_a[:, 1], np.concatenate([[0], np.cumsum(rays_a[:, 2])[:-1]]))
    ray_of = np.repeat(np.arange(n), rays_a[:, 2])
    # samples lie on their ray, inside an occupied cell (cell centre within the sphere, up to one cell diagonal)
    np.testing.assert_allclose(xyzs, o[ray_of] + ts[:, None] * d[ray_of], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(dirs, d[ray_of])
    assert np.all(np.linalg.norm(xyzs, axis=-1) < 0.35 + np.sqrt(3) / 128)
    np.testing.assert_allclose(deltas, np.float32(np.sqrt(3) / 1024), rtol=1e-6)
    # ts strictly increasing along every ray, first sample jittered by at most one step
    same_ray = ray_of[1:] == ray_of[:-1]
    assert np.all(np.diff(ts)[same_ray] > 0)
    # rays that miss the box (hits = -1 -> clamped t1 > t2) have no samples
    assert np.all(rays_a[hits[:, 1] < hits[:, 0], 2] == 0)
    # a central ray crosses the whole sphere: chord 0.7 / dt samples (+- cell quantisation)
    centre = rays_a[(40 // 2) * 48 + 48 // 2, 2]
    assert abs(centre - 0.7 / (np.sqrt(3) / 1024)) < 12


def test_raymarching_test_chunks_reproduce_train_samples():
    """Marching a ray in chunks with raymarching_test (hits_t advanced in place) yields the same t's as the train march
    with noise 0 (the test kernel's dt quirk is invisible at exp_step_factor = 0)."""
    o, d, hits, bitfield = _march_inputs(32, 24)
    n = o.shape[0]
    rays_a, xyzs, dirs, deltas, ts, counter = oracle.raymarching_train(o, d, hits, bitfield, 1, 0.5, 0.0, np.zeros(n, np.float32), 128, 1024)
    h = hits.copy()
    alive = np.arange(n, dtype=np.int64)
    got = [[] for _ in range(n)]
    for _ in range(40):
        if len(alive) == 0:
            break
        x, dd, dl, t, n_eff = oracle.raymarching_test(o, d, h, alive, bitfield, 1, 0.5, 0.0, 128, 1024, 16)
        for j, r in enumerate(alive):
            got[r].extend(t[j, :n_eff[j]].tolist())
            assert np.all(x[j, n_eff[j]:] == 0) and np.all(dd[j, n_eff[j]:] == 0)
        alive = alive[n_eff == 16]
    for r in range(n):
        st, ln = rays_a[r, 1], rays_a[r, 2]
        assert len(got[r]) == ln, r
        np.testing.assert_array_equal(np.asarray(got[r], np.float32), ts[st:st + ln])


def test_raymarching_train_cascades_and_max_samples():
    o, d, hits, bitfield = _march_inputs(24, 24, cascades=3, scale=2.0)
    n = o.shape[0]
    noise = RNG.random(n).astype(np.float32)
    rays_a, xyzs, dirs, deltas, ts, counter = oracle.raymarching_train(o, d, hits, bitfield, 3, 2.0, 1.0 / 256, noise, 128, 64)
    assert rays_a[:, 2].max() <= 64 and counter[0] > 0
    assert np.all(deltas >= np.float32(np.sqrt(3) / 64) * (1 - 1e-6))


def test_threaded_march_and_compositor_equal_the_single_thread_run():
    """Round 4: oracle_raymarching_train (both passes) and oracle_composite_train_fw run their rays on all host cores for bench.py's cpu_baseline.  A
    ray's arithmetic does not depend on who runs it and the outputs keep the ray-index order (counts -> serial prefix -> second pass), so every
    array must be bit-identical whatever the thread count -- single cascade with constant steps, and three cascades with exponential steps."""
    from tests import scenes
    rng = np.random.default_rng(12)
    o, _, d = scenes.numpy_rays(72, 56, scenes.orbit_pose(0.9, 0.3, scenes.LEGO_RADIUS))
    for cascades, scale, esf, bits in ((1, 0.5, 0.0, scenes.sphere_bitfield(128, 0.5, 0.35, 1)), (3, 2.0, 1 / 256, scenes.layered_bitfield(2.0, 3))):
        _, ht, _ = oracle.ray_aabb_intersect(o, d, np.zeros((1, 3), np.float32), np.full((1, 3), scale, np.float32), 1)
        hits = ht[:, 0].copy()
        hits[:, 0] = np.maximum(hits[:, 0], np.float32(0.2))
        noise = rng.random(len(o)).astype(np.float32)
        runs = []
        for threads in (1, 0, 3):
            before = oracle.set_threads(threads)
            try:
                march = oracle.raymarching_train(o, d, hits, bits, cascades, scale, esf, noise, 128, 1024)
                m = march[1].shape[0]
                sig = (np.random.default_rng(5).random(m) * 4).astype(np.float32)
                rgb = np.random.default_rng(6).random((m, 3)).astype(np.float32)
                comp = oracle.composite_train_fw(sig, rgb, march[3], march[4], march[0], 1e-4)
            finally:
                oracle.set_threads(before)
            runs.append(list(march) + list(comp))
        assert int(runs[0][5][0]) > 50_000
        for other in runs[1:]:
            for a, b in zip(runs[0], other):
                np.testing.assert_array_equal(a, b)


def test_philox_reference_reproduces_the_random123_known_answers():
    """tests/philox_ref.py (the checker of the fused iteration's in-kernel generator) against the known-answer vectors of Random123 for philox4x32, 10 rounds."""
    from tests import philox_ref
    for ctr, key, want in philox_ref.KAT:
        got = philox_ref.philox4x32_10([np.array([c]) for c in ctr], [np.array([k]) for k in key])
        assert tuple(int(g[0]) for g in got) == want
    u = philox_ref.u01(np.array([0, 0xff, 0x100, 0xffffffff], dtype=np.uint32))
    assert u[0] == 0.0 and u[1] == 0.0 and u[2] == np.float32(2.0 ** -24) and u[3] < 1.0 and u[3] == np.float32(1.0 - 2.0 ** -24)
