#!/usr/bin/env python3
"""bench.py -- headline benchmark: InstantNGP image rendering throughput (Mrays/s) on the lego-shaped synthetic workload.

A "step" = one pass of the hot path (ray generation -> box test -> DDA march -> hash-grid encode -> tiny MLPs ->
alpha compositing -> finalisation) over ONE 800x800 image (640 000 rays) per rank, through the C-ABI HIP library
(InstantNGPRenderer.render_image_fused).  Inputs (occupancy bitfield, fp16 parameter copies) are resident in HBM
before the timed region.  With N ranks every rank renders its own pose of the seeded orbit each step (weak scaling:
rays are independent units, no data-path collective -- SURVEY.md 8e); the barrier + max-over-ranks timing follows the
driver contract.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (metric/value/... + "roofline" for the dominant kernel + "cpu_baseline").
"""
from __future__ import annotations

import argparse
import gc
import json
import math
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

W = H = 800
N_POSES = 100
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CHUNK_ROWS = 131072  # rows of 64 slots per encode/MLP launch = the library's NRC_QUERY_CHUNK (8 Mi slots)
# dominant kernel = k_grid_encode.  Algorithmic bytes per LIVE sample, SURVEY 8(d): 16 levels x 8 corners x 2 features x 2 B of table reads
# = 512 B.  (The kernel also reads a 4 B sample record and writes 64 B of encoded features -- artefacts of the two-kernel split, not part of the
# algorithm: they are reported as `kernel_bytes_per_sample`, not used for `achieved`.)
ENC_BYTES_PER_SAMPLE = 512
ENC_KERNEL_BYTES_PER_SAMPLE = 512 + 4 + 64
MLP_FLOP_PER_SAMPLE = 20480   # 10 240 MAC per sample (layer widths padded to 16)
MFMA_PEAK_TFLOPS = 2500.0      # dense fp16 MFMA peak, MI355X_MICROARCH.md
DOMINANT_KERNEL = 'k_grid_encode<SRC_TILED>'


def build_scene(device):
    import torch
    from nerficg_amd.instant_ngp import Camera, InstantNGPModel, InstantNGPRenderer
    from tests import scenes
    model = InstantNGPModel(RANDOM_SEED=0, device=device)  # tcnn-style init: Xavier MLPs, table U(-1e-4, 1e-4), seed 0
    with torch.no_grad():
        model.occupancy_bitfield.copy_(torch.from_numpy(scenes.sphere_bitfield(128, 0.5, 0.35, 1)).to(device))  # solid |x| < 0.35
    renderer = InstantNGPRenderer(model)
    fx, fy, cx, cy = scenes.lego_intrinsics(W, H)
    cam = Camera(width=W, height=H, focal_x=fx, focal_y=fy, center_x=cx, center_y=cy, near_plane=0.2, far_plane=1000.0,
                 background_color=torch.ones(3))
    rng = np.random.default_rng(0)
    poses = [scenes.orbit_pose(float(rng.uniform(0, 2 * math.pi)), float(rng.uniform(-0.5, 0.9)), scenes.LEGO_RADIUS) for _ in range(N_POSES)]
    return model, renderer, cam, poses


C4_W, C4_H = 1600, 1060       # BASELINE configs[3]: InstantNGP on Mip-NeRF360 garden, 1600x1060 rays


def build_c4_scene(device):
    """The garden-SHAPED workload of BASELINE configs[3] (no dataset on the box): SCALE 2 -> three occupancy cascades (InstantNGP/Model.py:46-52),
    EXPONENTIAL_STEPS -> exp_step_factor 1/256 (Renderer.py:44), content in every cascade (ball, two shells: tests/scenes.layered_bitfield), a
    1600x1060 camera INSIDE the box on a seeded orbit between the shells, random-init networks (seed 0)."""
    import torch
    from nerficg_amd.instant_ngp import Camera, InstantNGPModel, InstantNGPRenderer
    from tests import scenes
    model = InstantNGPModel(SCALE=2.0, RANDOM_SEED=0, device=device)
    with torch.no_grad():
        model.occupancy_bitfield.copy_(torch.from_numpy(scenes.layered_bitfield(2.0, model.cascades)).to(device))
    renderer = InstantNGPRenderer(model, EXPONENTIAL_STEPS=True)
    cam = Camera(width=C4_W, height=C4_H, focal_x=0.9 * C4_W, focal_y=0.9 * C4_W, center_x=C4_W / 2, center_y=C4_H / 2, near_plane=0.2, far_plane=1000.0,
                 background_color=torch.ones(3))
    rng = np.random.default_rng(4)
    poses = [scenes.orbit_pose(float(rng.uniform(0, 2 * math.pi)), float(rng.uniform(-0.3, 0.6)), 1.15) for _ in range(N_POSES)]
    return model, renderer, cam, poses


def strong_scaling_frames(renderer, cam, poses, rank, world, steps, warmup, barrier, red_dev):
    """`--scaling strong` / the `config_c4` entry: ONE 1600x1060 frame per step, cut into `world` contiguous tile shards
    (parallel.shard_range + render_image_fused(tile_begin, n_tiles): the protocol of scripts/inference.py:63-97 with the frame as the unit),
    followed by the all-gather of the (N / world, 5) pixel blocks (parallel.gather_image_shards) so that every rank holds the frame.
    Timing contract of the driver: barrier + synchronize on both sides of exactly `steps` frames, max over ranks."""
    import torch
    import torch.distributed as dist
    from nerficg_amd import parallel
    nt = renderer.n_image_tiles(cam)
    b, e = parallel.shard_range(nt, rank, world)
    cache = {}
    ev = []
    samples = 0

    def frame(i, timed):
        nonlocal samples
        out = renderer.render_image_fused(cam, poses[i % N_POSES], tile_begin=b, n_tiles=e - b, return_stats=True)
        if timed:
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
        parallel.gather_image_shards(out, cam.width, cam.height, nt, cache=cache)
        if timed:
            a1.record()
            ev.append((a0, a1))
            samples += out['n_samples']
        return out

    for i in range(warmup):
        frame(i, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        out = frame(warmup + i, True)
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0          # this rank's render + gather, before waiting for the others
    barrier()
    elapsed = time.perf_counter() - t0
    gather_ms = sum(a.elapsed_time(c) for a, c in ev) / max(steps, 1)
    finite = bool(torch.isfinite(out['rgb']).all())
    vals = torch.tensor([elapsed, t_local, gather_ms], device=red_dev, dtype=torch.float64)
    tot = torch.tensor([samples], device=red_dev, dtype=torch.int64)
    if world > 1:
        mx = vals.clone(); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        mn = vals.clone(); dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        dist.all_reduce(tot)
    else:
        mx = mn = vals
    rays = cam.width * cam.height
    gathered = rays * 5 * 4 * (world - 1) // world if world > 1 else 0    # bytes a rank RECEIVES per frame
    return {'elapsed': float(mx[0]), 'ms_per_frame': round(float(mx[0]) / steps * 1e3, 4), 'mrays_per_s': round(rays * steps / float(mx[0]) / 1e6, 3),
            'per_rank_ms_max': round(float(mx[1]) / steps * 1e3, 4), 'per_rank_ms_min': round(float(mn[1]) / steps * 1e3, 4),
            'gather_ms_max': round(float(mx[2]), 4) if world > 1 else None, 'bytes_gathered_per_rank_per_frame': gathered,
            'tiles_per_rank': e - b, 'tiles': nt, 'samples_per_ray': round(int(tot[0]) / (rays * steps), 3), 'samples': int(tot[0]), 'finite': finite,
            'collective': (f'all_gather of padded (rays / {world}, 5) f32 pixel blocks over {dist.get_backend()}' if world > 1 else None)}


def nerf_c1_cpu_baseline(size=64, threads=None):
    """BASELINE configs[0] / SURVEY 8(d) C1: vanilla NeRF (configs/nerf_lego.yaml: 8 x 256 MLP, 64 coarse + 192 fine samples, near 2, far 6,
    white background), a size x size lego-intrinsics image through nerficg_amd.nerf -- the pure-PyTorch statement of src/Methods/NeRF/Renderer.py:132
    -- on the HOST cores (the reference's own CPU mode, GLOBAL.GPU_INDICES: null).  No kernel of this repository runs: it is the number BASELINE.md 3.1
    wants beside every GPU number."""
    import torch
    from nerficg_amd import nerf
    from tests import scenes
    before = torch.get_num_threads()
    import oracle as _oracle
    threads = threads or min(_oracle.usable_cpus(), 32)   # what the cgroup grants (16 on the pool's boxes); torch's CPU GEMMs stop scaling far below 256 threads anyway
    torch.set_num_threads(threads)
    try:
        torch.manual_seed(0)
        coarse, fine = nerf.NeRFBlock(), nerf.NeRFBlock()
        fx, fy, cx, cy = scenes.lego_intrinsics(size, size)
        c2w = np.eye(4); c2w[2, 3] = -4.0
        o, d, vd = (torch.from_numpy(a) for a in scenes.numpy_rays(size, size, c2w, fx, fy, cx, cy))
        t0 = time.perf_counter()
        with torch.no_grad():
            out = nerf.render_rays(coarse, fine, o, d, vd, 2.0, 6.0, torch.ones(3), ray_batch_size=8192, n_samples_coarse_nerf=64, n_samples_nerf=192)
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(before)
    return {'value': round(size * size / dt / 1e3, 4), 'unit': 'Krays/s', 'cores': threads, 'kind': 'port',
            'sample': f'{size}x{size} image, 64 + 192 samples per ray, 8x256 MLPs, nerficg_amd.nerf (pure PyTorch, = src/Methods/NeRF) on {threads} host threads, {dt:.2f} s',
            'finite': bool(torch.isfinite(out['rgb']).all())}


def trained_scene_leg(device, iters=1500, n_poses=20):
    """`secondary_trained`: the headline renders an UNTRAINED model -- 120 samples per ray, nothing ever saturates -- which is SURVEY 8(d)'s
    synthetic input but the opposite regime of what `scripts/inference.py -b` times on a trained scene (InstantNGP/Renderer.py:118-132: rays
    leave the loop when they saturate).  Here a model is trained inside the benchmark -- the analytic shaded sphere of
    tests/test_gpu_convergence.py (closed-form ground-truth views, the Trainer.py:79-94 iteration with occupancy maintenance), `iters`
    iterations -- and 800x800 frames of it are timed through render_image_fused with early_termination='auto' (depth slabs, finished tiles
    skipped) and with the single pass: Mrays/s, samples per ray that were marched, and the share of sample rows the slab order skipped."""
    import torch
    from nerficg_amd.amp import GradScaler
    from nerficg_amd.apex_optimizers import FusedAdam
    from nerficg_amd.instant_ngp import Camera, InstantNGPModel, InstantNGPRenderer
    from nerficg_amd.raygen import generate_rays
    from tests import scenes
    from tests.test_gpu_convergence import analytic_view, orbit, psnr
    tw = th = 100
    fx, fy, cx, cy = scenes.lego_intrinsics(tw, th)
    tcam = Camera(width=tw, height=th, focal_x=fx, focal_y=fy, center_x=cx, center_y=cy, near_plane=0.2, far_plane=1000.0, background_color=torch.ones(3))
    origins, dirs, colours, alphas = [], [], [], []
    for p in orbit(24, 0):
        rays = generate_rays(tw, th, fx, fy, cx, cy, p, device=device, want_direction=False)
        img, hit = analytic_view(tw, th, p, fx, fy, cx, cy, bg=(0.0, 0.0, 0.0))
        origins.append(rays['origin']); dirs.append(rays['view_direction'])
        colours.append(torch.from_numpy(img * hit[..., None]).reshape(-1, 3).to(device)); alphas.append(torch.from_numpy(hit.astype(np.float32)).reshape(-1).to(device))
    origins, dirs, colours, alphas = torch.cat(origins), torch.cat(dirs), torch.cat(colours), torch.cat(alphas)
    model = InstantNGPModel(RANDOM_SEED=0, device=device)
    renderer = InstantNGPRenderer(model)
    opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)
    scaler = GradScaler(init_scale=128.0, growth_interval=10 ** 9)
    perm = torch.randperm(origins.shape[0], generator=torch.Generator().manual_seed(0)).to(device)
    batch = 4096
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(iters):
        if it % 16 == 0:
            renderer.update_occupancy_grid(warmup=it < 256)
        ids = perm[(it * batch) % (perm.numel() - batch):][:batch]
        with torch.amp.autocast('cuda'):
            bg = torch.rand(3, device=device)
            out = renderer.render_rays(origins[ids], dirs[ids], tcam, train_mode=True, custom_bg_color=bg)
            target = colours[ids] + (1 - alphas[ids])[:, None] * bg
            loss = torch.nn.functional.mse_loss(out['rgb'].float(), target) + 0.5e-6 * model.weight_decay_mlp()
        scaler.scale(loss).backward()
        scaler.step(opt); scaler.update(); opt.zero_grad()
    torch.cuda.synchronize(); t_train = time.perf_counter() - t0
    fx8, fy8, cx8, cy8 = scenes.lego_intrinsics(W, H)
    cam = Camera(width=W, height=H, focal_x=fx8, focal_y=fy8, center_x=cx8, center_y=cy8, near_plane=0.2, far_plane=1000.0, background_color=torch.ones(3))
    poses = orbit(n_poses + 3, 5)
    gt, _ = analytic_view(W, H, poses[0], fx8, fy8, cx8, cy8)
    quality = psnr(renderer.render_image_fused(cam, poses[0])['rgb'].cpu().numpy().reshape(H, W, 3), gt)
    res = {'scene': 'analytic shaded sphere (r = 0.3) in front of white, trained in this run', 'training_iterations': iters, 'training_s': round(t_train, 2),
           'psnr_800x800_dB': round(quality, 2)}
    for label, et in (('slab_order_auto', 'auto'), ('single_pass', False)):
        for i in range(3):
            renderer.render_image_fused(cam, poses[i], early_termination=et)
        samples = rows = skipped = 0
        skipped_dev = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n_poses):
            out = renderer.render_image_fused(cam, poses[3 + i], return_stats=True, early_termination=et)
            samples += out['n_samples']; rows += out['n_rows']
            if et == 'auto':   # rows of finished tiles the slab order never queried: a device count, kept on the device until the timed loop is over
                skipped_dev.append(next(iter(renderer._fused_ws.values()))['skipped'].clone())   # (a read here would drain the GPU after every frame)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n_poses
        skipped = sum(int(t.item()) for t in skipped_dev)
        res[label] = {'ms_per_frame': round(dt * 1e3, 3), 'mrays_per_s': round(W * H / dt / 1e6, 2), 'samples_per_ray_marched': round(samples / (n_poses * W * H), 2),
                      'rows_skipped_frac': round(skipped / max(rows, 1), 4) if et == 'auto' else 0.0}
    return res


def time_dominant_kernel(renderer, cam, pose_list, reps=2):
    """Launch durations of the dominant kernel (k_grid_encode: 128 hash-grid gathers per sample) and of its partner (k_ngp_mlp) over ALL
    chunks (launches) of the images of `pose_list` -- the poses the timed region rendered --, measured with HIP events on the launch stream
    (torch's current stream = our launch stream).  Returns a dict: launch-weighted mean ms per launch of both kernels, the per-pose means
    (spread: the hash is only local along x, poses differ by up to 30 %), live samples and slots per launch."""
    import torch
    from nerficg_amd import _lib
    import ctypes
    m = renderer.model
    lib = _lib.load()
    f3 = lambda t: (ctypes.c_float * 3)(*[float(v) for v in t.reshape(-1).tolist()])
    mn, sz = f3(m.xyz_min), f3(m.xyz_size)
    g = m.encoding_xyz.grid_cfg
    nt = renderer.n_image_tiles(cam)
    vp = ctypes.c_void_p
    feat = sh_ws = None
    enc_ms, mlp_ms, launches, live_total, slots_total = [], [], 0, 0, 0
    for pose in pose_list:
        out = renderer.render_image_fused(cam, pose, return_stats=True, early_termination=False)
        ws = next(iter(renderer._fused_ws.values()))
        n_rows = out['n_rows']
        chunks = [(r0, min(CHUNK_ROWS, n_rows - r0)) for r0 in range(0, n_rows, CHUNK_ROWS)]  # the launches of one image
        st = _lib.stream_of(ws['ts'])
        dev = ws['ts'].device
        if feat is None:
            feat = torch.empty(CHUNK_ROWS * 64 * 64 + 256, dtype=torch.uint8, device=dev)
            sh_ws = torch.empty(nt * 2048, dtype=torch.uint8, device=dev)

        arena = renderer.ARENA_IN_PLACE and ws.get('ts_prov') is not None   # the form the frame ran its kernels in (samples read from the count pass's arena)
        ts_buf = ws['ts_prov'] if arena else ws['ts']
        a_off, a_rows = (_lib.ptr(ws['tile_off']), renderer.MAX_SAMPLES) if arena else (None, 0)

        def encode(r0, rows):
            _lib.check(lib.nrc_ngp_encode_samples(
                _lib.ptr(ts_buf), _lib.ptr(ws['row_tile']), _lib.ptr(ws['ray_od']), r0, rows, ctypes.cast(mn, vp),
                ctypes.cast(sz, vp), _lib.ptr(m.encoding_xyz._table16()), g['n_levels'], g['log2_hashmap_size'],
                g['base_resolution'], float(g['per_level_scale']), _lib.ptr(feat), a_off, a_rows, st), 'ngp_encode_samples')

        def mlp(r0, rows, n_ray_tiles=0):  # 0: the per-ray SH coefficients are in sh_ws already (once per image, like the product path)
            _lib.check(lib.nrc_ngp_mlp_samples(
                _lib.ptr(ts_buf), _lib.ptr(ws['row_tile']), _lib.ptr(ws['ray_od']), r0, rows, n_ray_tiles, _lib.ptr(feat),
                _lib.ptr(m.encoding_xyz._half_params()), _lib.ptr(m.color_mlp_with_encoding._half_params()),
                _lib.ptr(ws['packed']), _lib.ptr(sh_ws), a_off, a_rows, st), 'ngp_mlp_samples')

        def timed(fn):  # every chunk of the image, `reps` times, back to back between ONE pair of events on the launch stream
            fn(*chunks[0])
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                for c in chunks:
                    fn(*c)
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / (reps * len(chunks))

        enc_ms.append(timed(encode))
        mlp(*chunks[0], n_ray_tiles=nt)  # SH of this pose's rays; the timed launches below are the MLP kernel alone, as in the product's chunk loop
        mlp_ms.append(timed(mlp))
        launches += len(chunks)
        live_total += int(out['n_samples'])   # slots that hold a sample (the count pass's total; the rest of the rows' slots are holes)
        slots_total += n_rows * 64
    mean = lambda v: sum(v) / len(v)
    return {'enc_ms': mean(enc_ms), 'enc_ms_min': min(enc_ms), 'enc_ms_max': max(enc_ms), 'mlp_ms': mean(mlp_ms), 'mlp_ms_min': min(mlp_ms),
            'mlp_ms_max': max(mlp_ms), 'live_per_launch': live_total // launches, 'slots_per_launch': slots_total // launches, 'poses': len(pose_list),
            'launches_per_image': launches / len(pose_list)}


# ------------------------------------------------------------------------------------------------ 3DGS leg (secondary metric)
GS_W, GS_H = 1297, 840  # gs_garden.yaml: IMAGE_SCALE_FACTOR 0.25 of Mip-NeRF360 garden
C3B_W, C3B_H = 1600, 1060  # the size BASELINE.json names for the garden frames (SURVEY 8d lists both)
N_SIMDS = 256 * 4          # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
VALU_ISSUE_CYCLES_LO, VALU_ISSUE_CYCLES_HI = 2.8, 3.1   # measured cycles per dependent f32 VALU instruction and wave (tools/micro/pk_rate.hip)


def build_gs_scene(device, n=1_000_000, seed=0, w=None, h=None):
    """1 M synthetic Gaussians (positions U([-1.5,1.5]^3) + ground-plane cluster, log-scales N(log 0.01, 0.5^2), unit
    quaternions, opacity logits N(0, 2^2), SH degree 3), 1297x840 camera, black background."""
    import torch
    from nerficg_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from tests import scenes
    sc = scenes.gs_random_scene(n, seed=seed)
    w, h = w or GS_W, h or GS_H
    cam = scenes.gs_camera(w, h, scenes.orbit_pose(0.8, 0.35, 4.5))
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    settings = GaussianRasterizationSettings(
        image_height=h, image_width=w, tanfovx=cam['tanfovx'], tanfovy=cam['tanfovy'], bg=torch.zeros(3, device=device), scale_modifier=1.0,
        viewmatrix=T(cam['viewmatrix']), projmatrix=T(cam['projmatrix']), sh_degree=3, campos=T(cam['campos']), prefiltered=False, debug=False)
    t = {k: T(v) for k, v in sc.items() if k != 'sh_degree'}
    t['opacities'] = t['opacities'][:, None].contiguous()
    return dict(rast=GaussianRasterizer(settings), tensors=t, n=n, scene=sc, cam=cam, w=w, h=h)


def time_gs(gs, reps=5, barrier=None):
    import torch
    if barrier is None:
        barrier = torch.cuda.synchronize
    t = gs['tensors']
    rast = gs['rast']
    n = gs['n']

    def fwd(grad):
        args = {k: (v.detach().requires_grad_(grad)) for k, v in t.items()}
        m2d = torch.zeros_like(args['means3D'], requires_grad=grad)
        color, radii = rast(means3D=args['means3D'], means2D=m2d, opacities=args['opacities'], shs=args['shs'], scales=args['scales'],
                            rotations=args['rotations'])
        return color, radii

    def timed(body, trials=5):
        """Median of `trials` averages over `reps` back-to-back calls (one sync per trial): a single average is thrown off by the odd slow trial
        on a shared box (seen: 1.25 / 1.25 / 1.8 ms for the same frame, and 1.25 / 1.32 / 0.91 on a lease whose host stalled: five trials since); all are reported."""
        body()
        barrier()
        avgs = []
        for _ in range(trials):
            t0 = time.perf_counter()
            for _ in range(reps):
                body()
            barrier()
            avgs.append((time.perf_counter() - t0) / reps)
        return sorted(avgs)[len(avgs) // 2], avgs

    color, radii = fwd(False)
    t_fwd, fwd_trials = timed(lambda: fwd(False))
    g = torch.rand_like(color)

    def fwd_bwd():
        c, _ = fwd(True)
        c.backward(g)

    t_fb, fb_trials = timed(fwd_bwd)
    # per-kernel durations of THIS run: the library records a HIP event on the launch stream behind each of its kernels (stage timer,
    # include/nerficg_hip.h group 12) over `reps` more forward + backward frames, outside the timed loops above
    from nerficg_amd import _lib
    with _lib.stage_timer() as st:
        for _ in range(reps):
            fwd_bwd()
    stage_ms = {k: (tot / reps, cnt // reps) for k, (tot, cnt) in st.by_name().items()}   # name -> (ms per frame, launches per frame)
    color, radii = fwd(True)
    n_inst = color.grad_fn.num_rendered if color.grad_fn is not None else -1
    return {'stage_ms': stage_ms, 'stage_frames': reps, 'msplats_per_s_fwd': round(n / t_fwd / 1e6, 2), 'msplats_per_s_fwd_bwd': round(n / t_fb / 1e6, 2), 'ms_fwd': round(t_fwd * 1e3, 3),
            'ms_fwd_bwd': round(t_fb * 1e3, 3), 'trials_ms_fwd': [round(v * 1e3, 3) for v in fwd_trials],
            'trials_ms_fwd_bwd': [round(v * 1e3, 3) for v in fb_trials], 'gaussians': n, 'visible': int((radii > 0).sum().item()), 'instances': int(n_inst),
            'image': f"{gs.get('w', GS_W)}x{gs.get('h', GS_H)}", 'pixels': gs.get('w', GS_W) * gs.get('h', GS_H)}


def _second_ruler(kernel, in_frame_ms, units_per_launch, peak, ms_to_s):
    """frac of a kernel by its IN-FRAME duration (stage timer of this run) and by its rocprofv3 duration (profiles/kernel_durations.json, sha-gated), next to the
    back-to-back figure the entry's `frac` is computed from.  units_per_launch: GB or TFLOP per average launch."""
    out = {'why': 'kernel_ms above = the kernel back to back with itself between one pair of events; in_frame = behind / in front of its partner kernel as the frame issues them '
                  '(the encoder then starts behind 0.5 GB of feature traffic, the MLP kernel reads features written a moment ago): the pair sums agree, the split moves'}
    if in_frame_ms:
        out['in_frame_ms'] = round(in_frame_ms, 4)
        out['in_frame_frac'] = round(units_per_launch / (in_frame_ms * ms_to_s) / peak, 4)
    ms, src = kernel_durations_entry(kernel, 'ngp_net.hip')
    out['rocprof_ms'] = round(ms, 4) if ms else None
    out['rocprof_frac'] = round(units_per_launch / (ms * ms_to_s) / peak, 4) if ms else None
    out['rocprof_source'] = src
    return out


def time_kernels_in_frame(renderer, cam, pose_list):
    """The second ruler for the pair of kernels (round-5 review, "weak" 5): their durations INSIDE the frame -- encode, MLP, encode, MLP, ... as the product issues
    them -- from the library's stage timer (a HIP event behind every kernel of the frame's query loop), over the same poses as time_dominant_kernel.  That function
    times each kernel back to back with itself: the encoder then finds its table in L2 and the MLP kernel finds its features evicted by the encoder launches of the
    other chunks; in the frame it is the other way round (the MLP kernel reads features the encoder wrote a moment ago, the encoder starts behind 0.5 GB of feature
    traffic).  The SUMS agree; the split does not -- both are reported."""
    import torch
    from nerficg_amd import _lib
    tot = {'k_grid_encode': [0.0, 0], 'k_ngp_mlp': [0.0, 0]}
    for pose in pose_list:
        with _lib.stage_timer() as st:
            renderer.render_image_fused(cam, pose, early_termination=False)
            torch.cuda.synchronize()
        for name, (ms, n) in st.by_name().items():
            if name in tot:
                tot[name][0] += ms; tot[name][1] += n
    return {k: (v[0] / v[1] if v[1] else None) for k, v in tot.items()}


def kernel_durations_entry(kernel_prefix, source_file):
    """(mean ms per launch or None, provenance) of a kernel from profiles/kernel_durations.json -- rocprofv3 --kernel-trace --stats of THIS command line, written by
    tools/make_profile_summary.py next to pmc_summary.json and gated on the same source digests."""
    f = ROOT / 'profiles' / 'kernel_durations.json'
    if not f.exists():
        return None, 'no profiles/kernel_durations.json'
    try:
        d = json.loads(f.read_text())
    except Exception:
        return None, 'profiles/kernel_durations.json unreadable'
    meta = d.get('_meta', {})
    if meta.get('csrc_sha', {}).get(source_file) != csrc_digests().get(source_file):
        return None, f'profiles/kernel_durations.json predates the current {source_file}: not quoted'
    name = next((k for k in sorted(d) if k.startswith(kernel_prefix)), None)
    if name is None:
        return None, f'no {kernel_prefix} in profiles/kernel_durations.json'
    return d[name]['avg_us'] / 1e3, f"profiles/kernel_durations.json ({meta.get('round', '?')}: rocprofv3 --kernel-trace --stats of `{meta.get('command', 'python bench.py')}`, {d[name]['calls']} launches, same {source_file})"


def csrc_digests():
    """sha256[:16] of every translation unit / header of the library: profiles/pmc_summary.json records them at collection time, and a counter
    entry is only quoted for a kernel whose source file has not changed since (otherwise the line would carry numbers of another build)."""
    import hashlib
    return {p.name: hashlib.sha256(p.read_bytes()).hexdigest()[:16] for p in sorted((ROOT / 'nerficg_amd' / 'csrc').glob('*.h*'))}


def pmc_entry(pmc_all, kernel, source_file):
    """(counter entry or {}, provenance string) of `kernel` from profiles/pmc_summary.json, {} when its source changed since the collection."""
    meta = pmc_all.get('_meta', {})
    if kernel not in pmc_all:   # template instances are listed with their arguments (k_preprocess<0>, k_radix_pass<false>): the first one of that kernel
        kernel = next((k for k in sorted(pmc_all) if k.startswith(kernel + '<')), kernel)
    if kernel not in pmc_all:
        return {}, 'no counter entry in profiles/pmc_summary.json'
    if meta.get('csrc_sha', {}).get(source_file) != csrc_digests().get(source_file):
        return {}, f'profiles/pmc_summary.json predates the current {source_file}: not quoted'
    return pmc_all[kernel], f"profiles/pmc_summary.json ({meta.get('round', '?')}, rocprofv3 --pmc, same {source_file})"


def gs_kernel_rooflines(gs_res, pmc_all):
    """Per-kernel entries of the 3DGS leg on SURVEY 8(d)'s per-stage byte model: blend 40 B per instance + 20 B per pixel; backward blend
    76 B per instance + 20 B per pixel; preprocess 308 B and its backward 472 B per visible Gaussian; binning + sort 108 B per instance (the
    reference's key / value traffic -- this build moves less).  Durations are measured IN THIS RUN: HIP events the library records on the
    launch stream behind each of its kernels (stage timer), averaged over gs_res['stage_frames'] forward + backward frames.  `traffic`
    (memory-side bytes from rocprofv3 --pmc) can only come from a profiler run: it is quoted from profiles/pmc_summary.json when the
    kernel's source file is unchanged since that collection, else null."""
    stage = gs_res.get('stage_ms') or {}
    if not stage:
        return None
    P, D, HW = gs_res['visible'], gs_res['instances'], gs_res.get('pixels', GS_W * GS_H)
    model = {'k_render': 40 * D + 20 * HW, 'k_render_bw': 76 * D + 20 * HW, 'k_preprocess': 308 * P, 'k_preprocess_bw': 472 * P}
    out = {}
    for k, nbytes in model.items():
        if k in stage:
            ms = stage[k][0]
            entry, src = pmc_entry(pmc_all, k, 'gs_raster.hip')
            out[k] = {'ms': round(ms, 4), 'algorithmic_bytes': nbytes, 'achieved': round(nbytes / ms / 1e6, 1),
                      'frac': round(nbytes / ms / 1e6 / HBM_PEAK_GBS, 4), 'traffic': entry.get('hbm_bytes_per_launch'), 'traffic_source': src}
            if k in ('k_render', 'k_render_bw'):
                # The blend kernels are NOT byte-bound (counter traffic 0.4-0.6 x the byte model): the ruler that binds is VALU issue.  From the
                # counter collection of the same source (sha-gated like `traffic`): wave-instructions issued per SIMD-cycle, times the cycles one
                # dependent f32 instruction holds its wave's issue slot at this kernel's occupancy (tools/micro/valu_rate.hip / pk_rate.hip: 2.8-3.1).
                insts, cycles = entry.get('SQ_INSTS_VALU'), entry.get('kernel_cycles')
                if insts and cycles:
                    per_simd_clk = insts / (cycles * N_SIMDS)
                    out[k]['roofline_valu'] = {'bound': 'valu issue', 'valu_wave_instructions_per_launch': insts, 'gpu_cycles_per_launch': cycles, 'simds': N_SIMDS,
                                               'wave_instructions_per_simd_cycle': round(per_simd_clk, 4), 'issue_cycles_per_instruction': [VALU_ISSUE_CYCLES_LO, VALU_ISSUE_CYCLES_HI],
                                               'frac': [round(per_simd_clk * VALU_ISSUE_CYCLES_LO, 3), round(per_simd_clk * VALU_ISSUE_CYCLES_HI, 3)],
                                               'source': src + '; issue cycles: tools/micro/pk_rate.hip (round 3)'}
                else:
                    out[k]['roofline_valu'] = {'bound': 'valu issue', 'frac': None, 'source': src}
    binning = [k for k in stage if k.startswith(('k_depth_keys', 'k_radix', 'k_span', 'k_item', 'k_scan_tiles'))]
    if binning:
        t = sum(stage[k][0] for k in binning)          # ms per frame, all launches of the kernel (4 per radix kernel)
        n_launch = sum(stage[k][1] for k in binning)
        out['binning (depth sort + span scatter, %d launches)' % n_launch] = {
            'ms': round(t, 4), 'algorithmic_bytes': 108 * D, 'achieved': round(108 * D / t / 1e6, 1), 'frac': round(108 * D / t / 1e6 / HBM_PEAK_GBS, 4),
            'traffic': None, 'per_kernel_ms': {k: round(stage[k][0], 4) for k in sorted(binning)}}
    out['_all_kernels_ms_per_frame'] = round(sum(v[0] for v in stage.values()), 4)
    out['_timing'] = f"HIP events on the launch stream behind every kernel, {gs_res.get('stage_frames')} forward + backward frames of this run"
    return out


def gs_cpu_baseline(n=1_000_000, w=GS_W, h=GS_H):
    """CPU oracle (kind "port": the reference refuses CPU mode for GaussianSplatting, Renderer.py:32-33) on ONE frame of the workload itself
    (1 M Gaussians, 1297x840: ~10-20 s of CPU work), OpenMP over Gaussians / pixel rows on all host cores (the instance sort is a serial qsort)."""
    import oracle
    from tests import scenes
    sc = scenes.gs_random_scene(n, seed=0)
    cam = scenes.gs_camera(w, h, scenes.orbit_pose(0.8, 0.35, 4.5))
    cores = oracle.usable_cpus()      # affinity capped by the cgroup quota (16 of the 256 logical CPUs on the pool's boxes)
    before = oracle.set_threads(0)
    try:
        t0 = time.perf_counter()
        col, radii, st = oracle.gs_forward(sc['means3D'], sc['opacities'], cam['viewmatrix'], cam['projmatrix'], cam['campos'], cam['tanfovx'], cam['tanfovy'],
                                           w, h, np.zeros(3, np.float32), sh_degree=3, shs=sc['shs'], scales=sc['scales'], rotations=sc['rotations'])
        t_f = time.perf_counter() - t0
        g = np.ones((3, h, w), np.float32)
        t0 = time.perf_counter()
        oracle.gs_backward(st, g, threads=0)
        t_b = time.perf_counter() - t0
    finally:
        oracle.set_threads(before)
    return {'value': round(n / t_f / 1e6, 5), 'value_fwd_bwd': round(n / (t_f + t_b) / 1e6, 5), 'unit': 'Msplats/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} Gaussians, {w}x{h} image, {st.num_rendered} instances, oracle/gs_oracle.c with OpenMP on {cores} threads (the CPUs the cgroup grants of {os.cpu_count()} logical), '
                      f'fwd {t_f:.2f} s + bwd {t_b:.2f} s'}


def cpu_baseline(cam_full, pose, model_params, stride=4):
    """The CPU oracle (kind "port": the reference has no CPU path for InstantNGP) on a bounded sample of THE SAME FRAME: every stride-th pixel of every
    stride-th row of the 800x800 image (40 000 rays spread over the whole picture, so samples per ray equal the frame's ~120 -- rounds 3-4 took a
    central crop, whose rays all cross the object: 397 samples per ray, a Mrays/s figure 3.3 x off), same camera / pose / scene, on the CPUs the
    process may use (oracle.usable_cpus(): affinity capped by the cgroup quota) -- every stage is OpenMP-parallel over rays or samples (march, hash-grid
    encode, both MLPs, compositing), fp16 roundings through F16C.  `scaling`: the encode + MLP leg (98 % of the work) on one thread (a sub-sample) and
    on all of them."""
    import oracle
    from tests import scenes
    pd, pc, bitfield = model_params
    fx, fy, cx, cy = scenes.lego_intrinsics(W, H)
    cores = oracle.usable_cpus()
    before = oracle.set_threads(0)
    grid_kw = dict(n_levels=16, log2_hashmap_size=19, base_resolution=16, per_level_scale=float(math.exp(math.log(2048 * 1.0 / 16) / 15)))
    try:
        o_all, _, d_all = scenes.numpy_rays(W, H, pose, fx, fy, cx, cy)
        pick = (np.arange(0, H, stride)[:, None] * W + np.arange(0, W, stride)[None, :]).reshape(-1)
        o, d = np.ascontiguousarray(o_all[pick]), np.ascontiguousarray(d_all[pick])
        t0 = time.perf_counter()
        _, ht, _ = oracle.ray_aabb_intersect(o, d, np.zeros((1, 3), np.float32), np.full((1, 3), 0.5, np.float32), 1)
        hits = ht[:, 0].copy()
        hits[:, 0] = np.maximum(hits[:, 0], np.float32(0.2))
        rays_a, xyzs, dirs, deltas, ts, counter = oracle.raymarching_train(o, d, hits, bitfield, 1, 0.5, 0.0, np.zeros(len(o), np.float32), 128, 1024)
        t_march = time.perf_counter() - t0
        x01 = (xyzs + np.float32(0.5)) / np.float32(1.0)
        t1 = time.perf_counter()
        sig, rgb, _ = oracle.ngp_query(x01, dirs, pd[:3072], pc, pd[3072:].reshape(-1, 2), **grid_kw)
        t_query = time.perf_counter() - t1
        t2 = time.perf_counter()
        oracle.composite_train_fw(sig, rgb, deltas, ts, rays_a, 1e-4)
        t_comp = time.perf_counter() - t2
        dt = time.perf_counter() - t0
        n_all = int(counter[0])
        sub = slice(0, max(n_all // 16, 1))
        oracle.set_threads(1)
        t3 = time.perf_counter()
        oracle.ngp_query(x01[sub], dirs[sub], pd[:3072], pc, pd[3072:].reshape(-1, 2), **grid_kw)
        t_one = time.perf_counter() - t3
    finally:
        oracle.set_threads(before)
    n_sub = len(x01[sub])
    one, allc = n_sub / t_one / 1e6, n_all / t_query / 1e6
    return {'value': round(len(o) / dt / 1e6, 6), 'unit': 'Mrays/s', 'cores': cores, 'kind': 'port',
            'sample': f'every {stride}th pixel of every {stride}th row of one 800x800 pose ({len(o)} rays, {n_all} samples = {n_all / len(o):.1f} per ray, the frame: ~120), '
                      f'oracle/*.c with OpenMP in every stage on {cores} threads (what the cgroup grants of {os.cpu_count()} logical CPUs; F16C conversions: {oracle.has_f16c()}), '
                      f'{dt:.2f} s = march {t_march:.2f} + encode/MLP {t_query:.2f} + composite {t_comp:.2f}',
            'samples_per_ray': round(n_all / len(o), 1), 'msamples_per_s': round(n_all / dt / 1e6, 4),
            'scaling': {'leg': 'hash-grid encode + both MLPs (oracle.ngp_query)', 'msamples_per_s_1_thread': round(one, 4), 'msamples_per_s_all_threads': round(allc, 4),
                        'threads': cores, 'speedup': round(allc / one, 2), 'speedup_per_core': round(allc / one / cores, 3),
                        'one_thread_sample': f'{n_sub} samples, {t_one:.2f} s'}}


def time_train(model, renderer, cam, poses, n_rays=2200, iters=60):
    """InstantNGP training iteration through the drop-in modules (Trainer.py:79-94 sequence: sample rays -> render_rays training
    path -> MSE + weight decay -> GradScaler(128) backward -> Adam), rank 0 only, reported next to the headline metric."""
    import torch
    from nerficg_amd.raygen import generate_rays
    dev = model.encoding_xyz.params.device
    rays = [generate_rays(cam.width, cam.height, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, p, device=dev, want_direction=False)
            for p in poses[:2]]
    origin = torch.cat([r['origin'] for r in rays]); vdir = torch.cat([r['view_direction'] for r in rays])
    perm = torch.randperm(origin.shape[0], generator=torch.Generator(device='cpu').manual_seed(0)).to(dev)
    saved = [p.detach().clone() for p in model.parameters()]
    from nerficg_amd.apex_optimizers import FusedAdam
    opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)  # Trainer.py:35
    from nerficg_amd.amp import GradScaler   # torch.amp.GradScaler with the inf check as one streaming kernel
    scaler = GradScaler(init_scale=128.0, growth_interval=10 ** 9)
    target = torch.rand(origin.shape[0], 3, device=dev)

    from nerficg_amd.instant_ngp import InstantNGPLoss    # Loss.py:11-26 (MSE + 0.5e-6 x mean squared MLP weight) as one node
    from nerficg_amd.ngp import gather_ray_batch          # RayPoolSampler.get (DatasetSamplers.py:53-66): ray_pool[ids], every field in one launch
    criterion = InstantNGPLoss(model)

    def step(i):
        ids = perm[(i * n_rays) % (perm.numel() - n_rays):][:n_rays]
        batch = gather_ray_batch(ids, origin, vdir, target)
        with torch.amp.autocast('cuda'):
            bg = torch.rand(3, device=dev)
            out = renderer.render_rays(batch['origin'], batch['view_direction'], cam, train_mode=True, custom_bg_color=bg)
            loss = criterion(out, batch, bg)
        scaler.scale(loss).backward()
        scaler.step(opt); scaler.update(); opt.zero_grad()
        return int(out['rm_samples'].item())

    # what earlier legs left behind (a torn-down HIP graph, 6 M-Gaussian scenes) is released HERE, not by a collection inside the timed loop: this leg is ~90 host
    # launches per iteration, and one 75 ms collection in sixty iterations had it at 1.9 ms instead of 0.6-0.7 on one lease (profiles/r06_bench_runs.md)
    gc.collect()
    for i in range(3):
        step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter(); tot = 0
    for i in range(iters):
        tot += step(3 + i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters

    def restore():  # the benchmark must not depend on the order of its legs
        with torch.no_grad():
            for p, q in zip(model.parameters(), saved):
                p.copy_(q)

    # the training kernels on their rooflines, timed in THIS run (stage timer: HIP events behind every kernel of the query forward / backward):
    # SURVEY 8(d) per sample -- hash-grid encode 512 B of table reads; grid backward 512 B read + 512 B read-modify-write = 1024 B;
    # MLP forward 20 480 FLOP, backward 2 x forward = 40 960 FLOP (both networks together)
    # Two regimes.  The cost of the grid backward follows the number of samples whose upstream gradient is not exactly zero: a fresh model
    # has a gradient on every sample, a trained one (here: after ~70 iterations on random targets, rays saturating early) on a fraction
    # of them, and the kernels skip the others.  The rooflines are taken in the DENSE regime (restored parameters, the conservative
    # one: SURVEY's bytes assume every sample contributes); the late per-kernel times are reported beside them.
    from nerficg_amd import _lib
    n_prof = 10
    with _lib.stage_timer() as st_late:
        for i in range(n_prof):
            step(3 + iters + i)
    stage_late = {k: tot_ms / n_prof for k, (tot_ms, _) in st_late.by_name().items()}
    restore()
    opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)   # fresh moments (step() reads this name)
    for i in range(3):
        step(i)
    with _lib.stage_timer() as st:
        m_prof = sum(step(3 + i) for i in range(n_prof)) / n_prof
    stage = {k: tot_ms / n_prof for k, (tot_ms, _) in st.by_name().items()}   # ms per iteration
    roof = {}
    def hbm(name, kernels, bytes_per_sample):
        ms = sum(stage.get(k, 0.0) for k in kernels)
        if ms > 0:
            roof[name] = {'bound': 'hbm', 'kernels': [k for k in kernels if k in stage], 'ms': round(ms, 4), 'algorithmic_bytes_per_sample': bytes_per_sample,
                          'achieved': round(bytes_per_sample * m_prof / ms / 1e6, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                          'frac': round(bytes_per_sample * m_prof / ms / 1e6 / HBM_PEAK_GBS, 4)}
    def mfma(name, kernels, flop_per_sample):
        ms = sum(stage.get(k, 0.0) for k in kernels)
        if ms > 0:
            roof[name] = {'bound': 'mfma', 'kernels': [k for k in kernels if k in stage], 'ms': round(ms, 4), 'flop_per_sample': flop_per_sample,
                          'achieved': round(flop_per_sample * m_prof / ms / 1e9, 1), 'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                          'frac': round(flop_per_sample * m_prof / ms / 1e9 / MFMA_PEAK_TFLOPS, 4)}
    hbm('grid_encode', ['k_grid_encode<train>'], 512)
    hbm('grid_backward', ['k_grid_bwd', 'k_gb_split', 'k_gb_accumulate'], 1024)
    mfma('mlp_forward', ['k_nwie_fwd<density>', 'k_nwie_fwd<colour>'], MLP_FLOP_PER_SAMPLE)
    mfma('mlp_backward', ['k_nwie_bwd<colour>', 'k_nwie_bwd<density>'], 2 * MLP_FLOP_PER_SAMPLE)
    # The limiter-correct ruler for the two training MLP entries.  SURVEY 8(d) prices a15 in FLOP, and at 77 M samples per frame (k_ngp_mlp) that is the
    # bound; a TRAINING forward also WRITES what the backward needs (fp16 inputs and post-ReLU activations of both networks) and the backward reads it
    # back -- per sample, forward: density 64 (features) + 64 (inputs kept) + 128 (hidden kept) + 32 (outputs) and colour 12 + 32 + 64 + 256 + 8 + 16
    # (f32 sigma / rgb) = 676 B; backward: colour 64 + 256 + 16 + 32 and density 64 + 128 + 32 + 128 (pair-major f32 feature gradients) = 720 B --
    # the chip's balance point is 2.5 PFLOP/s / 8 TB/s = 312 FLOP per byte; these kernels do 20 480 / 676 = 30 and 40 960 / 720 = 57: they stream.
    def mlp_bytes(name, bytes_per_sample):
        r = roof.get(name)
        if r:
            r['roofline_hbm'] = {'bound': 'hbm', 'algorithmic_bytes_per_sample': bytes_per_sample, 'achieved': round(bytes_per_sample * m_prof / r['ms'] / 1e6, 1),
                                 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(bytes_per_sample * m_prof / r['ms'] / 1e6 / HBM_PEAK_GBS, 4),
                                 'what': 'inputs + the forward state kept for the backward (fp16, fragment-major) + outputs; the FLOP fraction above is SURVEY 8(d)\'s ruler, this one is what the kernel waits for'}
    mlp_bytes('mlp_forward', 676)
    mlp_bytes('mlp_backward', 720)
    roof['_per_kernel_ms'] = {k: round(v, 4) for k, v in sorted(stage.items())}
    roof['_per_kernel_ms_late'] = {k: round(v, 4) for k, v in sorted(stage_late.items())}
    roof['_timing'] = (f'HIP events on the launch stream behind every kernel of the query forward / backward, {n_prof} op-by-op iterations of this run, '
                       f'{round(m_prof)} samples each; rooflines and _per_kernel_ms: iterations 4-13 from the initial parameters (a gradient on every sample); '
                       f'_per_kernel_ms_late: iterations {4 + iters}-{3 + iters + n_prof} (most samples behind a saturated ray: zero gradient, skipped by the grid backward)')
    restore()
    res = {'metric': 'InstantNGP training iteration (drop-in modules, fwd + bwd + Adam); hip_graph = the same modules recorded; fused = nerficg_amd.ngp_trainer', 'ms_per_iteration': round(dt * 1e3, 3), 'rays': n_rays,
           'samples_per_iteration': int(tot / iters), 'msamples_per_s': round(tot / iters / dt / 1e6, 1), 'roofline': roof}
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():  # N > 1: no recording next to a live RCCL communicator (its watchdog thread polls events)
        res['hip_graph'] = None
        return res
    # the same iteration recorded once in a HIP graph (nerficg_amd.graphs): fixed sample capacity, batch gathered from the resident ray pool
    # inside the recording, step counter / learning rate on the device.  One graph launch + one index copy per iteration.
    from nerficg_amd.graphs import instant_ngp_iteration
    capacity = (int(1.15 * tot / iters) + 4095) // 4096 * 4096
    opt_g = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False, capturable=True)
    scaler_g = GradScaler(init_scale=128.0, growth_interval=10 ** 9)
    graphed = instant_ngp_iteration(model, renderer, opt_g, scaler_g, cam, n_rays, capacity, ray_pool={'origin': origin, 'view_direction': vdir, 'rgb': target},
                                    fold_weight_decay=True)
    batch = lambda i: perm[(i * n_rays) % (perm.numel() - n_rays):][:n_rays]
    for i in range(3):
        graphed(ids=batch(i))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    marched = torch.zeros((), dtype=torch.int64, device=dev); cut = torch.zeros((), dtype=torch.int64, device=dev)
    third = max(iters // 3, 1)
    marks = {}
    for i in range(iters):
        if i in (third, iters - third):
            torch.cuda.synchronize(); marks[i] = time.perf_counter()
        out = graphed(ids=batch(3 + i))
        marched += out['rm_samples']; cut += out['sample_overflow']
    torch.cuda.synchronize(); t1 = time.perf_counter(); dt_g = (t1 - t0) / iters
    dt_first = (marks.get(third, t1) - t0) / third; dt_last = (t1 - marks.get(iters - third, t0)) / third
    restore()
    res['hip_graph'] = {'ms_per_iteration': round(dt_g * 1e3, 3), 'ms_per_iteration_first_third': round(dt_first * 1e3, 3),
                        'ms_per_iteration_last_third': round(dt_last * 1e3, 3), 'weight_decay': 'in the Adam kernel (FusedAdam.set_l2_slice)', 'sample_capacity': capacity, 'samples_per_iteration': int(marched.item() / iters),
                        'samples_cut': int(cut.item()), 'msamples_per_s': round(marched.item() / iters / dt_g / 1e6, 1)}
    # The same iteration as FOUR library calls / 13 launches on device-resident state (nerficg_amd.ngp_trainer, include/nerficg_hip.h group 13), the next
    # batch marched ahead on a side stream; no recording -- the calls only enqueue.  Same regime as the recorded leg: 60 iterations from the restored
    # parameters (a gradient on every sample at the start).
    from nerficg_amd.ngp_trainer import FusedTrainingIteration
    del graphed
    gc.collect(); torch.cuda.synchronize()   # the recording (its closures form cycles) is torn down HERE: collected inside the timed loop below, the release of
                                             # the graph's private pool stalled the device for ~80 ms once (first third 4.1 ms per iteration in one run of five)
    opt_f = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False, capturable=True)
    scaler_f = GradScaler(init_scale=128.0, growth_interval=10 ** 9)
    fused = FusedTrainingIteration(model, renderer, opt_f, scaler_f, cam, {'origin': origin, 'view_direction': vdir, 'rgb': target}, n_rays, capacity, order=perm)
    marched = torch.zeros((), dtype=torch.int64, device=dev); cut = torch.zeros((), dtype=torch.int64, device=dev)
    for i in range(3):
        out = fused()
        marched += out['rm_samples']; cut += out['sample_overflow']     # (also loads the accumulation kernels before the clock starts)
    marched.zero_(); cut.zero_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    marks = {}
    for i in range(iters):
        if i in (third, iters - third):
            torch.cuda.synchronize(); marks[i] = time.perf_counter()
        out = fused()
        marched += out['rm_samples']; cut += out['sample_overflow']
    torch.cuda.synchronize(); t1 = time.perf_counter(); dt_f = (t1 - t0) / iters
    dt_first = (marks.get(third, t1) - t0) / third; dt_last = (t1 - marks.get(iters - third, t0)) / third
    loss_f = float(out['loss'])
    restore()
    res['fused'] = {'what': 'FusedTrainingIteration: batch + clip + draws + march | encode + MLPs | compositing + loss fwd/bwd | MLP + grid backward + Adam; 13 launches, '
                            'next batch marched ahead on a side stream, eager calls (no HIP graph)',
                    'ms_per_iteration': round(dt_f * 1e3, 3), 'ms_per_iteration_first_third': round(dt_first * 1e3, 3),
                    'ms_per_iteration_last_third': round(dt_last * 1e3, 3), 'sample_capacity': capacity, 'samples_per_iteration': int(marched.item() / iters),
                    'samples_cut': int(cut.item()), 'msamples_per_s': round(marched.item() / iters / dt_f / 1e6, 1), 'loss_last': round(loss_f, 4),
                    'weight_decay': 'L2 slice inside the Adam launch', 'launches_per_iteration': 13}
    return res


# ------------------------------------------------------------------------------------------------ data-parallel training legs
def _max_over_ranks(values, device, world):
    import torch
    import torch.distributed as dist
    t = torch.tensor(values, device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def _replica_drift(flat, world):
    """max |parameter - rank 0's parameter|: replicas that saw the same reduced gradients and took the same step differ by exactly 0"""
    import torch.distributed as dist
    if world == 1:
        return 0.0
    ref = flat.clone()
    dist.broadcast(ref, src=0)
    return float((flat.detach() - ref.detach()).abs().max())


def dp_ingp_leg(model, renderer, cam, poses, rank, world, device, rays_per_rank=2200, iters=10):
    """InstantNGP data-parallel training iteration: every rank draws the SAME seeded global batch (ray ids and march jitter), renders its share
    ray_ids[rank::world] (parallel.shard_ray_ids: the global ray set is the single-GPU one), the encoding / MLP gradients are averaged with ONE
    flat reduce-scatter + all-gather over RCCL (parallel.allreduce_gradients), every rank applies the same fused Adam step.  Weak scaling:
    rays_per_rank per GPU.  Reports the iteration time, the collective alone (HIP events around it), the bytes it reduces and the bus
    bandwidth 2 (N-1)/N * bytes / time per GPU; asserts that the replicas end bit-identical and saw the same loss."""
    import torch
    import torch.distributed as dist
    from nerficg_amd import parallel
    from nerficg_amd.apex_optimizers import FusedAdam
    from nerficg_amd.raygen import generate_rays
    saved = [p.detach().clone() for p in model.parameters()]
    if world > 1:
        parallel.broadcast_parameters(model.parameters())
    rays = [generate_rays(cam.width, cam.height, cam.focal_x, cam.focal_y, cam.center_x, cam.center_y, p, device=device, want_direction=False) for p in poses[:2]]
    origin = torch.cat([r['origin'] for r in rays]); vdir = torch.cat([r['view_direction'] for r in rays])
    perm = torch.randperm(origin.shape[0], generator=torch.Generator(device='cpu').manual_seed(0)).to(device)  # identical on every rank
    target = torch.rand(origin.shape[0], 3, device=device, generator=torch.Generator(device=device).manual_seed(1))
    opt = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False)  # Trainer.py:35
    # found-inf OR-ed over the ranks and the marched sample count summed, in one small collective per iteration (Trainer.py:44,89-94)
    scaler = parallel.DataParallelGradScaler(init_scale=128.0, growth_interval=10 ** 9)
    n_global = rays_per_rank * world
    params = list(model.parameters())
    coll_ms, losses, global_samples = [], [], []

    def step(i, timed):
        gen = torch.Generator(device=device).manual_seed(1000 + i)  # the same background and jitter on every rank
        batch = perm[(i * n_global) % (perm.numel() - n_global):][:n_global]
        jitter = torch.rand(n_global, device=device, generator=gen)
        bg = torch.rand(3, device=device, generator=gen)
        ids, noise = parallel.shard_ray_ids(batch, rank, world), parallel.shard_ray_ids(jitter, rank, world)
        with torch.amp.autocast('cuda'):
            out = renderer.render_rays(origin[ids], vdir[ids], cam, train_mode=True, custom_bg_color=bg, noise=noise)
            loss = torch.nn.functional.mse_loss(out['rgb'].float(), target[ids]) + 0.5e-6 * model.weight_decay_mlp()
        scaler.scale(loss).backward()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        parallel.allreduce_gradients(params, average=True)
        b.record()
        scaler.piggyback = [out['rm_samples']]
        scaler.step(opt); scaler.update(); opt.zero_grad()
        if timed:
            coll_ms.append((a, b))
            losses.append(loss.detach())
            global_samples.append(scaler.reduced[0])
        return out['rm_samples']

    for i in range(3):
        step(i, False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    n_samples = [step(3 + i, True) for i in range(iters)]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / iters
    coll = sum(a.elapsed_time(b) for a, b in coll_ms) / iters
    drift = _replica_drift(torch.cat([p.detach().reshape(-1) for p in params]), world)
    # every rank optimises its own share of the batch, so losses differ between ranks; what must agree is what the NETWORK computes afterwards
    probe = model.encoding_xyz(torch.rand(4096, 3, device=device, generator=torch.Generator(device=device).manual_seed(7))).float()
    out_drift = _replica_drift(probe.reshape(-1).contiguous(), world)
    nbytes = sum(p.numel() * 4 for p in params)
    dt, coll = _max_over_ranks([dt, coll], device, world)
    samples = float(torch.stack(n_samples).double().mean())
    # the batch-size controller of Trainer.py:73-75 on the GLOBAL count: every rank computes the same next batch size
    total_global = float(torch.stack(global_samples).sum())
    next_rays = parallel.rays_per_batch_update(rays_per_rank, 262144, total_global, iters, world)
    agree = _max_over_ranks([next_rays, -next_rays], device, world)
    if agree[0] != -agree[1]:
        raise RuntimeError(f'InstantNGP data-parallel replicas drifted: next rays_per_batch {next_rays} differs between ranks')
    with torch.no_grad():
        for p, q in zip(model.parameters(), saved):
            p.copy_(q)
    if drift != 0.0 or out_drift != 0.0:
        raise RuntimeError(f'InstantNGP data-parallel replicas drifted: parameters {drift:.3e}, network output {out_drift:.3e}')
    # the same data-parallel iteration through the fused trainer (nerficg_amd.ngp_trainer, data_parallel: per-rank slices of the global batch, ONE
    # flat gradient buffer, one reduce-scatter + all-gather between the backward pass and the step); N > 1 only -- at N = 1 it is `training.fused`
    fused_dp = None
    if world > 1:
        from nerficg_amd.amp import GradScaler as _GS
        from nerficg_amd.ngp_trainer import FusedTrainingIteration

        def fused_leg(sharded, wire_dtype=torch.float32):
            """sharded=True: round 6's step (small all-reduce beside the grid backward, in-place reduce-scatter, Adam on the rank's shard, all-gather of the fp16
            table, next batch marched beside the collective); False: round 5's (one flat reduce-scatter + all-gather of the f32 gradient, Adam on everything)."""
            it, built = None, None
            try:
                opt_f = FusedAdam(model.parameters(), lr=1e-2, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False, capturable=True)
                it = FusedTrainingIteration(model, renderer, opt_f, _GS(init_scale=128.0, growth_interval=10 ** 9), cam, {'origin': origin, 'view_direction': vdir, 'rgb': target},
                                            rays_per_rank, (int(1.15 * samples) + 4095) // 4096 * 4096, order=perm, seed=5, sharded=sharded, dp_timing=bool(sharded), wire_dtype=wire_dtype)
            except Exception as e:
                built = repr(e)[:300]
            # every rank enters the iterations (they hold a collective) or none does: a rank that failed to build must not leave the others waiting
            n_failed = _max_over_ranks([0.0 if built is None else 1.0], device, world)[0]
            try:
                if n_failed:
                    raise RuntimeError(built or 'another rank failed to build the fused trainer')
                for _ in range(3):
                    it()
                it.dp_times()
                torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
                for _ in range(3 * iters):
                    out_f = it()
                torch.cuda.synchronize(); dist.barrier()
                dt_f = _max_over_ranks([(time.perf_counter() - t0) / (3 * iters)], device, world)[0]
                times = it.dp_times()
                it.gather_state()
                drift_f = _replica_drift(torch.cat([p.detach().reshape(-1) for p in params]), world)
                half_drift = _replica_drift(model.encoding_xyz._half_params().float(), world)
                leg = {'ms_per_iteration': round(dt_f * 1e3, 3), 'rays_per_iteration': n_global, 'mrays_per_s': round(n_global / dt_f / 1e6, 3),
                       'replica_drift': drift_f, 'fp16_table_drift': half_drift, 'samples_cut': int(out_f['sample_overflow']), 'sharded_step': bool(it.sharded),
                       'next_batch_marched_beside': it.prefetch_at if it.prefetch_default else None}
                wire = it.layout.wire_bytes(2 if it.wire is not None else 4)
                if it.wire is not None:
                    leg['wire_dtype'], leg['wire_values_saturated'] = 'fp16 (saturating, summed in fp16 over the ranks)', int(it.wire_saturated)
                if it.sharded:
                    tm = _max_over_ranks([times['reduce_scatter_ms'], times['adam_ms'], times['all_gather_ms'], times['exposed_ms']], device, world)
                    leg.update(wire_bytes_per_gpu=wire['total'], wire=wire, adam_elements_per_rank=wire['adam_elements_per_rank'],
                               collective_ms=round(tm[0] + tm[2], 4), reduce_scatter_ms=round(tm[0], 4), sharded_adam_ms=round(tm[1], 4), all_gather_ms=round(tm[2], 4),
                               exposed_ms=round(tm[3], 4), exposed_note='end of the backward pass (and of the small all-reduce) -> the main stream continues: reduce-scatter + settle + Adam + all-gather')
                else:
                    n_all = it.grads.numel()
                    leg.update(wire_bytes_per_gpu=int(2 * (world - 1) / world * n_all * 4), adam_elements_per_rank=n_all - it.aux.numel())
                if drift_f != 0.0 or half_drift != 0.0:
                    leg['error'] = f'replicas drifted: parameters {drift_f:.3e}, fp16 table {half_drift:.3e}'
                return leg
            except Exception as e:   # an extra leg must never take the line down
                return {'error': repr(e)[:300]}
            finally:
                with torch.no_grad():
                    for p, q in zip(model.parameters(), saved):
                        p.copy_(q)
        fused_dp = fused_leg(True)
        fused_dp['replicated_step'] = fused_leg(False)
        fused_dp['sharded_step_fp16_wire'] = fused_leg(True, torch.float16)      # optional: half the reduce-scatter's bytes, fp16 summation (off by default)
    return {'fused_trainer': fused_dp, 'ms_per_iteration': round(dt * 1e3, 3), 'rays_per_iteration': n_global, 'samples_per_iteration_per_gpu': round(samples), 'mrays_per_s': round(n_global / dt / 1e6, 3),
            'collective': f'one flat f32 bucket, {"all_reduce" if dist.get_backend() == "gloo" else "reduce-scatter + all-gather"} over {dist.get_backend()}' if world > 1 else None, 'bytes_reduced_per_iteration': nbytes if world > 1 else 0,
            'collective_ms': round(coll, 3) if world > 1 else None,
            'bus_GBps_per_gpu': round(2 * (world - 1) / world * nbytes / (coll * 1e-3) / 1e9, 1) if world > 1 and coll > 0 else None,
            'scalar_collective': 'rm_samples (sum) + GradScaler found-inf (or), one packed all-reduce per iteration' if world > 1 else None,
            'global_samples_per_iteration': round(total_global / iters), 'next_rays_per_batch': next_rays,
            'replica_drift': drift, 'network_output_drift': out_drift, 'final_loss': round(float(losses[-1]), 6)}


def dp_gs_leg(rank, world, device, n_gaussians=1_000_000, iters=20):
    """3DGS view-parallel optimisation step: replicated Gaussians, every rank rasterizes ANOTHER view of the orbit, 0.8 L1 + 0.2 DSSIM, backward,
    visibility-sparse gradient reduction (parallel.sparse_allreduce_gradients: only rows seen by at least one rank travel), the same fused Adam
    step everywhere.  Weak scaling: one view per GPU and step."""
    import torch
    import torch.distributed as dist
    from nerficg_amd import parallel
    from nerficg_amd.gaussian_splatting import Gaussians, PerspectiveCamera, render_image_training, training_loss
    from tests import scenes
    sc = scenes.gs_random_scene(n_gaussians, seed=0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)  # noqa: E731
    g = Gaussians(T(sc['means3D']), torch.log(T(sc['scales'])), T(sc['rotations']), torch.logit(T(sc['opacities']).clamp(1e-4, 1 - 1e-4))[:, None].contiguous(),
                  T(sc['shs'][:, :1]), T(sc['shs'][:, 1:]))
    g.training_setup(training_cameras_extent=4.5)
    g.fuse_rest_step = False           # the reference's step order first (backward, then optimizer.step() on all six groups); the in-backward f_rest step is timed below
    params = [grp['params'][0] for grp in g.optimizer.param_groups]
    if world > 1:
        parallel.broadcast_parameters(params)
    cam = PerspectiveCamera(GS_W, GS_H, 1.2 * GS_W, 1.2 * GS_W, background_color=torch.zeros(3, device=device))
    target = torch.rand(3, GS_H, GS_W, device=device, generator=torch.Generator(device=device).manual_seed(1))
    rows, coll_ms = [0, 0], []

    # poses as device tensors: make_raster_settings assembles the camera on the GPU, the rasterizer reads it from a device block
    pose_of = lambda i: torch.from_numpy(np.asarray(scenes.orbit_pose(0.8 + 0.7 * (i * world + rank), 0.35, 4.5), dtype=np.float32)).to(device)  # noqa: E731
    poses_dev = [pose_of(i) for i in range(2 + iters)]

    exchange = parallel.UnionRowExchange(n_gaussians, device)

    def step(i, timed):
        out = render_image_training(g, cam, poses_dev[i])
        exchange.begin(out['radii'])      # (visible = radius > 0) mask max-reduce + device compaction + count to the host, beside the backward pass
        training_loss(out['rgb'], target).backward()
        if world > 1:       # (timing events are stream markers of their own: not in the single-GPU step, which has no collective to time)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
        n_union = exchange.finish(params, average=True)   # pack (1 launch) -> reduce-scatter + all-gather -> unpack (1 launch)
        if world > 1:
            b.record()
        g.optimizer.step(); g.optimizer.zero_grad()
        if timed:
            rows[0] += max(n_union, 0); rows[1] += 1
            if world > 1:
                coll_ms.append((a, b))

    gc.collect()      # before the warm-up steps, not between them and the timed loop: the chip must not sit idle (and clock down) in front of the first timed step
    for i in range(4):    # (four: the plain step takes its 236 MB of gradient tensors from torch's allocator, emptied behind the previous leg -- 2 of 20 leases had this leg at 1.45 / 1.56 ms with two)
        step(i, False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(iters):
        step(2 + i, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / iters
    coll = sum(a.elapsed_time(b) for a, b in coll_ms) / iters if coll_ms else 0.0
    drift = _replica_drift(torch.cat([p.detach().reshape(-1) for p in params]), world)
    dt, coll = _max_over_ranks([dt, coll], device, world)
    union_rows = rows[0] / max(rows[1], 1)
    nbytes = int(union_rows * 59 * 4)
    if drift != 0.0:
        raise RuntimeError(f'3DGS view-parallel replicas drifted: {drift:.3e}')
    graphed = None
    rest_in_backward = None
    if world == 1:
        # opt-in variant: Adam of the f_rest group inside the preprocessing backward.  Same parameters as the plain step on every iteration in
        # which optimizer.step() follows the backward pass directly; the reference's trainer lets densify / reset_opacities run between the two on some
        # iterations (Trainer.py:100-128), which drops that iteration's update there -- a loop using the variant switches it off for those iterations.
        g.fuse_rest_step = True
        for i in range(4):
            step(i, False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(iters):
            step(2 + i, False)
        torch.cuda.synchronize()
        rest_in_backward = {'ms_per_step': round((time.perf_counter() - t0) / iters * 1e3, 3), 'what': 'Gaussians.fuse_rest_step = True: nrc_gs_backward_rest_step + nrc_adam_step_multi; with rest_step_schedule it applies to 29 856 of the 30 000 iterations of the reference schedule (all but those in which densify_and_prune runs between backward and optimizer.step())'}
        g.fuse_rest_step = False
        # the same step (plus the densification statistics of the reference's trainer) recorded in a HIP graph: fixed list / span capacities
        # from the counts of an op-by-op frame, pose and target as device inputs, Adam's step counters and learning rates on the device
        from nerficg_amd import diff_gaussian_rasterization as dgr
        from nerficg_amd.graphs import gaussian_splatting_step
        n_inst, n_spans = dgr.last_counts().tolist()
        caps = (int(1.3 * n_inst), int(1.3 * n_spans) + 65536)
        g.optimizer.capturable = True
        gstep = gaussian_splatting_step(g, cam, instance_capacity=caps[0], span_capacity=caps[1])
        worst = torch.zeros(2, dtype=torch.int64, device=device)
        for i in range(3):   # op by op, recording + first replay, second replay (and the first use of every torch kernel of the loop below)
            worst = torch.maximum(worst, gstep(c2w=poses_dev[i % 2], target=target)['counts'])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(iters):
            worst = torch.maximum(worst, gstep(c2w=poses_dev[2 + i], target=target)['counts'])
        torch.cuda.synchronize(); dt_g = (time.perf_counter() - t0) / iters
        worst = worst.tolist()
        graphed = {'ms_per_step': round(dt_g * 1e3, 3), 'msplats_per_s': round(n_gaussians / dt_g / 1e6, 1), 'instance_capacity': caps[0],
                   'span_capacity': caps[1], 'max_instances_seen': worst[0], 'max_spans_seen': worst[1],
                   'dropped': bool(worst[0] > caps[0] or worst[1] > caps[1]), 'includes': 'densification statistics (nrc_gs_densify_stats)'}
    return {'hip_graph': graphed, 'ms_per_step': round(dt * 1e3, 3), 'rest_step_in_backward': rest_in_backward, 'views_per_step': world, 'gaussians': n_gaussians, 'msplats_per_s': round(world * n_gaussians / dt / 1e6, 1),
            'collective': f'max-reduce of the visibility mask beside the backward pass (device compaction, count read under it) + one packed reduction of the union rows over {dist.get_backend()}' if world > 1 else None,
            'wire_bytes_per_gpu': int(2 * (world - 1) / world * nbytes + (world - 1) / world * 2 * n_gaussians) if world > 1 else 0,
            'union_rows_per_step': round(union_rows), 'bytes_reduced_per_step': nbytes if world > 1 else 0, 'collective_ms': round(coll, 3) if world > 1 else None,
            'bus_GBps_per_gpu': round(2 * (world - 1) / world * nbytes / (coll * 1e-3) / 1e9, 1) if world > 1 and coll > 0 else None, 'replica_drift': drift}


def spawn_ranks(n: int) -> int:
    """Runs this script as n ranks under torch.distributed.run (the driver's own command line) and returns its exit code.  No GPU call has
    happened in this process; the children inherit stdout, so rank 0's JSON line is the output."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1', '--master-port', str(port),
           str(Path(__file__).resolve()), *sys.argv[1:]]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-gs', action='store_true', help='skip the secondary 3DGS leg')
    ap.add_argument('--no-train', action='store_true', help='skip the InstantNGP training-iteration leg')
    ap.add_argument('--no-gs-large', action='store_true', help='skip the 6 M-Gaussian run of the 3DGS leg')
    ap.add_argument('--no-dp', action='store_true', help='skip the data-parallel training legs (gradient collectives over RCCL)')
    ap.add_argument('--gs-gaussians', type=int, default=1_000_000)
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend for N > 1 (nccl = RCCL; gloo only to exercise the code path)')
    ap.add_argument('--pipeline', type=int, default=1, help='tile ranges per frame of the pipelined image path (1 = render_image_fused, one pass over the frame)')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='weak (default, the headline): every rank renders its own 800x800 frame per step; strong: ONE 1600x1060 garden-shaped frame per step '
                         'cut into contiguous tile shards over the ranks + all-gather of the pixels (BASELINE configs[3])')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (torch.distributed.run, one process per GPU) BEFORE anything in
        # this process touches the GPU -- the parent only relays the children's output and exit code
        raise SystemExit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist
    # The cyclic collector stays OFF for the whole run and runs at the leg boundaries instead (gc.collect() below and inside the legs): a collection inside a
    # timed loop -- e.g. the teardown of an earlier leg's HIP graph, tens of milliseconds -- is not part of any path under test (r05: fused leg; r06: the
    # drop-in training leg at 1.9 instead of 0.6-0.7 ms on one lease).  Reference counting frees tensors as usual.
    gc.disable()
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world} -- launch with --nproc-per-node {args.gpus} (or without a launcher: '
                         f'`python bench.py --gpus {args.gpus}` starts the ranks itself)')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the product path has no CPU fallback')
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    device = torch.device('cuda', local_dev)
    if world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(args.backend)
    red_dev = device if args.backend == 'nccl' else torch.device('cpu')  # where the tiny timing reductions live

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.scaling == 'strong':
        c4_model, c4_renderer, c4_cam, c4_poses = build_c4_scene(device)
        gc.collect(); gc.disable()
        r = strong_scaling_frames(c4_renderer, c4_cam, c4_poses, rank, world, args.steps, args.warmup, barrier, red_dev)
        if rank == 0:
            elapsed = r.pop('elapsed')
            print(json.dumps({
                'metric': 'Mrays/s (INGP garden-shaped frame, strong scaling)', 'value': r['mrays_per_s'], 'unit': 'Mrays/s', 'n_gpus': world, 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
                'dtype': 'f16 (tables, weights, activations) / f32 (accumulate, march, composite)', 'data': 'synthetic',
                'config': {'workload': f'ingp_garden_shape: ONE {C4_W}x{C4_H} frame per step ({C4_W * C4_H} rays) cut into {world} contiguous tile shards, SCALE 2 '
                                       '(3 occupancy cascades, content in each), exponential steps 1/256, camera inside the box, random-init hash grid + MLPs seed 0; '
                                       'every rank ends with the whole frame (all-gather of (rays / N, 5) pixel blocks)',
                           'rays_per_step': C4_W * C4_H, 'samples_per_ray': r['samples_per_ray'], 'parallelism': f'tiles of one frame x{world} (strong)'},
                'strong': r, 'roofline': None, 'cpu_baseline': None,
                'note': 'roofline / cpu_baseline are carried by the default (weak, N = 1) line: same kernels, same per-sample rate'}), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    model, renderer, cam, poses = build_scene(device)

    def step(i):
        pose = poses[(i * world + rank) % N_POSES]
        if args.pipeline > 1:   # the march of tile range k + 1 next to the encode / MLP kernels of range k (InstantNGPRenderer.render_image_pipelined)
            return renderer.render_image_pipelined(cam, pose, shards=args.pipeline, return_stats=True)
        return renderer.render_image_fused(cam, pose, return_stats=True)

    samples = 0
    gc.collect()  # (the collector is off for the whole run: a full collection in the middle of the timed steps costs tens of milliseconds and is not part of the path
                  # under test; collected in FRONT of the warm-up frames so that nothing stands between them and the timed ones)
    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    _dbg = []
    for i in range(args.steps):
        _t = time.perf_counter()
        samples += step(args.warmup + i)['n_samples']
        _dbg.append(time.perf_counter() - _t)
    barrier()
    elapsed_now = time.perf_counter() - t0
    if os.environ.get('NRC_BENCH_DEBUG'):
        print('per-step ms:', ' '.join(f'{x * 1e3:.1f}' for x in _dbg), file=sys.stderr)
    elapsed = elapsed_now
    if world > 1:
        t = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        s = torch.tensor([samples], device=red_dev, dtype=torch.int64)
        dist.all_reduce(s)
        samples = int(s.item())

    # ---- secondary metric: 3DGS rasterizer forward / forward+backward on the 1 M-Gaussian synthetic scene (every rank its own copy)
    gs_res = None
    if not args.no_gs:
        gs = build_gs_scene(device, args.gs_gaussians)
        gs_res = time_gs(gs, reps=max(3, args.steps // 2), barrier=barrier)
        if world > 1:
            t = torch.tensor([gs_res['ms_fwd'], gs_res['ms_fwd_bwd']], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            gs_res['ms_fwd'], gs_res['ms_fwd_bwd'] = float(t[0]), float(t[1])
        n_g = gs_res['gaussians'] * world
        gs_res['msplats_per_s_fwd'] = round(n_g / gs_res['ms_fwd'] / 1e3, 2)
        gs_res['msplats_per_s_fwd_bwd'] = round(n_g / gs_res['ms_fwd_bwd'] / 1e3, 2)
        del gs
        # BASELINE config C5 size (6 M Gaussians), single-GPU runs only: the same kernels with 50 M instances per frame
        if world == 1 and not args.no_gs_large:
            torch.cuda.empty_cache()
            big = build_gs_scene(device, 6_000_000)
            gs_res['large'] = time_gs(big, reps=8, barrier=barrier)
            del big
            torch.cuda.empty_cache()
        # the same 1 M Gaussians at BASELINE's 1600x1060 (SURVEY 8d names both image sizes for C3 / C5)
        if world == 1:
            wide = build_gs_scene(device, args.gs_gaussians, w=C3B_W, h=C3B_H)
            gs_res['c3_1600x1060'] = time_gs(wide, reps=8, barrier=barrier)
            del wide
            torch.cuda.empty_cache()

    def headline_result():
        """Rank 0: the line's headline part -- value, rooflines of the frame's two kernels, the 3DGS secondary -- without the training legs."""
        rays = W * H * args.steps * world
        value = rays / elapsed / 1e6
        # the dominant kernel, timed on the poses the timed region of THIS rank rendered (all of them up to 20, else an even subset)
        timed_poses = [poses[((args.warmup + i) * world + rank) % N_POSES] for i in range(args.steps)]
        if len(timed_poses) > 20:
            timed_poses = timed_poses[::max(1, len(timed_poses) // 20)][:20]
        kt = time_dominant_kernel(renderer, cam, timed_poses)
        k_ms, mlp_ms, k_live = kt['enc_ms'], kt['mlp_ms'], kt['live_per_launch']
        in_frame = time_kernels_in_frame(renderer, cam, timed_poses)
        achieved = ENC_BYTES_PER_SAMPLE * k_live / (k_ms * 1e-3) / 1e9
        pmc_all = {}
        pmc = ROOT / 'profiles' / 'pmc_summary.json'
        if pmc.exists():
            try:
                pmc_all = json.loads(pmc.read_text())
            except Exception:
                pmc_all = {}
        enc_pmc, enc_pmc_src = pmc_entry(pmc_all, DOMINANT_KERNEL, 'ngp_net.hip')
        mlp_pmc, mlp_pmc_src = pmc_entry(pmc_all, 'k_ngp_mlp<SRC_TILED>', 'ngp_net.hip')
        result = {
            'metric': 'Mrays/s (INGP lego)', 'value': round(value, 4), 'unit': 'Mrays/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f16 (tables, weights, activations) / f32 (accumulate, march, composite)', 'data': 'synthetic',
            'config': {'workload': 'ingp_lego: 800x800 image render per rank per step (640000 rays), solid-sphere occupancy r<0.35, '
                                   'random-init hash grid (T=2^19, L=16, F=2) + 64-wide MLPs, seed 0, 100 seeded orbit poses',
                       'rays_per_step_per_gpu': W * H, 'samples_per_ray': round(samples / rays, 3), 'parallelism': f'rays x{world} (weak)'},
            'msamples_per_s': round(samples / elapsed / 1e6, 3),
            # dominant kernel on SURVEY 8(d)'s algorithmic bytes (512 B of table reads per live sample) against the HBM peak.  The 24.4 MB table
            # is L2 / Infinity-Cache resident, so HBM is NOT what limits the kernel: `limiter` names the resource that does (L1 tag lookups,
            # from the PMC counters in profiles/), and `traffic` is the memory-side traffic per launch measured there.
            'roofline': {'bound': 'hbm', 'kernel': DOMINANT_KERNEL, 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 4), 'traffic': enc_pmc.get('hbm_bytes_per_launch'),
                         'algorithmic_bytes_per_launch': ENC_BYTES_PER_SAMPLE * k_live, 'algorithmic_bytes_per_sample': ENC_BYTES_PER_SAMPLE,
                         'kernel_bytes_per_sample': ENC_KERNEL_BYTES_PER_SAMPLE,
                         'kernel_ms': round(k_ms, 4), 'kernel_ms_min_pose': round(kt['enc_ms_min'], 4), 'kernel_ms_max_pose': round(kt['enc_ms_max'], 4),
                         'timed_over': f"{kt['poses']} poses of the timed region, {kt['launches_per_image']:.1f} launches per image, HIP events on the launch stream",
                         'second_ruler': _second_ruler('k_grid_encode<1', in_frame.get('k_grid_encode'), ENC_BYTES_PER_SAMPLE * k_live / 1e9, HBM_PEAK_GBS, 1e-3),
                         'samples_per_launch': k_live, 'slots_per_launch': kt['slots_per_launch'],
                         'limiter': {'resource': 'L1 (TCP): tag lookups, and behind them the L1 misses of the four finest levels (7.3 of the 7.8 L2 requests per sample, 47 % of the kernel: profiles/r05_encoder_levels.md)',
                                     'achieved': enc_pmc.get('tcp_accesses_per_clk_per_cu'), 'peak': 1.0,
                                     'unit': 'cache-line lookups per clock per CU', 'l1_hit_rate': enc_pmc.get('l1_hit_rate'),
                                     'l2_hit_rate': enc_pmc.get('l2_hit_rate'), 'source': enc_pmc_src},
                         'traffic_source': enc_pmc_src},
            # second kernel of the pair: the tiny-MLP chain on MFMA (20 480 FLOP per sample, padded layer widths)
            'roofline_mfma': {'bound': 'mfma', 'kernel': 'k_ngp_mlp<SRC_TILED>', 'achieved': round(MLP_FLOP_PER_SAMPLE * k_live / (mlp_ms * 1e-3) / 1e12, 2),
                              'peak': MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(MLP_FLOP_PER_SAMPLE * k_live / (mlp_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                              'traffic': mlp_pmc.get('hbm_bytes_per_launch'), 'traffic_source': mlp_pmc_src, 'kernel_ms': round(mlp_ms, 4),
                              'kernel_ms_min_pose': round(kt['mlp_ms_min'], 4), 'kernel_ms_max_pose': round(kt['mlp_ms_max'], 4),
                              'flop_per_sample': MLP_FLOP_PER_SAMPLE, 'samples_per_launch': k_live,
                              'second_ruler': _second_ruler('k_ngp_mlp<1', in_frame.get('k_ngp_mlp'), MLP_FLOP_PER_SAMPLE * k_live / 1e12, MFMA_PEAK_TFLOPS, 1e-3)},
        }
        if gs_res is not None:
            # bytes_fwd = 308 P_vis + 148 D + 20 H W ; bytes_bwd ~ 76 D + 472 P_vis + 20 H W
            b_fwd = 308 * gs_res['visible'] + 148 * gs_res['instances'] + 20 * GS_W * GS_H
            b_bwd = 76 * gs_res['instances'] + 472 * gs_res['visible'] + 20 * GS_W * GS_H
            result['secondary'] = {
                'metric': 'Msplats/s (3DGS, 1 M synthetic Gaussians, 1297x840)', 'value_fwd': gs_res['msplats_per_s_fwd'],
                'value_fwd_bwd': gs_res['msplats_per_s_fwd_bwd'], 'unit': 'Msplats/s', 'ms_fwd': gs_res['ms_fwd'], 'ms_fwd_bwd': gs_res['ms_fwd_bwd'],
                'timing': 'median of five averages over %d back-to-back frames each' % max(3, args.steps // 2),
                'trials_ms_fwd': gs_res.get('trials_ms_fwd'), 'trials_ms_fwd_bwd': gs_res.get('trials_ms_fwd_bwd'),
                'gaussians_per_gpu': gs_res['gaussians'], 'visible': gs_res['visible'], 'instances': gs_res['instances'], 'dtype': 'f32',
                'roofline': {'bound': 'hbm', 'scope': 'whole forward / forward+backward (all rasterizer kernels)',
                             'achieved_fwd': round(b_fwd / (gs_res['ms_fwd'] * 1e-3) / 1e9, 2),
                             'achieved_fwd_bwd': round((b_fwd + b_bwd) / (gs_res['ms_fwd_bwd'] * 1e-3) / 1e9, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                             'frac_fwd': round(b_fwd / (gs_res['ms_fwd'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             'frac_fwd_bwd': round((b_fwd + b_bwd) / (gs_res['ms_fwd_bwd'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             'algorithmic_bytes_fwd': b_fwd, 'algorithmic_bytes_bwd': b_bwd,
                             'kernels': gs_kernel_rooflines(gs_res, pmc_all)}}
            if 'large' in gs_res:
                lg = gs_res['large']
                lb_fwd = 308 * lg['visible'] + 148 * lg['instances'] + 20 * GS_W * GS_H
                lb_bwd = 76 * lg['instances'] + 472 * lg['visible'] + 20 * GS_W * GS_H
                result['secondary']['six_million'] = {
                    'value_fwd': lg['msplats_per_s_fwd'], 'value_fwd_bwd': lg['msplats_per_s_fwd_bwd'], 'unit': 'Msplats/s', 'ms_fwd': lg['ms_fwd'],
                    'ms_fwd_bwd': lg['ms_fwd_bwd'], 'trials_ms_fwd_bwd': lg.get('trials_ms_fwd_bwd'), 'gaussians': lg['gaussians'], 'visible': lg['visible'],
                    'instances': lg['instances'],
                    'frac_fwd': round(lb_fwd / (lg['ms_fwd'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    'frac_fwd_bwd': round((lb_fwd + lb_bwd) / (lg['ms_fwd_bwd'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            if 'c3_1600x1060' in gs_res:
                wd = gs_res['c3_1600x1060']
                wb_fwd = 308 * wd['visible'] + 148 * wd['instances'] + 20 * wd['pixels']
                wb_bwd = 76 * wd['instances'] + 472 * wd['visible'] + 20 * wd['pixels']
                result['secondary']['c3_1600x1060'] = {
                    'workload': f"{wd['gaussians']} Gaussians, {wd['image']} (the image size BASELINE.json names; the reference yaml's IMAGE_SCALE_FACTOR 0.25 gives 1297x840)",
                    'value_fwd': wd['msplats_per_s_fwd'], 'value_fwd_bwd': wd['msplats_per_s_fwd_bwd'], 'unit': 'Msplats/s', 'ms_fwd': wd['ms_fwd'],
                    'ms_fwd_bwd': wd['ms_fwd_bwd'], 'trials_ms_fwd': wd.get('trials_ms_fwd'), 'trials_ms_fwd_bwd': wd.get('trials_ms_fwd_bwd'),
                    'visible': wd['visible'], 'instances': wd['instances'],
                    'frac_fwd': round(wb_fwd / (wd['ms_fwd'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    'frac_fwd_bwd': round((wb_fwd + wb_bwd) / (wd['ms_fwd_bwd'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    'per_kernel_ms': {k: round(v[0], 4) for k, v in sorted((wd.get('stage_ms') or {}).items())}}
        return result

    # N > 1: rank 0 builds the headline part BEFORE the collectives of the data-parallel legs, and every rank arms a watchdog around them: a rank that
    # fails alone leaves the others waiting inside a collective (try / except cannot reach that), and the line must not be lost over an extra leg.
    result = None
    watchdog = None
    if world > 1 and not args.no_dp:
        if rank == 0:
            result = headline_result()
        deadline = float(os.environ.get('NRC_BENCH_DP_DEADLINE', 300))

        def bail():
            if rank == 0:
                print(json.dumps({**result, 'dp_training': {'error': f'watchdog: the data-parallel legs did not return within {deadline:.0f} s; the headline, the rooflines '
                                                                     'and the 3DGS secondary above were measured before them'}, 'cpu_baseline': None}), flush=True)
            os._exit(0)
        import threading
        watchdog = threading.Timer(deadline + (0.0 if rank == 0 else 10.0), bail)
        watchdog.daemon = True
        watchdog.start()

    # ---- data-parallel training legs: the collectives of SURVEY 8(e) inside a timed iteration (every rank takes part)
    dp = None
    if not args.no_dp:
        dp = {}
        for name, fn in (('ingp', lambda: dp_ingp_leg(model, renderer, cam, poses, rank, world, device)), ('gs', lambda: dp_gs_leg(rank, world, device))):
            gc.collect()
            try:
                dp[name] = fn()
            except Exception as e:   # replica drift included: reported in the line (every rank sees the same reduced drift, so every rank leaves the leg together)
                dp[name] = {'error': repr(e)[:300]}
            torch.cuda.empty_cache()
    if watchdog is not None:
        watchdog.cancel()

    if rank == 0:
        if result is None:
            result = headline_result()
        gc.collect()
        if not args.no_train:
            try:
                result['training'] = time_train(model, renderer, cam, poses)
            except Exception as e:  # never lose the headline line over the extra leg
                result['training'] = {'error': repr(e)[:200]}
        result['dp_training'] = dp
        gc.collect()
        if world == 1 and not args.no_train:
            try:
                result['secondary_trained'] = trained_scene_leg(device)
            except Exception as e:
                result['secondary_trained'] = {'error': repr(e)[:300]}
            torch.cuda.empty_cache()
        # BASELINE configs[3] shape on this one GPU (the single-rank point of `--scaling strong`) and one of eight shards of the same frames
        gc.collect()
        if world == 1:
            try:
                torch.cuda.empty_cache()
                c4_model, c4_renderer, c4_cam, c4_poses = build_c4_scene(device)
                whole = strong_scaling_frames(c4_renderer, c4_cam, c4_poses, 0, 1, 5, 2, barrier, red_dev)
                nt = c4_renderer.n_image_tiles(c4_cam)
                from nerficg_amd import parallel as _par
                b3, e3 = _par.shard_range(nt, 3, 8)
                for i in range(2):
                    c4_renderer.render_image_fused(c4_cam, c4_poses[i], tile_begin=b3, n_tiles=e3 - b3)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for i in range(5):
                    c4_renderer.render_image_fused(c4_cam, c4_poses[2 + i], tile_begin=b3, n_tiles=e3 - b3)
                torch.cuda.synchronize(); dt_shard = (time.perf_counter() - t0) / 5
                whole.pop('elapsed', None)
                result['config_c4'] = {'workload': f'{C4_W}x{C4_H} garden-shaped frame (3 cascades, exponential steps), one GPU', **whole,
                                       'one_of_eight_shards_ms': round(dt_shard * 1e3, 4), 'one_of_eight_shards_tiles': e3 - b3,
                                       'strong_scaling': 'python bench.py --gpus N --scaling strong'}
                del c4_model, c4_renderer
                torch.cuda.empty_cache()
            except Exception as e:
                result['config_c4'] = {'error': repr(e)[:300]}
        result['cpu_baseline'] = None  # timed on rank 0 of single-GPU runs only (the host cores are shared by all ranks otherwise)
        if not args.no_cpu_baseline and world == 1:
            if gs_res is not None:
                result['secondary']['cpu_baseline'] = gs_cpu_baseline()
            pd = model.encoding_xyz.params.detach().half().float().cpu().numpy()
            pc = model.color_mlp_with_encoding.params.detach().half().float().cpu().numpy()
            result['cpu_baseline'] = cpu_baseline(cam, poses[0], (pd, pc, model.occupancy_bitfield.cpu().numpy()))
            try:
                result['cpu_baseline_c1'] = nerf_c1_cpu_baseline()
            except Exception as e:
                result['cpu_baseline_c1'] = {'error': repr(e)[:300]}
        print(json.dumps(result), flush=True)
    if world > 1:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:   # a peer left through its watchdog: rank 0's line is out (or will never be), nothing is gained by a traceback here
            print(f'bench.py rank {rank}: the closing barrier failed ({repr(e)[:120]})', file=sys.stderr, flush=True)
            os._exit(0)


if __name__ == '__main__':
    main()
