#!/usr/bin/env python3
"""tools/stress_binning.py [SECONDS] -- random Gaussian counts / cameras through the rasterizer forward, tile lists compared with the independent
torch statement of tests/test_gpu_gs_parity.py every frame.  The depth sort and the span sweep hand counts from workgroup to workgroup
(published words, polling, tickets above one tile per CU): this looks for the rare interleaving a fixed test set would miss."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from tests import scenes
from tests.test_gpu_gs_parity import _run, _saved, _torch_tile_lists

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(time.time()) & 0xffff)
t0 = time.time()
frames = bad = 0
sizes = []
while time.time() - t0 < budget:
    n = int(rng.choice([rng.integers(1, 5000), rng.integers(4000, 300_000), rng.integers(250_000, 1_400_000), 4096 * int(rng.integers(1, 300)) + int(rng.integers(-2, 3))]))
    n = max(n, 1)
    w, h = int(rng.integers(64, 700)), int(rng.integers(64, 500))
    sc = scenes.gs_random_scene(n, seed=int(rng.integers(0, 1 << 30)), extent=float(rng.uniform(0.8, 1.6)), log_scale_mean=float(np.log(rng.uniform(0.006, 0.04))))
    cam = scenes.gs_camera(w, h, scenes.orbit_pose(float(rng.uniform(0, 6.28)), float(rng.uniform(-0.4, 0.8)), float(rng.uniform(2.5, 5.0))))
    for rep in range(int(rng.integers(1, 4))):
        color, radii, _, _ = _run(sc, cam, [0, 0, 0], requires_grad=True)
        sv, fn = _saved(color)
        want_list, want_ranges = _torch_tile_lists(sv['radii'][:n], sv['points_xy'][:n], fn.debug_state['depths'][:n], w, h)
        ok = fn.num_rendered == want_list.numel() and torch.equal(sv['ranges'].view(-1, 2).to(torch.int32), want_ranges) and \
            torch.equal(sv['point_list'][:want_list.numel()], want_list)
        frames += 1
        if not ok:
            bad += 1
            print('MISMATCH', n, w, h, rep, fn.num_rendered, want_list.numel(), flush=True)
    sizes.append(n)
print(f'{frames} frames, {len(sizes)} scenes (Gaussians {min(sizes)} .. {max(sizes)}), mismatches: {bad}')
sys.exit(1 if bad else 0)
