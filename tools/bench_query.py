#!/usr/bin/env python3
"""tools/bench_query.py -- developer micro-benchmark: times the stages of the fused InstantNGP image pipeline on one pose.
Usage (GPU box): python tools/bench_query.py [reps]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device('cuda', 0)
model, renderer, cam, poses = bench.build_scene(dev)
for i in range(2):
    renderer.render_image_fused(cam, poses[i])
torch.cuda.synchronize()
kt = bench.time_dominant_kernel(renderer, cam, [poses[2]], reps=reps)
ms, slots, n, ms_mlp = kt['enc_ms'], kt['slots_per_launch'], kt['live_per_launch'], kt['mlp_ms']
t0 = time.perf_counter()
for i in range(reps):
    renderer.render_image_fused(cam, poses[2])
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / reps * 1e3
print(f'encode kernel {ms:.3f} ms, mlp kernel {ms_mlp:.3f} ms per launch of {n} live samples ({slots} slots) = {n / ms / 1e6:.3f} Gsamples/s ; whole image {tot:.3f} ms = {800 * 800 / tot / 1e3:.2f} Mrays/s')
