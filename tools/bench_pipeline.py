#!/usr/bin/env python3
"""tools/bench_pipeline.py -- the 800x800 bench frame through render_image_fused (one pass) and render_image_pipelined with 2 .. 8 tile ranges:
ms per frame over the bench poses, and that the pictures are identical."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench

dev = torch.device('cuda', 0)
model, renderer, cam, poses = bench.build_scene(dev)
ref = {k: v.clone() for k, v in renderer.render_image_fused(cam, poses[7], early_termination=False).items()}


def run(fn, n=20):
    for i in range(3):
        fn(poses[i])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        fn(poses[3 + i])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print('one pass      %.3f ms' % run(lambda p: renderer.render_image_fused(cam, p, return_stats=True)))
for shards in (2, 3, 4, 6, 8):
    ms = run(lambda p: renderer.render_image_pipelined(cam, p, shards=shards, return_stats=True))
    out = renderer.render_image_pipelined(cam, poses[7], shards=shards, return_stats=True)
    same = all(torch.equal(out[k], ref[k]) for k in ('rgb', 'alpha', 'depth'))
    print('%d tile ranges %.3f ms = %.2f Mrays/s, identical picture: %s, samples %d' % (shards, ms, 0.64 / ms * 1e3, same, out['n_samples']))
print('one pass      %.3f ms' % run(lambda p: renderer.render_image_fused(cam, p, return_stats=True)))
