#!/usr/bin/env python3
"""tools/frame_host_timeline.py -- host-side timeline of render_image_fused around its one wait: when the count pass was enqueued, when the count
arrived in the mailbox, when nrc_ngp_query_samples returned (= its kernels are enqueued), when the frame call returned; microseconds, mean of N frames."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from nerficg_amd import _lib, instant_ngp

dev = torch.device('cuda', 0)
model, renderer, cam, poses = bench.build_scene(dev)
for i in range(3):
    renderer.render_image_fused(cam, poses[i], early_termination=False)
torch.cuda.synchronize()
marks = {}
orig_wait = _lib.HostMailbox.wait
def wait(self, ticket):
    marks['wait_in'] = time.perf_counter()
    r = orig_wait(self, ticket)
    marks['wait_out'] = time.perf_counter()
    return r
_lib.HostMailbox.wait = wait
orig_fwq = instant_ngp.InstantNGPRenderer._fused_write_query
def fwq(self, *a, **k):
    marks['query_in'] = time.perf_counter()
    r = orig_fwq(self, *a, **k)
    marks['query_out'] = time.perf_counter()
    return r
instant_ngp.InstantNGPRenderer._fused_write_query = fwq
acc = {}
N = 12
for i in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    renderer.render_image_fused(cam, poses[3 + i % 8], early_termination=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for k, v in (('count enqueued -> wait entered', marks['wait_in'] - t0), ('waiting for the count', marks['wait_out'] - marks['wait_in']),
                 ('count arrived -> query call entered', marks['query_in'] - marks['wait_out']), ('query call (marshalled) -> returned', marks['query_out'] - marks['query_in']),
                 ('query returned -> frame call returned', t1 - marks['query_out']), ('frame call returned -> GPU done', t2 - t1), ('whole frame', t2 - t0)):
        acc[k] = acc.get(k, 0.0) + v
for k, v in acc.items():
    print('%-42s %8.1f us' % (k, v / N * 1e6))
