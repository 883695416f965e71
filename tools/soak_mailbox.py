#!/usr/bin/env python3
"""tools/soak_mailbox.py [SECONDS] -- frames of both renderers back to back for a while: every count that arrives through the host mailbox is compared with
the device counter of the same frame (read afterwards), across image sizes / poses / Gaussian counts; reports frames and mismatches."""
import sys, time, warnings
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
from nerficg_amd import _lib
from nerficg_amd.instant_ngp import InstantNGPRenderer
from tests import scenes
from tests.test_gpu_render_parity import make_camera, make_model
from tests.test_gpu_mailbox import _gs_frame

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
warnings.simplefilter('error')          # a mailbox warning (timeout / retire) is a failure here
rng = np.random.default_rng(1)
model = make_model()
r = InstantNGPRenderer(model)
cams = [make_camera(int(w), int(h)) for w, h in ((96, 72), (200, 136), (333, 257), (640, 480))]
gs = [_gs_frame(n=int(n), seed=int(s)) for n, s in ((3000, 1), (30000, 2), (200000, 3))]
seen = []
orig = _lib.HostMailbox.counts
def spy(self, ticket, device):
    got = orig(self, ticket, device)
    seen.append(got)
    return got
_lib.HostMailbox.counts = spy
frames = bad = 0
t0 = time.time()
while time.time() - t0 < budget:
    cam = cams[int(rng.integers(len(cams)))]
    pose = scenes.orbit_pose(float(rng.uniform(0, 6.28)), float(rng.uniform(-0.3, 0.7)), scenes.LEGO_RADIUS)
    seen.clear()
    out = r.render_image_fused(cam, pose, return_stats=True, early_termination=bool(rng.integers(2)))
    ws = next(iter(r._fused_ws.values()))
    dev_counts = tuple(ws['counter'].tolist())
    frames += 1
    if not seen or seen[0] is None or tuple(seen[0]) != dev_counts or out['n_rows'] != dev_counts[0]:
        bad += 1
    f = gs[int(rng.integers(len(gs)))]
    seen.clear()
    color, n_inst, plist, ranges = f()
    frames += 1
    if seen and (seen[0] is None or seen[0][0] != n_inst):
        bad += 1
print(f'{frames} frames, mismatches / silent mailboxes: {bad}, mailbox still in use: {_lib.HostMailbox.for_device(torch.device("cuda", 0)) is not None}')
