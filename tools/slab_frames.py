#!/usr/bin/env python3
"""tools/slab_frames.py -- a short training of bench.py's analytic sphere, then slab-order frames only (for tools/kseq.sh)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
bench.trained_scene_leg.__defaults__ = (800, 2)
import nerficg_amd.instant_ngp as ingp
keep = {}
orig = ingp.InstantNGPRenderer.render_image_fused
def spy(self, cam, pose, *a, **k):
    keep['r'], keep['cam'], keep['pose'] = self, cam, pose
    return orig(self, cam, pose, *a, **k)
ingp.InstantNGPRenderer.render_image_fused = spy
bench.trained_scene_leg(torch.device('cuda', 0))
ingp.InstantNGPRenderer.render_image_fused = orig
for _ in range(4):
    keep['r'].render_image_fused(keep['cam'], keep['pose'], early_termination=True)
torch.cuda.synchronize()
