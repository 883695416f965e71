#!/usr/bin/env python3
"""tools/trained_frames.py [ITERS] [POSES] -- bench.py's trained-scene leg on its own (for tools/kseq.sh: the kernels of a slab-order frame)."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
poses = int(sys.argv[2]) if len(sys.argv) > 2 else 6
r = bench.trained_scene_leg(torch.device('cuda', 0), iters=iters, n_poses=poses)
print(json.dumps({k: r[k] for k in ('psnr_800x800_dB', 'slab_order_auto', 'single_pass')}))
