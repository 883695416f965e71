#!/usr/bin/env python3
"""tools/bench_gs.py -- developer micro-benchmark of the 3DGS rasterizer on the synthetic 1 M-Gaussian scene."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda', 0)
gs = bench.build_gs_scene(dev, n)
res = bench.time_gs(gs, reps)
print(res)
