"""Experiment: the 3DGS frame with SH degree 1 (M = 4) -- to compare library builds that differ only in the LDS they reserve per workgroup."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench
from nerficg_amd.diff_gaussian_rasterization import GaussianRasterizer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device('cuda', 0)
gs = bench.build_gs_scene(dev, n)
gs['tensors']['shs'] = gs['tensors']['shs'][:, :4].contiguous()
gs['rast'] = GaussianRasterizer(gs['rast'].raster_settings._replace(sh_degree=1))
print(bench.time_gs(gs, 10))
