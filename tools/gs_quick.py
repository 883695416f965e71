#!/usr/bin/env python3
"""tools/gs_quick.py [N ...] -- 3DGS rasterizer frame times and the per-kernel stage times (library stage timer) at the given Gaussian counts."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import bench

dev = torch.device('cuda', 0)
for n in [int(a) for a in sys.argv[1:]] or [1_000_000]:
    gs = bench.build_gs_scene(dev, n)
    r = bench.time_gs(gs, reps=10)
    st = r['stage_ms']
    fwd = [k for k in st if not k.endswith('_bw') and k != 'k_zero_grads']
    print(n, 'fwd', r['ms_fwd'], 'fwd+bwd', r['ms_fwd_bwd'], 'trials', r['trials_ms_fwd'], r['trials_ms_fwd_bwd'])
    print('   ', ' '.join(f'{k}={v[0] * 1e3:.1f}us/{v[1]}' for k, v in st.items()))
    print('    forward kernels sum %.1f us, binning %.1f us' % (sum(st[k][0] for k in fwd) * 1e3,
          sum(st[k][0] for k in fwd if k.startswith(('k_depth', 'k_radix', 'k_span', 'k_item', 'k_scan', 'k_bin'))) * 1e3))
    del gs
    torch.cuda.empty_cache()
