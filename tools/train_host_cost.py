#!/usr/bin/env python3
"""tools/train_host_cost.py -- cProfile of the op-by-op InstantNGP training iteration (tools/bench_train.py's step): where the host time goes."""
import sys, time, cProfile, pstats, runpy
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.argv = [sys.argv[0], '2200', '5']
ns = runpy.run_path(str(Path(__file__).with_name('bench_train.py')))
import torch
step = ns['step']
for i in range(10):
    step(50 + i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(200):
    step(100 + i)
torch.cuda.synchronize()
print('iteration %.1f us' % ((time.perf_counter() - t0) / 200 * 1e6))
pr = cProfile.Profile(); pr.enable()
for i in range(200):
    step(400 + i)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
st.sort_stats('cumtime').print_stats(30)
