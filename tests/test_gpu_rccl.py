"""GPU: the RCCL branch of nerficg_amd.parallel executed on the one GPU of the box.

The CPU suite runs the data-parallel host logic over gloo (tests/test_distributed_cpu.py), which takes the `all_reduce` branch of
`allreduce_flat`; the branch RCCL takes -- `init_process_group('nccl', device_id=...)`, `reduce_scatter_tensor` + `all_gather_into_tensor` on a
padded flat buffer, the packed f64 scalar all-reduce, the byte-mask MAX all-reduce and the packed-row reduction of the view-parallel 3DGS
gradients, `all_gather` of ragged pixel blocks, `broadcast` -- would otherwise run for the first time on the driver's 8-GPU node.  A process
group of ONE rank with `parallel.single_rank_collectives(True)` issues every one of those collectives on this ROCm / RCCL build; each is then a
sum / gather over one contribution, so the expected values are the inputs themselves.  Runs in a child process: the default process group
must not leak into the other tests.  (The reference has no collective at all: src/Methods/Base/Renderer.py:24-33 is a DataParallel no-op.)
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]

CHILD = r'''
import json, os, sys
sys.path.insert(0, os.environ['NRC_ROOT'])
import torch
import torch.distributed as dist
from nerficg_amd import parallel

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
rank, world = parallel.init_distributed('nccl', dev, single_rank_group=True)
assert dist.is_initialized() and dist.get_backend() == 'nccl' and parallel.world_info() == (0, 1)
parallel.single_rank_collectives(True)
assert parallel._has_tensor_collectives()
out = {}
g = torch.Generator(device=dev).manual_seed(0)

# reduce-scatter + all-gather of a flat buffer: a length that divides the world and the InstantNGP payload size (12.2 M f32)
for n in (11, 12_206_480):
    buf = torch.randn(n, device=dev, generator=g)
    ref = buf.clone()
    parallel.allreduce_flat(buf, average=True)
    torch.cuda.synchronize()
    out[f'flat_{n}'] = bool(torch.equal(buf, ref))

# bucketed gradients: several buckets, several tensors per bucket
params = [torch.nn.Parameter(torch.zeros(s, device=dev)) for s in ((1000, 3), (17,), (64, 64), (5,))]
for p in params:
    p.grad = torch.randn(p.shape, device=dev, generator=g)
refs = [p.grad.clone() for p in params]
parallel.allreduce_gradients(params, average=True, bucket_bytes=8192)
out['buckets'] = all(bool(torch.equal(p.grad, r)) for p, r in zip(params, refs))

# per-iteration scalars and the data-parallel GradScaler on top of them
sums, flags = parallel.allreduce_scalars([torch.tensor(123456, device=dev), torch.tensor(2.5, device=dev)], [torch.tensor(1.0, device=dev), torch.tensor(0.0, device=dev)])
out['scalars'] = [float(v) for v in sums] + [float(v) for v in flags]
p = torch.nn.Parameter(torch.ones(4, device=dev))
opt = torch.optim.SGD([p], lr=0.5)
scaler = parallel.DataParallelGradScaler('cuda', init_scale=128.0, growth_interval=10 ** 6)
trace = []
for it in range(3):
    x = torch.full((4,), float('inf') if it == 1 else 2.0, device=dev)
    scaler.scale((p * x).sum()).backward()
    scaler.piggyback = [torch.tensor(100.0, device=dev)]
    scaler.step(opt); scaler.update(); opt.zero_grad()
    trace.append((p.detach().tolist(), float(scaler.get_scale()), float(scaler.reduced[0])))
out['scaler'] = trace

# view-parallel 3DGS: byte-mask MAX all-reduce, packed rows, reduce-scatter + all-gather
P = 5000
visible = torch.rand(P, device=dev, generator=g) < 0.3
shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 3), (P, 4)]
gp = []
for s in shapes:
    q = torch.nn.Parameter(torch.zeros(s, device=dev))
    q.grad = torch.randn(s, device=dev, generator=g) * visible.view(-1, *([1] * (len(s) - 1)))
    gp.append(q)
refs = [q.grad.clone() for q in gp]
n_union = parallel.sparse_allreduce_gradients(gp, visible, average=True)
out['sparse'] = [int(n_union), int(visible.sum()), all(bool(torch.equal(q.grad, r)) for q, r in zip(gp, refs))]

# ragged pixel gather (SURVEY 8e: the (N / world, 5) all-gather of a sharded frame) and the parameter broadcast
local = torch.rand(1234, 5, device=dev, generator=g)
full = parallel.all_gather_pixels(local, [1234])
out['gather'] = bool(torch.equal(full, local))
w = torch.rand(100, device=dev, generator=g); w0 = w.clone()
parallel.broadcast_parameters([w])
out['broadcast'] = bool(torch.equal(w, w0))

class S: pass
st = S(); st.densification_gradient_accum = torch.full((P, 1), 2.0, device=dev); st.n_observations = torch.full((P, 1), 3, dtype=torch.int32, device=dev)
parallel.allreduce_densification_stats(st)
out['stats'] = [float(st.densification_gradient_accum[0, 0]), int(st.n_observations[0, 0])]
torch.cuda.synchronize()
dist.destroy_process_group()
print('RESULT ' + json.dumps(out))
'''


def test_every_rccl_collective_of_the_data_parallel_path_runs_on_one_gpu(tmp_path):
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, NRC_ROOT=str(ROOT), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    script = tmp_path / 'rccl_child.py'
    script.write_text(CHILD)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT ')][-1]
    out = json.loads(line[len('RESULT '):])
    assert out['flat_11'] and out['flat_12206480'] and out['buckets'] and out['gather'] and out['broadcast']
    assert out['scalars'] == [123456.0, 2.5, 1.0, 0.0]
    t = out['scaler']
    assert t[0][0] == [0.0] * 4 and t[0][1] == 128.0 and t[0][2] == 100.0
    assert t[1][0] == t[0][0] and t[1][1] == 64.0          # the overflowing step was skipped, the scale halved
    assert t[2][0] == [-1.0] * 4
    n_union, n_visible, same = out['sparse']
    assert n_union == n_visible > 0 and same
    assert out['stats'] == [2.0, 3]


TRAINER_CHILD = r'''
import json, os, sys
sys.path.insert(0, os.environ['NRC_ROOT'])
import torch
from nerficg_amd import parallel
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
parallel.init_distributed('nccl', dev, single_rank_group=True)
parallel.single_rank_collectives(True)
from tests.test_gpu_ngp_trainer import _fused, _pool
from tests.test_gpu_graphs import _train_pair
cam, pool = _pool(size=96)
order = torch.randperm(pool['origin'].shape[0], generator=torch.Generator().manual_seed(4)).to(dev)
res = {}
for mode in ('plain', 'dp'):
    model, renderer, _ = _train_pair(seed=3)
    it, opt, _ = _fused(model, renderer, cam, pool, 1024, 200_000, prefetch=True, graph=False, order=order, seed=21, fused_step=False, data_parallel=(mode == 'dp'))
    losses = [float(it()['loss']) for _ in range(6)]
    res[mode] = dict(losses=losses, cursor=int(it.cursor), dp=bool(it.data_parallel), checksum=float(model.encoding_xyz.params.double().abs().sum()))
print(json.dumps(res))
'''


def test_fused_trainer_data_parallel_path_on_a_one_rank_group():
    """nerficg_amd.ngp_trainer with data_parallel=True: per-rank order, one flat gradient buffer, parallel.allreduce_flat between the backward pass and the
    step -- executed on a one-rank RCCL group (the average over one rank is the identity): the same batches, losses and (to the atomics' run-to-run noise)
    parameters as the plain trainer."""
    env = {**os.environ, 'NRC_ROOT': str(ROOT), 'MASTER_PORT': '29517'}
    r = subprocess.run([sys.executable, '-c', TRAINER_CHILD], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads(next(l for l in reversed(r.stdout.strip().splitlines()) if l.startswith('{')))     # (RCCL prints its library path behind it)
    assert res['dp']['dp'] and not res['plain']['dp'] and res['dp']['cursor'] == res['plain']['cursor'] == 7 * 1024     # six iterations + the batch marched ahead
    for a, b in zip(res['plain']['losses'], res['dp']['losses']):
        assert abs(a - b) <= 2e-3 * abs(a), (res['plain']['losses'], res['dp']['losses'])
    assert abs(res['plain']['checksum'] - res['dp']['checksum']) <= 1e-4 * res['plain']['checksum']
