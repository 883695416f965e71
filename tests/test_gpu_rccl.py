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

# the three in-place collectives of the sharded optimizer step (round 6)
buf = torch.randn(12_196_240, device=dev, generator=g); ref = buf.clone()
shard = parallel.reduce_scatter_sum_(buf)
half = buf.half(); href = half.clone()
parallel.all_gather_(half)
small = torch.randn(10_272, device=dev, generator=g); sref = small.clone()
parallel.allreduce_sum_(small)
torch.cuda.synchronize()
out['inplace'] = [bool(torch.equal(buf, ref)), shard.data_ptr() == buf.data_ptr() and shard.numel() == buf.numel(), bool(torch.equal(half, href)), bool(torch.equal(small, sref))]

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
    assert out['flat_11'] and out['flat_12206480'] and out['buckets'] and out['gather'] and out['broadcast'] and all(out['inplace'])
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
for mode in ('plain', 'dp', 'dp_sharded', 'dp_sharded_f16'):
    model, renderer, _ = _train_pair(seed=3)
    it, opt, _ = _fused(model, renderer, cam, pool, 1024, 200_000, prefetch=True, graph=False, order=order, seed=21, fused_step=False, data_parallel=(mode != 'plain'),
                        sharded=mode.startswith('dp_sharded') if mode != 'plain' else None, dp_timing=(mode == 'dp_sharded'),
                        wire_dtype=torch.float16 if mode == 'dp_sharded_f16' else torch.float32)
    losses = [float(it()['loss']) for _ in range(6)]
    it.gather_state()
    res[mode] = dict(losses=losses, cursor=int(it.cursor), dp=bool(it.data_parallel), sharded=bool(it.sharded), times=it.dp_times(),
                     saturated=int(it.wire_saturated) if it.wire is not None else None, wire=it.layout.wire_bytes(2 if it.wire is not None else 4)['reduce_scatter'],
                     checksum=float(model.encoding_xyz.params.double().abs().sum()), step=int(opt.effective_step(opt.param_groups[0])), prefetch_at=it.prefetch_at)
print(json.dumps(res))
'''


def test_fused_trainer_data_parallel_path_on_a_one_rank_group():
    """nerficg_amd.ngp_trainer with data_parallel=True: per-rank order, one flat gradient buffer, and between the backward pass and the step either
    parallel.allreduce_flat (sharded=False) or the sharded step of round 6 (small all-reduce beside the grid backward, in-place reduce-scatter, Adam on the shard,
    in-place all-gather of the fp16 table, the next batch marched beside the collective) -- executed on a one-rank RCCL group (a sum over one rank is the identity):
    the same batches, losses and (to the atomics' run-to-run noise) parameters as the plain trainer."""
    env = {**os.environ, 'NRC_ROOT': str(ROOT), 'MASTER_PORT': '29517'}
    r = subprocess.run([sys.executable, '-c', TRAINER_CHILD], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.loads(next(l for l in reversed(r.stdout.strip().splitlines()) if l.startswith('{')))     # (RCCL prints its library path behind it)
    assert res['dp']['dp'] and not res['plain']['dp'] and res['dp']['cursor'] == res['plain']['cursor'] == res['dp_sharded']['cursor'] == 7 * 1024     # six iterations + the batch marched ahead
    assert res['dp_sharded']['sharded'] and not res['dp']['sharded'] and res['dp_sharded']['prefetch_at'] == 'collective' and res['dp']['prefetch_at'] == 'forward'
    assert res['dp_sharded']['step'] == res['dp']['step'] == res['plain']['step'] == 6
    t = res['dp_sharded']['times']
    assert t['iterations'] == 6 and all(t[k] >= 0 for k in ('reduce_scatter_ms', 'adam_ms', 'all_gather_ms', 'exposed_ms'))
    assert res['dp_sharded_f16']['sharded'] and res['dp_sharded_f16']['saturated'] == 0 and res['dp_sharded_f16']['step'] == 6
    assert res['dp_sharded_f16']['wire'] == 0 and res['dp_sharded']['wire'] == 0      # one rank: (world - 1) / world = 0 of the bytes leave the GPU
    for mode in ('dp', 'dp_sharded', 'dp_sharded_f16'):
        for a, b in zip(res['plain']['losses'], res[mode]['losses']):
            assert abs(a - b) <= 2e-3 * abs(a), (mode, res['plain']['losses'], res[mode]['losses'])
        # fp16 wire: a gradient below fp16's smallest step reaches Adam as zero, and Adam (eps = 1e-15) turns any non-zero gradient into a full-size step -- entries
        # that receive only such gradients move in the f32 run and stay put in the fp16 one (tiny-cuda-nn's own fp16 gradients behave the same way): the losses
        # agree (above), the parameter sums to a few per cent
        tol = 0.1 if mode == 'dp_sharded_f16' else 1e-4
        assert abs(res['plain']['checksum'] - res[mode]['checksum']) <= tol * res['plain']['checksum']


TWO_RANK_CHILD = r'''
import copy, hashlib, json, os, sys
sys.path.insert(0, os.environ['NRC_ROOT'])
import torch
from nerficg_amd import parallel
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
rank, world = parallel.init_distributed('gloo')
from tests.test_gpu_ngp_trainer import _fused, _pool
from tests.test_gpu_graphs import _train_pair
from tests.noise import mismatch_fraction
cam, pool = _pool(size=128)      # 16 384 rays: eight global batches of 2 x 1 024
order = torch.randperm(pool['origin'].shape[0], generator=torch.Generator().manual_seed(4)).to(dev)
POISON_AT = 2


def run(sharded, iterations=5, resume=None):
    model, renderer, _ = _train_pair(seed=3)
    it, opt, scaler = _fused(model, renderer, cam, pool, 1024, 200_000, graph=False, order=order, seed=21, fused_step=False, data_parallel=True, sharded=sharded)
    if resume is not None:        # a checkpoint written behind gather_state(): parameters, moments, step counter, scale
        model.load_state_dict(resume['model']); opt.load_state_dict(resume['optimizer']); scaler.load_state_dict(resume['scaler'])
        it.cursor.fill_(resume['cursor']); it._cursor_host = resume['cursor']; it.rng.copy_(resume['rng'])
    trace = []
    for k in range(iterations):
        saved = None
        if resume is None and k == POISON_AT and rank == 1:      # ONE rank's targets are inf in ONE iteration: its loss gradient overflows, rank 0's does not
            saved = it.pool['rgb'].clone(); it.pool['rgb'].fill_(float('inf'))
        out = it()
        torch.cuda.synchronize()
        if saved is not None:
            it.pool['rgb'].copy_(saved)
        trace.append((float(scaler.get_scale()), int(opt.effective_step(opt.param_groups[0]))))
    stale = bool(it._master_stale)
    it.gather_state()
    torch.cuda.synchronize()
    params = [p.detach().clone() for p in model.parameters()]
    moments = [opt.state[p][k].clone() for p in model.parameters() for k in ('exp_avg', 'exp_avg_sq')]
    ckpt = dict(model={k: v.clone() for k, v in model.state_dict().items()}, optimizer=copy.deepcopy(opt.state_dict()), scaler=scaler.state_dict(), cursor=int(it.cursor), rng=it.rng.clone())      # (state_dict() hands out the LIVE moment tensors)
    return dict(it=it, trace=trace, params=params, moments=moments, ckpt=ckpt, stale=stale, half=model.encoding_xyz._half_params().clone(), wire=it.layout.wire_bytes())


def agree(tensors):       # every rank holds the same bits?
    h = hashlib.sha1()
    for t in tensors:
        h.update(t.detach().cpu().numpy().tobytes())
    mine = torch.tensor(list(h.digest()), dtype=torch.int64)
    both = [torch.zeros_like(mine) for _ in range(world)]
    torch.distributed.all_gather(both, mine)
    return all(bool(torch.equal(b, both[0])) for b in both)


rep, rep2, sh = run(False), run(False), run(True)
res = dict(rank=rank, world=world, trace_replicated=rep['trace'], trace_sharded=sh['trace'], stale=[rep['stale'], sh['stale']], wire=sh['wire'],
           replicas_agree=[agree(rep['params'] + rep['moments']), agree(sh['params'] + sh['moments'] + [sh['half']])],
           half_matches_master=bool(torch.equal(sh['half'], sh['params'][0].half())))
# sharded against replicated, with the replicated step's own run-to-run spread (float atomics in the dense levels' / MLP weights' gradients) as the yardstick
res['mismatch'] = [[mismatch_fraction(r, t, 1e-5, 1e-3), mismatch_fraction(r, r2, 1e-5, 1e-3),
                    float(((r - t).abs() / (r.abs() + 1e-2)).flatten()[:: max(1, r.numel() // 2 ** 20)].quantile(0.5))] for r, r2, t in zip(rep['params'], rep2['params'], sh['params'])]
# checkpoint round trip: the state gathered after 5 sharded iterations, loaded into a NEW model / optimizer / trainer, continues like the live trainer does
live = [float(sh['it']()['loss']) for _ in range(2)]
sh['it'].gather_state(); torch.cuda.synchronize()
live_params = [p.detach().clone() for p in sh['it'].model.parameters()]
again = run(True, iterations=2, resume=sh['ckpt'])
res['resume'] = dict(trace=again['trace'], mismatch=[mismatch_fraction(a, b, 1e-5, 1e-3) for a, b in zip(live_params, again['params'])],
                     median=[float(((a - b).abs() / (a.abs() + 1e-2)).flatten()[:: max(1, a.numel() // 2 ** 20)].quantile(0.5)) for a, b in zip(live_params, again['params'])])
torch.distributed.barrier()
torch.distributed.destroy_process_group()
print('RESULT ' + json.dumps(res))
'''


def test_two_ranks_sharing_the_gpu_sharded_step_follows_the_replicated_one(tmp_path):
    """Two gloo ranks on the ONE GPU of the box (device tensors staged through the host: the code path, not a speed), five iterations of the fused trainer
    with rank 1's targets poisoned in iteration 2: the sharded step (flag in the small all-reduce, in-place reduce-scatter, Adam on the rank's shard, all-gather
    of the fp16 table) skips the same step, moves the scale the same way and leaves the parameters of the replicated step to within the replicated step's own
    run-to-run spread; both ranks hold bit-identical state after gather_state(); a checkpoint written then resumes like the live trainer."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    script = tmp_path / 'two_rank_child.py'
    script.write_text(TWO_RANK_CHILD)
    procs = []
    for r in range(2):
        env = dict(os.environ, NRC_ROOT=str(ROOT), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(r), WORLD_SIZE='2', LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for pr in procs:
        try:
            so, se = pr.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert pr.returncode == 0, so[-2000:] + se[-4000:]
        outs.append(json.loads([ln for ln in so.splitlines() if ln.startswith('RESULT ')][-1][len('RESULT '):]))
    for o in outs:
        assert o['world'] == 2
        assert o['trace_sharded'] == o['trace_replicated'] == [[128.0, 1], [128.0, 2], [64.0, 2], [64.0, 3], [64.0, 4]]      # the poisoned step skipped on BOTH ranks
        assert o['stale'] == [False, True] and all(o['replicas_agree']) and o['half_matches_master']
        for got, noise, median in o['mismatch']:
            assert got <= 4 * noise + 2e-3 and median < 1e-4, o['mismatch']
        assert o['resume']['trace'] == [[64.0, 5], [64.0, 6]] and all(m <= 0.02 for m in o['resume']['mismatch']) and all(m < 1e-4 for m in o['resume']['median']), o['resume']
        assert o['wire']['reduce_scatter'] == 12_196_240 * 4 // 2 and o['wire']['all_gather'] == 12_196_240 * 2 // 2
    assert outs[0]['trace_sharded'] == outs[1]['trace_sharded']
