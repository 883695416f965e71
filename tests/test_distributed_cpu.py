"""world_size-2 tests of the data-parallel host logic on CPU (gloo, 127.0.0.1): ray / tile sharding and the bucketed gradient
reduction reproduce the single-process result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from nerficg_amd import parallel
    parallel.init_distributed('gloo')
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    ctx = mp.get_context('spawn')
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, fn, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        return dict(ret)


def _t_shard_rays(rank, world):
    from nerficg_amd import parallel
    torch.manual_seed(0)  # same seeded permutation on every rank (src/Optim/Samplers/utils.py:8-34)
    perm = torch.randperm(1001)
    batch = perm[:333]
    mine = parallel.shard_ray_ids(batch)
    return mine.tolist(), batch.tolist(), parallel.shard_range(63), parallel.shard_range(10000)


def test_ray_and_tile_sharding_partition_the_batch():
    out = _run(_t_shard_rays)
    a, b = out[0], out[1]
    assert a[1] == b[1]
    assert sorted(a[0] + b[0]) == sorted(a[1]) and not set(a[0]) & set(b[0])
    assert a[2] == (0, 32) and b[2] == (32, 63)
    assert a[3] == (0, 5000) and b[3] == (5000, 10000)


def _t_grad_reduce(rank, world):
    from nerficg_amd import parallel
    torch.manual_seed(1)
    model = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    parallel.broadcast_parameters(model.parameters())
    x = torch.randn(64, 7)
    y = torch.randn(64, 3)
    ids = parallel.shard_ray_ids(torch.arange(64))
    loss = torch.nn.functional.mse_loss(model(x[ids]), y[ids])
    loss.backward()
    parallel.allreduce_gradients(model.parameters(), average=True, bucket_bytes=64)  # tiny buckets: exercises several collectives
    got = [p.grad.clone() for p in model.parameters()]
    model.zero_grad()
    torch.nn.functional.mse_loss(model(x), y).backward()
    ref = [p.grad.clone() for p in model.parameters()]
    return max(float((g - r).abs().max()) for g, r in zip(got, ref))


def test_bucketed_gradient_allreduce_equals_full_batch_gradient():
    out = _run(_t_grad_reduce)
    assert out[0] < 1e-6 and out[1] < 1e-6


def _t_flat_and_gather(rank, world):
    from nerficg_amd import parallel
    buf = torch.arange(11, dtype=torch.float32) * (rank + 1)   # length not divisible by world: padded path
    parallel.allreduce_flat(buf, average=False)
    local = torch.full((3 + rank, 2), float(rank))
    full = parallel.all_gather_pixels(local, [3, 4])
    return buf.tolist(), full.tolist()


def test_flat_allreduce_and_ragged_pixel_gather():
    out = _run(_t_flat_and_gather)
    exp = [i * 3.0 for i in range(11)]
    assert out[0][0] == exp and out[1][0] == exp
    assert out[0][1] == [[0.0, 0.0]] * 3 + [[1.0, 1.0]] * 4 == out[1][1]
