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


def _t_frame_shards(rank, world):
    """One frame cut into contiguous tile shards (BASELINE configs[3] decomposition): after gather_image_shards every rank holds every pixel."""
    from nerficg_amd import parallel
    w, h = 52, 29                      # 7 x 4 tiles of 8 x 8 pixels, the last column / row of tiles cut by the image border
    nt = 7 * 4
    truth = {'rgb': torch.arange(w * h * 3, dtype=torch.float32).reshape(w * h, 3), 'alpha': torch.arange(w * h, dtype=torch.float32) * 0.5,
             'depth': torch.arange(w * h, dtype=torch.float32) * 0.25}
    b, e = parallel.shard_range(nt)
    mine = parallel.tile_pixel_indices(w, h, b, e - b)
    out = {k: torch.full_like(v, -1.0) for k, v in truth.items()}
    for k in out:
        out[k][mine] = truth[k][mine]   # what render_image_fused(tile_begin=b, n_tiles=e - b, out=out) leaves
    cache = {}
    parallel.gather_image_shards(out, w, h, nt, cache=cache)
    again = parallel.gather_image_shards(out, w, h, nt, cache=cache)   # cached index lists
    return all(bool(torch.equal(again[k], truth[k])) for k in truth), cache['counts'], int(mine.numel())


def test_frame_shards_gather_to_the_whole_frame():
    out = _run(_t_frame_shards)
    assert out[0][0] and out[1][0]
    assert out[0][1] == out[1][1] and sum(out[0][1]) == 52 * 29 and out[0][2] == out[0][1][0] and out[1][2] == out[1][1][1]


def _t_sparse_reduce(rank, world):
    """View-parallel 3DGS: each rank 'sees' another subset of 500 Gaussians; sparse reduction == dense all-reduce, rows seen by nobody stay 0."""
    from types import SimpleNamespace
    from nerficg_amd import parallel
    g = torch.Generator().manual_seed(100 + rank)
    P = 500
    visible = torch.rand(P, generator=g) < (0.3 if rank == 0 else 0.2)
    shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 3), (P, 4)]
    params = []
    for s in shapes:
        p = torch.nn.Parameter(torch.zeros(s))
        p.grad = torch.randn(s, generator=g) * visible.view(-1, *([1] * (len(s) - 1)))
        params.append(p)
    dense = [p.grad.clone() for p in params]
    for d in dense:
        dist.all_reduce(d)
        d.div_(world)
    n_union = parallel.sparse_allreduce_gradients(params, visible, average=True)
    err = max(float((p.grad - d).abs().max()) for p, d in zip(params, dense))
    u = visible.to(torch.uint8)
    dist.all_reduce(u, op=dist.ReduceOp.MAX)
    untouched = max(float(p.grad[u == 0].abs().max()) for p in params)
    stats = SimpleNamespace(densification_gradient_accum=torch.full((P, 1), float(rank + 1)), n_observations=torch.full((P, 1), rank + 2, dtype=torch.int32))
    parallel.allreduce_densification_stats(stats)
    noise = parallel.synchronized_noise(7, seed=5, device='cpu')
    return err, untouched, n_union, int(u.sum()), float(stats.densification_gradient_accum[3, 0]), int(stats.n_observations[3, 0]), noise.tolist()


def test_sparse_gaussian_gradient_reduction_equals_dense_allreduce():
    out = _run(_t_sparse_reduce)
    for r in (0, 1):
        err, untouched, n_union, n_mask, acc, nobs, _ = out[r]
        assert err < 1e-6 and untouched == 0.0 and n_union == n_mask and 0 < n_union < 500
        assert acc == 3.0 and nobs == 5
    assert out[0][6] == out[1][6]  # identical split noise on every rank


def _t_scalars_and_scaler(rank, world):
    """SURVEY 8(e)'s auxiliary reductions: rm_samples summed, found-inf OR-ed in one collective; the GradScaler built on it skips the SAME
    step on every rank when only one rank overflowed, and a rank without gradients still enters the sparse reduction."""
    from nerficg_amd import parallel
    sums, flags = parallel.allreduce_scalars([torch.tensor(1000 + rank), torch.tensor(2.5)], [torch.tensor(float(rank == 1)), torch.tensor(0.0)])
    out = {'sums': [float(v) for v in sums], 'flags': [float(v) for v in flags]}
    out['rays'] = parallel.rays_per_batch_update(4096, 262144, float(sums[0]) * 16, 16)   # 1000.5 samples per rank and iteration from 4096 rays
    # one parameter, three iterations; rank 1's loss overflows in iteration 1 only
    p = torch.nn.Parameter(torch.ones(4))
    opt = torch.optim.SGD([p], lr=0.5)
    scaler = parallel.DataParallelGradScaler('cpu', init_scale=128.0, growth_interval=10 ** 6)
    trace = []
    for it in range(3):
        x = torch.full((4,), float('inf') if (it == 1 and rank == 1) else 1.0 + rank)
        loss = (p * x).sum()
        scaler.scale(loss).backward()
        parallel.allreduce_gradients([p], average=True) if not (it == 1) else None   # iteration 1: NO gradient collective, only the flag can tell rank 0
        scaler.piggyback = [torch.tensor(100.0 * (rank + 1))]
        scaler.step(opt)
        scaler.update()
        opt.zero_grad()
        trace.append((p.detach().tolist(), float(scaler.get_scale()), float(scaler.reduced[0])))
    out['trace'] = trace
    # a rank whose optimizer holds NO gradient this step still enters the scaler's collective (round-3 advisor finding: it used to skip it and the
    # other rank waited for ever) and learns about the other rank's overflow
    q2 = torch.nn.Parameter(torch.ones(2))
    opt2 = torch.optim.SGD([q2], lr=0.5)
    scaler2 = parallel.DataParallelGradScaler('cpu', init_scale=8.0, growth_interval=10 ** 6)
    lone = []
    for it in range(2):
        if rank == 0:
            scaler2.scale((q2 * (float('inf') if it == 1 else 1.0)).sum()).backward()
        else:
            scaler2.scale(torch.zeros(()))   # initialises the scale; no backward: q2.grad stays None on this rank
        scaler2.piggyback = [torch.tensor(1.0 + rank)]
        scaler2.step(opt2)
        scaler2.update()
        opt2.zero_grad()
        lone.append((float(scaler2.get_scale()), float(scaler2.reduced[0])))
    out['lone'] = lone
    # sparse reduction with a rank that holds no gradient at all (its view saw nothing)
    P = 50
    q = torch.nn.Parameter(torch.zeros(P, 3))
    visible = torch.zeros(P, dtype=torch.bool)
    if rank == 0:
        visible[::5] = True
        q.grad = torch.ones(P, 3) * visible[:, None]
    n_union = parallel.sparse_allreduce_gradients([q], visible, average=False)
    out['sparse'] = (n_union, float(q.grad.sum()))
    return out


def test_scalar_reductions_and_the_data_parallel_grad_scaler():
    out = _run(_t_scalars_and_scaler)
    for r in (0, 1):
        assert out[r]['sums'] == [2001.0, 5.0] and out[r]['flags'] == [1.0, 0.0]
        assert out[r]['sparse'] == (10, 30.0)
    assert out[0]['rays'] == out[1]['rays'] == 262144   # 4096 * 262144 / 1000.5 is far above the cap
    assert out[0]['lone'] == out[1]['lone'] == [(8.0, 3.0), (4.0, 3.0)]   # rank 1 never had a gradient; both halved the scale on rank 0's overflow
    t0, t1 = out[0]['trace'], out[1]['trace']
    assert t0 == t1                                         # identical parameters, scale and global sample count after every iteration
    assert t0[0][0] == [1.0 - 0.5 * 1.5] * 4 and t0[0][1] == 128.0 and t0[0][2] == 300.0
    assert t0[1][0] == t0[0][0] and t0[1][1] == 64.0       # iteration 1: rank 1 overflowed -> BOTH ranks skipped the step and halved the scale
    assert t0[2][0] == [t0[1][0][0] - 0.5 * 1.5] * 4


def test_rays_per_batch_update_is_the_reference_rule_for_one_rank():
    from nerficg_amd import parallel
    # Trainer.py:73-75: measured /= interval; rays = min(next_multiple(rays * target / measured, 256), target)
    assert parallel.rays_per_batch_update(4096, 262144, 16 * 300_000.0, 16, world=1) == 3584   # 4096 * 262144 / 300000 = 3579.1 -> 3584
    assert parallel.rays_per_batch_update(4096, 262144, 16 * 300_000.0 * 8, 16, world=8) == 3584
    assert parallel.rays_per_batch_update(4096, 262144, 16 * 10.0, 16, world=1) == 262144
    assert parallel.rays_per_batch_update(4096, 262144, 0.0, 16, world=1) == 4096


def test_rank_batch_orders_tile_the_global_batches():
    """parallel.rank_batch_order (host logic of the data-parallel fused trainer): for every iteration the ranks' slices are disjoint and their union, in
    rank order, IS the global batch order[k W n : (k + 1) W n]; a tail that does not fill a global batch is dropped."""
    import torch
    from nerficg_amd import parallel
    order = torch.randperm(10_007, generator=torch.Generator().manual_seed(3))
    for world, n in ((2, 512), (8, 100), (1, 1000)):
        per_rank = [parallel.rank_batch_order(order, n, r, world) for r in range(world)]
        n_iter = 10_007 // (world * n)
        assert all(p.numel() == n_iter * n for p in per_rank)
        for k in range(n_iter):
            union = torch.cat([p[k * n:(k + 1) * n] for p in per_rank])
            assert torch.equal(union, order[k * world * n:(k + 1) * world * n])
