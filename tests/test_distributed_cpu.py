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


def _t_sharded_step(rank, world):
    """Five optimizer steps of a 2-rank job, an overflow on ONE rank in the middle, through (A) the replicated step -- allreduce_flat(average) of the whole
    gradient, inf / NaN scan of the result, Adam on everything on every rank -- and (B) parallel.sharded_step -- small all-reduce carrying the ranks' flags,
    in-place reduce-scatter, Adam on the rank's table shard with 1 / (scale x world), in-place all-gather of the fp16 copy.  Adam is oracle/adam_oracle.c in both."""
    import numpy as np
    import oracle
    from nerficg_amd import parallel
    n_c, n_dm, n_t = 96, 64, 4096
    L = parallel.ShardedStepLayout(n_c, n_dm, n_t)
    assert (L.rank, L.world, L.sharded, L.shard, L.shard_begin) == (rank, world, True, n_t // world, rank * n_t // world)
    lr, betas, eps, growth_interval = 1e-2, (0.9, 0.99), 1e-15, 2
    init = np.random.default_rng(0).normal(size=n_c + n_dm + n_t).astype(np.float32)       # [colour | density MLP | table], the same on every rank

    def fresh():
        return dict(p=init.copy(), m=np.zeros_like(init), v=np.zeros_like(init), h=init.astype(np.float16), step=0, scale=128.0, tracker=0, skipped=0)
    A, B = fresh(), fresh()

    def scale_rule(S, overflow):      # torch's amp_update_scale
        if overflow:
            S['scale'] *= 0.5; S['tracker'] = 0; S['skipped'] += 1
        else:
            S['step'] += 1
            S['tracker'] += 1
            if S['tracker'] == growth_interval:
                S['scale'] *= 2.0; S['tracker'] = 0

    def adam(S, lo, hi, g, grad_scale):
        S['p'][lo:hi], S['m'][lo:hi], S['v'][lo:hi] = oracle.adam_step(S['p'][lo:hi], g, S['m'][lo:hi], S['v'][lo:hi], S['step'], lr, betas, eps, grad_scale=grad_scale)
        S['h'][lo:hi] = S['p'][lo:hi].astype(np.float16)

    for it in range(5):
        g = (np.random.default_rng(100 * it + rank).normal(size=n_c + n_dm + n_t) * A['scale']).astype(np.float32)   # this rank's scaled gradient
        assert A['scale'] == B['scale']
        flag = 0.0
        if it == 2 and rank == 1:
            g[n_c + n_dm + 7] = np.inf       # in rank 0's table shard, produced on rank 1: only the flag (B) / the sum (A) can tell rank 0
            flag = 1.0
        # (A) replicated
        buf = torch.from_numpy(g.copy())
        parallel.allreduce_flat(buf, average=True)
        ga = buf.numpy()
        overflow = not np.isfinite(ga).all()
        scale_a = A['scale']
        scale_rule(A, overflow)
        if not overflow:
            adam(A, 0, ga.size, ga, scale_a)
        # (B) sharded: the layout's buffer, flag in aux[0]
        grads = torch.zeros(L.total)
        grads[:n_c] = torch.from_numpy(g[:n_c]); grads[L.off_aux] = flag
        grads[L.off_density:] = torch.from_numpy(g[n_c:])
        half_table = torch.from_numpy(B['h'][n_c + n_dm:].copy())
        ctx = {}

        def settle(aux):
            ctx['overflow'] = float(aux[0]) != 0.0
            ctx['grad_scale'] = B['scale'] * world          # 1 / (scale x world) on a SUM = 1 / scale on the mean (world a power of two: the same bits)
            scale_rule(B, ctx['overflow'])

        def adam_small():
            if not ctx['overflow']:
                adam(B, 0, n_c, grads[:n_c].numpy(), ctx['grad_scale'])
                adam(B, n_c, n_c + n_dm, grads[L.off_density:L.off_table].numpy(), ctx['grad_scale'])

        def adam_table(begin, count):
            if not ctx['overflow']:
                lo = n_c + n_dm + begin
                adam(B, lo, lo + count, grads[L.off_table + begin:L.off_table + begin + count].numpy(), ctx['grad_scale'])
                half_table[begin:begin + count] = torch.from_numpy(B['h'][lo:lo + count].copy())
        parallel.sharded_step(L, grads, half_table, settle, adam_small, adam_table)
        B['h'][n_c + n_dm:] = half_table.numpy()
        assert ctx['overflow'] == overflow == (it == 2)
    # the fp16 copy is whole on every rank after every step; master and moments of the table are gathered on demand (FusedTrainingIteration.gather_state)
    same_half = bool(np.array_equal(A['h'], B['h'])) if world == 2 else bool(np.allclose(A['h'].astype(np.float32), B['h'].astype(np.float32), rtol=2e-3, atol=1e-6))
    others = np.ones(n_t, dtype=bool)
    others[rank * L.shard:(rank + 1) * L.shard] = False
    stale_before = bool(np.array_equal(B['p'][n_c + n_dm:][others], init[n_c + n_dm:][others]))      # the other ranks' shards of the master were never touched here
    for k in ('p', 'm', 'v'):
        t = torch.from_numpy(B[k][n_c + n_dm:].copy())
        parallel.all_gather_(t)
        B[k][n_c + n_dm:] = t.numpy()
    # two ranks: a + b has one order, the two statements agree to the bit; four ranks: gloo's all-reduce and reduce-scatter add the four contributions in different
    # orders, so the sums -- and with them the moments -- may differ in the last place
    if world == 2:
        same = all(bool(np.array_equal(A[k], B[k])) for k in ('p', 'm', 'v', 'h'))
    else:
        same = all(bool(np.allclose(A[k].astype(np.float64), B[k].astype(np.float64), rtol=2e-5, atol=1e-7)) for k in ('p', 'm', 'v', 'h'))
    wire = L.wire_bytes()
    import hashlib
    digest = hashlib.sha256(b''.join(B[k].tobytes() for k in ('p', 'm', 'v', 'h'))).hexdigest()    # what the replicas hold must agree to the bit, whatever the world size
    return same_half, stale_before, same, (A['step'], A['scale'], A['skipped']), (B['step'], B['scale'], B['skipped']), wire, digest


@pytest.mark.parametrize('world', [2, 4, 8])
def test_sharded_optimizer_step_is_bit_identical_to_the_replicated_one(world):
    out = _run(_t_sharded_step, world=world)
    for r in range(world):
        same_half, stale_before, same, a, b, wire, digest = out[r]
        assert same_half and stale_before and same
        assert digest == out[0][6]
        assert a == b == (4, 256.0, 1)        # two clean steps grow the scale to 256, the overflow halves it and skips the step, two clean steps grow it again
        assert wire == out[0][5]
    w = out[0][5]
    assert w['reduce_scatter'] == 4096 * 4 * (world - 1) // world and w['all_gather'] == 4096 * 2 * (world - 1) // world
    assert w['adam_elements_per_rank'] == 96 + 64 + 4096 // world


def test_sharded_layout_at_the_instant_ngp_sizes():
    """The shipped InstantNGP configuration at 8 ranks: what leaves a GPU per iteration and how many elements its Adam touches."""
    from nerficg_amd import parallel
    L = parallel.ShardedStepLayout(7168, 3072, 12_196_240, rank=3, world=8)
    assert L.sharded and L.shard == 1_524_530 and L.shard_begin == 3 * 1_524_530 and (L.off_density * 4) % 128 == 0 and (L.off_table * 4) % 128 == 0
    w = L.wire_bytes()
    assert w['reduce_scatter'] == 42_686_840 and w['all_gather'] == 21_343_420 and w['small_allreduce'] == int(2 * 7 / 8 * (7168 + 32 + 3072) * 4)
    assert w['adam_elements_per_rank'] == 7168 + 3072 + 1_524_530            # 1.53 M instead of 12.2 M
    replicated = parallel.ShardedStepLayout(7168, 3072, 12_196_240, rank=0, world=3)
    assert not replicated.sharded and replicated.wire_bytes()['adam_elements_per_rank'] == 7168 + 3072 + 12_196_240


def _t_sharded_wire(rank, world):
    """parallel.sharded_step with the optional 16-bit wire: pack -> reduce-scatter of the fp16 buffer -> unpack of the rank's shard; the shard the Adam callback sees is
    the fp16-rounded sum of the fp16-rounded contributions, the small piece and the flag still travel in f32."""
    from nerficg_amd import parallel
    L = parallel.ShardedStepLayout(8, 8, 64)
    g = torch.Generator().manual_seed(rank)
    grads = torch.randn(L.total, generator=g)
    grads[L.off_aux:L.off_density] = 0
    mine = grads.clone()
    wire = torch.zeros(L.n_table, dtype=torch.float16)
    half_table = torch.full((L.n_table,), float(rank), dtype=torch.float16)
    seen = {}
    parallel.sharded_step(L, grads, half_table, settle=lambda aux: seen.setdefault('flag', float(aux[0])), adam_small=lambda: None,
                          adam_table=lambda b, n: seen.setdefault('shard', grads[L.off_table + b:L.off_table + b + n].clone()),
                          wire=wire, pack=lambda: wire.copy_(grads[L.off_table:].clamp(-65504, 65504).half()),
                          unpack=lambda b, n: grads[L.off_table + b:L.off_table + b + n].copy_(wire[b:b + n].float()))
    others = [torch.randn(L.total, generator=torch.Generator().manual_seed(r)) for r in range(world)]
    want = sum(o[L.off_table:].half() for o in others).float()[L.shard_begin:L.shard_begin + L.shard]      # fp16 addends, fp16 sum
    small = sum(o[:L.off_table] for o in others)
    small[L.off_aux:L.off_density] = 0
    return bool(torch.equal(seen['shard'], want)), bool(torch.allclose(grads[:L.off_table], small)), seen['flag'], half_table.tolist()[::L.shard], L.wire_bytes(2)['reduce_scatter']


def test_sharded_step_with_the_16_bit_wire():
    out = _run(_t_sharded_wire)
    for r in (0, 1):
        shard_ok, small_ok, flag, gathered, rs_bytes = out[r]
        assert shard_ok and small_ok and flag == 0.0 and gathered == [0.0, 1.0] and rs_bytes == 64 * 2 // 2


def _t_four_ranks(rank, world):
    """The host logic above with expectations that hold for ANY world size: run at 4 ranks (a ragged last shard, a padded flat reduction, three remote peers)."""
    from nerficg_amd import parallel
    out = {}
    torch.manual_seed(0)
    batch = torch.randperm(1001)[:333]
    out['rays'] = parallel.shard_ray_ids(batch).tolist()
    out['batch'] = batch.tolist()
    out['tiles'] = parallel.shard_range(63)
    buf = torch.arange(11, dtype=torch.float32) * (rank + 1)       # 11 elements over 4 ranks: the padded reduce-scatter + all-gather
    parallel.allreduce_flat(buf, average=False)
    out['flat'] = buf.tolist()
    counts = [3 + r for r in range(world)]
    out['pixels'] = parallel.all_gather_pixels(torch.full((counts[rank], 2), float(rank)), counts).tolist()
    # view-parallel 3DGS: sparse reduction of the union rows == dense all-reduce
    g = torch.Generator().manual_seed(100 + rank)
    P = 300
    visible = torch.rand(P, generator=g) < 0.15
    params = []
    for s in [(P, 3), (P, 15, 3), (P, 1), (P, 4)]:
        p = torch.nn.Parameter(torch.zeros(s))
        p.grad = torch.randn(s, generator=g) * visible.view(-1, *([1] * (len(s) - 1)))
        params.append(p)
    dense = [p.grad.clone() for p in params]
    for d in dense:
        dist.all_reduce(d)
        d.div_(world)
    n_union = parallel.sparse_allreduce_gradients(params, visible, average=True)
    u = visible.to(torch.uint8)
    dist.all_reduce(u, op=dist.ReduceOp.MAX)
    out['sparse'] = (max(float((p.grad - d).abs().max()) for p, d in zip(params, dense)), n_union, int(u.sum()),
                     max(float(p.grad[u == 0].abs().max()) for p in params))
    sums, flags = parallel.allreduce_scalars([torch.tensor(1000 + rank)], [torch.tensor(float(rank == world - 1))])
    out['scalars'] = (float(sums[0]), float(flags[0]))
    # a table that does not divide by the world: the layout says so (FusedTrainingIteration then takes the replicated step) and sharded_step refuses it
    L = parallel.ShardedStepLayout(8, 8, 66)
    try:
        parallel.sharded_step(L, torch.zeros(L.total), torch.zeros(L.n_table, dtype=torch.float16), lambda aux: None, lambda: None, lambda b, n: None)
        refused = False
    except RuntimeError as e:
        refused = 'does not divide' in str(e)
    out['unsharded'] = (L.sharded, L.shard, refused)
    return out


def test_the_data_parallel_host_logic_at_four_ranks():
    world = 4
    out = _run(_t_four_ranks, world=world)
    batch = out[0]['batch']
    assert sorted(sum((out[r]['rays'] for r in range(world)), [])) == sorted(batch)             # the shards partition the batch
    assert [out[r]['tiles'] for r in range(world)] == [(0, 16), (16, 32), (32, 48), (48, 63)]
    pixels = sum(([[float(r)] * 2] * (3 + r) for r in range(world)), [])
    for r in range(world):
        assert out[r]['batch'] == batch
        assert out[r]['flat'] == [i * 10.0 for i in range(11)]                                  # 1 + 2 + 3 + 4
        assert out[r]['pixels'] == pixels
        err, n_union, n_mask, untouched = out[r]['sparse']
        assert err < 1e-6 and n_union == n_mask and 0 < n_union < 300 and untouched == 0.0
        assert out[r]['scalars'] == (4006.0, 1.0)
        assert out[r]['unsharded'] == (False, 66, True)
