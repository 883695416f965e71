"""nerficg_amd.parallel -- data parallelism for the hot path on one node of MI355X GPUs: one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference has no working multi-GPU path: its DataParallel wrapper is a no-op for RayBatch
(src/Methods/Base/Renderer.py:24-33, src/Methods/NeRF/Renderer.py:31).  What shards naturally:

  * rays / image tiles   -- independent units.  Inference: every rank renders a contiguous range of 8x8-pixel tiles (or its own
                            views); NO data-path collective, only an optional all-gather of the finished pixels.
  * 3DGS training views  -- view-parallel: every rank rasterizes ANOTHER training view against replicated Gaussians; a Gaussian's gradient is
                            non-zero only on ranks that saw it, so the reduction moves only the rows visible on at least one rank
                            (sparse_allreduce_gradients); densification statistics are summed when densification is due and
                            densify_and_prune runs redundantly on every rank from identically seeded noise.
  * training rays        -- rank r takes ray_ids[r::world] of the batch every rank draws from the same seeded permutation
                            (src/Optim/Samplers/utils.py:8-34), so the global ray set is bit-identical to the single-GPU run;
                            after backward the encoding / MLP (InstantNGP) or Gaussian (3DGS) gradients are summed with ONE
                            bucketed collective.  xGMI is point-to-point (7 links per GPU): reduce-scatter + all-gather lets
                            every link carry 1/world of the payload instead of a ring's per-link bound.
  * per-iteration scalars -- the marched sample count that drives update_batch_size (src/Methods/InstantNGP/Trainer.py:73-75,94) and the
                            GradScaler's found-inf flag (Trainer.py:44,89-93) must be the SAME on every rank, or the replicas pick different
                            batch sizes / skip different steps: allreduce_scalars moves them in ONE tiny collective, DataParallelGradScaler
                            is the torch.amp.GradScaler that uses it.
"""
from __future__ import annotations

import os
from typing import Iterable

import torch
import torch.distributed as dist

from .amp import GradScaler as _FastGradScaler

__all__ = ['init_distributed', 'world_info', 'shard_ray_ids', 'shard_range', 'allreduce_flat', 'allreduce_gradients',
           'all_gather_pixels', 'tile_pixel_indices', 'gather_image_shards', 'broadcast_parameters', 'sparse_allreduce_gradients', 'allreduce_densification_stats', 'synchronized_noise',
           'allreduce_scalars', 'DataParallelGradScaler', 'rays_per_batch_update', 'single_rank_collectives', 'ShardedStepLayout', 'allreduce_sum_',
           'reduce_scatter_sum_', 'all_gather_', 'sharded_step', 'UnionRowExchange']


_SINGLE_RANK_COLLECTIVES = False


def single_rank_collectives(on: bool) -> None:
    """Testing switch: with a process group of ONE rank, issue every collective of this module anyway (each is then a no-op sum / gather of
    the rank's own data).  This is how the RCCL branches -- reduce_scatter_tensor, all_gather_into_tensor, the device_id initialisation -- are
    executed on a box with a single GPU (tests/test_gpu_rccl.py); production runs never set it."""
    global _SINGLE_RANK_COLLECTIVES
    _SINGLE_RANK_COLLECTIVES = bool(on)


def _no_collective(world: int) -> bool:
    return world == 1 and not (_SINGLE_RANK_COLLECTIVES and dist.is_available() and dist.is_initialized())


def init_distributed(backend: str | None = None, device: torch.device | None = None, single_rank_group: bool = False) -> tuple[int, int]:
    """Initialises the default process group from the torchrun environment (RANK / WORLD_SIZE / MASTER_*). Returns (rank, world).
    `single_rank_group`: create the group for WORLD_SIZE = 1 too (see single_rank_collectives)."""
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', str(rank))
        os.environ.setdefault('WORLD_SIZE', str(world))
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC only on this platform
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {'device_id': device} if (backend == 'nccl' and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return rank, world


def world_info() -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_ray_ids(ray_ids: torch.Tensor, rank: int | None = None, world: int | None = None) -> torch.Tensor:
    """Rank r's share of a batch of ray indices: ray_ids[r::world].  The union over ranks is exactly the batch."""
    r, w = world_info()
    rank = r if rank is None else rank
    world = w if world is None else world
    return ray_ids[rank::world]


def rank_batch_order(order: torch.Tensor, n_rays: int, rank: int | None = None, world: int | None = None) -> torch.Tensor:
    """The sampling order of ONE rank when `world` ranks each consume n_rays rows per iteration from a GLOBAL order (the RandomSequentialSampler's
    permutation, identical on every rank): iteration k's global batch is order[k W n : (k + 1) W n] and rank r takes rows [r n, (r + 1) n) of it, so the
    rank's own order is the concatenation of those slices -- consumed front to back with a local cursor (nerficg_amd.ngp_trainer).  The union over the
    ranks of iteration k is exactly the global batch; a tail that does not fill a global batch is dropped (the sampler rewinds there anyway)."""
    r, w = world_info()
    rank = r if rank is None else rank
    world = w if world is None else world
    n_iter = order.numel() // (world * n_rays)
    return order[:n_iter * world * n_rays].reshape(n_iter, world, n_rays)[:, rank].reshape(-1).contiguous()


def shard_range(n: int, rank: int | None = None, world: int | None = None) -> tuple[int, int]:
    """Contiguous, balanced [begin, end) share of n units (image tiles, views) for this rank."""
    r, w = world_info()
    rank = r if rank is None else rank
    world = w if world is None else world
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def _has_tensor_collectives() -> bool:
    """reduce_scatter_tensor / all_gather_into_tensor exist on RCCL ("nccl"); allreduce_flat keeps gloo on one all_reduce (the round-3 wire format of
    the CPU tests).  Decided from the backend NAME, once per call site and identically on every rank -- never by catching an exception from a
    collective: a rank-local failure would send that rank alone into another collective and leave the job desynchronised."""
    return dist.get_backend() != 'gloo'


# ---- the three collectives of the sharded optimizer step ("reduce-scatter + sharded Adam + all-gather"), all IN PLACE on resident buffers -------
def _staged(t: torch.Tensor) -> bool:
    """gloo moves host memory: a device tensor is staged through the host (the 2-rank tests that share ONE GPU; production runs are RCCL)."""
    return t.is_cuda and dist.get_backend() == 'gloo'


def allreduce_sum_(buf: torch.Tensor) -> torch.Tensor:
    """In-place SUM of a small flat tensor over the ranks (the MLP-weight gradients + the overflow flag of a data-parallel iteration: 41 KB)."""
    rank, world = world_info()
    if _no_collective(world):
        return buf
    if _staged(buf):
        host = buf.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        buf.copy_(host)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def reduce_scatter_sum_(buf: torch.Tensor, rank: int | None = None, world: int | None = None) -> torch.Tensor:
    """In-place reduce-scatter (SUM) of a flat contiguous tensor whose length is a multiple of the world size: afterwards rows
    [rank * n / world, (rank + 1) * n / world) of `buf` hold the sum over the ranks (the rest of `buf` is this rank's own contribution still).  Returns that
    shard (a view).  RCCL's in-place form (receive buffer = send buffer + rank * count): no second 48.8 MB buffer, each xGMI link carries 1 / world."""
    r, w = world_info()
    rank, world = (r if rank is None else rank), (w if world is None else world)
    n = buf.numel()
    if n % world:
        raise RuntimeError(f'reduce_scatter_sum_: {n} elements do not divide by {world} ranks')
    shard = buf[rank * (n // world):(rank + 1) * (n // world)]
    if _no_collective(world):
        return shard
    if _staged(buf):
        host = buf.cpu()
        out = torch.empty(n // world, dtype=buf.dtype)
        dist.reduce_scatter_tensor(out, host, op=dist.ReduceOp.SUM)
        shard.copy_(out)
    else:
        dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM)
    return shard


def all_gather_(buf: torch.Tensor, rank: int | None = None, world: int | None = None) -> torch.Tensor:
    """In-place all-gather: every rank contributes rows [rank * n / world, (rank + 1) * n / world) of its `buf` and ends with all of them."""
    r, w = world_info()
    rank, world = (r if rank is None else rank), (w if world is None else world)
    n = buf.numel()
    if n % world:
        raise RuntimeError(f'all_gather_: {n} elements do not divide by {world} ranks')
    if _no_collective(world):
        return buf
    shard = buf[rank * (n // world):(rank + 1) * (n // world)]
    if _staged(buf):
        host = torch.empty(n, dtype=buf.dtype)
        dist.all_gather_into_tensor(host, shard.cpu())
        buf.copy_(host)
    else:
        dist.all_gather_into_tensor(buf, shard)
    return buf


def sharded_step(layout: 'ShardedStepLayout', grads: torch.Tensor, half_table: torch.Tensor, settle, adam_small, adam_table,
                 before_table=None, mark=None, wire: torch.Tensor | None = None, pack=None, unpack=None) -> None:
    """The optimizer step of one data-parallel rank on a gradient buffer laid out as `layout` says -- the ORDER of collectives and updates, shared by
    nerficg_amd.ngp_trainer (callbacks = library calls on the communication stream) and the CPU tests (callbacks = the CPU oracle):

        1. all-reduce (SUM) of [colour MLP | aux | density MLP]: aux[0] becomes the number of ranks that saw an inf / NaN
        2. before_table(): the caller waits for its grid backward here (the small collective above ran beside it)
        3. reduce-scatter (SUM, in place) of the table gradient: this rank's shard now holds the sum over the ranks
        4. settle(aux): found_inf = aux[0] != 0, step counter, 1 / (scale x world), scale update -- identical on every rank, no further agreement needed
        5. adam_small(): the MLP weights, on every rank redundantly;  adam_table(begin, count): this rank's shard (offsets relative to the table)
        6. all-gather (in place) of `half_table`, the fp16 (or whatever the kernels read) copy of the table, whose shard step 5 rewrote

    mark(k), optional: called behind steps 2, 3, 5, 6 (k = 0..3) -- the trainer records timing events there.
    wire / pack / unpack (optional, all three): the 16-bit wire of step 3 -- pack() converts the table gradient into `wire` (n_table elements of a 16-bit
    dtype), the reduce-scatter runs on `wire`, unpack(begin, count) brings this rank's reduced shard back into grads[table] for step 5."""
    L = layout
    if not L.sharded:
        raise RuntimeError('sharded_step: the table does not divide by the world size (ShardedStepLayout.sharded)')
    allreduce_sum_(grads[:L.off_table])
    if before_table is not None:
        before_table()
    if mark: mark(0)
    if wire is not None:
        pack()
        reduce_scatter_sum_(wire, L.rank, L.world)
        unpack(L.shard_begin, L.shard)
    else:
        reduce_scatter_sum_(grads[L.off_table:L.total], L.rank, L.world)
    if mark: mark(1)
    settle(grads[L.off_aux:L.off_density])
    adam_small()
    adam_table(L.shard_begin, L.shard)
    if mark: mark(2)
    all_gather_(half_table, L.rank, L.world)
    if mark: mark(3)


class ShardedStepLayout:
    """Where the pieces of ONE flat f32 gradient buffer of an InstantNGP model live, in the order a data-parallel iteration finishes them:

        [ colour-MLP gradient (n_color) | aux (AUX floats: [0] = overflow flag) | density-MLP gradient (n_density_mlp) | hash-table gradient (n_table) ]
          `-------------------------- small: ONE all-reduce right behind the network backward ---------------'   `-- reduce-scatter behind the grid backward

    The table is cut into `world` equal shards; rank r runs Adam on table[r s, (r + 1) s) only (its moments and fp32 master are current THERE only) and on the
    10 240 MLP weights, which every rank keeps in step redundantly; the fp16 table copy the kernels read is all-gathered.  `sharded` is False when the table
    does not divide by the world size (the caller then reduces the whole buffer and steps redundantly -- a world of 3, 6 or 7 GPUs).
    wire_bytes(): bytes a GPU SENDS per iteration, the (world - 1) / world factors of ring / direct reduce-scatter and all-gather included."""
    AUX = 32    # floats: keeps the density vector 128-byte aligned behind the colour vector

    def __init__(self, n_color: int, n_density_mlp: int, n_table: int, rank: int | None = None, world: int | None = None) -> None:
        r, w = world_info()
        self.rank, self.world = (r if rank is None else int(rank)), (w if world is None else int(world))
        self.n_color, self.n_density_mlp, self.n_table = int(n_color), int(n_density_mlp), int(n_table)
        self.off_aux = self.n_color
        self.off_density = self.n_color + self.AUX
        self.off_table = self.off_density + self.n_density_mlp
        self.total = self.off_table + self.n_table
        self.sharded = self.n_table % self.world == 0
        self.shard = self.n_table // self.world if self.sharded else self.n_table
        self.shard_begin = self.rank * self.shard if self.sharded else 0      # relative to the table

    def wire_bytes(self, table_wire_bytes: int = 4, half_bytes: int = 2) -> dict:
        f = (self.world - 1) / self.world
        small = 2 * f * self.off_table * 4
        if self.sharded:
            rs, ag = f * self.n_table * table_wire_bytes, f * self.n_table * half_bytes
        else:
            rs, ag = f * self.n_table * 4, f * self.n_table * 4
        return {'small_allreduce': int(small), 'reduce_scatter': int(rs), 'all_gather': int(ag), 'total': int(small + rs + ag),
                'adam_elements_per_rank': self.off_table - self.AUX + self.shard}


def allreduce_flat(buf: torch.Tensor, average: bool = True) -> torch.Tensor:
    """In-place sum (or mean) of a flat contiguous tensor over all ranks as reduce-scatter + all-gather (each of the 7 xGMI
    links of a GPU then carries 1/world of the payload); one all_reduce on gloo.  Errors of a collective propagate."""
    rank, world = world_info()
    if _no_collective(world):
        return buf
    if not _has_tensor_collectives():
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        if average:
            buf.div_(world)
        return buf
    n = buf.numel()
    pad = (-n) % world
    work = buf if pad == 0 else torch.cat([buf, buf.new_zeros(pad)])
    shard = torch.empty(work.numel() // world, dtype=work.dtype, device=work.device)
    dist.reduce_scatter_tensor(shard, work, op=dist.ReduceOp.SUM)
    if average:
        shard.div_(world)
    dist.all_gather_into_tensor(work, shard)
    if pad:
        buf.copy_(work[:n])
    return buf


def allreduce_gradients(params: Iterable[torch.nn.Parameter], average: bool = True, bucket_bytes: int = 256 << 20) -> None:
    """Sums (averages) .grad of the given parameters over all ranks, bucketed into few large flat collectives.  InstantNGP: two
    parameters (48.8 MB hash table + MLP, 28 KB colour MLP) -> one bucket; 3DGS: 59 floats x P in 5 tensors -> 1.4 GB at 6 M
    Gaussians in 256 MB buckets."""
    rank, world = world_info()
    if _no_collective(world):
        return
    grads = [p.grad for p in params if p.grad is not None]
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket]) if len(bucket) > 1 else bucket[0].reshape(-1)
        allreduce_flat(flat, average)
        if len(bucket) > 1:
            off = 0
            for g in bucket:
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()
        bucket, size = [], 0

    for g in grads:
        if not g.is_contiguous():
            raise RuntimeError('allreduce_gradients expects contiguous gradients')
        nbytes = g.numel() * g.element_size()
        if size and (size + nbytes > bucket_bytes or g.dtype != bucket[0].dtype):
            flush()
        bucket.append(g)
        size += nbytes
    flush()


def all_gather_pixels(local: torch.Tensor, counts: list[int]) -> torch.Tensor:
    """Concatenates per-rank pixel blocks of different lengths (dim 0) on every rank."""
    rank, world = world_info()
    if _no_collective(world):
        return local
    m = max(counts)
    padded = local if local.shape[0] == m else torch.cat([local, local.new_zeros((m - local.shape[0],) + tuple(local.shape[1:]))])
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded.contiguous())
    return torch.cat([o[:c] for o, c in zip(out, counts)])


def tile_pixel_indices(width: int, height: int, tile_begin: int, n_tiles: int, tile_w: int = 8, tile_h: int = 8, device=None) -> torch.Tensor:
    """Flat pixel indices (y * width + x, int64) of the image pixels inside the tiles [tile_begin, tile_begin + n_tiles) of the row-major tile
    grid render_image_fused shards over, in tile order then row-major inside a tile; pixels of border tiles that fall outside the image are
    left out.  A pure function of the frame shape, so every rank can compute every other rank's list (and its length) without a message."""
    gx = (width + tile_w - 1) // tile_w
    t = torch.arange(tile_begin, tile_begin + n_tiles, dtype=torch.int64, device=device)
    ty, tx = t // gx, t % gx
    py = (ty[:, None, None] * tile_h + torch.arange(tile_h, dtype=torch.int64, device=device)[None, :, None]).expand(-1, tile_h, tile_w)
    px = (tx[:, None, None] * tile_w + torch.arange(tile_w, dtype=torch.int64, device=device)[None, None, :]).expand(-1, tile_h, tile_w)
    ok = (py < height) & (px < width)
    return (py * width + px)[ok]


def gather_image_shards(out: dict[str, torch.Tensor], width: int, height: int, n_tiles: int, tile_w: int = 8, tile_h: int = 8,
                        cache: dict | None = None) -> dict[str, torch.Tensor]:
    """One frame rendered as contiguous tile shards (shard_range over the ranks; render_image_fused(tile_begin, n_tiles) wrote this rank's pixels
    into the flat (H*W, C) buffers of `out`): packs this rank's pixels as (n_r, 5) rows [r, g, b, alpha, depth], all-gathers the blocks
    ("all-gather of (N / world, 5)") and scatters every other rank's block into `out`, so that every rank ends with the whole
    frame.  `cache`: a dict the caller keeps per frame shape (the index lists are static).  world = 1: nothing to do."""
    rank, world = world_info()
    if _no_collective(world):
        return out
    dev = out['rgb'].device
    key = (width, height, n_tiles, world, str(dev))
    if cache is None or cache.get('key') != key:
        idx = [tile_pixel_indices(width, height, *(lambda b, e: (b, e - b))(*shard_range(n_tiles, r, world)), tile_w, tile_h, dev) for r in range(world)]
        entry = {'key': key, 'idx': idx, 'counts': [int(i.numel()) for i in idx]}
        if cache is not None:
            cache.clear(); cache.update(entry)
    else:
        entry = cache
    mine = entry['idx'][rank]
    local = torch.cat([out['rgb'][mine], out['alpha'][mine, None], out['depth'][mine, None]], dim=1)
    full = all_gather_pixels(local, entry['counts'])
    off = 0
    for r, (ids, c) in enumerate(zip(entry['idx'], entry['counts'])):
        if r != rank:
            block = full[off:off + c]
            out['rgb'][ids] = block[:, :3]; out['alpha'][ids] = block[:, 3]; out['depth'][ids] = block[:, 4]
        off += c
    return out


def broadcast_parameters(params: Iterable[torch.Tensor], src: int = 0) -> None:
    """Makes every rank start from rank `src`'s parameters (seeded init is already identical; this guards checkpoints)."""
    rank, world = world_info()
    if _no_collective(world):
        return
    for p in params:
        dist.broadcast(p.data if isinstance(p, torch.nn.Parameter) else p, src)


class UnionRowExchange:
    """View-parallel 3DGS: the per-Gaussian gradients of `params` (each (P, ...)) are summed (averaged) over the ranks, moving only the rows
    some rank saw.  Two calls per step, shaped so that nothing waits for the wire or the host that does not have to:

        begin(visible)    (a boolean mask, or the rasterizer's integer radii) right after the forward pass (the backward pass is not enqueued yet), on a communication stream:
                          max-all-reduce of the byte visibility mask -> nrc_compact_mask (device scan: the same ascending union list on every rank) ->
                          the count goes to pinned host memory behind an event.  The caller now enqueues the backward pass; mask traffic and scan run beside it.
        finish(params)    behind the backward pass: the host reads the count (the device is busy with the backward pass meanwhile -- the read that used to
                          stall the step, torch.nonzero, sat between backward and collective), nrc_gather_rows packs the union rows of ALL tensors into one
                          flat buffer in ONE launch (tensor-major: n rows of tensor 0, n rows of tensor 1, ...), reduce-scatter + all-gather
                          (allreduce_flat), nrc_scatter_rows puts them back in ONE launch.  Rows outside the union keep their (zero) gradient.

    Payload: n_union x 236 B instead of P x 236 B.  Every rank enters both collectives whatever it holds: a rank without gradients contributes zero rows.
    Adam stays replicated here: a row's moments would have to live with a fixed owner while the union changes every step, and shipping the updated
    parameters of the owner's rows costs what the all-gather of the reduced gradients costs."""

    def __init__(self, n_rows: int, device) -> None:
        from . import _lib
        self.P, self.dev = int(n_rows), torch.device(device)
        self._cuda = self.dev.type == 'cuda'
        self.union = torch.zeros(self.P, dtype=torch.uint8, device=self.dev)
        self.n: int | None = None
        if self._cuda:
            lib = _lib.load()
            self.idx = torch.empty(self.P, dtype=torch.int32, device=self.dev)
            self.count = torch.zeros(1, dtype=torch.int32, device=self.dev)
            self.ws = torch.empty(int(lib.nrc_compact_mask_ws_bytes(self.P)), dtype=torch.uint8, device=self.dev)
            self.count_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self.comm = torch.cuda.Stream(device=self.dev)
            self.ready = torch.cuda.Event()
            self.packed = None

    def begin(self, visible: torch.Tensor) -> None:
        rank, world = world_info()
        if visible.shape[0] != self.P:
            raise RuntimeError(f'UnionRowExchange: visibility mask with {visible.shape[0]} rows, built for {self.P}')
        self.n = None
        if _no_collective(world):
            self.n = -1
            return
        if not self._cuda:      # CPU tensors (the gloo tests): the same protocol with torch ops
            self.union.copy_((visible if visible.dtype == torch.bool else visible > 0).to(torch.uint8))
            dist.all_reduce(self.union, op=dist.ReduceOp.MAX)
            self.idx = torch.nonzero(self.union, as_tuple=False).flatten().to(torch.int32)
            self.n = int(self.idx.numel())
            return
        from . import _lib
        main = torch.cuda.current_stream(self.dev)
        self.comm.wait_stream(main)
        visible.record_stream(self.comm)
        with torch.cuda.stream(self.comm):
            self.union.copy_(visible if visible.dtype == torch.bool else visible > 0)     # bool (or the rasterizer's radii: visible = radius > 0) -> u8
            if _staged(self.union):
                host = self.union.cpu(); dist.all_reduce(host, op=dist.ReduceOp.MAX); self.union.copy_(host)
            else:
                dist.all_reduce(self.union, op=dist.ReduceOp.MAX)
            _lib.check(_lib.load().nrc_compact_mask(_lib.ptr(self.union), self.P, _lib.ptr(self.idx), _lib.ptr(self.count), _lib.ptr(self.ws),
                                                    _lib.stream_of(self.union)), 'compact_mask')
            self.count_host.copy_(self.count, non_blocking=True)
            self.ready.record(self.comm)

    def finish(self, params: Iterable[torch.nn.Parameter], average: bool = True) -> int:
        params = list(params)
        if self.n == -1:
            return -1
        if self.n is None:
            if not self._cuda:
                raise RuntimeError('UnionRowExchange.finish() without begin()')
            self.ready.synchronize()
            self.n = int(self.count_host[0])
            torch.cuda.current_stream(self.dev).wait_stream(self.comm)
        n = self.n
        if n == 0 or not params:   # the same decision on every rank: the union and the parameter list are identical everywhere
            return n
        rows, widths = [], []
        for p in params:
            if p.shape[0] != self.P:
                raise RuntimeError(f'UnionRowExchange: parameter with {p.shape[0]} rows, visibility mask has {self.P}')
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            if not p.grad.is_contiguous():
                raise RuntimeError('UnionRowExchange expects contiguous gradients')
            rows.append(p.grad.reshape(self.P, -1))
            widths.append(rows[-1].shape[1])
        total = sum(widths)
        if not self._cuda:
            idx = self.idx.long()
            packed = torch.cat([r[idx].reshape(-1) for r in rows])
            allreduce_flat(packed, average)
            off = 0
            for r, w in zip(rows, widths):
                r[idx] = packed[off:off + n * w].view(n, w)
                off += n * w
            return n
        import ctypes
        from . import _lib
        lib = _lib.load()
        if self.packed is None or self.packed.numel() < n * total:
            self.packed = torch.empty(max(n * total, self.P * total // 2), dtype=torch.float32, device=self.dev)      # grows to the largest union seen
        packed = self.packed[:n * total]
        k = len(rows)
        offs = [0]
        for w in widths:
            offs.append(offs[-1] + n * w)
        a_grad = (ctypes.c_void_p * k)(*[r.data_ptr() for r in rows])
        a_pack = (ctypes.c_void_p * k)(*[packed.data_ptr() + 4 * offs[t] for t in range(k)])
        a_row = (ctypes.c_int32 * k)(*widths)
        a_zero = (ctypes.c_int32 * k)(*([0] * k))
        cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
        s = _lib.stream_of(packed)
        _lib.check(lib.nrc_gather_rows(cast(a_grad), cast(a_pack), cast(a_row), cast(a_zero), k, _lib.ptr(self.idx), None, n, s), 'gather_rows')
        if _staged(packed):
            host = packed.cpu(); allreduce_flat(host, average); packed.copy_(host)
        else:
            allreduce_flat(packed, average)
        _lib.check(lib.nrc_scatter_rows(cast(a_pack), cast(a_grad), cast(a_row), k, _lib.ptr(self.idx), n, s), 'scatter_rows')
        return n


def sparse_allreduce_gradients(params: Iterable[torch.nn.Parameter], visible: torch.Tensor, average: bool = True, exchange: UnionRowExchange | None = None) -> int:
    """UnionRowExchange.begin + finish back to back (callers that cannot split the two around their backward pass).  `visible`: this rank's (P,) boolean
    mask (radii > 0); rows outside it must have zero gradient here (the rasterizer backward guarantees that).  Returns n_union (-1 in a single-process run).
    `exchange`: a UnionRowExchange to reuse (its buffers are sized once); by default one is kept per (P, device)."""
    key = (int(visible.shape[0]), str(visible.device))
    ex = exchange or _EXCHANGES.get(key)
    if ex is None:
        ex = _EXCHANGES[key] = UnionRowExchange(visible.shape[0], visible.device)
        while len(_EXCHANGES) > 4:
            _EXCHANGES.pop(next(iter(_EXCHANGES)))
    ex.begin(visible)
    return ex.finish(params, average)


_EXCHANGES: dict = {}


def allreduce_scalars(sums: Iterable[torch.Tensor] = (), flags: Iterable[torch.Tensor] = ()) -> tuple[list[torch.Tensor], list[torch.Tensor]]:
    """The per-iteration scalars of SURVEY 8(e) in ONE collective, device tensors in and out (no host read): `sums` are added up over the ranks
    (e.g. rm_samples, the marched sample count update_batch_size works from, InstantNGP/Trainer.py:73-75,94), `flags` are OR-ed (e.g. the
    GradScaler's found-inf, Trainer.py:44,89-93: one rank's overflow must skip the step on every rank).  All values travel as float64 in one
    packed all-reduce (SUM; a flag is set iff the sum of the ranks' flags is positive -- counts up to 2^53 are exact).  Returns
    ([global sums, f64 scalars], [global flags, f32 0/1 scalars like GradScaler's found_inf]).  Single process: the inputs, converted."""
    sums, flags = list(sums), list(flags)
    if not sums and not flags:
        return [], []
    dev = (sums + flags)[0].device
    pack = torch.stack([t.detach().reshape(-1)[0].to(device=dev, dtype=torch.float64) for t in sums + flags])
    rank, world = world_info()
    if not _no_collective(world):
        dist.all_reduce(pack, op=dist.ReduceOp.SUM)
    out_sums = [pack[i] for i in range(len(sums))]
    out_flags = [(pack[len(sums) + j] > 0).to(torch.float32) for j in range(len(flags))]
    return out_sums, out_flags


def rays_per_batch_update(rays_per_batch: int, target_samples: int, samples_since_update: float, iterations: int, world: int | None = None) -> int:
    """The batch-size controller of InstantNGP/Trainer.py:73-75 for replicas: `samples_since_update` is the GLOBAL marched sample count of the
    last `iterations` iterations (allreduce_scalars of rm_samples, accumulated), every rank marches rays_per_batch rays per iteration and
    aims at target_samples samples per iteration ON ITS OWN GPU (weak scaling), so the per-rank mean is global / (iterations * world).  The
    input is identical on all ranks, hence so is the result.  world = 1 is the reference's rule unchanged: next multiple of 256 of
    rays * target / measured, capped at the target."""
    world = world_info()[1] if world is None else world
    measured = samples_since_update / (iterations * world)
    if measured <= 0:
        return rays_per_batch
    wanted = rays_per_batch * target_samples / measured
    return int(min(-(-wanted // 256) * 256, target_samples))


class DataParallelGradScaler(_FastGradScaler):
    """GradScaler (nerficg_amd.amp: torch.amp.GradScaler with the streaming inf check) whose found-inf decision is global: whenever the scaler has looked at an optimizer's gradients (unscale_, or the
    check inside step() for optimizers that apply the scale themselves, like FusedAdam), the per-device found-inf tensors are replaced by
    their OR over all ranks -- so every replica skips the same steps and update() moves the scale identically.  When the gradients were
    already summed over the ranks an inf / NaN has reached every rank through the sum; the agreement then costs one tiny all-reduce and
    protects the other orders of operations (unscale before the gradient collective, sharded optimizers).
    `piggyback`: device scalars to add up over the ranks in the SAME collective (the marched sample count); the global values are in
    `reduced` after step()."""

    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.piggyback: list[torch.Tensor] = []
        self.reduced: list[torch.Tensor] = []

    def _agree(self, optimizer) -> None:
        # EVERY rank enters the collective with a vector of the same length -- the piggyback values (the caller passes the same number on every
        # rank) plus ONE flag: the OR of this rank's per-device found-inf tensors, zero when this rank's optimizer had no gradient this step (its
        # per-device dictionary is then empty; returning early there would leave the other ranks waiting in the all-reduce).
        found = self._per_optimizer_states[id(optimizer)]['found_inf_per_device']
        flags = list(found.values())
        if flags:
            dev = flags[0].device
            local = flags[0] if len(flags) == 1 else torch.stack([f.reshape(-1)[0].to(dev) for f in flags]).max().reshape(1)
        else:
            dev = self._scale.device if self._scale is not None else (self.piggyback[0].device if self.piggyback else torch.device('cpu'))
            local = torch.zeros(1, dtype=torch.float32, device=dev)
        sums, agreed = allreduce_scalars([t.to(dev) for t in self.piggyback], [local])    # (a count the host already knows travels as a host scalar)
        for t in flags:
            t.copy_(agreed[0].to(t.device))
        if not flags:   # torch's step() insists on a recorded check: this rank's is the agreed flag (its own step has nothing to apply)
            found[dev] = agreed[0].reshape(1)
        self.reduced, self.piggyback = sums, []

    def _check_inf_per_device(self, optimizer):
        out = super()._check_inf_per_device(optimizer)
        self._agree(optimizer)
        return out

    def unscale_(self, optimizer) -> None:
        super().unscale_(optimizer)
        self._agree(optimizer)


def allreduce_densification_stats(gaussians) -> None:
    """Sums densification_gradient_accum / n_observations (GaussianSplatting/Model.py:243-246) over the ranks; call right before
    densify_and_prune so that every rank classifies from the statistics of ALL views since the last densification."""
    rank, world = world_info()
    if _no_collective(world):
        return
    dist.all_reduce(gaussians.densification_gradient_accum, op=dist.ReduceOp.SUM)
    dist.all_reduce(gaussians.n_observations, op=dist.ReduceOp.SUM)


def synchronized_noise(n_rows: int, seed: int, device) -> torch.Tensor:
    """(n_rows, 3) standard-normal draws that are identical on every rank (host generator, then one upload): the `noise` argument of
    Gaussians.densify_and_prune when the model is replicated -- every rank then performs the same split (Model.py:196-198)."""
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn((n_rows, 3), generator=g, dtype=torch.float32).to(device)
