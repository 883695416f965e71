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
"""
from __future__ import annotations

import os
from typing import Iterable

import torch
import torch.distributed as dist

__all__ = ['init_distributed', 'world_info', 'shard_ray_ids', 'shard_range', 'allreduce_flat', 'allreduce_gradients',
           'all_gather_pixels', 'broadcast_parameters', 'sparse_allreduce_gradients', 'allreduce_densification_stats', 'synchronized_noise']


def init_distributed(backend: str | None = None, device: torch.device | None = None) -> tuple[int, int]:
    """Initialises the default process group from the torchrun environment (RANK / WORLD_SIZE / MASTER_*). Returns (rank, world)."""
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
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


def shard_range(n: int, rank: int | None = None, world: int | None = None) -> tuple[int, int]:
    """Contiguous, balanced [begin, end) share of n units (image tiles, views) for this rank."""
    r, w = world_info()
    rank = r if rank is None else rank
    world = w if world is None else world
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def _has_tensor_collectives() -> bool:
    """reduce_scatter_tensor / all_gather_into_tensor exist on RCCL ("nccl"); gloo has all_reduce only.  Decided from the backend NAME, once
    per call site and identically on every rank -- never by catching an exception from a collective: a rank-local failure would send that
    rank alone into another collective and leave the job desynchronised."""
    return dist.get_backend() != 'gloo'


def allreduce_flat(buf: torch.Tensor, average: bool = True) -> torch.Tensor:
    """In-place sum (or mean) of a flat contiguous tensor over all ranks as reduce-scatter + all-gather (each of the 7 xGMI
    links of a GPU then carries 1/world of the payload); one all_reduce on gloo.  Errors of a collective propagate."""
    rank, world = world_info()
    if world == 1:
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
    if world == 1:
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
    if world == 1:
        return local
    m = max(counts)
    padded = local if local.shape[0] == m else torch.cat([local, local.new_zeros((m - local.shape[0],) + tuple(local.shape[1:]))])
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded.contiguous())
    return torch.cat([o[:c] for o, c in zip(out, counts)])


def broadcast_parameters(params: Iterable[torch.Tensor], src: int = 0) -> None:
    """Makes every rank start from rank `src`'s parameters (seeded init is already identical; this guards checkpoints)."""
    rank, world = world_info()
    if world == 1:
        return
    for p in params:
        dist.broadcast(p.data if isinstance(p, torch.nn.Parameter) else p, src)


def sparse_allreduce_gradients(params: Iterable[torch.nn.Parameter], visible: torch.Tensor, average: bool = True) -> int:
    """View-parallel 3DGS: sums (averages) the per-Gaussian gradients of `params` (each (P, ...)) over all ranks, moving only
    the rows that are visible on at least one rank.  `visible`: this rank's (P,) boolean mask (radii > 0); rows outside it must have zero
    gradient here (the rasterizer backward guarantees that).  Protocol: one all-reduce (max) of the byte mask -> the same ascending
    union index list on every rank -> the union rows of all tensors packed into ONE flat buffer -> reduce-scatter + all-gather ->
    unpack.  Payload: n_union x 236 B instead of P x 236 B (59 floats per Gaussian).  Returns n_union (-1 in a single-process run)."""
    rank, world = world_info()
    params = [p for p in params if p.grad is not None]
    if world == 1 or not params:
        return -1  # nothing travels; the count of visible rows would cost a host read per step
    P = visible.shape[0]
    union = visible.to(torch.uint8).contiguous()
    dist.all_reduce(union, op=dist.ReduceOp.MAX)
    idx = torch.nonzero(union, as_tuple=False).flatten()
    n = idx.numel()
    if n == 0:
        return 0
    rows = []
    for p in params:
        if p.grad.shape[0] != P:
            raise RuntimeError(f'sparse_allreduce_gradients: gradient with {p.grad.shape[0]} rows, visibility mask has {P}')
        rows.append(p.grad.reshape(P, -1))
    packed = torch.cat([r[idx] for r in rows], dim=1).contiguous()  # (n_union, 59)
    allreduce_flat(packed.view(-1), average)
    off = 0
    for r in rows:
        w = r.shape[1]
        r[idx] = packed[:, off:off + w]
        off += w
    return n


def allreduce_densification_stats(gaussians) -> None:
    """Sums densification_gradient_accum / n_observations (GaussianSplatting/Model.py:243-246) over the ranks; call right before
    densify_and_prune so that every rank classifies from the statistics of ALL views since the last densification."""
    rank, world = world_info()
    if world == 1:
        return
    dist.all_reduce(gaussians.densification_gradient_accum, op=dist.ReduceOp.SUM)
    dist.all_reduce(gaussians.n_observations, op=dist.ReduceOp.SUM)


def synchronized_noise(n_rows: int, seed: int, device) -> torch.Tensor:
    """(n_rows, 3) standard-normal draws that are identical on every rank (host generator, then one upload): the `noise` argument of
    Gaussians.densify_and_prune when the model is replicated -- every rank then performs the same split (Model.py:196-198)."""
    g = torch.Generator().manual_seed(int(seed))
    return torch.randn((n_rows, 3), generator=g, dtype=torch.float32).to(device)
