"""nerficg_amd.rays -- host-side containers of the ray path: RayBatch (src/Datasets/utils.py:537-670), RayCollection (:673-690) and the
ray-pool sampler that DEFINES the ray indices of a training batch (src/Optim/Samplers/DatasetSamplers.py:53-66).

A RayBatch is a struct of per-ray tensors (origin, direction and optional view_direction / rgb / alpha / depth / timestamp) that all
share length, dtype and device; indexing gathers every present field.  `rays_of_view` builds one with the device ray generator
(nerficg_amd.raygen) the way View.get_rays does (:1053-1074): annotations flattened HWC, timestamp broadcast per ray.
"""
from __future__ import annotations

from dataclasses import dataclass, fields, replace

import numpy as np
import torch

from .samplers import RandomSequentialSampler

__all__ = ['compute_all_rays', 'RayBatch', 'RayCollection', 'RayPoolSampler', 'rays_of_view']

_RAY_FIELDS = ('origin', 'direction', 'view_direction', 'rgb', 'alpha', 'depth', 'timestamp')


@dataclass(frozen=True)
class RayBatch:
    origin: torch.Tensor
    direction: torch.Tensor
    view_direction: torch.Tensor | None = None
    rgb: torch.Tensor | None = None
    alpha: torch.Tensor | None = None
    depth: torch.Tensor | None = None
    timestamp: torch.Tensor | None = None
    _skip_post_init: bool = False

    def __post_init__(self):
        if self._skip_post_init:
            return
        ref = self.origin
        for name in _RAY_FIELDS:
            t = getattr(self, name)
            if t is None:
                continue
            if t.shape[0] != ref.shape[0] or t.dtype != ref.dtype or t.device != ref.device:
                raise ValueError(f'RayBatch.{name}: expected {ref.shape[0]} rays of {ref.dtype} on {ref.device}, '
                                 f'got {t.shape[0]} of {t.dtype} on {t.device}')

    def _map(self, fn) -> 'RayBatch':
        return RayBatch(**{n: (None if getattr(self, n) is None else fn(getattr(self, n))) for n in _RAY_FIELDS}, _skip_post_init=True)

    def __len__(self) -> int:
        return self.origin.shape[0]

    @property
    def dtype(self) -> torch.dtype:
        return self.origin.dtype

    @property
    def device(self) -> torch.device:
        return self.origin.device

    @property
    def annotations(self):
        return self.view_direction, self.rgb, self.alpha, self.depth, self.timestamp

    @property
    def has_annotations(self) -> bool:
        return any(a is not None for a in self.annotations)

    @property
    def as_tensor(self) -> torch.Tensor:
        return torch.cat([self.origin, self.direction] + [a for a in self.annotations if a is not None], dim=-1)

    def __getitem__(self, idx) -> 'RayBatch':
        if idx is Ellipsis or (isinstance(idx, slice) and idx == slice(None)):
            return self
        if isinstance(idx, int):
            idx = slice(idx, idx + 1)
        return self._map(lambda t: t[idx])

    def to(self, dtype: torch.dtype | None = None, device=None, non_blocking: bool = False) -> 'RayBatch':
        if (dtype is None or dtype == self.dtype) and (device is None or torch.device(device) == self.device):
            return self
        return self._map(lambda t: t.to(dtype=dtype, device=device, non_blocking=non_blocking))

    def cpu(self, non_blocking: bool = False) -> 'RayBatch':
        return self.to(device='cpu', non_blocking=non_blocking)

    def cuda(self, non_blocking: bool = False) -> 'RayBatch':
        return self.to(device='cuda', non_blocking=non_blocking)

    def split(self, chunk_size: int) -> list['RayBatch']:
        return [self[i:i + chunk_size] for i in range(0, len(self), chunk_size)]

    @classmethod
    def cat(cls, batches: list['RayBatch']) -> 'RayBatch':
        if not batches:
            raise ValueError('no RayBatch instances to concatenate')
        out = {}
        for name in _RAY_FIELDS:
            present = [getattr(b, name) is not None for b in batches]
            if any(present) and not all(present):
                raise ValueError(f'RayBatch field "{name}" is not present in some batches')
            out[name] = torch.cat([getattr(b, name) for b in batches], dim=0) if all(present) else None
        return cls(**out)


@dataclass(frozen=True)
class RayCollection:
    """All rays of a dataset split: one flat RayBatch + the ray count of every view (utils.py:673-690)."""
    rays: RayBatch
    rays_per_view: tuple[int, ...]

    def __len__(self) -> int:
        return len(self.rays_per_view)

    def __getitem__(self, index: int) -> RayBatch:
        start = int(np.sum(self.rays_per_view[:index], dtype=np.int64))
        return self.rays[start:start + self.rays_per_view[index]]

    @property
    def all_rays(self) -> RayBatch:
        return self.rays


def rays_of_view(camera, c2w: np.ndarray, rgb: torch.Tensor | None = None, alpha: torch.Tensor | None = None, depth: torch.Tensor | None = None,
                 timestamp: float | None = None, device='cuda') -> RayBatch:
    """View.get_rays (utils.py:1053-1074) on the device: images arrive (C, H, W) and are flattened to (H*W, C)."""
    from .raygen import generate_rays
    r = generate_rays(camera.width, camera.height, camera.focal_x, camera.focal_y, camera.center_x, camera.center_y, c2w, device=device)
    flat = lambda img: None if img is None else img.to(r['origin'].device).permute(1, 2, 0).reshape(camera.width * camera.height, -1).contiguous()
    ts = None if timestamp is None else torch.full((camera.width * camera.height, 1), float(timestamp), device=r['origin'].device)
    return RayBatch(origin=r['origin'], direction=r['direction'], view_direction=r['view_direction'], rgb=flat(rgb), alpha=flat(alpha),
                    depth=flat(depth), timestamp=ts)


def compute_all_rays(views, store_on_cpu: bool = False, as_ray_collection: bool = False, device='cuda'):
    """BaseDataset.compute_all_rays / precompute_rays (src/Datasets/Base.py:172-216): the rays of every view of a split, generated on the
    device (C ABI group 5), concatenated into one RayBatch -- optionally parked in host memory -- or a RayCollection that remembers the
    per-view ray counts.  `views`: iterable of dicts with the View fields the reference reads: camera, c2w and optionally rgb, alpha,
    depth ((C, H, W) tensors) and timestamp."""
    batches, counts = [], []
    for view in views:
        batch = rays_of_view(view['camera'], view['c2w'], view.get('rgb'), view.get('alpha'), view.get('depth'), view.get('timestamp'), device=device)
        counts.append(len(batch))
        batches.append(batch.cpu() if store_on_cpu else batch)
    rays = RayBatch.cat(batches)
    return RayCollection(rays, tuple(counts)) if as_ray_collection else rays


class RayPoolSampler:
    """RayPoolSampler.get (DatasetSamplers.py:53-66): one shuffled permutation over ALL rays of the split (torch CPU RNG), consumed
    sequentially; with the same torch.manual_seed every data-parallel rank draws the same ids (parallel.shard_ray_ids splits them)."""

    def __init__(self, all_rays: RayBatch, img_sampler_cls=RandomSequentialSampler) -> None:
        self.all_rays = all_rays
        self.image_sampler = img_sampler_cls(num_elements=len(all_rays))

    def get(self, ray_batch_size: int, device=None) -> dict:
        ray_ids = self.image_sampler.get(ray_batch_size).to(self.all_rays.device)
        batch = self.all_rays[ray_ids]
        if device is not None:
            batch = batch.to(device=device)
        return {'sample_id': None, 'view': None, 'image_sampler': self.image_sampler, 'ray_ids': ray_ids, 'ray_batch': batch}
