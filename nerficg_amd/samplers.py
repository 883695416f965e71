"""nerficg_amd.samplers -- the index samplers that DEFINE which rays a training batch contains (parity contract for ray indices):
src/Optim/Samplers/utils.py:8-34.  The permutation comes from torch's CPU generator (`torch.randperm`), consumed sequentially and
reshuffled when a request would run past the end; with the same `torch.manual_seed` every data-parallel rank draws the same batch
(nerficg_amd.parallel.shard_ray_ids then splits it)."""
from __future__ import annotations

import torch

__all__ = ['SequentialSampler', 'RandomSequentialSampler']


class SequentialSampler:
    def __init__(self, num_elements: int) -> None:
        self.num_elements = num_elements
        self.indices = torch.arange(num_elements)
        self.reset()

    def shuffle(self) -> None:
        pass

    def reset(self) -> None:
        self.current_id = 0
        self.shuffle()

    def get(self, num_samples: int) -> torch.Tensor:
        if num_samples > self.num_elements:
            raise RuntimeError(f'cannot draw {num_samples} samples from {self.num_elements} elements')
        if self.current_id + num_samples > self.num_elements:
            self.reset()
        out = self.indices[self.current_id:self.current_id + num_samples]
        self.current_id += num_samples
        return out


class RandomSequentialSampler(SequentialSampler):
    def shuffle(self) -> None:
        self.indices = self.indices[torch.randperm(self.num_elements)]
