"""nerficg_amd.samplers -- the index samplers that DEFINE which rays a training batch contains (parity contract for ray indices;
behaviour of src/Optim/Samplers/utils.py:8-56, pinned by the reference's own draws in tests/golden/misc.npz).

One cursor walks an index tensor front to back; a request that would run past the end rewinds first.  The random variant re-permutes
the CURRENT order with one `torch.randperm` from torch's CPU generator per rewind (so successive epochs compose permutations, like the
reference), which makes the draws a function of `torch.manual_seed` alone: every data-parallel rank draws the same batch and
nerficg_amd.parallel.shard_ray_ids splits it.  The incremental variant opens its window by one element per rewind."""
from __future__ import annotations

import torch

__all__ = ['SequentialSampler', 'RandomSequentialSampler', 'IncrementalSequentialSampler']


class _Cursor:
    """Shared mechanics: `indices` (the order), `current_id` (next position), `_limit()` (how much of `indices` is in play)."""

    permute_on_rewind = False

    def __init__(self, num_elements: int) -> None:
        self.num_elements = int(num_elements)
        self.indices = torch.arange(self.num_elements)
        self.current_id = 0
        self.reset()

    def _limit(self) -> int:
        return self.num_elements

    def _window(self) -> torch.Tensor:
        return self.indices

    def shuffle(self) -> None:
        if self.permute_on_rewind:
            self.indices = self.indices.index_select(0, torch.randperm(self.num_elements))

    def reset(self) -> None:
        self.current_id = 0
        self.shuffle()

    def get(self, num_samples: int) -> torch.Tensor:
        want = int(num_samples)
        if want > self._limit():
            raise RuntimeError(f'cannot draw {want} samples from {self._limit()} elements')  # Framework.SamplerError in the reference
        stop = self.current_id + want
        if stop > self._limit():
            self.reset()
            stop = want
        start, self.current_id = stop - want, stop
        return self._window()[start:stop]


class SequentialSampler(_Cursor):
    """0, 1, 2, ... in order, wrapping to 0."""


class RandomSequentialSampler(_Cursor):
    """A fresh permutation of the previous order at construction and at every wrap."""
    permute_on_rewind = True


class IncrementalSequentialSampler(_Cursor):
    """Serves indices [0, size) in order; size starts at 1 and grows by one at every rewind until it covers all elements."""

    def __init__(self, num_elements: int) -> None:
        self.current_size = 0
        super().__init__(num_elements)

    def _limit(self) -> int:
        return self.current_size

    def _window(self) -> torch.Tensor:
        return self.indices[:self.current_size]

    def reset(self) -> None:
        self.current_size = min(self.current_size + 1, self.num_elements)
        self.current_id = 0
