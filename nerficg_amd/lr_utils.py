"""nerficg_amd.lr_utils -- learning-rate decay policy of src/Optim/lr_utils.py:8-35 (the 3DGS position group uses it, Model.py:138-150):
log-linear interpolation lr_init -> lr_final over max_steps, optionally eased in by a sine ramp over the first lr_delay_steps."""
from __future__ import annotations

import math
from dataclasses import dataclass

__all__ = ['LRDecayPolicy']


@dataclass(frozen=True)
class LRDecayPolicy:
    lr_init: float = 1.0
    lr_final: float = 1.0
    lr_delay_steps: int = 0
    lr_delay_mult: float = 1.0
    max_steps: int = 1_000_000

    def __call__(self, iteration: int) -> float:
        if iteration < 0 or (self.lr_init == 0.0 and self.lr_final == 0.0):
            return 0.0
        ramp = 1.0
        if self.lr_delay_steps > 0 and iteration < self.lr_delay_steps:
            phase = min(max(iteration / self.lr_delay_steps, 0.0), 1.0)
            ramp = self.lr_delay_mult + (1.0 - self.lr_delay_mult) * math.sin(0.5 * math.pi * phase)
        t = min(max(iteration / self.max_steps, 0.0), 1.0)
        return float(ramp * math.exp((1.0 - t) * math.log(self.lr_init) + t * math.log(self.lr_final)))
