"""nerficg_amd.lr_utils -- learning-rate decay policy with the call signature of src/Optim/lr_utils.py:8-35 (`LRDecayPolicy(lr_init, lr_final,
lr_delay_steps, lr_delay_mult, max_steps)(iteration) -> float`; the 3DGS position group uses it, Model.py:138-150), pinned on the reference's
values in tests/golden/misc.npz.

The schedule is a straight line in log space from lr_init (iteration 0) to lr_final (iteration max_steps and beyond); during the first
lr_delay_steps iterations it is additionally scaled by a quarter sine wave that rises from lr_delay_mult to 1."""
from __future__ import annotations

import math

__all__ = ['LRDecayPolicy']


def _unit(x: float) -> float:
    return 0.0 if x < 0.0 else (1.0 if x > 1.0 else x)


class LRDecayPolicy:
    __slots__ = ('lr_init', 'lr_final', 'lr_delay_steps', 'lr_delay_mult', 'max_steps', '_log_a', '_log_b')

    def __init__(self, lr_init: float = 1.0, lr_final: float = 1.0, lr_delay_steps: int = 0, lr_delay_mult: float = 1.0, max_steps: int = 1_000_000) -> None:
        self.lr_init, self.lr_final = float(lr_init), float(lr_final)
        self.lr_delay_steps, self.lr_delay_mult, self.max_steps = int(lr_delay_steps), float(lr_delay_mult), int(max_steps)
        disabled = self.lr_init == 0.0 and self.lr_final == 0.0
        self._log_a = None if disabled else math.log(self.lr_init)
        self._log_b = None if disabled else math.log(self.lr_final)

    def warmup_factor(self, iteration: int) -> float:
        if self.lr_delay_steps <= 0 or iteration >= self.lr_delay_steps:
            return 1.0
        rise = math.sin(0.5 * math.pi * _unit(iteration / self.lr_delay_steps))
        return self.lr_delay_mult + (1.0 - self.lr_delay_mult) * rise

    def __call__(self, iteration: int) -> float:
        if iteration < 0 or self._log_a is None:
            return 0.0  # a negative iteration or an all-zero schedule switches the parameter group off
        t = _unit(iteration / self.max_steps)
        return float(self.warmup_factor(iteration) * math.exp(self._log_a * (1.0 - t) + self._log_b * t))

    def __repr__(self) -> str:
        return (f'LRDecayPolicy(lr_init={self.lr_init}, lr_final={self.lr_final}, lr_delay_steps={self.lr_delay_steps}, '
                f'lr_delay_mult={self.lr_delay_mult}, max_steps={self.max_steps})')
