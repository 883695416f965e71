"""Run-to-run noise of order-dependent f32 sums (float atomics), measured instead of guessed.

Several product kernels add with float atomics (dense hash-grid levels, weight-gradient flush, per-Gaussian sums across tiles): two runs
of the SAME code on the SAME inputs differ in the last bits of a gradient, and Adam with eps = 1e-15 turns a gradient that is rounding
noise into a full-size step of either sign.  A test that compares two formulations of one computation (replayed graph vs op by op, folded
vs loss-term weight decay) therefore has no fixed noise floor to put a threshold on: it runs the reference formulation TWICE, measures how
far those two runs are apart, and asks the other formulation to be no further away than a small multiple of that.
"""
from __future__ import annotations

import torch


def mismatch_fraction(a: torch.Tensor, b: torch.Tensor, atol: float, rtol: float) -> float:
    """fraction of the entries of b that lie outside atol + rtol*|a| of a"""
    return float(((a - b).abs() > atol + rtol * a.abs()).float().mean())


def assert_within_run_to_run_noise(test, ref, ref_again, *, atol: float, rtol: float, factor: float = 4.0, floor: float = 2e-3, what: str = ''):
    """test / ref / ref_again: lists of tensors.  Per tensor: mismatch(test, ref) <= factor * mismatch(ref_again, ref) + floor, and the BULK of
    the entries agrees (median relative deviation < 1e-4) -- a formulation that really differs (a missed step, a wrong batch, a stale
    buffer) moves most entries by about lr and fails the median whatever the noise."""
    for i, (t, r, r2) in enumerate(zip(test, ref, ref_again)):
        noise = mismatch_fraction(r, r2, atol, rtol)
        got = mismatch_fraction(r, t, atol, rtol)
        assert got <= factor * noise + floor, f'{what}[{i}]: {got:.5f} of the entries differ, run-to-run noise of the reference is {noise:.5f}'
        med = float(((r - t).abs() / (r.abs() + 1e-2)).flatten().float().quantile(0.5)) if r.numel() <= 2 ** 24 else \
            float(((r - t).abs() / (r.abs() + 1e-2)).flatten().float()[:: max(1, r.numel() // 2 ** 22)].quantile(0.5))
        assert med < 1e-4, f'{what}[{i}]: median relative deviation {med:.2e}'
