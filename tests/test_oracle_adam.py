"""CPU: pins oracle/adam_oracle.c against torch.optim.Adam (the reference's own fallback when apex is absent, Trainer.py:37-38)."""
import numpy as np
import torch

import oracle


def test_adam_oracle_matches_torch_adam_over_several_steps():
    rng = np.random.default_rng(0)
    p0 = rng.normal(size=1000).astype(np.float32)
    tp = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([tp], lr=1e-2, eps=1e-15, betas=(0.9, 0.99))
    p, m, v = p0.copy(), np.zeros_like(p0), np.zeros_like(p0)
    for step in range(1, 6):
        g = (rng.normal(size=1000) * 10.0 ** rng.integers(-6, 1, size=1000)).astype(np.float32)
        tp.grad = torch.from_numpy(g.copy())
        opt.step()
        p, m, v = oracle.adam_step(p, g, m, v, step, 1e-2, (0.9, 0.99), 1e-15)
        np.testing.assert_allclose(p, tp.detach().numpy(), rtol=2e-6, atol=1e-7)
    st = opt.state[tp]
    np.testing.assert_allclose(m, st['exp_avg'].numpy(), rtol=1e-6, atol=1e-7 * np.abs(m).max())  # torch uses lerp: m + (1-b1)(g-m)
    np.testing.assert_allclose(v, st['exp_avg_sq'].numpy(), rtol=1e-6, atol=1e-7 * np.abs(v).max())
    # weight decay (L2 mode) + found_inf skip
    q, _, _ = oracle.adam_step(p, g, m, v, 6, 1e-2, weight_decay=0.1, found_inf=True)
    np.testing.assert_array_equal(q, p)
