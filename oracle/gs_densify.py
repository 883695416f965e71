"""oracle/gs_densify.py -- CPU restatement (numpy, f32) of the reference's 3DGS densification bookkeeping.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows the reference step by step -- clone, then split, then prune, each with its own
concatenations and mask copies -- so that it checks the single-plan formulation of nerficg_amd/csrc/gs_densify.hip independently:
  add_densification_stats   src/Methods/GaussianSplatting/Model.py:243-246
  duplicate                 Model.py:211-224
  split                     Model.py:186-209   (torch.normal(0, std) = z * std with z the standard-normal draws handed in as `noise`)
  densify_and_prune         Model.py:226-241
  prune_points / densification_postfix     Model.py:157-184
  prune_param_groups / extend_param_groups / sort_param_groups / replace_param_group_data / reset_state   src/Optim/adam_utils.py:6-98
Pinned by tests/golden/gs_densify.npz, produced by the reference's own Gaussians class and adam_utils run on CPU (tests/golden/make_golden.py).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
GROUPS = ('positions', 'f_dc', 'f_rest', 'opacities', 'scales', 'rotations')


def add_densification_stats(accum, n_obs, viewspace_grad, radii):
    """Model.py:243-246.  accum (P,1) f32, n_obs (P,1) i32, viewspace_grad (P,>=2), radii (P,) -> updated copies."""
    accum, n_obs = accum.copy(), n_obs.copy()
    vis = radii > 0
    g = viewspace_grad[vis, :2].astype(f32)
    accum[vis, 0] += np.sqrt(g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1], dtype=f32)
    n_obs[vis, 0] += 1
    return accum, n_obs


def _expit(x):
    return (f32(1) / (f32(1) + np.exp(-x.astype(f32), dtype=f32))).astype(f32)


def quaternion_to_rotation_matrix(q):
    """src/Cameras/utils.py:180-208 with normalisation (torch.nn.functional.normalize, eps 1e-12)."""
    q = q.astype(f32)
    n = np.maximum(np.sqrt((q * q).sum(axis=1, dtype=f32), dtype=f32), f32(1e-12))
    q = q / n[:, None]
    r, i, j, k = q.T
    two = f32(2)
    ii2, jj2, kk2 = i * i * two, j * j * two, k * k * two
    ij2, ik2, jk2 = i * j * two, i * k * two, j * k * two
    ri2, rj2, rk2 = r * i * two, r * j * two, r * k * two
    R = np.empty((q.shape[0], 3, 3), f32)
    R[:, 0, 0] = 1 - (jj2 + kk2); R[:, 0, 1] = ij2 - rk2; R[:, 0, 2] = ik2 + rj2
    R[:, 1, 0] = ij2 + rk2; R[:, 1, 1] = 1 - (ii2 + kk2); R[:, 1, 2] = jk2 - ri2
    R[:, 2, 0] = ik2 - rj2; R[:, 2, 1] = jk2 + ri2; R[:, 2, 2] = 1 - (ii2 + jj2)
    return R


class _Groups:
    """Six single-tensor parameter groups with their Adam moments (None = optimizer state not created yet)."""

    def __init__(self, params, moments):
        self.p = {k: np.array(params[k], f32) for k in GROUPS}
        self.m = None if moments is None else {k: (np.array(moments[k][0], f32), np.array(moments[k][1], f32)) for k in GROUPS}

    def extend(self, extra):  # adam_utils.py:42-61
        for k in GROUPS:
            e = np.asarray(extra[k], f32)
            self.p[k] = np.concatenate((self.p[k], e), axis=0)
            if self.m is not None:
                z = np.zeros_like(e)
                self.m[k] = (np.concatenate((self.m[k][0], z), axis=0), np.concatenate((self.m[k][1], z), axis=0))

    def prune(self, valid):  # adam_utils.py:21-39
        for k in GROUPS:
            self.p[k] = self.p[k][valid]
            if self.m is not None:
                self.m[k] = (self.m[k][0][valid], self.m[k][1][valid])

    def sort(self, order):  # adam_utils.py:81-98
        self.prune(order)


def densify_and_prune(params, moments, accum, n_obs, grad_threshold, min_opacity, prune_large, percent_dense, extent, noise):
    """Model.py:226-241.  params: {group: array}; moments: {group: (exp_avg, exp_avg_sq)} or None; noise: (2 * n_split, 3) standard-normal
    draws (n_split known only after classification: hand in at least that many rows, the first 2 * n_split are used the way
    torch.normal fills its (2 * n_split, 3) output).  Returns (params, moments, n_split)."""
    G = _Groups(params, moments)
    grads = (accum.astype(f32) / np.maximum(n_obs, 1).astype(f32)).astype(f32)  # (P, 1)
    dense_extent = f32(f32(percent_dense) * f32(extent))
    thr = f32(grad_threshold)

    # duplicate (Model.py:211-224)
    sel = np.abs(grads[:, 0]) >= thr
    sel &= np.exp(G.p['scales'], dtype=f32).max(axis=1) <= dense_extent
    G.extend({k: G.p[k][sel] for k in GROUPS})

    # split (Model.py:186-209)
    n_init = G.p['positions'].shape[0]
    padded = np.zeros(n_init, f32)
    padded[:grads.shape[0]] = grads[:, 0]
    sel = padded >= thr
    scales = np.exp(G.p['scales'], dtype=f32)
    sel &= scales.max(axis=1) > dense_extent
    n_split = int(sel.sum())
    stds = np.tile(scales[sel], (2, 1))
    samples = (np.asarray(noise, f32)[:2 * n_split] * stds).astype(f32)
    rots = np.tile(quaternion_to_rotation_matrix(G.p['rotations'][sel]), (2, 1, 1))
    new_pos = (rots[:, :, 0] * samples[:, None, 0] + rots[:, :, 1] * samples[:, None, 1] + rots[:, :, 2] * samples[:, None, 2]).astype(f32) \
        + np.tile(G.p['positions'][sel], (2, 1))
    extra = {k: np.tile(G.p[k][sel], (2,) + (1,) * (G.p[k].ndim - 1)) for k in GROUPS}
    extra['positions'] = new_pos.astype(f32)
    extra['scales'] = np.log(np.tile(scales[sel], (2, 1)) / f32(1.6), dtype=f32)
    G.extend(extra)
    G.prune(~np.concatenate((sel, np.zeros(2 * n_split, bool))))

    # final prune (Model.py:235-239)
    prune = _expit(G.p['opacities']).reshape(-1) < f32(min_opacity)
    if prune_large:
        prune |= np.exp(G.p['scales'], dtype=f32).max(axis=1) > f32(f32(0.1) * f32(extent))
    G.prune(~prune)
    return G.p, G.m, n_split


def prune(params, moments, valid_mask):
    G = _Groups(params, moments)
    G.prune(np.asarray(valid_mask, bool))
    return G.p, G.m


def sort(params, moments, order):
    G = _Groups(params, moments)
    G.sort(np.asarray(order))
    return G.p, G.m
