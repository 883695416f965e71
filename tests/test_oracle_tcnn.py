"""Pins for oracle/tcnn_oracle.c (CPU).  tiny-cuda-nn is not under /root/reference (parity unpinned, see the file header);
what is pinned: the level geometry recorded in SURVEY.md Appendix C.1, the SH basis against the reference's own
convert_sh_features (which cites tiny-cuda-nn as its source, src/Methods/GaussianSplatting/utils.py:26-31), and the internal
consistency of forward/backward (numpy matmul, finite differences)."""
import numpy as np

import oracle

PLS = float(np.exp(np.log(2048 * 1.0 / 16) / 15))
RNG = np.random.default_rng(3)


def test_grid_layout_matches_survey_appendix_c1():
    total, offsets, scales, res = oracle.grid_layout(16, 19, 16, PLS)
    assert list(res) == [16, 23, 31, 43, 59, 81, 112, 154, 213, 295, 407, 562, 777, 1073, 1483, 2048]
    sizes = np.diff(offsets.astype(np.int64))
    assert list(sizes[:5]) == [4096, 12168, 29792, 79512, 205384] and np.all(sizes[5:] == 2 ** 19)
    assert total == 6098120 and 2 * total == 12196240


def test_grid_encoding_interpolates_and_is_differentiable():
    total, offsets, scales, res = oracle.grid_layout(16, 19, 16, PLS)
    table = oracle.round_half(RNG.uniform(-1, 1, size=(total, 2)))
    # at a dense level's lattice point (x*scale + 0.5 integer) the feature equals the table entry (trilinear weights 1,0,...)
    g = np.array([3, 5, 7])
    x = ((g + 0.0 - 0.5) / scales[0]).astype(np.float32)[None]
    enc = oracle.grid_encode_fw(x, table, 16, 19, 16, PLS)
    idx = offsets[0] + g[0] + g[1] * res[0] + g[2] * res[0] ** 2
    np.testing.assert_allclose(enc[0, :2], table[idx], rtol=0, atol=2e-3)
    # backward = transpose of forward: <enc(x), d> == <table, scatter(d)>
    xs = RNG.random((50, 3)).astype(np.float32)
    d = RNG.normal(size=(50, 32)).astype(np.float32)
    t32 = RNG.uniform(-1, 1, size=(total, 2)).astype(np.float32)
    t16 = oracle.round_half(t32)
    lhs = (oracle.grid_encode_fw(xs, t16, 16, 19, 16, PLS).astype(np.float64) * d).sum()
    rhs = (t16.astype(np.float64) * oracle.grid_encode_bw(xs, d, total, 16, 19, 16, PLS)).sum()
    assert abs(lhs - rhs) <= 2e-3 * abs(lhs) + 1e-3


def test_sh4_basis_matches_reference_convert_sh_features(golden_dir):
    g = np.load(golden_dir / 'gs_utils.npz')
    vd = g['view_dirs']
    basis = oracle.sh4_encode(vd * 0.5 + 0.5)  # (n,16) fp16-rounded
    # reference: rgb = 0.5 + sum_k B_k(dir) * sh_k (before the clamp) -> compare where no channel is clamped
    sh = g['sh']  # (n,3,16)
    ref = g['rgb_deg3']
    mine = 0.5 + np.einsum('nk,nck->nc', basis.astype(np.float64), sh.astype(np.float64))
    ok = (ref > 1e-3).all(-1)
    np.testing.assert_allclose(mine[ok], ref[ok], rtol=0, atol=4e-3)  # fp16 rounding of the 16 basis values


def test_mlp_forward_is_a_relu_mlp_and_backward_matches_fd():
    m = 40
    W = oracle.round_half(RNG.normal(size=64 * 32 + 64 * 64 + 16 * 64) * 0.2)
    x = oracle.round_half(RNG.normal(size=(m, 32)) * 0.5)
    out, acts = oracle.mlp_fw(x, W, n_hidden=2, out_act=1, want_acts=True)
    W0, W1, W2 = W[:2048].reshape(64, 32), W[2048:2048 + 4096].reshape(64, 64), W[6144:].reshape(16, 64)
    h0 = np.maximum(x @ W0.T, 0).astype(np.float16).astype(np.float32)
    h1 = np.maximum(h0 @ W1.T, 0).astype(np.float16).astype(np.float32)
    ref = 1 / (1 + np.exp(-(h1 @ W2.T)))
    np.testing.assert_allclose(out, ref, rtol=0, atol=2e-3)
    # backward vs finite differences of a smooth surrogate (no fp16 rounding): compare against autograd-free analytic formula
    g = oracle.round_half(RNG.normal(size=(m, 16)) * 0.1)
    dW, d_in = oracle.mlp_bw(x, W, out, acts, g, n_hidden=2, out_act=1)
    dz2 = g * out * (1 - out)
    np.testing.assert_allclose(dW[6144:].reshape(16, 64), dz2.T @ acts[1], rtol=0, atol=2e-3 * np.abs(dW).max())
    dh1 = (dz2 @ W2) * (acts[1] > 0)
    np.testing.assert_allclose(dW[2048:6144].reshape(64, 64), dh1.T @ acts[0], rtol=0, atol=4e-3 * np.abs(dW).max())
    dh0 = (dh1 @ W1) * (acts[0] > 0)
    np.testing.assert_allclose(dW[:2048].reshape(64, 32), dh0.T @ x, rtol=0, atol=4e-3 * np.abs(dW).max())
    np.testing.assert_allclose(d_in, dh0 @ W0, rtol=0, atol=4e-3 * np.abs(d_in).max())


def test_round_half_matches_numpy_float16():
    v = np.concatenate([RNG.normal(size=2000) * 10.0 ** RNG.integers(-9, 6, size=2000), [0.0, -0.0, 65504.0, 65520.0, 1e-8, 6e-8, 5.96e-8]]).astype(np.float32)
    out = np.empty_like(v)
    oracle._call('oracle_round_to_half', v, v.size, out)
    with np.errstate(over='ignore'):
        np.testing.assert_array_equal(out, v.astype(np.float16).astype(np.float32))
