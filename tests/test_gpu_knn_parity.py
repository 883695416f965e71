"""GPU: nerficg_amd.simple_knn.distCUDA2 (HIP, C ABI group 9) against oracle/knn_oracle.c -- exact neighbours, identical f32 distance
arithmetic, so the results are bit-equal."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


@pytest.mark.parametrize('n,clustered', [(4, False), (1000, False), (5000, True), (20000, True)])
def test_dist2_matches_oracle(n, clustered):
    from nerficg_amd.simple_knn import _C
    rng = np.random.default_rng(n)
    p = rng.normal(size=(n, 3)).astype(np.float32)
    if clustered:
        p[: n // 3] = p[: n // 3] * 0.001 + 3.0   # a tight cluster far from the rest + duplicates
        p[n // 2] = p[n // 2 + 1]
    got = _C.distCUDA2(torch.from_numpy(p).to(DEV)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.knn3_mean_sq_dist(p))


def test_dist2_errors_and_empty():
    from nerficg_amd.simple_knn import distCUDA2
    assert distCUDA2(torch.zeros(0, 3, device=DEV)).shape == (0,)
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(3, 3, device=DEV))
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(10, 3))
