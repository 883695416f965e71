"""CPU: pins oracle/knn_oracle.c against scikit-learn's exact nearest neighbours, the reference's own fallback (knn_utils.py:24-27)."""
import numpy as np
from sklearn.neighbors import NearestNeighbors

import oracle


def test_knn_oracle_matches_sklearn():
    rng = np.random.default_rng(0)
    p = np.concatenate([rng.normal(size=(1500, 3)), rng.normal(size=(500, 3)) * 0.01 + 2.0]).astype(np.float32)
    d, _ = NearestNeighbors(n_neighbors=3).fit(p.astype(np.float64)).kneighbors()
    ref = np.square(d).mean(axis=-1)
    np.testing.assert_allclose(oracle.knn3_mean_sq_dist(p), ref, rtol=2e-5, atol=1e-12)
