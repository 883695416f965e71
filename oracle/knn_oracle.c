/* oracle/knn_oracle.c -- brute-force statement of simple_knn's distCUDA2 as the reference uses it (src/Optim/knn_utils.py:34-38):
 * per point the mean of the three smallest squared Euclidean distances to OTHER points.  TEST INFRASTRUCTURE ONLY.  simple-knn is
 * not under /root/reference (parity unpinned against its binary); pinned against scikit-learn's exact NearestNeighbors -- the
 * reference's own fallback (knn_utils.py:24-27) -- in tests/test_oracle_knn.py. */
#include <math.h>
#include <stdint.h>

void oracle_knn3_mean_sq_dist(const float* p, int64_t n, float* out) {
#pragma omp parallel for
    for (int64_t i = 0; i < n; i++) {
        float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
        for (int64_t j = 0; j < n; j++) {
            if (j == i) continue;
            const float dx = p[3 * i] - p[3 * j], dy = p[3 * i + 1] - p[3 * j + 1], dz = p[3 * i + 2] - p[3 * j + 2];
            const float d = (dx * dx + dy * dy) + dz * dz;
            if (d < b2) {
                if (d < b1) { b2 = b1; if (d < b0) { b1 = b0; b0 = d; } else b1 = d; } else b2 = d;
            }
        }
        out[i] = (b0 + b1 + b2) / 3.0f;
    }
}
