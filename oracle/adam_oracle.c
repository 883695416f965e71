/* oracle/adam_oracle.c -- CPU restatement of one Adam step as the reference's optimizers perform it
 * (apex.optimizers.FusedAdam with adam_w_mode=False, or its fallback torch.optim.Adam(fused=True): src/Methods/InstantNGP/Trainer.py:33-38,
 * src/Methods/GaussianSplatting/Model.py:131-136).  TEST INFRASTRUCTURE ONLY.  Apex is not under /root/reference (parity unpinned
 * against its binary); the restatement is pinned against torch.optim.Adam in tests/test_oracle_adam.py.
 *   g' = g / scale (+ wd p in L2 mode);  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;  p -= lr (m / bc1) / (sqrt(v / bc2) + eps)   */
#include <math.h>
#include <stdint.h>

void oracle_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps, float weight_decay,
                      int64_t adam_w_mode, float bc1, float bc2, float grad_scale, int64_t found_inf) {
    if (found_inf) return;
    const float inv_scale = 1.0f / grad_scale;
    for (int64_t i = 0; i < n; i++) {
        float gr = g[i] * inv_scale;
        if (!adam_w_mode) gr += weight_decay * p[i];
        m[i] = beta1 * m[i] + (1.f - beta1) * gr;
        v[i] = beta2 * v[i] + (1.f - beta2) * gr * gr;
        const float m_hat = m[i] / bc1, v_hat = v[i] / bc2;
        float update = m_hat / (sqrtf(v_hat) + eps);
        if (adam_w_mode) update += weight_decay * p[i];
        p[i] -= lr * update;
    }
}
