// adam_math.h -- the per-element Adam update of adam.hip's kernels (k_adam, k_amp_adam): ONE statement of the arithmetic, built without FMA
// contraction wherever it is included, so that every path leaves bit-identical parameters and moments (oracle/adam_oracle.c states the same sequence).
#pragma once
#include <hip/hip_runtime.h>

struct NrcAdamHyper {
    float lr, beta1, beta2, eps, weight_decay;
    int adam_w_mode;
    float bc1, bc2, inv_scale;
};
// p, m, v updated in place from the (scaled) gradient g; l2_coeff: extra L2 term on this element (0 = none)
__device__ __forceinline__ void nrc_adam_update(float& p, float g, float& m, float& v, const NrcAdamHyper& h, bool l2, float l2_coeff) {
#pragma clang fp contract(off)
    float gr = g * h.inv_scale;
    if (!h.adam_w_mode) gr += h.weight_decay * p;  // L2 mode (apex multi_tensor_adam ADAM_MODE_0)
    if (l2) gr += l2_coeff * p;                     // L2 term of a leading slice only (the MLP weights in front of a hash table)
    m = h.beta1 * m + (1.f - h.beta1) * gr;
    v = h.beta2 * v + (1.f - h.beta2) * gr * gr;
    const float m_hat = m / h.bc1, v_hat = v / h.bc2;
    float update = m_hat / (sqrtf(v_hat) + h.eps);
    if (h.adam_w_mode) update += h.weight_decay * p;
    p -= h.lr * update;
}

