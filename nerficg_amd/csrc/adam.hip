// adam.hip -- fused Adam step: replaces apex.optimizers.FusedAdam as the reference constructs it
// (src/Methods/InstantNGP/Trainer.py:33-38: lr, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False;
//  src/Methods/GaussianSplatting/Model.py:131-136: six single-tensor parameter groups, eps=1e-15).
// Pure HBM streaming, 28 B per parameter (p, g, m, v read; p, m, v written); the GradScaler's 1/scale and its found-inf skip are
// folded in (device scalars, no host round trip), which removes the separate unscale pass of the reference's step.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

template <bool ALIGNED>
__global__ void __launch_bounds__(256) k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                                              float lr, float beta1, float beta2, float eps, float weight_decay, int adam_w_mode, float bc1, float bc2,
                                              const float* __restrict__ grad_scale, const float* __restrict__ found_inf) {
    if (found_inf && *found_inf != 0.f) return;  // GradScaler: skip the step, keep the state
    const float inv_scale = grad_scale ? 1.0f / *grad_scale : 1.0f;
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    float pv[4], gv[4], mv[4], vv[4];
    const bool full = ALIGNED && i0 + 3 < n;  // 16-byte vector access needs all four pointers aligned (views into larger tensors may not be)
    if (full) {
        *reinterpret_cast<float4*>(pv) = *reinterpret_cast<const float4*>(p + i0);
        *reinterpret_cast<float4*>(gv) = *reinterpret_cast<const float4*>(g + i0);
        *reinterpret_cast<float4*>(mv) = *reinterpret_cast<const float4*>(m + i0);
        *reinterpret_cast<float4*>(vv) = *reinterpret_cast<const float4*>(v + i0);
    } else {
        for (int k = 0; k < 4; k++) {
            const bool in = i0 + k < n;
            pv[k] = in ? p[i0 + k] : 0.f; gv[k] = in ? g[i0 + k] : 0.f; mv[k] = in ? m[i0 + k] : 0.f; vv[k] = in ? v[i0 + k] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        float gr = gv[k] * inv_scale;
        if (!adam_w_mode) gr += weight_decay * pv[k];  // L2 mode (apex multi_tensor_adam ADAM_MODE_0)
        mv[k] = beta1 * mv[k] + (1.f - beta1) * gr;
        vv[k] = beta2 * vv[k] + (1.f - beta2) * gr * gr;
        const float m_hat = mv[k] / bc1, v_hat = vv[k] / bc2;
        float update = m_hat / (sqrtf(v_hat) + eps);
        if (adam_w_mode) update += weight_decay * pv[k];
        pv[k] -= lr * update;
    }
    if (full) {
        *reinterpret_cast<float4*>(p + i0) = *reinterpret_cast<const float4*>(pv);
        *reinterpret_cast<float4*>(m + i0) = *reinterpret_cast<const float4*>(mv);
        *reinterpret_cast<float4*>(v + i0) = *reinterpret_cast<const float4*>(vv);
    } else {
        for (int k = 0; k < 4; k++)
            if (i0 + k < n) { p[i0 + k] = pv[k]; m[i0 + k] = mv[k]; v[i0 + k] = vv[k]; }
    }
}

}  // namespace

extern "C" {

int nrc_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int32_t adam_w_mode, float bias_correction1, float bias_correction2, const float* grad_scale,
                  const float* found_inf, nrc_stream_t stream) {
    NRC_ENTER();
    if (n < 0 || !(bias_correction1 > 0.f) || !(bias_correction2 > 0.f)) return NRC_ERR_INVALID;
    if (n == 0) return NRC_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return NRC_ERR_INVALID;
    const bool aligned = ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15u) == 0);
    const dim3 grid((unsigned)nrc_cdiv(nrc_cdiv(n, 4), 256));
    if (aligned)
        hipLaunchKernelGGL(k_adam<true>, grid, dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                           (int)adam_w_mode, bias_correction1, bias_correction2, grad_scale, found_inf);
    else
        hipLaunchKernelGGL(k_adam<false>, grid, dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                           (int)adam_w_mode, bias_correction1, bias_correction2, grad_scale, found_inf);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

}  // extern "C"
