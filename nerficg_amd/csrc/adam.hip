// adam.hip -- fused Adam step: replaces apex.optimizers.FusedAdam as the reference constructs it
// (src/Methods/InstantNGP/Trainer.py:33-38: lr, eps=1e-15, betas=(0.9, 0.99), adam_w_mode=False;
//  src/Methods/GaussianSplatting/Model.py:131-136: six single-tensor parameter groups, eps=1e-15).
// Pure HBM streaming, 28 B per parameter (p, g, m, v read; p, m, v written); the GradScaler's 1/scale and its found-inf skip are
// folded in (device scalars, no host round trip), which removes the separate unscale pass of the reference's step.
// Optional extras of the step: (a) the fp16 compute copy of the parameters that the tinycudann module reads is written by the same
// kernel (2 B per parameter more, instead of a separate 6 B per parameter conversion pass before the next forward); (b) the bias
// corrections can come from a device pair written by k_adam_prepare, which holds the step counter back on the device whenever the
// GradScaler found an overflow (apex: the scaler then skips optimizer.step() altogether, so the count must not advance).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "common.h"
#include "adam_math.h"

namespace {

__global__ void k_adam_prepare(int32_t host_step, float beta1, float beta2, const float* __restrict__ found_inf, int32_t* __restrict__ skipped,
                               int32_t* __restrict__ device_step, float* __restrict__ bc) {
    if (threadIdx.x | blockIdx.x) return;
    const bool overflow = found_inf && *found_inf != 0.f;
    int32_t step;
    if (device_step) {  // capturable mode: the count itself lives on the device and stands still on a skipped step
        step = *device_step;
        if (!overflow) *device_step = ++step;
    } else {
        int32_t sk = *skipped;
        if (overflow) *skipped = ++sk;
        step = host_step - sk;
    }
    if (step < 1) step = 1;
    bc[0] = (float)(1.0 - pow((double)beta1, (double)step));
    bc[1] = (float)(1.0 - pow((double)beta2, (double)step));
}

// four consecutive elements starting at i0 (the body of both Adam kernels)
// GUARD (k_amp_adam): an element whose gradient is inf / NaN is left alone (parameter, moments, fp16 copy) and *guard_flag is raised.  The fused training step takes
// found_inf from its producers' flags (state4[0]) without a pass over the summed gradients; a sum of finite values that overflows behind those flags -- it needs
// |g| ~ 1e38 -- would otherwise enter the moments as inf and leave NaN parameters behind (advisor, round 5).  Bit-identical for finite gradients.
template <bool ALIGNED, bool GUARD = false>
__device__ __forceinline__ void adam_four(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                                          int64_t i0, float lr, float beta1, float beta2, float eps, float weight_decay, int adam_w_mode, float bc1, float bc2,
                                          float inv_scale, __half* __restrict__ p16, float l2_coeff, int64_t l2_count, float* __restrict__ guard_flag = nullptr) {
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
    const NrcAdamHyper h{lr, beta1, beta2, eps, weight_decay, adam_w_mode, bc1, bc2, inv_scale};
    if constexpr (GUARD) {
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool finite = (__float_as_uint(gv[k]) & 0x7f800000u) != 0x7f800000u;
            if (finite) nrc_adam_update(pv[k], gv[k], mv[k], vv[k], h, i0 + k < l2_count, l2_coeff);
            bad = bad || !finite;
        }
        if (bad && guard_flag) *guard_flag = 1.0f;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) nrc_adam_update(pv[k], gv[k], mv[k], vv[k], h, i0 + k < l2_count, l2_coeff);
    }
    if (full) {
        *reinterpret_cast<float4*>(p + i0) = *reinterpret_cast<const float4*>(pv);
        *reinterpret_cast<float4*>(m + i0) = *reinterpret_cast<const float4*>(mv);
        *reinterpret_cast<float4*>(v + i0) = *reinterpret_cast<const float4*>(vv);
    } else {
        for (int k = 0; k < 4; k++)
            if (i0 + k < n) { p[i0 + k] = pv[k]; m[i0 + k] = mv[k]; v[i0 + k] = vv[k]; }
    }
    if (p16) {  // same rounding as k_f32_to_f16 (round to nearest even)
        if (full && (((uintptr_t)(p16 + i0)) & 7u) == 0) {
            __half2 h[2] = {__floats2half2_rn(pv[0], pv[1]), __floats2half2_rn(pv[2], pv[3])};
            *reinterpret_cast<uint2*>(p16 + i0) = *reinterpret_cast<const uint2*>(h);
        } else {
            for (int k = 0; k < 4; k++)
                if (i0 + k < n) p16[i0 + k] = __float2half_rn(pv[k]);
        }
    }
}

template <bool ALIGNED>
__global__ void __launch_bounds__(256) k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int64_t n,
                                              float lr, float beta1, float beta2, float eps, float weight_decay, int adam_w_mode, float bc1, float bc2,
                                              const float* __restrict__ bc_dev, const float* __restrict__ lr_dev, const float* __restrict__ grad_scale,
                                              const float* __restrict__ found_inf, __half* __restrict__ p16, float l2_coeff, int64_t l2_count) {
    if (found_inf && *found_inf != 0.f) return;  // GradScaler: skip the step, keep the state (and the fp16 copy, which still matches)
    if (bc_dev) { bc1 = bc_dev[0]; bc2 = bc_dev[1]; }
    if (lr_dev) lr = *lr_dev;
    const float inv_scale = grad_scale ? 1.0f / *grad_scale : 1.0f;
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    adam_four<ALIGNED>(p, g, m, v, n, i0, lr, beta1, beta2, eps, weight_decay, adam_w_mode, bc1, bc2, inv_scale, p16, l2_coeff, l2_count);
}

// ---- several tensors (parameter groups with their own learning rate / step count) in ONE launch: what apex's multi_tensor_apply does for FusedAdam over the six
// single-tensor groups of src/Methods/GaussianSplatting/Model.py:121-138 (positions, dc, rest, opacity, scaling, rotation).  Six launches of 9-250 us cost their
// tails: the five small tensors (14 M of 59 M floats) ran at 4.2 TB/s, the one launch streams all of them at the rate of the big one.
#define ADAM_MULTI_MAX 12
struct MultiTensor { float* p; const float* g; float* m; float* v; int64_t n; float lr, bc1, bc2; int block0; };
struct MultiList { MultiTensor t[ADAM_MULTI_MAX]; int n; };
__global__ void __launch_bounds__(256) k_adam_multi(MultiList l, float beta1, float beta2, float eps, float weight_decay, int adam_w_mode) {
    int k = 0;
#pragma unroll
    for (int q = 1; q < ADAM_MULTI_MAX; q++) k += (q < l.n && (int)blockIdx.x >= l.t[q].block0) ? 1 : 0;   // block0 ascending: the last tensor that starts at or before this block
    const MultiTensor t = l.t[k];
    const int64_t i0 = ((int64_t)((int)blockIdx.x - t.block0) * 256 + threadIdx.x) * 4;
    if (i0 >= t.n) return;
    const bool aligned = ((((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15u) == 0);   // uniform per tensor
    if (aligned) adam_four<true>(t.p, t.g, t.m, t.v, t.n, i0, t.lr, beta1, beta2, eps, weight_decay, adam_w_mode, t.bc1, t.bc2, 1.0f, nullptr, 0.f, 0);
    else adam_four<false>(t.p, t.g, t.m, t.v, t.n, i0, t.lr, beta1, beta2, eps, weight_decay, adam_w_mode, t.bc1, t.bc2, 1.0f, nullptr, 0.f, 0);
}

// ---- GradScaler.step + FusedAdam.step + GradScaler.update of one parameter group as two launches (include/nerficg_hip.h: nrc_amp_adam_step) -------
struct AmpTensor { float* p; const float* g; float* m; float* v; __half* p16; int64_t n; float l2_coeff; int64_t l2_count; };
struct AmpList { AmpTensor t[2]; };
struct AmpState {
    int32_t* device_step; float* bc;          // step counter (advanced unless an overflow was found), bias corrections (2)
    int32_t* skipped;                         // optional: counts the overflow-skipped steps (FusedAdam's non-capturable bookkeeping)
    float* scale; int32_t* growth_tracker;    // the GradScaler's scalars (scale may be NULL: no scaler)
    float growth_factor, backoff_factor; int32_t growth_interval;
    float beta1, beta2;
    float* state;                             // f32[4]: [0] raw flag of the check (0 on entry, 0 on exit), [1] found_inf of this step, [2] 1 / scale of this step
    uint32_t* ticket;
    float grad_divisor;                       // the gradients are a SUM over this many ranks (data parallel): state[2] = 1 / (scale * divisor); 1 otherwise
};
// what ONE thread does once it is known whether the gradients are finite (state[0]): k_adam_prepare (capturable form) + torch's amp_update_scale kernel
__device__ __forceinline__ void amp_prepare_thread(const AmpState& a) {
    const bool overflow = __hip_atomic_load(&a.state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0.f;
    int32_t step = *a.device_step;
    if (!overflow) *a.device_step = ++step;
    else if (a.skipped) *a.skipped += 1;
    if (step < 1) step = 1;
    a.bc[0] = (float)(1.0 - pow((double)a.beta1, (double)step));
    a.bc[1] = (float)(1.0 - pow((double)a.beta2, (double)step));
    a.state[1] = overflow ? 1.f : 0.f;
    float scale = a.scale ? *a.scale : 1.f;
    a.state[2] = 1.0f / (scale * a.grad_divisor);   // (divisor a power of two: the same bits as dividing the gradient first)
    if (a.scale) {   // torch/aten/src/ATen/native/cuda/AmpKernels.cu, amp_update_scale_cuda_kernel: back off on an overflow, grow after growth_interval clean steps
        if (overflow) { *a.scale = scale * a.backoff_factor; *a.growth_tracker = 0; }
        else {
            const int32_t ok = *a.growth_tracker + 1;
            if (ok == a.growth_interval) {
                const float grown = scale * a.growth_factor;
                if (isfinite(grown)) *a.scale = grown;
                *a.growth_tracker = 0;
            } else *a.growth_tracker = ok;
        }
    }
    __hip_atomic_store(&a.state[0], 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (1) k_nonfinite_check4 over the gradients; the last workgroup to finish is k_adam_prepare (capturable form) + torch's amp_update_scale kernel
__global__ void __launch_bounds__(256) k_amp_check_prepare(AmpList l, AmpState a) {
    const uint32_t* __restrict__ g = reinterpret_cast<const uint32_t*>(l.t[blockIdx.y].g);
    const int64_t n = l.t[blockIdx.y].n;
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(g) & 15u) == 0) ? n / 4 : 0;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const uint4 q = reinterpret_cast<const uint4*>(g)[i];
        bad |= ((q.x & 0x7f800000u) == 0x7f800000u) | ((q.y & 0x7f800000u) == 0x7f800000u) | ((q.z & 0x7f800000u) == 0x7f800000u) |
               ((q.w & 0x7f800000u) == 0x7f800000u);
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) bad |= (g[i] & 0x7f800000u) == 0x7f800000u;
    const bool any = __syncthreads_or(bad);
    if (threadIdx.x == 0 && any) __hip_atomic_store(&a.state[0], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!nrc_last_workgroup(a.ticket, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y) || threadIdx.x != 0) return;
    amp_prepare_thread(a);
}
// the same closing step as a launch of its own: for callers whose PRODUCERS flag non-finite gradients (state[0]) -- nrc_ngp_train_backward_step
__global__ void k_amp_prepare(AmpState a) {
    if (threadIdx.x | blockIdx.x) return;
    amp_prepare_thread(a);
}
// ---- 16-bit wire of the sharded step's reduce-scatter (optional): f32 gradient -> fp16, SATURATING at +-65504 (a finite f32 value must not become an inf
// that no producer flagged; NaN stays NaN), and the reduced shard back to f32 for the Adam launch.  tiny-cuda-nn itself keeps hash-grid gradients in fp16
// under the same loss scale; here it is the wire format only -- gradients are produced, and the moments kept, in f32.
__global__ void __launch_bounds__(256) k_wire_pack_f16(const float* __restrict__ src, __half* __restrict__ dst, int64_t n, unsigned long long* __restrict__ saturated) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    float v[4];
    const bool full = i + 3 < n && ((reinterpret_cast<uintptr_t>(src + i) & 15u) == 0) && ((reinterpret_cast<uintptr_t>(dst + i) & 7u) == 0);
    if (full) *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(src + i);
    else for (int k = 0; k < 4; k++) v[k] = i + k < n ? src[i + k] : 0.f;
    int sat = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float c = fminf(fmaxf(v[k], -65504.f), 65504.f);      // (fminf / fmaxf return the other operand for a NaN: restore it)
        sat += (c != v[k] && v[k] == v[k]) ? 1 : 0;
        v[k] = v[k] == v[k] ? c : v[k];
    }
    if (full) {
        __half2 h[2] = {__floats2half2_rn(v[0], v[1]), __floats2half2_rn(v[2], v[3])};
        *reinterpret_cast<uint2*>(dst + i) = *reinterpret_cast<const uint2*>(h);
    } else {
        for (int k = 0; k < 4; k++) if (i + k < n) dst[i + k] = __float2half_rn(v[k]);
    }
    if (saturated && sat) atomicAdd(saturated, (unsigned long long)sat);
}
__global__ void __launch_bounds__(256) k_wire_unpack_f16(const __half* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = __half2float(src[i]);
}

// the closing step for a data-parallel rank: the flag is a SUM of the ranks' flags that arrived with the small all-reduce (nrc_amp_settle)
__global__ void k_amp_settle(AmpState a, const float* __restrict__ flag) {
    if (threadIdx.x | blockIdx.x) return;
    const float f = *flag;
    __hip_atomic_store(&a.state[0], (f != 0.f) ? 1.f : 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // NaN != 0 as well: a poisoned flag is an overflow
    amp_prepare_thread(a);
}
// (2) Adam on both tensors (blockIdx.y), everything it needs in device scalars
__global__ void __launch_bounds__(256) k_amp_adam(AmpList l, const float* __restrict__ state, const float* __restrict__ bc, const float* __restrict__ lr_dev,
                                                  float lr, float beta1, float beta2, float eps, float weight_decay, int adam_w_mode) {
    if (state[1] != 0.f) return;
    const AmpTensor t = l.t[blockIdx.y];
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= t.n) return;
    if (lr_dev) lr = *lr_dev;
    const bool aligned = ((((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15u) == 0);   // uniform per tensor
    float* guard = const_cast<float*>(state) + 3;      // state4[3]: sticky "an inf / NaN gradient reached the Adam launch and was left out" (the caller looks when it likes)
    if (aligned) adam_four<true, true>(t.p, t.g, t.m, t.v, t.n, i0, lr, beta1, beta2, eps, weight_decay, adam_w_mode, bc[0], bc[1], state[2], t.p16, t.l2_coeff, t.l2_count, guard);
    else adam_four<false, true>(t.p, t.g, t.m, t.v, t.n, i0, lr, beta1, beta2, eps, weight_decay, adam_w_mode, bc[0], bc[1], state[2], t.p16, t.l2_coeff, t.l2_count, guard);
}

// found_inf |= any element of g is inf / NaN.  One 16-byte load per lane and round, exponent test on the raw bits (all ones = inf or NaN), one
// store per workgroup that saw one -- the GradScaler's check of a 48.8 MB gradient in ~10 us (torch's multi-tensor unscale-and-check pass: 37 us).
__global__ void __launch_bounds__(256) k_nonfinite_check(const uint32_t* __restrict__ g, int64_t n, float* __restrict__ found_inf) {
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(g) & 15u) == 0) ? n / 4 : 0;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const uint4 q = reinterpret_cast<const uint4*>(g)[i];
        bad |= ((q.x & 0x7f800000u) == 0x7f800000u) | ((q.y & 0x7f800000u) == 0x7f800000u) | ((q.z & 0x7f800000u) == 0x7f800000u) |
               ((q.w & 0x7f800000u) == 0x7f800000u);
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) bad |= (g[i] & 0x7f800000u) == 0x7f800000u;
    if (__syncthreads_or(bad) && threadIdx.x == 0) *found_inf = 1.0f;
}

// the same over up to four tensors in one launch: a workgroup row (blockIdx.y) per tensor
struct FiniteList { const uint32_t* g[4]; int64_t n[4]; };
__global__ void __launch_bounds__(256) k_nonfinite_check4(FiniteList l, float* __restrict__ found_inf) {
    const uint32_t* __restrict__ g = l.g[blockIdx.y];
    const int64_t n = l.n[blockIdx.y];
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(g) & 15u) == 0) ? n / 4 : 0;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const uint4 q = reinterpret_cast<const uint4*>(g)[i];
        bad |= ((q.x & 0x7f800000u) == 0x7f800000u) | ((q.y & 0x7f800000u) == 0x7f800000u) | ((q.z & 0x7f800000u) == 0x7f800000u) |
               ((q.w & 0x7f800000u) == 0x7f800000u);
    }
    for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) bad |= (g[i] & 0x7f800000u) == 0x7f800000u;
    if (__syncthreads_or(bad) && threadIdx.x == 0) *found_inf = 1.0f;
}

}  // namespace

// internal launchers for nrc_ngp_train_backward_step (ngp_net.hip); declared in common.h
void nrc_launch_amp_prepare(int32_t* device_step, float* bias_corrections, float* scale, int32_t* growth_tracker, float growth_factor, float backoff_factor,
                            int32_t growth_interval, float beta1, float beta2, float* state4, hipStream_t s) {
    AmpState a;
    a.device_step = device_step; a.bc = bias_corrections; a.scale = scale; a.growth_tracker = growth_tracker; a.growth_factor = growth_factor;
    a.backoff_factor = backoff_factor; a.growth_interval = growth_interval; a.beta1 = beta1; a.beta2 = beta2; a.state = state4; a.ticket = nullptr; a.skipped = nullptr;
    a.grad_divisor = 1.f;
    hipLaunchKernelGGL(k_amp_prepare, dim3(1), dim3(64), 0, s, a);
}
void nrc_launch_amp_adam(float* pa, const float* ga, float* ma, float* va, void* ha, int64_t na, float l2c_a, int64_t l2n_a, float* pb, const float* gb, float* mb,
                         float* vb, void* hb, int64_t nb, float l2c_b, int64_t l2n_b, const float* state4, const float* bias_corrections, const float* lr_dev, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int adam_w_mode, hipStream_t s) {
    AmpList l;
    l.t[0] = AmpTensor{pa, ga, ma, va, (__half*)ha, na, l2c_a, l2n_a};
    l.t[1] = AmpTensor{pb, gb, mb, vb, (__half*)hb, nb, l2c_b, l2n_b};
    const int64_t largest = na > nb ? na : nb;
    hipLaunchKernelGGL(k_amp_adam, dim3((unsigned)nrc_cdiv(nrc_cdiv(largest, 4), 256), (unsigned)(nb > 0 ? 2 : 1)), dim3(256), 0, s, l, state4, bias_corrections, lr_dev, lr,
                       beta1, beta2, eps, weight_decay, adam_w_mode);
}

extern "C" {

int nrc_nonfinite_check4(const float* g0, int64_t n0, const float* g1, int64_t n1, const float* g2, int64_t n2, const float* g3, int64_t n3,
                         float* found_inf, nrc_stream_t stream) {
    NRC_ENTER();
    if (!found_inf || n0 < 0 || n1 < 0 || n2 < 0 || n3 < 0) return NRC_ERR_INVALID;
    const float* gs[4] = {g0, g1, g2, g3};
    const int64_t ns[4] = {n0, n1, n2, n3};
    FiniteList l;
    int count = 0;
    int64_t largest = 0;
    for (int k = 0; k < 4; k++) {
        if (ns[k] == 0) continue;
        if (!gs[k]) return NRC_ERR_INVALID;
        l.g[count] = reinterpret_cast<const uint32_t*>(gs[k]); l.n[count] = ns[k]; count++;
        largest = ns[k] > largest ? ns[k] : largest;
    }
    if (count == 0) return NRC_OK;
    for (int k = count; k < 4; k++) { l.g[k] = l.g[0]; l.n[k] = 0; }
    const int64_t blocks = nrc_cdiv(nrc_cdiv(largest, 4), 256);
    hipLaunchKernelGGL(k_nonfinite_check4, dim3((unsigned)(blocks < 2048 ? blocks : 2048), (unsigned)count), dim3(256), 0, (hipStream_t)stream, l, found_inf);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_nonfinite_check(const float* grad, int64_t n, float* found_inf, nrc_stream_t stream) {
    NRC_ENTER();
    if (n < 0 || !found_inf) return NRC_ERR_INVALID;
    if (n == 0) return NRC_OK;
    if (!grad) return NRC_ERR_INVALID;
    const int64_t blocks = nrc_cdiv(nrc_cdiv(n, 4), 256);
    hipLaunchKernelGGL(k_nonfinite_check, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)grad, n, found_inf);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_adam_prepare(int32_t host_step, float beta1, float beta2, const float* found_inf, int32_t* skipped_steps, int32_t* device_step,
                     float* bias_corrections, nrc_stream_t stream) {
    NRC_ENTER();
    if (!bias_corrections || (!device_step && (host_step < 1 || !skipped_steps))) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_adam_prepare, dim3(1), dim3(64), 0, (hipStream_t)stream, host_step, beta1, beta2, found_inf, skipped_steps, device_step,
                       bias_corrections);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int32_t adam_w_mode, float bias_correction1, float bias_correction2, const float* bias_corrections_dev,
                  const float* lr_dev, const float* grad_scale, const float* found_inf, void* param_f16_out, float l2_slice_coeff,
                  int64_t l2_slice_count, nrc_stream_t stream) {
    NRC_ENTER();
    if (n < 0 || (!bias_corrections_dev && (!(bias_correction1 > 0.f) || !(bias_correction2 > 0.f)))) return NRC_ERR_INVALID;
    if (n == 0) return NRC_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq) return NRC_ERR_INVALID;
    const bool aligned = ((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15u) == 0);
    const dim3 grid((unsigned)nrc_cdiv(nrc_cdiv(n, 4), 256));
    if (aligned)
        hipLaunchKernelGGL(k_adam<true>, grid, dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                           (int)adam_w_mode, bias_correction1, bias_correction2, bias_corrections_dev, lr_dev, grad_scale, found_inf, (__half*)param_f16_out, l2_slice_coeff,
                           l2_slice_count);
    else
        hipLaunchKernelGGL(k_adam<false>, grid, dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay,
                           (int)adam_w_mode, bias_correction1, bias_correction2, bias_corrections_dev, lr_dev, grad_scale, found_inf, (__half*)param_f16_out, l2_slice_coeff,
                           l2_slice_count);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_adam_step_multi(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                        const int64_t* sizes, const float* lrs, const float* bias_correction1, const float* bias_correction2, float beta1, float beta2,
                        float eps, float weight_decay, int32_t adam_w_mode, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_tensors < 0 || n_tensors > ADAM_MULTI_MAX) return NRC_ERR_INVALID;
    if (n_tensors == 0) return NRC_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !sizes || !lrs || !bias_correction1 || !bias_correction2) return NRC_ERR_INVALID;
    MultiList l;
    l.n = 0;
    int64_t blocks = 0;
    for (int k = 0; k < n_tensors; k++) {
        if (sizes[k] < 0 || !(bias_correction1[k] > 0.f) || !(bias_correction2[k] > 0.f)) return NRC_ERR_INVALID;
        if (sizes[k] == 0) continue;
        if (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k]) return NRC_ERR_INVALID;
        if (blocks > 0x7fffffff) return NRC_ERR_INVALID;
        l.t[l.n++] = MultiTensor{params[k], grads[k], exp_avg[k], exp_avg_sq[k], sizes[k], lrs[k], bias_correction1[k], bias_correction2[k], (int)blocks};
        blocks += nrc_cdiv(nrc_cdiv(sizes[k], 4), 256);
    }
    if (l.n == 0) return NRC_OK;
    if (blocks > 0x7fffffff) return NRC_ERR_INVALID;
    for (int k = l.n; k < ADAM_MULTI_MAX; k++) l.t[k] = MultiTensor{nullptr, nullptr, nullptr, nullptr, 0, 0.f, 1.f, 1.f, 0x7fffffff};
    hipLaunchKernelGGL(k_adam_multi, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, l, beta1, beta2, eps, weight_decay, (int)adam_w_mode);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_amp_adam_step(float* param_a, const float* grad_a, float* exp_avg_a, float* exp_avg_sq_a, void* param_f16_a, int64_t n_a, float l2_coeff_a,
                      int64_t l2_count_a, float* param_b, const float* grad_b, float* exp_avg_b, float* exp_avg_sq_b, void* param_f16_b, int64_t n_b,
                      float l2_coeff_b, int64_t l2_count_b, float lr, const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                      int32_t adam_w_mode, int32_t* device_step, float* bias_corrections, float* scale, int32_t* growth_tracker, float growth_factor,
                      float backoff_factor, int32_t growth_interval, float* state4, void* ticket, int32_t* skipped_steps, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_a < 1 || n_b < 0 || !param_a || !grad_a || !exp_avg_a || !exp_avg_sq_a || (n_b && (!param_b || !grad_b || !exp_avg_b || !exp_avg_sq_b)) ||
        !device_step || !bias_corrections || !state4 || !ticket || (scale && (!growth_tracker || growth_interval < 1)))
        return NRC_ERR_INVALID;
    AmpList l;
    l.t[0] = AmpTensor{param_a, grad_a, exp_avg_a, exp_avg_sq_a, (__half*)param_f16_a, n_a, l2_coeff_a, l2_count_a};
    l.t[1] = AmpTensor{param_b, grad_b, exp_avg_b, exp_avg_sq_b, (__half*)param_f16_b, n_b, l2_coeff_b, l2_count_b};
    const int count = n_b > 0 ? 2 : 1;
    AmpState a;
    a.device_step = device_step; a.bc = bias_corrections; a.scale = scale; a.growth_tracker = growth_tracker; a.growth_factor = growth_factor;
    a.backoff_factor = backoff_factor; a.growth_interval = growth_interval; a.beta1 = beta1; a.beta2 = beta2; a.state = state4; a.ticket = (uint32_t*)ticket;
    a.skipped = skipped_steps; a.grad_divisor = 1.f;
    const int64_t largest = n_a > n_b ? n_a : n_b;
    const int64_t blocks = nrc_cdiv(nrc_cdiv(largest, 4), 256);
    hipStream_t s = (hipStream_t)stream;
    NRC_STAGE(s, nullptr);
    hipLaunchKernelGGL(k_amp_check_prepare, dim3((unsigned)(blocks < 2048 ? blocks : 2048), (unsigned)count), dim3(256), 0, s, l, a);
    NRC_STAGE(s, "k_amp_check_prepare");
    hipLaunchKernelGGL(k_amp_adam, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, s, l, (const float*)state4, (const float*)bias_corrections, lr_dev, lr,
                       beta1, beta2, eps, weight_decay, (int)adam_w_mode);
    NRC_STAGE(s, "k_amp_adam");
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

/* ---- group 14: the optimizer step of a data-parallel rank, in pieces (include/nerficg_hip.h) ---- */
int nrc_wire_pack_f16(const float* src, void* dst_f16, int64_t n, uint64_t* saturated, nrc_stream_t stream) {
    NRC_ENTER();
    if (n < 0 || (n > 0 && (!src || !dst_f16))) return NRC_ERR_INVALID;
    if (n == 0) return NRC_OK;
    hipLaunchKernelGGL(k_wire_pack_f16, dim3((unsigned)nrc_cdiv(nrc_cdiv(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, src, (__half*)dst_f16, n, (unsigned long long*)saturated);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_wire_unpack_f16(const void* src_f16, float* dst, int64_t n, nrc_stream_t stream) {
    NRC_ENTER();
    if (n < 0 || (n > 0 && (!src_f16 || !dst))) return NRC_ERR_INVALID;
    if (n == 0) return NRC_OK;
    hipLaunchKernelGGL(k_wire_unpack_f16, dim3((unsigned)nrc_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, (const __half*)src_f16, dst, n);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_amp_settle(const float* flag_sum, float grad_divisor, float beta1, float beta2, int32_t* device_step, float* bias_corrections, float* scale,
                   int32_t* growth_tracker, float growth_factor, float backoff_factor, int32_t growth_interval, float* state4, int32_t* skipped_steps,
                   nrc_stream_t stream) {
    NRC_ENTER();
    if (!flag_sum || !(grad_divisor >= 1.f) || !device_step || !bias_corrections || !state4 || (scale && (!growth_tracker || growth_interval < 1))) return NRC_ERR_INVALID;
    AmpState a;
    a.device_step = device_step; a.bc = bias_corrections; a.scale = scale; a.growth_tracker = growth_tracker; a.growth_factor = growth_factor;
    a.backoff_factor = backoff_factor; a.growth_interval = growth_interval; a.beta1 = beta1; a.beta2 = beta2; a.state = state4; a.ticket = nullptr;
    a.skipped = skipped_steps; a.grad_divisor = grad_divisor;
    hipLaunchKernelGGL(k_amp_settle, dim3(1), dim3(64), 0, (hipStream_t)stream, a, flag_sum);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_amp_adam_slices(float* param_a, const float* grad_a, float* exp_avg_a, float* exp_avg_sq_a, void* param_f16_a, int64_t n_a, float l2_coeff_a,
                        int64_t l2_count_a, float* param_b, const float* grad_b, float* exp_avg_b, float* exp_avg_sq_b, void* param_f16_b, int64_t n_b,
                        float l2_coeff_b, int64_t l2_count_b, float lr, const float* lr_dev, float beta1, float beta2, float eps, float weight_decay,
                        int32_t adam_w_mode, const float* bias_corrections, const float* state4, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_a < 0 || n_b < 0 || !bias_corrections || !state4 || (n_a && (!param_a || !grad_a || !exp_avg_a || !exp_avg_sq_a)) ||
        (n_b && (!param_b || !grad_b || !exp_avg_b || !exp_avg_sq_b)))
        return NRC_ERR_INVALID;
    if (n_a == 0 && n_b == 0) return NRC_OK;
    if (n_a == 0)   // the kernel's grid is sized by the larger slice and its second row is optional: an empty first slice moves the second one up
        nrc_launch_amp_adam(param_b, grad_b, exp_avg_b, exp_avg_sq_b, param_f16_b, n_b, l2_coeff_b, l2_count_b, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0,
                            state4, bias_corrections, lr_dev, lr, beta1, beta2, eps, weight_decay, (int)adam_w_mode, (hipStream_t)stream);
    else
        nrc_launch_amp_adam(param_a, grad_a, exp_avg_a, exp_avg_sq_a, param_f16_a, n_a, l2_coeff_a, l2_count_a, param_b, grad_b, exp_avg_b, exp_avg_sq_b, param_f16_b, n_b,
                            l2_coeff_b, l2_count_b, state4, bias_corrections, lr_dev, lr, beta1, beta2, eps, weight_decay, (int)adam_w_mode, (hipStream_t)stream);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

}  // extern "C"
