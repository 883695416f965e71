// ngp_net.h -- device building blocks of the fused "input encoding + tiny MLP" kernels (gfx950, wave64, MFMA).
//
// Formulation (transposed GEMM): for a tile of 32 samples a wave computes  H^T = act(W . X^T)  with
//   A operand = weights  (rows = output neurons, k = input neurons)         -> v_mfma_f32_32x32x16_f16
//   B operand = activations (k = input neurons, columns = the 32 samples)
//   C/D       = 32 neurons x 32 samples, f32: lane l holds column (sample) l&31, rows (reg&3)+8*(reg>>2)+4*(l>>5).
// Because a D tile has the sample on the lane and the neurons in the registers, it IS the B operand of the next
// layer after a register-local f32->f16 conversion: no LDS round trip, no cross-lane traffic between layers.
// The only price is a fixed permutation of the k index inside each 16-wide k-step ("ACC order"):
//   element j of lane half h  <->  neuron 16s + 8*(j>>2) + 4*h + (j&3)
// which is applied once, when the weights are loaded into their A fragments.
// The first layer's B fragments are produced by the encoding directly in registers ("NATURAL order":
//   element j of lane half h  <->  input feature 16s + 8*h + j ).
#pragma once
#include <hip/hip_fp16.h>

#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define NRC_MAX_LEVELS 16

struct GridCfg {
    uint32_t offset[NRC_MAX_LEVELS];  // first entry of the level (in entries of n_features halves)
    uint32_t size[NRC_MAX_LEVELS];    // entries in the level
    uint32_t res[NRC_MAX_LEVELS];
    uint32_t hashed[NRC_MAX_LEVELS];  // 1: spatial hash (size is then a power of two), 0: dense x + y*res + z*res^2
    float scale[NRC_MAX_LEVELS];
};

__device__ __forceinline__ f16v zero16() {
    f16v z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0.f;
    return z;
}

// ---- weight (A operand) fragments ------------------------------------------------------------------------------
// W: row-major [n_rows][ld] fp16.  Fragment of m-tile mt, k-step s for this lane (r = lane&31, hh = lane>>5).
template <bool ACC_ORDER>
__device__ __forceinline__ h8 load_w_frag(const __half* __restrict__ W, int ld, int n_rows, int mt, int s, int r, int hh) {
    h8 f;
    const int row = 32 * mt + r;
    if (row >= n_rows) {
#pragma unroll
        for (int j = 0; j < 8; j++) f[j] = (_Float16)0.f;
        return f;
    }
    const _Float16* p = reinterpret_cast<const _Float16*>(W) + (size_t)row * ld + 16 * s;
    if constexpr (!ACC_ORDER) {
        const h8 v = *reinterpret_cast<const h8*>(p + 8 * hh);
        return v;
    } else {
        const h4 a = *reinterpret_cast<const h4*>(p + 4 * hh);
        const h4 b = *reinterpret_cast<const h4*>(p + 8 + 4 * hh);
#pragma unroll
        for (int j = 0; j < 4; j++) { f[j] = a[j]; f[4 + j] = b[j]; }
        return f;
    }
}
// transposed weights as A operand (backward: dH_prev^T = W^T . dZ^T): rows = input neurons k, k = output neurons o.
// element j <-> o = omap(s,hh,j) ; value W[o][32*mt + r]
template <bool ACC_ORDER>
__device__ __forceinline__ h8 load_wT_frag(const __half* __restrict__ W, int ld, int n_out_rows, int n_in_cols, int mt, int s, int r, int hh) {
    h8 f;
    const int col = 32 * mt + r;
    const _Float16* p = reinterpret_cast<const _Float16*>(W);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int o = ACC_ORDER ? (16 * s + 8 * (j >> 2) + 4 * hh + (j & 3)) : (16 * s + 8 * hh + j);
        f[j] = (o < n_out_rows && col < n_in_cols) ? p[(size_t)o * ld + col] : (_Float16)0.f;
    }
    return f;
}

// ---- D tile -> next layer's B fragments --------------------------------------------------------------------------
__device__ __forceinline__ h8 acc_to_frag_relu(const f16v& acc, int g) {
    h8 f;
#pragma unroll
    for (int j = 0; j < 8; j++) f[j] = (_Float16)fmaxf(acc[8 * g + j], 0.f);
    return f;
}
__device__ __forceinline__ h8 acc_to_frag(const f16v& acc, int g) {
    h8 f;
#pragma unroll
    for (int j = 0; j < 8; j++) f[j] = (_Float16)acc[8 * g + j];
    return f;
}

#define NRC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// ---- hash-grid encoding of one sample for one level -> (f0, f1) --------------------------------------------------
__device__ __forceinline__ uint32_t grid_entry(uint32_t x, uint32_t y, uint32_t z, uint32_t res, uint32_t size, bool hashed) {
    if (hashed) return ((x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u)) & (size - 1u);
    return (x + y * res + z * res * res) % size;
}
struct Corner8 {
    uint32_t e[8];
    float w[8];
};
__device__ __forceinline__ void grid_corners(float px, float py, float pz, float scale, uint32_t res, uint32_t size, uint32_t off,
                                             bool hashed, Corner8& c) {
    const float fx = fmaf(scale, px, 0.5f), fy = fmaf(scale, py, 0.5f), fz = fmaf(scale, pz, 0.5f);
    const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
    const uint32_t gx = (uint32_t)(int32_t)flx, gy = (uint32_t)(int32_t)fly, gz = (uint32_t)(int32_t)flz;
    const float wx1 = fx - flx, wy1 = fy - fly, wz1 = fz - flz;
    const float wx0 = 1.f - wx1, wy0 = 1.f - wy1, wz0 = 1.f - wz1;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t qx = gx + (k & 1), qy = gy + ((k >> 1) & 1), qz = gz + ((k >> 2) & 1);
        // weight product in the order x, y, z (matches the oracle's loop)
        c.w[k] = (((k & 1) ? wx1 : wx0) * ((k & 2) ? wy1 : wy0)) * ((k & 4) ? wz1 : wz0);
        c.e[k] = off + grid_entry(qx, qy, qz, res, size, hashed);
    }
}
__device__ __forceinline__ void grid_level_features(const __half2* __restrict__ table, const Corner8& c, float& f0, float& f1) {
    __half2 v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = table[c.e[k]];
    f0 = 0.f; f1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const float2 t = __half22float2(v[k]);
        f0 = fmaf(c.w[k], t.x, f0);
        f1 = fmaf(c.w[k], t.y, f1);
    }
}

// ---- SH degree 4 (16 coefficients) of a direction in [-1,1]^3 ------------------------------------------------------
__device__ __forceinline__ void sh4_eval(float x, float y, float z, float* o) {
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
}
