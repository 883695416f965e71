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
    uint32_t total_entries;           // sum of the level sizes
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
// ReLU is applied AFTER the f32->f16 conversion as a packed fp16 max (same value: rounding is monotone and keeps 0),
// which halves the VALU work of the activation.
__device__ __forceinline__ h8 acc_to_frag_relu(const f16v& acc, int g) {
    h8 f;
#pragma unroll
    for (int j = 0; j < 8; j++) f[j] = (_Float16)acc[8 * g + j];
    const h8 z = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    return __builtin_elementwise_max(f, z);
}
__device__ __forceinline__ h8 acc_to_frag(const f16v& acc, int g) {
    h8 f;
#pragma unroll
    for (int j = 0; j < 8; j++) f[j] = (_Float16)acc[8 * g + j];
    return f;
}

#define NRC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// ---- 16-row output layers on v_mfma_f32_16x16x32_f16 (round 6) ------------------------------------------------------------------------------------
// A 16-neuron head (density features, rgb) run as a 32-row tile spends half of its matrix passes on padding rows.  The 16x16x32 form has no padding:
// A = W (16 neurons x 32 k), B = activations (32 k x 16 samples), D = 16 x 16 f32 with lane l holding column (sample) l&15, rows 4 (l>>4) + reg.
// Its operands want lane l = 16 kg + n to hold k-slots 8 kg .. 8 kg + 7 of sample n, while the 64-wide layer in front leaves lane 32 hh + r with values of
// sample r.  In units of 16-lane rows the wave holds [samples 0-15 | 16-31 | 0-15 | 16-31] (hh = 0, 0, 1, 1); v_permlane16_swap_b32 X, Y swaps the odd rows
// of X with the even rows of Y:  X' = [X0 Y0 X2 Y2],  Y' = [X1 Y1 X3 Y3]  -- X' holds samples 0-15 in all four rows, Y' samples 16-31: the B operands of
// the two 16-sample blocks, k-groups (X hh=0, Y hh=0, X hh=1, Y hh=1).  One swap per dword pair, the k order is folded into the weight fragment.
typedef float f4v __attribute__((ext_vector_type(4)));
#define NRC_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

// H[0..3]: the four ACC-order fragments of a 64-wide layer (H[2 mt + gq]).  B[nb][ks]: block nb (samples 16 nb ..), k-step ks (neurons 32 ks ..).
__device__ __forceinline__ void head_split(const h8 (&H)[4], h8 (&B)[2][2]) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
        const uint4 x = *reinterpret_cast<const uint4*>(&H[2 * ks]), y = *reinterpret_cast<const uint4*>(&H[2 * ks + 1]);
        const auto s0 = __builtin_amdgcn_permlane16_swap(x.x, y.x, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(x.y, y.y, false, false);
        const auto s2 = __builtin_amdgcn_permlane16_swap(x.z, y.z, false, false);
        const auto s3 = __builtin_amdgcn_permlane16_swap(x.w, y.w, false, false);
        const uint4 a = make_uint4(s0[0], s1[0], s2[0], s3[0]), b = make_uint4(s0[1], s1[1], s2[1], s3[1]);
        B[0][ks] = *reinterpret_cast<const h8*>(&a);
        B[1][ks] = *reinterpret_cast<const h8*>(&b);
    }
}
// the A fragment that goes with head_split: W row-major [n_rows <= 16][ld], k-step ks; lane l = 16 kg + m, element e <-> neuron
// 32 ks + (e & 3) + 8 (2 (kg & 1) + (e >> 2)) + 4 (kg >> 1)   (the ACC order of fragment H[2 ks + (kg & 1)], lane half kg >> 1)
__device__ __forceinline__ h8 load_w_head_frag(const __half* __restrict__ W, int ld, int n_rows, int ks, int lane) {
    h8 f;
    const int m = lane & 15, kg = lane >> 4;
    const _Float16* p = reinterpret_cast<const _Float16*>(W) + (size_t)m * ld;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const int k = 32 * ks + (e & 3) + 8 * (2 * (kg & 1) + (e >> 2)) + 4 * (kg >> 1);
        f[e] = m < n_rows ? p[k] : (_Float16)0.f;
    }
    return f;
}
// two head accumulators (blocks 0 and 1: lane 16 kg + n holds neurons 4 kg .. 4 kg + 3 of sample 16 nb + n) -> fp16, back on the sample's own lanes:
// lane 32 hh + r receives neurons 8 hh .. 8 hh + 7 of sample r -- the NATURAL-order B fragment of a 16-wide k-step of the 32x32x16 form.
__device__ __forceinline__ h8 head_join(const f4v& o0, const f4v& o1) {
    const h2 p00 = {(_Float16)o0[0], (_Float16)o0[1]}, p01 = {(_Float16)o0[2], (_Float16)o0[3]};
    const h2 p10 = {(_Float16)o1[0], (_Float16)o1[1]}, p11 = {(_Float16)o1[2], (_Float16)o1[3]};
    const auto s0 = __builtin_amdgcn_permlane16_swap(*reinterpret_cast<const uint32_t*>(&p00), *reinterpret_cast<const uint32_t*>(&p10), false, false);
    const auto s1 = __builtin_amdgcn_permlane16_swap(*reinterpret_cast<const uint32_t*>(&p01), *reinterpret_cast<const uint32_t*>(&p11), false, false);
    const uint4 v = make_uint4(s0[0], s1[0], s0[1], s1[1]);
    return *reinterpret_cast<const h8*>(&v);
}
// neurons 0..2 of the two blocks (k-group 0: lanes 0-15 of each accumulator) as f32 on the sample's own lane (lanes 0-31): three swaps, so that the
// sigmoid runs ONCE per sample on three values instead of on both accumulators
__device__ __forceinline__ void head_join_rgb(const f4v& o0, const f4v& o1, float (&rgb)[3]) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const auto s = __builtin_amdgcn_permlane16_swap(__float_as_uint(o0[c]), __float_as_uint(o1[c]), false, false);
        rgb[c] = __uint_as_float(s[0]);      // [o0 row 0 | o1 row 0 | ...]: lane r < 32 holds sample r
    }
}

// ---- hash-grid encoding of one sample for one level -> (f0, f1) --------------------------------------------------
// Entry index of a corner.  Dense levels: x + y*res + z*res^2, wrapped once at `size` (inputs in [0,1] never exceed
// 2*size; anything beyond is clamped into the level so that the gather stays in bounds).  Hashed levels: the spatial hash
// x ^ y*2654435761 ^ z*805459861 masked to the power-of-two level size.  Both forms share per-axis partial terms that are
// computed once per level (the +1 corner is the term plus its multiplier), and the choice is a select, not a branch: the
// two halves of a wave work on different levels, so a branch on `hashed` would serialise them.
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
    // weight products in the order (x*y)*z, as in the oracle
    const float wxy[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
    const uint32_t my = hashed ? 2654435761u : res, mz = hashed ? 805459861u : res * res;
    const uint32_t ty0 = gy * my, tz0 = gz * mz;
    const uint32_t ty[2] = {ty0, ty0 + my}, tz[2] = {tz0, tz0 + mz};
    const uint32_t mask = size - 1u;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t qx = gx + (k & 1), a = ty[(k >> 1) & 1], b = tz[(k >> 2) & 1];
        const uint32_t h = (qx ^ a ^ b) & mask;
        uint32_t d = qx + a + b;
        d = d >= size ? d - size : d;
        d = min(d, mask);
        c.w[k] = wxy[k & 3] * ((k & 4) ? wz1 : wz0);
        c.e[k] = off + (hashed ? h : d);
    }
}
// Same corners when the level (hence `hashed`) is uniform over the wave: the index form is a template argument chosen by a
// scalar branch, which halves the integer work per corner (k_grid_encode / k_grid_bwd are VALU co-limited: profiles/).
template <bool HASHED>
__device__ __forceinline__ void grid_corners_u(float px, float py, float pz, float scale, uint32_t res, uint32_t size, uint32_t off, Corner8& c) {
    const float fx = fmaf(scale, px, 0.5f), fy = fmaf(scale, py, 0.5f), fz = fmaf(scale, pz, 0.5f);
    const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
    const uint32_t gx = (uint32_t)(int32_t)flx, gy = (uint32_t)(int32_t)fly, gz = (uint32_t)(int32_t)flz;
    const float wx1 = fx - flx, wy1 = fy - fly, wz1 = fz - flz;
    const float wx0 = 1.f - wx1, wy0 = 1.f - wy1, wz0 = 1.f - wz1;
    const float wxy[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
    const uint32_t my = HASHED ? 2654435761u : res, mz = HASHED ? 805459861u : res * res;
    const uint32_t ty0 = gy * my, tz0 = gz * mz;
    const uint32_t ty[2] = {ty0, ty0 + my}, tz[2] = {tz0, tz0 + mz};
    const uint32_t mask = size - 1u;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const uint32_t a = ty[p & 1], b = tz[p >> 1];
        const uint32_t yz = HASHED ? (a ^ b) : (a + b);
#pragma unroll
        for (int dx = 0; dx < 2; dx++) {
            const int k = 2 * p + dx;
            uint32_t e;
            if constexpr (HASHED) {
                e = ((gx + dx) ^ yz) & mask;
            } else {
                e = gx + dx + yz;
                e = e >= size ? e - size : e;
                e = min(e, mask);
            }
            c.w[k] = wxy[k & 3] * ((k & 4) ? wz1 : wz0);
            c.e[k] = off + e;
        }
    }
}
// fp16 feature pairs are fetched with buffer loads: one 128-bit descriptor for the table, 32-bit byte offsets per lane
// (no 64-bit address arithmetic per gather; out-of-range offsets return 0 instead of faulting)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_table_rsrc(const void* table, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(table), /*stride*/ 0, (int)bytes, 0x00020000);
}
// The two x-neighbours of a corner pair sit next to each other in the table (dense: e+1; hashed: (x ^ h) and ((x+1) ^ h)
// differ only in the low bits unless x crosses a power-of-two boundary), so one 16-byte aligned load usually returns both
// (3 out of 4 positions).  The texture-cache access rate (one line per clock per CU), not bandwidth, bounds the encoding:
// 4 wide loads + a predicated narrow load for the straddling quarter = 5 accesses per level instead of 8.
// f += w * (fp16 pair) in two v_fma_mix_f32 (f32 FMA reading its fp16 operand in place): the same arithmetic as convert + fmaf
// without the 16 conversions per level
__device__ __forceinline__ void fma_half2(float w, uint32_t packed, float& f0, float& f1) {
#if defined(NRC_NO_FMA_MIX)
    const __half2 hv = *reinterpret_cast<const __half2*>(&packed);
    const float2 t = __half22float2(hv);
    f0 = fmaf(w, t.x, f0);
    f1 = fmaf(w, t.y, f1);
#else
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(f0) : "v"(w), "v"(packed));
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(f1) : "v"(w), "v"(packed));
#endif
}
__device__ __forceinline__ uint32_t pick4(const uint4& q, uint32_t i) {
    const uint32_t lo = (i & 1) ? q.y : q.x, hi = (i & 1) ? q.w : q.z;
    return (i & 2) ? hi : lo;
}
__device__ __forceinline__ void grid_level_features(__amdgpu_buffer_rsrc_t rsrc, const Corner8& c, float& f0, float& f1) {
    uint32_t v[8];
    uint4 quad[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const uint32_t base = c.e[2 * p] & ~3u;
        const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base << 2, 0, 0);
        quad[p] = make_uint4(raw[0], raw[1], raw[2], raw[3]);
    }
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const uint32_t e0 = c.e[2 * p], e1 = c.e[2 * p + 1], base = e0 & ~3u;
        v[2 * p] = pick4(quad[p], e0 & 3u);
        uint32_t other = pick4(quad[p], e1 & 3u);
        if ((e1 & ~3u) != base) other = __builtin_amdgcn_raw_buffer_load_b32(rsrc, e1 << 2, 0, 0);
        v[2 * p + 1] = other;
    }
    f0 = 0.f; f1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) fma_half2(c.w[k], v[k], f0, f1);
}

// Hashed levels: both x-neighbours of ALL four (y, z) pairs straddle or not together -- (x ^ h) and ((x + 1) ^ h) differ by x ^ (x + 1),
// which does not depend on h.  One predicate, one predicated region for the four fix-up loads.
// AUX: cache-policy bits of the gathers (gfx940+ buffer loads: 1 = sc0, 2 = nt, 16 = sc1); 0 = default
template <int AUX = 0>
__device__ __forceinline__ void grid_level_features_hashed(__amdgpu_buffer_rsrc_t rsrc, const Corner8& c, float& f0, float& f1) {
    uint32_t v[8];
    uint4 quad[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const auto raw = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (c.e[2 * p] & ~3u) << 2, 0, AUX);
        quad[p] = make_uint4(raw[0], raw[1], raw[2], raw[3]);
    }
    const bool straddle = ((c.e[0] ^ c.e[1]) & ~3u) != 0u;
    uint32_t fix[4] = {0u, 0u, 0u, 0u};
    if (straddle) {
#pragma unroll
        for (int p = 0; p < 4; p++) fix[p] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, c.e[2 * p + 1] << 2, 0, AUX);
    }
#pragma unroll
    for (int p = 0; p < 4; p++) {
        v[2 * p] = pick4(quad[p], c.e[2 * p] & 3u);
        const uint32_t other = pick4(quad[p], c.e[2 * p + 1] & 3u);
        v[2 * p + 1] = straddle ? fix[p] : other;
    }
    f0 = 0.f; f1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) fma_half2(c.w[k], v[k], f0, f1);
}
// Hashed levels, 8-byte variant: the aligned pair {e0 & ~1, e0 | 1} holds both neighbours whenever x is even; odd x (one predicate for
// all four pairs, see above) fetches the partner with a 4-byte load.  6 instead of 5 line lookups per sample and level, half the
// returned bytes, a quarter of the select instructions.
__device__ __forceinline__ void grid_level_features_pair(__amdgpu_buffer_rsrc_t rsrc, const Corner8& c, float& f0, float& f1) {
    uint32_t v[8];
    uint2 pr[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const auto raw = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (c.e[2 * p] & ~1u) << 2, 0, 0);
        pr[p] = make_uint2(raw[0], raw[1]);
    }
    const bool apart = (c.e[0] ^ c.e[1]) != 1u;
    uint32_t fix[4] = {0u, 0u, 0u, 0u};
    if (apart) {
#pragma unroll
        for (int p = 0; p < 4; p++) fix[p] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, c.e[2 * p + 1] << 2, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const bool odd = (c.e[2 * p] & 1u) != 0u;
        v[2 * p] = odd ? pr[p].y : pr[p].x;
        const uint32_t other = odd ? pr[p].x : pr[p].y;
        v[2 * p + 1] = apart ? fix[p] : other;
    }
    f0 = 0.f; f1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) fma_half2(c.w[k], v[k], f0, f1);
}

// dense (coarse) levels: the lanes of a wave share a handful of cache lines, so the cost of a gather is its data return, not its
// tag lookups -- eight 4-byte loads (256 B per wave instruction) beat four 16-byte loads (1 KB per wave instruction) there
__device__ __forceinline__ void grid_level_features_narrow(__amdgpu_buffer_rsrc_t rsrc, const Corner8& c, float& f0, float& f1) {
    uint32_t v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, c.e[k] << 2, 0, 0);
    f0 = 0.f; f1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) fma_half2(c.w[k], v[k], f0, f1);
}

// Wave-uniform cell (coarse levels in the tile-interleaved layout: the 64 samples of a row are 8 x 8 neighbouring pixels at the same step, a
// footprint of ~0.01 of the box, smaller than the cells of levels 0-4 and about the size of those of levels 5-6): when every ACTIVE lane of the
// wave sits in the same cell, the eight corners are the same eight table entries for all of them -- they are fetched ONCE through the scalar
// cache (s_load_dword, no vector-memory instruction, no L1 tag lookups) and only the trilinear weights are per lane.  profiles/: levels 0-4
// cost 8.0 of the kernel's 41.7 L1 lookups per sample through the vector path although they touch a handful of lines (the texture path counts
// per 16-lane quad, not per distinct line).  Same entries, same weights, same order of the eight multiply-adds: bit-identical features.
// Returns false (nothing computed) when the active lanes disagree; the caller then takes the vector path.
template <bool HASHED>
__device__ __forceinline__ bool grid_level_features_uniform(const uint32_t* __restrict__ table32, float px, float py, float pz, float scale, uint32_t res,
                                                            uint32_t size, uint32_t off, float& f0, float& f1) {
    const float fx = fmaf(scale, px, 0.5f), fy = fmaf(scale, py, 0.5f), fz = fmaf(scale, pz, 0.5f);
    const float flx = floorf(fx), fly = floorf(fy), flz = floorf(fz);
    const uint32_t gx = (uint32_t)(int32_t)flx, gy = (uint32_t)(int32_t)fly, gz = (uint32_t)(int32_t)flz;
    const uint32_t ux = (uint32_t)__builtin_amdgcn_readfirstlane((int)gx), uy = (uint32_t)__builtin_amdgcn_readfirstlane((int)gy),
                   uz = (uint32_t)__builtin_amdgcn_readfirstlane((int)gz);
    if (__ballot(((gx ^ ux) | (gy ^ uy) | (gz ^ uz)) != 0u) != 0ull) return false;
    const float wx1 = fx - flx, wy1 = fy - fly, wz1 = fz - flz;
    const float wx0 = 1.f - wx1, wy0 = 1.f - wy1, wz0 = 1.f - wz1;
    const float wxy[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
    // entry indices exactly as grid_corners_u, on the wave-uniform cell: scalar arithmetic
    const uint32_t my = HASHED ? 2654435761u : res, mz = HASHED ? 805459861u : res * res;
    const uint32_t ty0 = uy * my, tz0 = uz * mz;
    const uint32_t ty[2] = {ty0, ty0 + my}, tz[2] = {tz0, tz0 + mz};
    const uint32_t mask = size - 1u;
    uint32_t v[8];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const uint32_t a = ty[p & 1], b = tz[p >> 1];
        const uint32_t yz = HASHED ? (a ^ b) : (a + b);
#pragma unroll
        for (int dx = 0; dx < 2; dx++) {
            uint32_t e;
            if constexpr (HASHED) {
                e = ((ux + dx) ^ yz) & mask;
            } else {
                e = ux + dx + yz;
                e = e >= size ? e - size : e;
                e = min(e, mask);
            }
            v[2 * p + dx] = table32[off + e];   // uniform address: s_load_dword
        }
    }
    f0 = 0.f; f1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) fma_half2(wxy[k & 3] * ((k & 4) ? wz1 : wz0), v[k], f0, f1);
    return true;
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

// SH degree 4 of a direction as the colour net's k-step-0 B fragments (lo: coefficients 0-7, hi: 8-15)
__device__ __forceinline__ void sh4_fragments(float dx, float dy, float dz, h8& lo, h8& hi) {
    // the reference feeds fp16(d*0.5+0.5) and tiny-cuda-nn maps it back with *2-1 (Renderer.py:52)
    float sh[16];
    const float ex = (float)(_Float16)__fadd_rn(__fmul_rn(dx, 0.5f), 0.5f) * 2.f - 1.f;
    const float ey = (float)(_Float16)__fadd_rn(__fmul_rn(dy, 0.5f), 0.5f) * 2.f - 1.f;
    const float ez = (float)(_Float16)__fadd_rn(__fmul_rn(dz, 0.5f), 0.5f) * 2.f - 1.f;
    sh4_eval(ex, ey, ez, sh);
#pragma unroll
    for (int jj = 0; jj < 8; jj++) { lo[jj] = (_Float16)sh[jj]; hi[jj] = (_Float16)sh[8 + jj]; }

}
