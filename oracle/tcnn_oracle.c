/*
 * oracle/tcnn_oracle.c -- CPU statement of the tiny-cuda-nn subset the reference's InstantNGP method uses
 * (src/Methods/InstantNGP/Model.py:58-114, queried at src/Methods/InstantNGP/Renderer.py:50-57):
 *   multiresolution hash-grid encoding (fw/bw), degree-4 spherical-harmonics encoding, and the 64-wide bias-free
 *   ReLU MLP with fp16 storage / f32 accumulation (fw/bw).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/ngp_oracle.c header for the rule).
 *
 * PARITY UNPINNED: tiny-cuda-nn is an un-vendored pip-from-git dependency of the reference
 * (src/Thirdparty/TinyCudaNN.py:10, unpinned HEAD); its source is not under /root/reference and the reference holds
 * no test or golden vector at this boundary.  This file restates the PUBLISHED algorithm (Mueller et al. 2022,
 * "Instant Neural Graphics Primitives with a Multiresolution Hash Encoding", sec. 3 + the open-source
 * tiny-cuda-nn conventions recorded in SURVEY.md Appendix C.1): level scale 2^(l*log2 pls)*base - 1, resolution
 * ceil(scale)+1, level size min(round_up_8(res^3), 2^log2_T), lookup position x*scale + 0.5, dense index
 * x + y*res + z*res^2 when the level fits else the spatial hash x ^ y*2654435761 ^ z*805459861, trilinear weights,
 * SH basis identical to the one the reference itself restates in src/Methods/GaussianSplatting/utils.py:21-59
 * (which cites tiny-cuda-nn as its source), weights row-major [out][in], MLP weights before the grid table in the
 * flat parameter vector (src/Methods/InstantNGP/Model.py:40,80-89 depends on that order).
 * Numerics chosen here (and mirrored by the HIP kernels): fp16 tables/weights/activations, f32 interpolation and
 * f32 dot-product accumulation, one fp16 rounding per layer output.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- IEEE binary16 helpers (round to nearest even), no compiler _Float16 needed ---- */
static inline uint16_t f32_to_f16_bits(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u)); /* NaN / inf */
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                                 /* overflow (>= 65520) */
    if (x < 0x33000001u) return (uint16_t)sign;                                              /* underflow to 0 (<= 2^-25) */
    if (x < 0x38800000u) { /* subnormal half */
        const int shift = 113 - (int)(x >> 23);
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const uint32_t rem_mask = (1u << (shift + 13)) - 1u, half_ulp = 1u << (shift + 12);
        uint32_t h = m >> (shift + 13);
        const uint32_t rem = m & rem_mask;
        if (rem > half_ulp || (rem == half_ulp && (h & 1u))) h++;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((x >> 23) - 112u) << 10 | ((x >> 13) & 0x3ffu);
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}
static inline float f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { e = 113; while (!(m & 0x400u)) { m <<= 1; e--; } x = sign | (e << 23) | ((m & 0x3ffu) << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4);
    return f;
}
static inline float round_half_soft(float f) { return f16_bits_to_f32(f32_to_f16_bits(f)); }
/* The same rounding through the host's F16C conversion instructions when the compiler has them (oracle/Makefile: -mf16c): one instruction each
 * way instead of ~25, which is most of what the CPU baseline of bench.py spent per value.  The bit-twiddled statement above stays the CHECKED
 * definition: tests/test_oracle_tcnn.py compares the two on every binary16 boundary and on random bit patterns (oracle_round_to_half_soft). */
#if defined(__F16C__)
#include <immintrin.h>
static inline float round_half(float f) { return _cvtsh_ss(_cvtss_sh(f, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC)); }
int oracle_has_f16c(void) { return 1; }
#else
static inline float round_half(float f) { return round_half_soft(f); }
int oracle_has_f16c(void) { return 0; }
#endif
void oracle_round_to_half(const float* in, int64_t n, float* out) { for (int64_t i = 0; i < n; i++) out[i] = round_half(in[i]); }
void oracle_round_to_half_soft(const float* in, int64_t n, float* out) { for (int64_t i = 0; i < n; i++) out[i] = round_half_soft(in[i]); }

/* ------------------------------------------------------------------------------------------------ hash grid */
#define MAX_LEVELS 32
typedef struct {
    int n_levels, n_features;
    uint32_t offsets[MAX_LEVELS + 1], resolution[MAX_LEVELS];
    float scale[MAX_LEVELS];
} grid_layout;

static uint32_t next_multiple_u32(uint32_t v, uint32_t m) { return ((v + m - 1) / m) * m; }

/* level geometry; returns total entry count; fills offsets[n_levels+1], scales, resolutions */
uint32_t oracle_grid_layout(int n_levels, int log2_hashmap_size, int base_resolution, float per_level_scale,
                            uint32_t* offsets, float* scales, uint32_t* resolutions) {
    const float log2_pls = log2f(per_level_scale);
    uint32_t off = 0;
    for (int l = 0; l < n_levels; l++) {
        const float scale = exp2f(l * log2_pls) * base_resolution - 1.0f;
        const uint32_t res = (uint32_t)ceilf(scale) + 1;
        const uint32_t max_params = 0xffffffffu / 2;
        uint32_t n = powf((float)res, 3.0f) > (float)max_params ? max_params : res * res * res;
        n = next_multiple_u32(n, 8u);
        if (n > (1u << log2_hashmap_size)) n = 1u << log2_hashmap_size;
        offsets[l] = off; scales[l] = scale; resolutions[l] = res;
        off += n;
    }
    offsets[n_levels] = off;
    return off;
}

static inline uint32_t grid_index(const uint32_t p[3], uint32_t res, uint32_t size) {
    uint32_t stride = 1, index = 0;
    for (int d = 0; d < 3 && stride <= size; d++) { index += p[d] * stride; stride *= res; }
    if (size < stride) index = (p[0] * 1u) ^ (p[1] * 2654435761u) ^ (p[2] * 805459861u);
    return index % size;
}

/* x (M,3) in [0,1]; table: (entries, 2) fp16-representable floats; out (M, 2*n_levels) = fp16-rounded features.
 * accumulate_half = 0 (the definition the HIP kernels are held to): f32 trilinear weights, f32 running sums, ONE fp16 rounding per feature.
 * accumulate_half = 1: what upstream tiny-cuda-nn's kernel is understood to do when its parameters are __half (it computes
 * `result += (T)weight * value` in T = __half; not verifiable here, the source is absent): the weight is rounded to fp16 and the running sum
 * is rounded to fp16 after every one of the eight fused multiply-adds.  tests/test_oracle_tcnn.py reports the distance between the two. */
/* oracle_grid_encode_fwF: n_features F values per table entry (1 .. 8; tiny-cuda-nn's n_features_per_level, src/Methods/InstantNGP/Model.py:63
 * forwards HASHGRID_N_FEATURES_PER_LEVEL), table (entries, F), out (M, F * n_levels) level-major.  oracle_grid_encode_fw2 is F = 2. */
void oracle_grid_encode_fwF(const float* x, int64_t M, const float* table, int n_levels, int n_features, int log2_hashmap_size,
                            int base_resolution, float per_level_scale, int accumulate_half, float* out) {
    const int F = n_features;
    uint32_t offsets[MAX_LEVELS + 1], res[MAX_LEVELS]; float scales[MAX_LEVELS];
    oracle_grid_layout(n_levels, log2_hashmap_size, base_resolution, per_level_scale, offsets, scales, res);
    /* Level-synchronous order: all threads work through level l of all samples before anyone touches level l + 1, so the 4 MB slice of
     * the table that a hashed level gathers from stays in the last-level caches of every core complex while it is in use (sample-major, every
     * gather of the 49 MB table went to DRAM -- and on a two-socket host to the other socket half of the time).  Each (sample, level) pair is
     * computed exactly as before; only the order of independent pairs changes. */
    for (int l = 0; l < n_levels; l++) {
        const uint32_t size = offsets[l + 1] - offsets[l];
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < M; i++) {
            float frac[3]; uint32_t g[3];
            for (int d = 0; d < 3; d++) {
                const float p = fmaf(scales[l], x[3 * i + d], 0.5f);
                const float fl = floorf(p);
                g[d] = (uint32_t)(int32_t)fl; frac[d] = p - fl;
            }
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < 8; c++) {
                float w = 1.f; uint32_t q[3];
                for (int d = 0; d < 3; d++) {
                    if (c & (1 << d)) { w *= frac[d]; q[d] = g[d] + 1; } else { w *= 1.f - frac[d]; q[d] = g[d]; }
                }
                const uint32_t e = offsets[l] + grid_index(q, res[l], size);
                if (accumulate_half) {
                    w = round_half(w);
                    for (int f = 0; f < F; f++) acc[f] = round_half(fmaf(w, table[F * (int64_t)e + f], acc[f]));
                } else {
                    for (int f = 0; f < F; f++) acc[f] = fmaf(w, table[F * (int64_t)e + f], acc[f]);
                }
            }
            for (int f = 0; f < F; f++) out[i * F * n_levels + F * l + f] = round_half(acc[f]);
        }
    }
}
void oracle_grid_encode_fw2(const float* x, int64_t M, const float* table, int n_levels, int log2_hashmap_size,
                            int base_resolution, float per_level_scale, int accumulate_half, float* out) {
    oracle_grid_encode_fwF(x, M, table, n_levels, 2, log2_hashmap_size, base_resolution, per_level_scale, accumulate_half, out);
}
void oracle_grid_encode_fw(const float* x, int64_t M, const float* table, int n_levels, int log2_hashmap_size,
                           int base_resolution, float per_level_scale, float* out) {
    oracle_grid_encode_fw2(x, M, table, n_levels, log2_hashmap_size, base_resolution, per_level_scale, 0, out);
}

/* scatter-add of d_out (M, 2*n_levels) f32 into grad_table (entries,2) f32 (zeroed by the caller) */
void oracle_grid_encode_bwF(const float* x, int64_t M, const float* d_out, int n_levels, int n_features, int log2_hashmap_size,
                            int base_resolution, float per_level_scale, float* grad_table) {
    const int F = n_features;
    uint32_t offsets[MAX_LEVELS + 1], res[MAX_LEVELS]; float scales[MAX_LEVELS];
    oracle_grid_layout(n_levels, log2_hashmap_size, base_resolution, per_level_scale, offsets, scales, res);
    for (int64_t i = 0; i < M; i++) {
        for (int l = 0; l < n_levels; l++) {
            const uint32_t size = offsets[l + 1] - offsets[l];
            float frac[3]; uint32_t g[3];
            for (int d = 0; d < 3; d++) {
                const float p = fmaf(scales[l], x[3 * i + d], 0.5f);
                const float fl = floorf(p);
                g[d] = (uint32_t)(int32_t)fl; frac[d] = p - fl;
            }
            const float* gl = d_out + i * F * n_levels + F * l;
            for (int c = 0; c < 8; c++) {
                float w = 1.f; uint32_t q[3];
                for (int d = 0; d < 3; d++) {
                    if (c & (1 << d)) { w *= frac[d]; q[d] = g[d] + 1; } else { w *= 1.f - frac[d]; q[d] = g[d]; }
                }
                const int64_t e = offsets[l] + grid_index(q, res[l], size);
                for (int f = 0; f < F; f++) grad_table[F * e + f] += w * gl[f];
            }
        }
    }
}
void oracle_grid_encode_bw(const float* x, int64_t M, const float* d_out, int n_levels, int log2_hashmap_size,
                           int base_resolution, float per_level_scale, float* grad_table) {
    oracle_grid_encode_bwF(x, M, d_out, n_levels, 2, log2_hashmap_size, base_resolution, per_level_scale, grad_table);
}

/* ------------------------------------------------------------------------------------------------ SH degree 4 */
/* d01 (M,3) in [0,1] (the reference feeds d*0.5+0.5, Renderer.py:52); out (M,16) fp16-rounded */
static inline void sh4(float x, float y, float z, float* o) {
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
void oracle_sh4_encode(const float* d01, int64_t M, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; i++) {
        float o[16];
        sh4(d01[3 * i] * 2.f - 1.f, d01[3 * i + 1] * 2.f - 1.f, d01[3 * i + 2] * 2.f - 1.f, o);
        for (int k = 0; k < 16; k++) out[16 * i + k] = round_half(o[k]);
    }
}

/* ------------------------------------------------------------------------------------------------ MLP */
/* in (M,n_in) fp16-representable; W = [W0 (width,n_in) | hidden (width,width)*(n_hidden-1) | Wout (n_out_pad,width)] row-major,
 * fp16-representable.  out_act: 0 none, 1 sigmoid.  out (M,n_out_pad) fp16-rounded.  acts (optional) receives the
 * post-ReLU hidden activations, (n_hidden, M, width).
 * Every output is the fused-multiply-add chain over k = 0, 1, ... of its weight row, as before; the loops run k-outer over a TRANSPOSED copy
 * of the weights so that the chains of all outputs advance together (contiguous, the compiler vectorises them) -- the per-output order of the
 * additions, and with it every bit of the result, is unchanged.
 * accumulate_half = 1: the running sums are rounded to fp16 after every multiply-add -- a model of upstream tiny-cuda-nn's FullyFusedMLP,
 * whose tensor-core products accumulate in __half (not verifiable here; see oracle_grid_encode_fw2). */
static int64_t mlp_weight_count(int n_in, int width, int n_hidden, int n_out_pad) {
    int64_t n_w = 0; int k = n_in;
    for (int l = 0; l < n_hidden; l++) { n_w += (int64_t)width * k; k = width; }
    return n_w + (int64_t)n_out_pad * width;
}
/* layer l: (rows x cols) row-major -> (cols x rows) */
static void mlp_transpose(const float* W, int n_in, int width, int n_hidden, int n_out_pad, float* WT) {
    const float* w = W; float* t = WT; int k = n_in;
    for (int l = 0; l <= n_hidden; l++) {
        const int rows = l < n_hidden ? width : n_out_pad;
        for (int o = 0; o < rows; o++) for (int c = 0; c < k; c++) t[(int64_t)c * rows + o] = w[(int64_t)o * k + c];
        w += (int64_t)rows * k; t += (int64_t)rows * k; k = width;
    }
}
/* one sample through the network (transposed weights WT); acts_i: this sample's slot of layer 0 in the (n_hidden, M, width) array or NULL */
static inline void mlp_one(const float* in_i, const float* WT, int n_in, int width, int n_hidden, int n_out_pad, int out_act, int accumulate_half,
                           float* out_i, float* acts_i, int64_t acts_layer_stride) {
    float h[2][128] __attribute__((aligned(64)));
    float acc[128] __attribute__((aligned(64)));
    const float* cur = in_i; int cur_n = n_in; const float* wt = WT;
    for (int l = 0; l <= n_hidden; l++) {
        const int rows = l < n_hidden ? width : n_out_pad;
        for (int o = 0; o < rows; o++) acc[o] = 0.f;
        if (accumulate_half) {
            for (int k = 0; k < cur_n; k++) {
                const float c = cur[k]; const float* col = wt + (int64_t)k * rows;
                for (int o = 0; o < rows; o++) acc[o] = round_half(fmaf(col[o], c, acc[o]));
            }
        } else {
            for (int k = 0; k < cur_n; k++) {
                const float c = cur[k]; const float* col = wt + (int64_t)k * rows;
                for (int o = 0; o < rows; o++) acc[o] = fmaf(col[o], c, acc[o]);
            }
        }
        if (l < n_hidden) {
            float* nxt = h[l & 1];
            for (int o = 0; o < rows; o++) nxt[o] = round_half(acc[o] > 0.f ? acc[o] : 0.f);
            if (acts_i) memcpy(acts_i + (int64_t)l * acts_layer_stride, nxt, sizeof(float) * width);
            cur = nxt;
        } else {
            for (int o = 0; o < rows; o++) {
                float a = acc[o];
                if (out_act == 1) a = 1.0f / (1.0f + expf(-a));
                out_i[o] = round_half(a);
            }
        }
        wt += (int64_t)rows * cur_n; cur_n = width;
    }
}
void oracle_mlp_fw2(const float* in, int64_t M, const float* W, int n_in, int width, int n_hidden, int n_out_pad, int out_act,
                    int accumulate_half, float* out, float* acts) {
    float* WT = (float*)malloc(sizeof(float) * (size_t)mlp_weight_count(n_in, width, n_hidden, n_out_pad));
    mlp_transpose(W, n_in, width, n_hidden, n_out_pad, WT);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; i++)
        mlp_one(in + i * n_in, WT, n_in, width, n_hidden, n_out_pad, out_act, accumulate_half, out + i * n_out_pad,
                acts ? acts + i * width : NULL, M * (int64_t)width);
    free(WT);
}
/* InstantNGPRayRenderingComponent.query_model (src/Methods/InstantNGP/Renderer.py:48-53) for M samples in ONE parallel loop over blocks of
 * samples -- hash-grid encode (level by level inside the block) -> density network (32 -> 64 -> 16) -> sigma = exp(h0) -> SH4 of the fp16-rounded
 * d * 0.5 + 0.5 next to h -> colour network (32 -> 64 -> 64 -> 16, sigmoid): the composition oracle.ngp_query used to assemble from the
 * stage functions above with numpy glue between them (single-threaded concatenations and conversions of M x 32 floats, which is what bounded
 * the CPU baseline on a many-core host).  Same stage arithmetic, same values (tests/test_oracle_tcnn.py compares the two).
 * sigma (M), rgb (M,3), h (M,16). */
void oracle_ngp_query(const float* x01, const float* dirs, int64_t M, const float* Wd, const float* Wc, const float* table, int n_levels,
                      int log2_hashmap_size, int base_resolution, float per_level_scale, int accumulate_half, float* sigma, float* rgb, float* h_out) {
    enum { B = 512 };
    uint32_t offsets[MAX_LEVELS + 1], res[MAX_LEVELS]; float scales[MAX_LEVELS];
    oracle_grid_layout(n_levels, log2_hashmap_size, base_resolution, per_level_scale, offsets, scales, res);
    const int n_enc = 2 * n_levels;   /* 32 */
    float* WdT = (float*)malloc(sizeof(float) * (size_t)mlp_weight_count(n_enc, 64, 1, 16));
    float* WcT = (float*)malloc(sizeof(float) * (size_t)mlp_weight_count(32, 64, 2, 16));
    mlp_transpose(Wd, n_enc, 64, 1, 16, WdT);
    mlp_transpose(Wc, 32, 64, 2, 16, WcT);
    const int64_t n_blocks = (M + B - 1) / B;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t blk = 0; blk < n_blocks; blk++) {
        const int64_t i0 = blk * B; const int m = (int)((M - i0) < B ? (M - i0) : B);
        float enc[B][64];
        for (int l = 0; l < n_levels; l++) {
            const uint32_t size = offsets[l + 1] - offsets[l];
            for (int j = 0; j < m; j++) {
                const float* x = x01 + 3 * (i0 + j);
                float frac[3]; uint32_t g[3];
                for (int d = 0; d < 3; d++) {
                    const float p = fmaf(scales[l], x[d], 0.5f);
                    const float fl = floorf(p);
                    g[d] = (uint32_t)(int32_t)fl; frac[d] = p - fl;
                }
                float acc0 = 0.f, acc1 = 0.f;
                for (int c = 0; c < 8; c++) {
                    float w = 1.f; uint32_t q[3];
                    for (int d = 0; d < 3; d++) {
                        if (c & (1 << d)) { w *= frac[d]; q[d] = g[d] + 1; } else { w *= 1.f - frac[d]; q[d] = g[d]; }
                    }
                    const uint32_t e = offsets[l] + grid_index(q, res[l], size);
                    if (accumulate_half) {
                        w = round_half(w);
                        acc0 = round_half(fmaf(w, table[2 * (int64_t)e], acc0));
                        acc1 = round_half(fmaf(w, table[2 * (int64_t)e + 1], acc1));
                    } else {
                        acc0 = fmaf(w, table[2 * (int64_t)e], acc0);
                        acc1 = fmaf(w, table[2 * (int64_t)e + 1], acc1);
                    }
                }
                enc[j][2 * l] = round_half(acc0); enc[j][2 * l + 1] = round_half(acc1);
            }
        }
        for (int j = 0; j < m; j++) {
            const int64_t i = i0 + j;
            float cin[32], out[16];
            mlp_one(enc[j], WdT, n_enc, 64, 1, 16, 0, accumulate_half, cin + 16, NULL, 0);
            memcpy(h_out + 16 * i, cin + 16, sizeof(float) * 16);
            sigma[i] = expf(cin[16]);
            float d01[3], o[16];
            for (int d = 0; d < 3; d++) d01[d] = round_half(dirs[3 * i + d] * 0.5f + 0.5f);
            sh4(d01[0] * 2.f - 1.f, d01[1] * 2.f - 1.f, d01[2] * 2.f - 1.f, o);
            for (int k = 0; k < 16; k++) cin[k] = round_half(o[k]);
            mlp_one(cin, WcT, 32, 64, 2, 16, 1, accumulate_half, out, NULL, 0);
            rgb[3 * i] = out[0]; rgb[3 * i + 1] = out[1]; rgb[3 * i + 2] = out[2];
        }
    }
    free(WdT); free(WcT);
}
void oracle_mlp_fw(const float* in, int64_t M, const float* W, int n_in, int width, int n_hidden, int n_out_pad, int out_act,
                   float* out, float* acts) {
    oracle_mlp_fw2(in, M, W, n_in, width, n_hidden, n_out_pad, out_act, 0, out, acts);
}
/* d_out (M,n_out_pad) fp16-representable upstream gradient w.r.t. the (activated) output; out = forward output.
 * Produces dW (same layout as W, f32) and d_in (M,n_in) f32. Gradients entering a matrix product are rounded to fp16
 * (they are MFMA operands on the device); accumulation is f32. */
void oracle_mlp_bw(const float* in, int64_t M, const float* W, int n_in, int width, int n_hidden, int n_out_pad, int out_act,
                   const float* out, const float* acts, const float* d_out, float* dW, float* d_in) {
    int64_t w_off[16]; int w_k[16]; int64_t off = 0; int k = n_in;
    for (int l = 0; l < n_hidden; l++) { w_off[l] = off; w_k[l] = k; off += (int64_t)width * k; k = width; }
    w_off[n_hidden] = off; w_k[n_hidden] = width; off += (int64_t)n_out_pad * width;
    memset(dW, 0, sizeof(float) * off);
    for (int64_t i = 0; i < M; i++) {
        float dz[128], dh[128];
        /* output layer */
        for (int o = 0; o < n_out_pad; o++) {
            float g = d_out[i * n_out_pad + o];
            if (out_act == 1) { const float y = out[i * n_out_pad + o]; g = g * y * (1.f - y); }
            dz[o] = round_half(g);
        }
        const float* hl = acts + ((int64_t)(n_hidden - 1) * M + i) * width;
        const float* w = W + w_off[n_hidden];
        float* gw = dW + w_off[n_hidden];
        for (int o = 0; o < n_out_pad; o++) for (int c = 0; c < width; c++) gw[o * width + c] += dz[o] * hl[c];
        for (int c = 0; c < width; c++) { float a = 0.f; for (int o = 0; o < n_out_pad; o++) a = fmaf(w[o * width + c], dz[o], a); dh[c] = a; }
        for (int l = n_hidden - 1; l >= 0; l--) {
            const float* h_this = acts + ((int64_t)l * M + i) * width;
            for (int c = 0; c < width; c++) dz[c] = round_half(h_this[c] > 0.f ? dh[c] : 0.f);
            const float* h_prev = l > 0 ? acts + ((int64_t)(l - 1) * M + i) * width : in + i * n_in;
            const int kk = w_k[l];
            w = W + w_off[l]; gw = dW + w_off[l];
            for (int o = 0; o < width; o++) for (int c = 0; c < kk; c++) gw[o * kk + c] += dz[o] * h_prev[c];
            for (int c = 0; c < kk; c++) { float a = 0.f; for (int o = 0; o < width; o++) a = fmaf(w[o * kk + c], dz[o], a); dh[c] = a; }
        }
        if (d_in) memcpy(d_in + i * n_in, dh, sizeof(float) * n_in);
    }
}
