// ngp_net.hip -- fused "input encoding + tiny MLP" forward kernels (tinycudann subset used by nerficg's InstantNGP:
// src/Methods/InstantNGP/Model.py:58-114, queried at src/Methods/InstantNGP/Renderer.py:48-60).
//
//   ENC_GRID  : 16-level x 2-feature multiresolution hash grid (fp16 table) -> 32 inputs
//   ENC_SH_ID : degree-4 spherical harmonics of 3 dims (16) + identity of 16 dims -> 32 inputs
//   MLP       : 32 -> 64 (xN_HIDDEN, ReLU) -> 16 (padded), fp16 weights/activations, f32 accumulation on MFMA
//
// One wave owns a tile of 32 samples: two lanes per sample gather 8 levels x 8 corners each straight into the B
// fragments of the first MFMA; activations never leave registers between layers (see ngp_net.h).  Weights (<= 14 KB)
// live in the wave's VGPRs for the whole persistent loop.  Per sample the kernel reads 12 B (+ the gathers, served by
// L2 / Infinity Cache: the 24.4 MB table never streams from HBM twice) and writes 8-32 B.
#include "ngp_net.h"

namespace {

enum { ENC_GRID = 0, ENC_SH_ID = 1, ENC_FEAT = 2, ENC_DIR_H = 3, ENC_GRID_F2 = 4, ENC_GRID_F4 = 5 };  // ENC_GRID_F2 / F4: any (levels, 2 or 4 features per level) with levels x features <= 32  // ENC_DIR_H: SH of f32 directions + the density net's 16 fp16 outputs (fused training query)  // ENC_FEAT: fragment-major features written by k_grid_encode (see there)
enum { ACT_NONE = 0, ACT_SIGMOID = 1 };

// ---- first-layer B fragments from the encodings -------------------------------------------------------------------
__device__ __forceinline__ void encode_grid(float px, float py, float pz, int hh, __amdgpu_buffer_rsrc_t table, const GridCfg& g, h8 (&B)[2]) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int l0 = 8 * s + q, l1 = 8 * s + 4 + q;  // level of lane half 0 / 1 (uniform indices -> SGPR loads + select)
            const float scale = hh ? g.scale[l1] : g.scale[l0];
            const uint32_t res = hh ? g.res[l1] : g.res[l0];
            const uint32_t size = hh ? g.size[l1] : g.size[l0];
            const uint32_t off = hh ? g.offset[l1] : g.offset[l0];
            const bool hashed = hh ? g.hashed[l1] : g.hashed[l0];
            Corner8 c;
            grid_corners(px, py, pz, scale, res, size, off, hashed, c);
            float f0, f1;
            grid_level_features(table, c, f0, f1);
            B[s][2 * q] = (_Float16)f0;
            B[s][2 * q + 1] = (_Float16)f1;
        }
    }
}
// Any grid the reference's yaml can ask for whose encoding fits the 32 first-layer inputs (src/Methods/InstantNGP/Model.py:18-29,58-70 forward
// HASHGRID_N_LEVELS and HASHGRID_N_FEATURES_PER_LEVEL into the tcnn config): F = 2 or 4 features per table entry, input k = level * F + c, zero behind
// n_levels * F (tiny-cuda-nn pads the encoding to a multiple of 16).  Plain 4- / 8-byte gathers, eight per level: the shipped 16 x 2 configuration keeps
// its own kernels (k_grid_encode and encode_grid above); this is the drop-in's general path, not the headline's.
template <int F>
__device__ __forceinline__ void encode_grid_general(float px, float py, float pz, int hh, __amdgpu_buffer_rsrc_t table, const GridCfg& g, int n_levels, h8 (&B)[2]) {
    constexpr int LPS = 8 / F;   // levels per lane half and k-step
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
        for (int q = 0; q < LPS; q++) {
            const int l0 = s * (16 / F) + q, l1 = l0 + LPS;  // level of lane half 0 / 1 (uniform indices -> SGPR loads + select)
            const float scale = hh ? g.scale[l1] : g.scale[l0];
            const uint32_t res = hh ? g.res[l1] : g.res[l0];
            const uint32_t size = hh ? g.size[l1] : g.size[l0];
            const uint32_t off = hh ? g.offset[l1] : g.offset[l0];
            const bool hashed = hh ? g.hashed[l1] : g.hashed[l0];
            const bool live = (hh ? l1 : l0) < n_levels;    // levels behind the grid: the defaults of make_grid_cfg (in bounds), then zeroed
            Corner8 c;
            grid_corners(px, py, pz, scale, res, size, off, hashed, c);
            float acc[F];
#pragma unroll
            for (int f = 0; f < F; f++) acc[f] = 0.f;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if constexpr (F == 2) {
                    const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(table, c.e[k] << 2, 0, 0);
                    fma_half2(c.w[k], v, acc[0], acc[1]);
                } else {
                    const auto v = __builtin_amdgcn_raw_buffer_load_b64(table, c.e[k] << 3, 0, 0);
                    fma_half2(c.w[k], v[0], acc[0], acc[1]);
                    fma_half2(c.w[k], v[1], acc[2], acc[3]);
                }
            }
#pragma unroll
            for (int f = 0; f < F; f++) B[s][F * q + f] = live ? (_Float16)acc[f] : (_Float16)0.f;
        }
    }
}
// input row: [d01 (3) | h (16)] fp16, row stride `ld` halves
__device__ __forceinline__ void encode_sh_id(const __half* __restrict__ in, int ld, int64_t i, int hh, h8 (&B)[2]) {
    const _Float16* row = reinterpret_cast<const _Float16*>(in) + i * ld;
    float sh[16];
    sh4_eval((float)row[0] * 2.f - 1.f, (float)row[1] * 2.f - 1.f, (float)row[2] * 2.f - 1.f, sh);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        B[0][j] = (_Float16)(hh ? sh[8 + j] : sh[j]);
        B[1][j] = row[3 + 8 * hh + j];
    }
}

#ifndef NRC_BWD_RECOMPUTE
#define NRC_BWD_RECOMPUTE 1   // the second hidden layer's activations are recomputed in the backward pass instead of saved by the forward pass (A/B builds: -DNRC_BWD_RECOMPUTE=0)
#endif
// ---- generic per-network forward -----------------------------------------------------------------------------------
template <int ENC, int N_HIDDEN, int OUT_ACT, bool SAVE>
__global__ void __launch_bounds__(256) k_nwie_fwd(const void* __restrict__ input, int in_ld, int64_t M, const __half* __restrict__ W,
                                                  const __half2* __restrict__ table, GridCfg g, int n_out_rows,
                                                  __half* __restrict__ out, int out_ld, int n_store, __half* __restrict__ save_in,
                                                  __half* __restrict__ save_acts, float* __restrict__ sigmas_f32 = nullptr, float* __restrict__ rgbs_f32 = nullptr,
                                                  const int32_t* __restrict__ m_live = nullptr, int grid_levels = 16) {
    // sigmas_f32 / rgbs_f32 (ENC_DIR_H only, training query): the f32 outputs query_model hands to the compositor -- sigma = exp(h0) (TruncExp forward,
    // custom_functions.py:201-204) from the density row this kernel reads anyway, rgb = the three fp16 sigmoid outputs widened -- written by the
    // epilogue instead of a separate element-wise kernel over the two fp16 tensors
    // m_live (optional, DEVICE): the rows that hold samples -- a launch sized for a row CAPACITY (fixed-capacity training batches) works on those only;
    // the layout of the saved state keeps following the capacity
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int64_t n_tiles_cap = (M + 31) / 32;
    if (m_live) M = min(M, (int64_t)max(*m_live, 0));
    const int64_t n_tiles = (M + 31) / 32;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    const __amdgpu_buffer_rsrc_t trs = make_table_rsrc(table, g.total_entries * (ENC == ENC_GRID_F4 ? 8u : 4u));

    h8 A0[2][2], AH[N_HIDDEN > 1 ? N_HIDDEN - 1 : 1][2][4], AO[2];
    const __half* Wp = W;
#pragma unroll
    for (int mt = 0; mt < 2; mt++)
#pragma unroll
        for (int s = 0; s < 2; s++) A0[mt][s] = load_w_frag<false>(Wp, 32, 64, mt, s, r, hh);
    Wp += 64 * 32;
#pragma unroll
    for (int l = 0; l < N_HIDDEN - 1; l++) {
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int s = 0; s < 4; s++) AH[l][mt][s] = load_w_frag<true>(Wp, 64, 64, mt, s, r, hh);
        Wp += 64 * 64;
    }
#pragma unroll
    for (int s = 0; s < 2; s++) AO[s] = load_w_head_frag(Wp, 64, n_out_rows, s, lane);   // 16-row output layer: v_mfma_f32_16x16x32_f16 (ngp_net.h), as in k_ngp_mlp

    // The raw inputs of a tile (two 16-byte vectors, or a direction + one vector) are requested ONE TILE AHEAD: a training batch is 4-16 tiles per
    // wave, and without it every tile began with an exposed memory round trip in front of its MFMA chain (round 3: the same finding as in k_ngp_mlp).
    struct RawIn { uint4 a, b; float d0, d1, d2; };
    auto fetch_raw = [&](int64_t tile, RawIn& q) {
        const int64_t i = tile * 32 + r;
        const int64_t ic = i < M ? i : M - 1;
        if constexpr (ENC == ENC_FEAT) {
            const uint4* fp = reinterpret_cast<const uint4*>(input) + tile * 128 + r;
            const int rot = (int)(tile & 3);
            q.a = fp[((hh + rot) & 3) * 32]; q.b = fp[((2 + hh + rot) & 3) * 32];
        } else if constexpr (ENC == ENC_DIR_H) {
            const float* dp = reinterpret_cast<const float*>(input) + 3 * ic;
            q.d0 = dp[0]; q.d1 = dp[1]; q.d2 = dp[2];
            q.b = *reinterpret_cast<const uint4*>(reinterpret_cast<const _Float16*>(table) + ic * 16 + 8 * hh);
        }
    };
    RawIn nxt_raw;
    if (wave0 < n_tiles) fetch_raw(wave0, nxt_raw);
    for (int64_t tile = wave0; tile < n_tiles; tile += n_waves) {
        const int64_t i = tile * 32 + r;
        const bool valid = i < M;
        const int64_t ic = valid ? i : M - 1;
        const RawIn raw = nxt_raw;
        if (tile + n_waves < n_tiles) fetch_raw(tile + n_waves, nxt_raw);
        h8 B[2];
        if constexpr (ENC == ENC_GRID) {
            const float* xp = reinterpret_cast<const float*>(input) + 3 * ic;
            encode_grid(xp[0], xp[1], xp[2], hh, trs, g, B);
        } else if constexpr (ENC == ENC_GRID_F2 || ENC == ENC_GRID_F4) {
            const float* xp = reinterpret_cast<const float*>(input) + 3 * ic;
            encode_grid_general<(ENC == ENC_GRID_F4 ? 4 : 2)>(xp[0], xp[1], xp[2], hh, trs, g, grid_levels, B);
        } else if constexpr (ENC == ENC_FEAT) {
            B[0] = *reinterpret_cast<const h8*>(&raw.a);
            B[1] = *reinterpret_cast<const h8*>(&raw.b);
        } else if constexpr (ENC == ENC_DIR_H) {
            // `input` = directions (M,3) f32, `table` = the density network's output rows (M,16) fp16
            h8 lo, hi;
            sh4_fragments(raw.d0, raw.d1, raw.d2, lo, hi);
            B[0] = hh ? hi : lo;
            B[1] = *reinterpret_cast<const h8*>(&raw.b);
        } else {
            encode_sh_id(reinterpret_cast<const __half*>(input), in_ld, ic, hh, B);
        }
        if constexpr (SAVE) {
            if (!valid) {   // lanes past M: a finite column (their inputs may be uninitialised workspace)
#pragma unroll
                for (int j = 0; j < 8; j++) { B[0][j] = (_Float16)0.f; B[1][j] = (_Float16)0.f; }
            }
            // Saved state, FRAGMENT-major: [tile][fragment][lane half][sample of the tile] x 16 bytes, rows padded to whole tiles
            // (nrc_nwie_save_rows).  A store instruction writes two contiguous 512-byte blocks; sample-major rows made every one of the ~20 stores
            // of a tile touch 64 different lines, 8 or 16 bytes each -- at ~4 clocks per line in the address path that, not the MFMA chain, was
            // the forward kernels' time on a training batch (and the backward's loads had the same shape).  Rows past M hold the activations of a
            // zero input (finite: the backward multiplies them by a zero gradient).
            _Float16* p = reinterpret_cast<_Float16*>(save_in) + ((tile * 4 + hh) * 32 + r) * 8;
            *reinterpret_cast<h8*>(p) = B[0];
            *reinterpret_cast<h8*>(p + 2 * 32 * 8) = B[1];
        }
        f16v acc[2] = {zero16(), zero16()};
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int s = 0; s < 2; s++) acc[mt] = NRC_MFMA(A0[mt][s], B[s], acc[mt]);
        h8 H[4];
#pragma unroll
        for (int l = 0; l < N_HIDDEN; l++) {
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int gq = 0; gq < 2; gq++) H[2 * mt + gq] = acc_to_frag_relu(acc[mt], gq);
            // (the SECOND hidden layer of a two-hidden-layer network is not saved: k_nwie_bwd recomputes it from the first with the same eight MFMAs --
            //  128 B per sample less to write here and to read there, NRC_BWD_RECOMPUTE)
            if constexpr (SAVE) if (!(NRC_BWD_RECOMPUTE && l == 1)) {
                _Float16* p = reinterpret_cast<_Float16*>(save_acts) + ((int64_t)l * n_tiles_cap * 32) * 64 + ((tile * 8 + hh) * 32 + r) * 8;
#pragma unroll
                for (int s = 0; s < 4; s++) *reinterpret_cast<h8*>(p + s * 2 * 32 * 8) = H[s];
            }
            if (l + 1 < N_HIDDEN) {
                acc[0] = zero16(); acc[1] = zero16();
#pragma unroll
                for (int mt = 0; mt < 2; mt++)
#pragma unroll
                    for (int s = 0; s < 4; s++) acc[mt] = NRC_MFMA(AH[l][mt][s], H[s], acc[mt]);
            }
        }
        h8 HB[2][2];
        head_split(H, HB);
        f4v o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
        for (int s = 0; s < 2; s++) { o0 = NRC_MFMA16(AO[s], HB[0][s], o0); o1 = NRC_MFMA16(AO[s], HB[1][s], o1); }
        if constexpr (OUT_ACT == ACT_SIGMOID) {
#pragma unroll
            for (int j = 0; j < 4; j++) { o0[j] = 1.f / (1.f + expf(-o0[j])); o1[j] = 1.f / (1.f + expf(-o1[j])); }
        }
        const h8 ov = head_join(o0, o1);     // element j = output neuron 8 hh + j of this lane's sample
        if (valid) {
            h4 lo, hi;
#pragma unroll
            for (int j = 0; j < 4; j++) { lo[j] = ov[j]; hi[j] = ov[4 + j]; }
            _Float16* p = reinterpret_cast<_Float16*>(out) + i * out_ld;
            if (8 * hh < n_store) *reinterpret_cast<h4*>(p + 8 * hh) = lo;
            if (8 * hh + 4 < n_store) *reinterpret_cast<h4*>(p + 8 * hh + 4) = hi;
            if constexpr (ENC == ENC_DIR_H) {
                if (sigmas_f32 && hh == 0) {
                    sigmas_f32[i] = expf((float)B[1][0]);   // B[1] of lane half 0 = density outputs 0-7 of this sample, as stored (fp16)
                    rgbs_f32[3 * i] = (float)lo[0]; rgbs_f32[3 * i + 1] = (float)lo[1]; rgbs_f32[3 * i + 2] = (float)lo[2];
                }
            }
        }
    }
}

// ---- query pipeline: encode kernel + MLP kernel --------------------------------------------------------------------
// Measured on MI355X (profiles/r01_*): a single kernel that gathers AND runs the MFMA chain is pinned at 1-2 waves per SIMD
// by its register footprint and cannot keep the texture-address path busy (18-36 ms for 77 M samples, whichever way it was
// scheduled), while its two halves run in 6.5 ms (gathers, latency hidden by 8 waves/SIMD) and 2.8 ms (MFMA) on their own.
// The pipeline is therefore split at the 32 encoded features (64 B/sample, fp16, level-major so that both sides are
// coalesced) and processed in chunks of NRC_QUERY_CHUNK slots.
//
//   k_grid_encode : one lane per sample, all 16 levels (128 gathers, issued as 16-byte pair loads).  Bound by the L1
//                   texture-cache access rate (~1 line / clk / CU: TCP_TOTAL_CACHE_ACCESSES / GRBM_GUI_ACTIVE in profiles/).
//                   An XCD-affine variant (blockIdx%8 -> level pair, so that each 4 MB L2 serves 2 levels) was measured:
//                   L2 hit rate 91 %, but the 8x repeated sample fetch cost more than it saved (12.2 vs 10.9 ms / image).
//   k_ngp_mlp     : density net -> sigma -> colour net from the level-major features; weights as LDS-resident A fragments.
enum { SRC_ARRAYS = 0, SRC_TILED = 1 };
struct QueryIn {
    const float* xyz01; const float* dirs;                          // SRC_ARRAYS: sample i = row i of both arrays
    float* x01_out; int normalise;                                  // SRC_ARRAYS, training: xyz01 holds WORLD positions, (x - mn) / sz is applied
                                                                    // here (torch's two roundings) and written to x01_out for the backward
    const float* ts; const int32_t* row_tile; const float* ray_od;  // SRC_TILED: slot i = (row i>>6, lane i&63), ray = row_tile[row]*64 + lane
                                                                    // SRC_TILED, slab order: a row whose tile has finished carries row_tile = -1 - tile (written by
                                                                    // the slab compositor): the query kernels skip it without a look at anything else
    const int32_t* n_rows_dev;                                      // SRC_TILED, optional: the number of rows the march produced, on the DEVICE -- a launch sized
                                                                    // for a row CAPACITY processes only the rows that exist (fixed-capacity image pipeline)
    const int32_t* tile_off; int32_t arena_rows;                    // SRC_TILED, optional: `ts` is the count pass's ARENA (arena_rows rows of 64 per ray tile,
                                                                    // sample k of tile rt in arena row rt * arena_rows + k) and slot i's t is read from there --
                                                                    // row i >> 6 is sample (i >> 6) - tile_off[rt] of its tile; the copy into compact rows is skipped
    const int32_t* row_k;                                           // ... or, for rows in slab-major order, sample row_k[i >> 6] of its tile (tile_off unused)
    float mn[3], sz[3];                                             // xyz_min, xyz_size of the model box (Renderer.py:50)
    const int32_t* m_live;                                          // SRC_ARRAYS, optional: rows that hold samples (DEVICE), see k_nwie_fwd
};
// index of slot i (row i >> 6 of ray tile rt) in `ts`: the slot itself, or its place in the count pass's arena
__device__ __forceinline__ int64_t ts_slot(const QueryIn& in, int64_t i, int32_t rt) {
    if (in.arena_rows <= 0) return i;
    const int64_t k = in.row_k ? (int64_t)in.row_k[i >> 6] : (i >> 6) - (int64_t)in.tile_off[rt];
    return (((int64_t)rt * in.arena_rows + k) << 6) + (i & 63);
}
// row_tile entry -> ray tile (finished tiles' rows hold -1 - tile)
__device__ __forceinline__ int32_t tile_of(int32_t rt) { return rt < 0 ? -1 - rt : rt; }
// 1: a sample; 0: a hole of the tiled layout (no sample in this slot); -1: a row of a finished tile (slab order): nothing of it is read or written
template <int SRC, bool UNIFORM_ROW = true>
__device__ __forceinline__ int fetch_pos(const QueryIn& in, int64_t i, float& px, float& py, float& pz) {
    if constexpr (SRC == SRC_ARRAYS) {
        px = in.xyz01[3 * i]; py = in.xyz01[3 * i + 1]; pz = in.xyz01[3 * i + 2];
        if (in.normalise) {
            px = __fdiv_rn(__fsub_rn(px, in.mn[0]), in.sz[0]); py = __fdiv_rn(__fsub_rn(py, in.mn[1]), in.sz[1]);
            pz = __fdiv_rn(__fsub_rn(pz, in.mn[2]), in.sz[2]);
            if (in.x01_out) { in.x01_out[3 * i] = px; in.x01_out[3 * i + 1] = py; in.x01_out[3 * i + 2] = pz; }
        }
        return 1;
    } else {
        // one wave = one row of the tiled layout: the row's ray tile is wave-uniform and comes through the scalar cache (no vector round trip in
        // front of the six ray loads that depend on it)
        const int32_t rt = UNIFORM_ROW ? in.row_tile[__builtin_amdgcn_readfirstlane((int)(i >> 6))] : in.row_tile[i >> 6];
        if (rt < 0) { px = py = pz = 0.f; return -1; }
        const float t = in.ts[ts_slot(in, i, rt)];
        if (t < 0.f) { px = py = pz = 0.f; return 0; }
        const float* od = in.ray_od + (int64_t)rt * 384 + (i & 63);  // per-tile SoA [6][64]
        // same roundings as the op-by-op path: xyz = o + t*d (mul, add: raymarching.cu:368), then (xyz - min) / size in torch
        px = __fsub_rn(__fadd_rn(od[0], __fmul_rn(t, od[192])), in.mn[0]);
        py = __fsub_rn(__fadd_rn(od[64], __fmul_rn(t, od[256])), in.mn[1]);
        pz = __fsub_rn(__fadd_rn(od[128], __fmul_rn(t, od[320])), in.mn[2]);
        if (in.sz[0] != 1.0f || in.sz[1] != 1.0f || in.sz[2] != 1.0f) {  // x / 1.0f == x exactly: skip the IEEE division for the unit box
            px = __fdiv_rn(px, in.sz[0]); py = __fdiv_rn(py, in.sz[1]); pz = __fdiv_rn(pz, in.sz[2]);
        }
        return 1;
    }
}
template <int SRC>
__device__ __forceinline__ void fetch_dir(const QueryIn& in, int64_t i, float& dx, float& dy, float& dz) {
    if constexpr (SRC == SRC_ARRAYS) {
        dx = in.dirs[3 * i]; dy = in.dirs[3 * i + 1]; dz = in.dirs[3 * i + 2];
    } else {
        const float* od = in.ray_od + (int64_t)tile_of(in.row_tile[i >> 6]) * 384 + (i & 63);
        dx = od[192]; dy = od[256]; dz = od[320];
    }
}

// feat: FRAGMENT-major fp16 features of the samples [base, base+n): the 16-byte vector
//   feat[((j >> 5) * 4 + ((g + (j >> 5)) & 3)) * 32 + (j & 31)]   holds levels 4g .. 4g+3 (2 features each) of sample j,
// which is, bit for bit, the first-layer B fragment of k-step g>>1 for lane half g&1 of the MFMA wave that owns samples
// 32*(j>>5) .. +31: the encoder writes 4 coalesced 16-byte stores per sample, the MLP kernel reads 2 coalesced 16-byte loads
// per lane and needs no unpacking.  The group position inside a tile's 2 KB record rotates with the tile index: all waves
// of the chip reach group g at about the same time, and un-rotated they would all write the same quarter of the memory
// channels (measured: 0.262 ms vs 0.22 ms per 2 Mi samples).  One lane per sample, all 16 levels.
template <int SRC>
__global__ void __launch_bounds__(256) k_grid_encode(QueryIn in, int64_t base, int64_t n, const __half2* __restrict__ table, GridCfg g,
                                                         uint4* __restrict__ feat, int narrow_levels, int hashed_mode, int lane_shape = 0) {
    int64_t bid = blockIdx.x;
    if constexpr (SRC == SRC_TILED) {
        // workgroups are dealt round-robin over the 8 XCDs; with NRC_ENC_XCD each XCD takes a CONTIGUOUS eighth of the launch's slots instead of
        // every eighth workgroup, so that neighbouring bricks (which share table lines) meet in the same L2
        if (hashed_mode & (1 << 28)) {
            const int64_t g8 = (int64_t)gridDim.x >> 3;
            if (bid < 8 * g8) bid = (bid & 7) * g8 + (bid >> 3);
        }
    }
    int64_t j = bid * 256 + threadIdx.x;
    bool remapped = false;
    if constexpr (SRC == SRC_TILED) {
        if (in.n_rows_dev) {   // uniform scalar load: the slots behind the marched rows belong to nobody
            const int64_t have = (int64_t)in.n_rows_dev[0] * 64 - base;
            n = have < n ? have : n;
        }
    }
    if constexpr (SRC == SRC_TILED) {
        if (lane_shape != 0) {
            // WHICH 64 samples a wave encodes.  Hashed entries are contiguous along x only: a wave whose samples share (y, z) columns shares cache
            // lines, one that covers 64 columns opens 64+ lines per gather -- and the miss traffic of the four finest levels is what bounds this
            // kernel.  With a wave = a row of the layout (8 x 8 pixels at ONE step) the time per launch followed the viewing direction: 0.50 ms
            // looking along z, 0.79 ms along x, correlation with |forward.x| 0.90 over the bench's poses.  The slots of 16 consecutive rows (the
            // same 64 rays at 16 consecutive steps) are therefore dealt to the 16 waves as BRICKS of 2^u pixels along the image row x 2^v image rows
            // x 2^(6-u-v) steps.  Measured, mean / worst pose of 24, us per launch of 7.7 M samples: 8x8x1 (a row) 680 / 795, 8x1x8 628 / 828,
            // 8x4x2 622 / 712, 4x4x4 615 / 685, 4x2x8 610 / 725, 8x2x4 604 / 723 -- the brick with the smallest cross-section next to its long
            // (image-row) side wins on almost every pose; choosing per pose between it, 8x1x8 and a row would gain another 1 %.  Default 8 x 2 x 4.
            // The slot a sample is stored in does not change, and neither does any value.
            const int lu = (lane_shape >> 4) & 7, lv = lane_shape & 7, ls = 6 - lu - lv;
            const int64_t g1024 = (base + j) & ~(int64_t)1023;
            const int w = (int)((j >> 6) & 15), l = (int)(j & 63);
            const int iu = l & ((1 << lu) - 1), iv = (l >> lu) & ((1 << lv) - 1), is = l >> (lu + lv);
            const int gu = w & ((NRC_TILE_W >> lu) - 1), gv = (w >> (NRC_TILE_W_LOG2 - lu)) & ((NRC_TILE_H >> lv) - 1), gs = w >> (6 - lu - lv);
            j = g1024 + ((int64_t)((gs << ls) + is) << 6) + NRC_TILE_W * ((gv << lv) + iv) + (gu << lu) + iu - base;
            remapped = true;
        }
    }
    if (j >= n || j < 0) return;
    float px, py, pz;
    if (blockIdx.y != 0) in.x01_out = nullptr;   // level groups split over workgroup rows: the first row writes the normalised positions
    const int state = remapped ? fetch_pos<SRC, false>(in, base + j, px, py, pz) : fetch_pos<SRC>(in, base + j, px, py, pz);
    if (state < 0) return;   // a row of a finished tile (slab order): the MLP kernel skips it too, nothing reads its features
    const bool live = state > 0;
    const __amdgpu_buffer_rsrc_t trs = make_table_rsrc(table, g.total_entries * 4u);
    uint4* out = feat + ((j >> 5) * 4) * 32 + (j & 31);
    const int rot = (int)((j >> 5) & 3);
    const int lvl_lo = (hashed_mode >> 4) & 0xff, lvl_hi = (hashed_mode >> 12) & 0xff;  // NRC_ENC_LEVELS (measurement only; 0 .. 16 normally)
    const int uniform_levels = (hashed_mode >> 20) & 0xff;                             // levels that try the wave-uniform scalar path first
    const bool skip_last_group = ((hashed_mode >> 29) & 1) != 0;                       // NRC_ENC_FINE_SPLIT: levels 12-15 come from k_grid_encode_fine
    hashed_mode &= 0xf;
    const uint32_t* __restrict__ table32 = reinterpret_cast<const uint32_t*>(table);
    // gridDim.y == 4: one group of four levels per workgroup row (small batches: four times the waves, a quarter of the dependent gathers each --
    // a 264 K-sample training batch is ~4 waves per SIMD in all and runs at the latency of ONE wave's sixteen gather rounds otherwise)
    const int grp_begin = gridDim.y == 4 ? (int)blockIdx.y : 0, grp_end = gridDim.y == 4 ? grp_begin + 1 : (skip_last_group ? 3 : 4);
#pragma unroll 1
    for (int grp = grp_begin; grp < grp_end; grp++) {
        uint32_t v[4] = {0u, 0u, 0u, 0u};
        if (live) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int level = 4 * grp + q;
                if (level < lvl_lo || level >= lvl_hi) continue;   // scalar, wave-uniform
                Corner8 c;
                float f0 = 0.f, f1 = 0.f;
                bool have = false;
                if constexpr (SRC == SRC_TILED) {   // rows of the tiled layout are spatially compact: try the scalar-cache path on the coarse levels
                    if (level < uniform_levels)
                        have = g.hashed[level] ? grid_level_features_uniform<true>(table32, px, py, pz, g.scale[level], g.res[level], g.size[level], g.offset[level], f0, f1)
                                               : grid_level_features_uniform<false>(table32, px, py, pz, g.scale[level], g.res[level], g.size[level], g.offset[level], f0, f1);
                }
                if (!have) {
                    if (g.hashed[level]) grid_corners_u<true>(px, py, pz, g.scale[level], g.res[level], g.size[level], g.offset[level], c);
                    else grid_corners_u<false>(px, py, pz, g.scale[level], g.res[level], g.size[level], g.offset[level], c);
                    if (level < narrow_levels) grid_level_features_narrow(trs, c, f0, f1);
                    else if (g.hashed[level] && hashed_mode == 1) grid_level_features_hashed(trs, c, f0, f1);
                    else if (g.hashed[level] && hashed_mode == 2) grid_level_features_pair(trs, c, f0, f1);
                    else grid_level_features(trs, c, f0, f1);
                }
                const __half2 h = __floats2half2_rn(f0, f1);
                v[q] = *reinterpret_cast<const uint32_t*>(&h);
                // finish this level before the next one starts: otherwise the compiler sinks all four interpolations below the
                // last gather and the 4 x 20 live registers cut the occupancy from 8 to 3 waves per SIMD
                // (two levels in flight per wave = 83 VGPRs / 5 waves per SIMD measured no faster: the kernel is throughput-bound)
                asm volatile("" : "+v"(v[q]));
            }
        }
        // (cache-policy variants of this store -- sc1, sc0 sc1, nt, so that the 64 B of features per sample would not take L2 lines from the hash
        // table -- measured 0.630-0.640 ms against 0.640: no effect, removed)
        out[((grp + rot) & 3) * 32] = make_uint4(v[0], v[1], v[2], v[3]);
    }
}

// Experiment (round-3 review, item 5b; NRC_ENC_FINE_SPLIT=1): the four finest levels in a launch of their own, ONE LEVEL PER XCD.  The whole table is
// 24.4 MB and an XCD's L2 4 MB: with all sixteen levels in one kernel every L2 sees all of it (hit rate 0.70, 7.5 of the 7.9 L1 misses per sample
// come from levels 12-15 and 30 % of those go on to the fabric).  Here workgroup b works on level 12 + (b & 3) for the slots of half (b >> 2) & 1
// (workgroups are dealt round-robin over the XCDs), so each L2 holds ONE 2 MB level.  Price: the position of a sample is derived five times instead
// of once and the level's 4 bytes are stored into the 16-byte group of the fragment-major record on their own.  Same values in the same places.
template <int SRC>
__global__ void __launch_bounds__(256) k_grid_encode_fine(QueryIn in, int64_t base, int64_t n, const __half2* __restrict__ table, GridCfg g, uint4* __restrict__ feat,
                                                              int first_level, int lane_shape) {
    const int xcd = (int)(blockIdx.x & 7u);
    const int level = first_level + (xcd & 3);
    const int64_t per_half = (int64_t)(gridDim.x >> 3);
    const int64_t bid = (int64_t)(xcd >> 2) * per_half + (int64_t)(blockIdx.x >> 3);
    int64_t j = bid * 256 + threadIdx.x;
    if constexpr (SRC == SRC_TILED) {
        if (in.n_rows_dev) {
            const int64_t have = (int64_t)in.n_rows_dev[0] * 64 - base;
            n = have < n ? have : n;
        }
        if (lane_shape != 0) {   // the same bricks as k_grid_encode
            const int lu = (lane_shape >> 4) & 7, lv = lane_shape & 7, ls = 6 - lu - lv;
            const int64_t g1024 = (base + j) & ~(int64_t)1023;
            const int w = (int)((j >> 6) & 15), l = (int)(j & 63);
            const int iu = l & ((1 << lu) - 1), iv = (l >> lu) & ((1 << lv) - 1), is = l >> (lu + lv);
            const int gu = w & ((NRC_TILE_W >> lu) - 1), gv = (w >> (NRC_TILE_W_LOG2 - lu)) & ((NRC_TILE_H >> lv) - 1), gs = w >> (6 - lu - lv);
            j = g1024 + ((int64_t)((gs << ls) + is) << 6) + NRC_TILE_W * ((gv << lv) + iv) + (gu << lu) + iu - base;
        }
    }
    if (j >= n || j < 0) return;
    float px, py, pz;
    in.x01_out = nullptr;
    const int state = fetch_pos<SRC, false>(in, base + j, px, py, pz);
    if (state < 0) return;
    const bool live = state > 0;
    const __amdgpu_buffer_rsrc_t trs = make_table_rsrc(table, g.total_entries * 4u);
    const int rot = (int)((j >> 5) & 3);
    uint32_t* out = reinterpret_cast<uint32_t*>(feat + ((j >> 5) * 4) * 32 + (j & 31) + ((3 + rot) & 3) * 32) + (level - first_level);
    uint32_t v = 0u;
    if (live) {
        Corner8 c;
        float f0, f1;
        grid_corners_u<true>(px, py, pz, g.scale[level], g.res[level], g.size[level], g.offset[level], c);
        grid_level_features_hashed(trs, c, f0, f1);
        const __half2 h = __floats2half2_rn(f0, f1);
        v = *reinterpret_cast<const uint32_t*>(&h);
    }
    *out = v;
}

// Small batches (a training iteration: ~264 K samples, once, right after the optimizer rewrote the fp16 table: every XCD's L2 is cold).  With one
// lane per sample and all sixteen levels each of the 8 XCDs pulls the whole 24.4 MB table through its own L2 (measured 63-67 us = 0.24 of the
// HBM roofline on 512 B per sample).  Here workgroup b encodes a level PAIR of sample block b >> 3: workgroups are dealt round-robin over the
// XCDs, so XCD x mostly sees two levels (L2 hit rate 0.87 against 0.61, profiles/).  The pairs are BALANCED, (p, 15 - p): a coarse level (few
// cache lines per wave) with a fine one (64 lines per gather) -- adjacent pairs (2p, 2p + 1) put both finest levels on one XCD, whose L1s then
// bound the launch: 94 us.  Measured: 63 -> 54 us.  The same split cost the inference path more than it saved (every sample record is
// read once per pair, 8 x 7.7 M records per launch); at this size the positions are 3 MB.  Placement is a speed assumption only: any
// workgroup-to-XCD mapping gives the same features.
__global__ void __launch_bounds__(256) k_grid_encode_pairs(QueryIn in, int64_t n, const __half2* __restrict__ table, GridCfg g, uint2* __restrict__ feat,
                                                           int narrow_levels, int hashed_mode, const int32_t* __restrict__ m_live = nullptr) {
    const int pair = (int)(blockIdx.x & 7u);
    const int64_t j = (int64_t)(blockIdx.x >> 3) * 256 + threadIdx.x;
    if (m_live) n = min(n, ((int64_t)max(*m_live, 0) + 31) / 32 * 32);   // whole MLP tiles of the live rows (the rows behind them hold finite, inert samples)
    if (j >= n) return;
    float px, py, pz;
    if (pair != 0) in.x01_out = nullptr;
    fetch_pos<SRC_ARRAYS>(in, j, px, py, pz);
    const __amdgpu_buffer_rsrc_t trs = make_table_rsrc(table, g.total_entries * 4u);
    // balanced pairs: a coarse level (few distinct cache lines per wave) with a fine one (64 lines per gather): levels p and 15 - p
    uint32_t* out32 = reinterpret_cast<uint32_t*>(feat);
    const int rot = (int)((j >> 5) & 3);
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int level = q == 0 ? pair : NRC_MAX_LEVELS - 1 - pair;
        Corner8 c;
        float f0, f1;
        if (g.hashed[level]) grid_corners_u<true>(px, py, pz, g.scale[level], g.res[level], g.size[level], g.offset[level], c);
        else grid_corners_u<false>(px, py, pz, g.scale[level], g.res[level], g.size[level], g.offset[level], c);
        if (level < narrow_levels) grid_level_features_narrow(trs, c, f0, f1);
        else if (g.hashed[level] && hashed_mode == 1) grid_level_features_hashed(trs, c, f0, f1);
        else if (g.hashed[level] && hashed_mode == 2) grid_level_features_pair(trs, c, f0, f1);
        else grid_level_features(trs, c, f0, f1);
        const __half2 h = __floats2half2_rn(f0, f1);
        const int grp = level >> 2;
        out32[(((j >> 5) * 4 + ((grp + rot) & 3)) * 32 + (j & 31)) * 4 + (level & 3)] = *reinterpret_cast<const uint32_t*>(&h);
    }
}

// SH degree 4 of the ray direction as the colour net's k-step-0 B fragments, once per RAY of the tiled layout:
//   ray_sh[(tile * 2 + hh) * 64 + lane] = fp16 coefficients 8hh .. 8hh+7 of ray (tile, lane)
// (a sample's direction is its ray's: evaluating per sample would repeat ~80 VALU instructions 120 times per ray)
// rows_tile_off / row_tile_out (arena frames): the same launch names the ray tile of every row -- rows tile_off[t] .. tile_off[t + 1] - 1 belong to
// tile t (what nrc_ngp_render_write does next to its copy; here there is no copy and no second launch)
__global__ void __launch_bounds__(256) k_ray_sh(const float* __restrict__ ray_od, int64_t n_ray_tiles, h8* __restrict__ ray_sh,
                                                const int32_t* __restrict__ rows_tile_off = nullptr, int32_t* __restrict__ row_tile_out = nullptr,
                                                int64_t n_rows = 0) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_ray_tiles * 64) return;
    const int64_t tile = i >> 6;
    const int lane = (int)(i & 63);
    if (row_tile_out) {
        const int64_t r1 = min((int64_t)rows_tile_off[tile + 1], n_rows);
        for (int64_t r = (int64_t)rows_tile_off[tile] + lane; r < r1; r += 64) row_tile_out[r] = (int32_t)tile;
    }
    const float* od = ray_od + tile * 384 + lane;
    h8 lo, hi;
    sh4_fragments(od[192], od[256], od[320], lo, hi);
    ray_sh[(tile * 2) * 64 + lane] = lo;
    ray_sh[(tile * 2 + 1) * 64 + lane] = hi;
}

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// NT = sample tiles (32 samples each) per wave iteration.  With NT = 2 every weight fragment read from LDS feeds two MFMAs and the
// two chains interleave (the chain of one tile is serial: MFMA -> convert -> MFMA ...).
// (the three wrappers below were the seams of the round-2/3 ablation builds -- no conversion, no MFMA, synthetic inputs; numbers in LABBOOK.md)
__device__ __forceinline__ h8 mlp_frag_relu(const f16v& acc, int g) { return acc_to_frag_relu(acc, g); }
__device__ __forceinline__ h8 mlp_frag(const f16v& acc, int g) { return acc_to_frag(acc, g); }
__device__ __forceinline__ f16v mlp_mfma(const h8& a, const h8& b, const f16v& c) { return NRC_MFMA(a, b, c); }
#ifndef NRC_MLP_WAVES
#define NRC_MLP_WAVES 2   // workgroups per CU the register budget is set for (= waves per SIMD); A/B builds: -DNRC_MLP_WAVES=3
#endif
template <int SRC, int NT>
__global__ void __launch_bounds__(256, NRC_MLP_WAVES) k_ngp_mlp(QueryIn in, int64_t base, int64_t n, const uint4* __restrict__ feat,
                                                    const h8* __restrict__ ray_sh, const __half* __restrict__ Wd,
                                                    const __half* __restrict__ Wc, float* __restrict__ sigmas, float* __restrict__ rgbs,
                                                    __half* __restrict__ packed) {
    enum { F_D0 = 0, F_DO = 4, F_C0 = 8, F_C1 = 12, F_CO = 20, N_FRAG = 24 };
    if constexpr (SRC == SRC_TILED) {
        if (in.n_rows_dev) {   // fixed row capacity: only the rows the march produced (uniform over the launch)
            const int64_t have = (int64_t)in.n_rows_dev[0] * 64 - base;
            n = have < n ? have : n;
            if (n <= 0) return;
        }
    }
    __shared__ h8 wlds[N_FRAG][64];
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    const int64_t n_tiles = (n + 31) / 32;
    const int64_t n_groups = (n_tiles + NT - 1) / NT;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    for (int f = threadIdx.x >> 6; f < N_FRAG; f += 4) {
        h8 v;
        if (f < F_DO) v = load_w_frag<false>(Wd, 32, 64, (f - F_D0) >> 1, (f - F_D0) & 1, r, hh);
        else if (f < F_C0) v = load_w_head_frag(Wd + 64 * 32, 64, 16, (f - F_DO) & 1, lane);     // 16-row head: two 16x16x32 k-steps (slots F_DO + 2, + 3 unused)
        else if (f < F_C1) {
            const int mt = (f - F_C0) >> 1, s = (f - F_C0) & 1;  // k-step 0 = SH coefficients, 1 = density features: both natural order (head_join)
            v = load_w_frag<false>(Wc, 32, 64, mt, s, r, hh);
        } else if (f < F_CO) v = load_w_frag<true>(Wc + 64 * 32, 64, 64, (f - F_C1) >> 2, (f - F_C1) & 3, r, hh);
        else v = load_w_head_frag(Wc + 64 * 32 + 64 * 64, 64, 16, (f - F_CO) & 1, lane);
        wlds[f][lane] = v;
    }
    __syncthreads();
#define D0(mt, s) wlds[F_D0 + 2 * (mt) + (s)][lane]
#define DO(s) wlds[F_DO + (s)][lane]
#define C0(mt, s) wlds[F_C0 + 2 * (mt) + (s)][lane]
#define C1(mt, s) wlds[F_C1 + 4 * (mt) + (s)][lane]
#define CO(s) wlds[F_CO + (s)][lane]
    // software pipeline: the loads of the NEXT group (features 2 x 16 B, SH 16 B or direction, hole flag) are issued before the MFMA
    // chain of the current one
    // A tile's row number (hence its ray tile) is uniform over the wave and known from the tile index alone, so it is fetched through the SCALAR
    // cache TWO wave iterations ahead (one SGPR per tile) -- when the vector loads of a tile are issued, every address is ready and nothing waits.
    // Round 2's form loaded row_tile with a vector load inside the prefetch and waited for it (s_waitcnt vmcnt) before it could address the SH
    // fragments and the hole flag: two exposed memory round trips per wave iteration in front of the MFMA chain; with all compute removed the
    // kernel still took 0.178 ms of its 0.197 (round-3 ablations).
    struct TileIn { uint4 b0, b1; h8 sh; float dx, dy, dz; float t; uint32_t alive; };
    auto tile_rt = [&](int64_t tile) -> int32_t {
        if constexpr (SRC == SRC_TILED) {
            const int64_t tc = tile < n_tiles ? tile : n_tiles - 1;
            const int row = __builtin_amdgcn_readfirstlane((int)(((base + (n > 0 ? (tc * 32 < n ? tc * 32 : n - 1) : 0)) >> 6)));
            return in.row_tile[row];   // uniform address: s_load_dword
        } else {
            return 0;
        }
    };
    auto fetch = [&](int64_t tile, int32_t rt, TileIn& ti, const TileIn* same_ray_tile = nullptr) {
        const int64_t tc = tile < n_tiles ? tile : n_tiles - 1;
        const int64_t j = tc * 32 + r;
        const int64_t i = base + (j < n ? j : n - 1);
        const uint4* fp = feat + tc * 128 + r;
        const int rot = (int)(tc & 3);
        if constexpr (SRC == SRC_TILED) {
            if (rt < 0) {   // (wave-uniform) a row of a finished tile: no load, no arithmetic, no store
                ti.b0 = make_uint4(0u, 0u, 0u, 0u); ti.b1 = ti.b0; ti.t = 0.f; ti.alive = 0u;
                for (int q = 0; q < 8; q++) ti.sh[q] = (_Float16)0.f;
                return;
            }
        }
        ti.b0 = fp[((hh + rot) & 3) * 32];
        ti.b1 = fp[((2 + hh + rot) & 3) * 32];
        ti.alive = 1u;
        if constexpr (SRC == SRC_TILED) {
            // (no look at the sample's t: a hole of the layout -- 4 % of the slots -- carries the encoder's all-zero features, runs through the
            // networks like any other slot and its output is never read (the compositor walks a ray's k < ray_cnt rows only); the 4-byte load per
            // slot and, on arena frames, the scalar load of tile_off in front of it cost more than the holes' arithmetic)
            ti.t = 0.f;
            // (contiguous rows per wave: the next row mostly belongs to the same ray tile -- its SH fragment is in registers already)
            if (same_ray_tile) ti.sh = same_ray_tile->sh;
            else ti.sh = ray_sh[((int64_t)rt * 2 + hh) * 64 + (i & 63)];
        } else {
            ti.t = 0.f;
            ti.dx = in.dirs[3 * i]; ti.dy = in.dirs[3 * i + 1]; ti.dz = in.dirs[3 * i + 2];
        }
    };
    // PF groups in flight per wave (ring slots indexed by the unrolled p).  Measured (round 3): two or three groups ahead cost registers (NT = 1:
    // 90 -> 126 -> 149 VGPRs, 5 -> 4 -> 3 waves per SIMD; NT = 2: 168 -> 209) and run 0.187 / 0.200 ms against 0.182 with one -- more bytes in
    // flight is not what the kernel lacks.  PF stays 1.
    constexpr int PF = 1;
    // Which groups a wave takes.  Strided (group = wave + k * n_waves) or, NRC_MLP_CONTIG, a CONTIGUOUS range of rows per wave (tiled layout):
    // consecutive rows of the layout belong to the same ray tile ~130 times in a row, so the 16-byte SH fragment of a lane's ray -- a fifth of
    // what the kernel streams per sample -- is loaded once per ray tile and wave instead of once per row.
    // Measured (800x800 bench frame, two boxes): 0.174-0.182 -> 0.171-0.174 ms per launch, frame 8.08-8.10 -> 8.05-8.08 ms.  -DNRC_MLP_CONTIG=0: strided.
#ifndef NRC_MLP_CONTIG
#define NRC_MLP_CONTIG 1
#endif
    constexpr bool CONTIG = NRC_MLP_CONTIG != 0 && SRC == SRC_TILED;
    const int64_t g_per = CONTIG ? (n_groups + n_waves - 1) / n_waves : 0;
    const int64_t g_first = CONTIG ? wave0 * g_per : wave0;
    const int64_t g_stop = CONTIG ? (g_first + g_per < n_groups ? g_first + g_per : n_groups) : n_groups;
    const int64_t g_step = CONTIG ? 1 : n_waves;
    TileIn ring[PF][NT];
    int32_t rt_ring[PF][NT];   // ray tiles of the group that the NEXT visit of the slot will prefetch
    int32_t rt_have[PF][NT];   // ray tiles of the data in the ring
#pragma unroll
    for (int p = 0; p < PF; p++) {
        const int64_t g0 = g_first + p * g_step;
#pragma unroll
        for (int u = 0; u < NT; u++) rt_have[p][u] = 0x7fffffff;
        if (g0 < g_stop) {
#pragma unroll
            for (int u = 0; u < NT; u++) {
                rt_have[p][u] = tile_rt(g0 * NT + u);
                fetch(g0 * NT + u, rt_have[p][u], ring[p][u]);
            }
        }
#pragma unroll
        for (int u = 0; u < NT; u++) rt_ring[p][u] = tile_rt((g0 + PF * g_step) * NT + u);
    }
    // The outputs of a group leave ONE ITERATION LATER, right behind the next prefetch: a wave's vector-memory operations complete in issue
    // order as far as s_waitcnt is concerned, and the wait for the prefetched inputs at the top of an iteration (vmcnt(0) across the loop edge)
    // otherwise also waits for the stores the previous iteration issued a moment ago -- a store round trip per iteration in front of the chain.
    h4 pend_pk[NT];
    int64_t pend_idx[NT];
    bool pend_ok[NT];
#pragma unroll
    for (int u = 0; u < NT; u++) { pend_ok[u] = false; pend_idx[u] = base; pend_pk[u] = h4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f}; }
    auto flush_pending = [&]() {
#pragma unroll
        for (int u = 0; u < NT; u++) {
            if (pend_ok[u]) {
                const int64_t i = pend_idx[u];
                if constexpr (SRC == SRC_ARRAYS) {
                    sigmas[i] = expf((float)pend_pk[u][0]);  // TruncExp forward (custom_functions.py:201-204)
#pragma unroll
                    for (int c = 0; c < 3; c++) rgbs[3 * i + c] = (float)pend_pk[u][1 + c];
                } else {
                    *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(packed) + 4 * i) = pend_pk[u];
                }
            }
            pend_ok[u] = false;
        }
    };
    for (int64_t grp0 = g_first; grp0 < g_stop; grp0 += PF * g_step) {
#pragma unroll
      for (int p = 0; p < PF; p++) {
        const int64_t grp = grp0 + p * g_step;
        if (grp >= g_stop) break;
        TileIn cur[NT];
#pragma unroll
        for (int u = 0; u < NT; u++) cur[u] = ring[p][u];
        if (grp + PF * g_step < g_stop) {
#pragma unroll
            for (int u = 0; u < NT; u++) {
                // (scalar) the next row of this wave is a row of the same ray tile -- AND slot u keeps its tile parity from group to group (NT even): with one
                // tile per group, consecutive groups are the two halves of a row and read different lanes of the ray's SH fragment (advisor finding, round 4)
                const bool same = CONTIG && (NT % 2 == 0) && rt_ring[p][u] == rt_have[p][u];
                fetch((grp + PF * g_step) * NT + u, rt_ring[p][u], ring[p][u], same ? &cur[u] : nullptr);
                rt_have[p][u] = rt_ring[p][u];
            }
        }
#pragma unroll
        for (int u = 0; u < NT; u++) rt_ring[p][u] = tile_rt((grp + 2 * PF * g_step) * NT + u);
        flush_pending();   // the previous group's outputs: issued behind this group's prefetch, complete long before the next wait
        bool valid[NT];
        int64_t idx[NT];
        bool any = false;
#pragma unroll
        for (int u = 0; u < NT; u++) {
            const int64_t tile = grp * NT + u;
            const int64_t j = tile * 32 + r;
            valid[u] = tile < n_tiles && j < n;
            idx[u] = base + (valid[u] ? j : n - 1);
            if constexpr (SRC == SRC_TILED) valid[u] = valid[u] && cur[u].t >= 0.f && cur[u].alive != 0u;
            any = any || valid[u];
        }
        if (__ballot(any) == 0ull) continue;  // nothing but holes
        // first-layer B fragments straight from the encoder's fragment-major records (512 contiguous bytes per half wave);
        // element (2q, 2q+1) of k-step s <- features of level 8s + 4hh + q
        h8 B[NT][2], X[NT][2], H[NT][4];
        f16v acc[NT][2];
        h8 HB[NT][2][2];      // head operands: [sample block][k-step] (head_split)
        f4v o[NT][2];         // head accumulators, one per 16-sample block
#pragma unroll
        for (int u = 0; u < NT; u++) {
            B[u][0] = *reinterpret_cast<const h8*>(&cur[u].b0);
            B[u][1] = *reinterpret_cast<const h8*>(&cur[u].b1);
            if constexpr (SRC == SRC_TILED) {
                X[u][0] = cur[u].sh;  // colour-net k-step 0 = SH(dir) (natural order)
            } else {
                h8 lo, hi;
                sh4_fragments(cur[u].dx, cur[u].dy, cur[u].dz, lo, hi);
                X[u][0] = hh ? hi : lo;
            }
            acc[u][0] = zero16(); acc[u][1] = zero16();
            o[u][0] = f4v{0.f, 0.f, 0.f, 0.f}; o[u][1] = o[u][0];
        }
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const h8 w = D0(mt, s);
#pragma unroll
                for (int u = 0; u < NT; u++) acc[u][mt] = mlp_mfma(w, B[u][s], acc[u][mt]);
            }
#pragma unroll
        for (int u = 0; u < NT; u++) {
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int gq = 0; gq < 2; gq++) H[u][2 * mt + gq] = mlp_frag_relu(acc[u][mt], gq);
            head_split(H[u], HB[u]);
        }
        // density head, 64 -> 16: two 16x16x32 k-steps per 16-sample block instead of four 32x32x16 steps over a half-empty tile
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const h8 w = DO(s);
#pragma unroll
            for (int u = 0; u < NT; u++)
#pragma unroll
                for (int nb = 0; nb < 2; nb++) o[u][nb] = NRC_MFMA16(w, HB[u][nb][s], o[u][nb]);
        }
        _Float16 h0[NT];
#pragma unroll
        for (int u = 0; u < NT; u++) {
            X[u][1] = head_join(o[u][0], o[u][1]);  // colour-net k-step 1 = fp16(h), natural order, back on the sample's lanes
            h0[u] = X[u][1][0];              // fp16 density feature 0 (lane half 0, element 0)
            acc[u][0] = zero16(); acc[u][1] = zero16();
        }
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const h8 w = C0(mt, s);
#pragma unroll
                for (int u = 0; u < NT; u++) acc[u][mt] = mlp_mfma(w, X[u][s], acc[u][mt]);
            }
#pragma unroll
        for (int u = 0; u < NT; u++) {
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int gq = 0; gq < 2; gq++) H[u][2 * mt + gq] = mlp_frag_relu(acc[u][mt], gq);
            acc[u][0] = zero16(); acc[u][1] = zero16();
        }
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                const h8 w = C1(mt, s);
#pragma unroll
                for (int u = 0; u < NT; u++) acc[u][mt] = mlp_mfma(w, H[u][s], acc[u][mt]);
            }
#pragma unroll
        for (int u = 0; u < NT; u++) {
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int gq = 0; gq < 2; gq++) H[u][2 * mt + gq] = mlp_frag_relu(acc[u][mt], gq);
            head_split(H[u], HB[u]);
            o[u][0] = f4v{0.f, 0.f, 0.f, 0.f}; o[u][1] = o[u][0];
        }
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const h8 w = CO(s);
#pragma unroll
            for (int u = 0; u < NT; u++)
#pragma unroll
                for (int nb = 0; nb < 2; nb++) o[u][nb] = NRC_MFMA16(w, HB[u][nb][s], o[u][nb]);
        }
#pragma unroll
        for (int u = 0; u < NT; u++) {
            pend_ok[u] = valid[u] && hh == 0;
            pend_idx[u] = idx[u];
            pend_pk[u][0] = h0[u];
            // rgb = neurons 0..2: lanes 0-15 of each block (k-group 0); a row swap puts sample r's triple on lane r, the sigmoid runs there
            float rgb[3];
            head_join_rgb(o[u][0], o[u][1], rgb);
#pragma unroll
            for (int c = 0; c < 3; c++) pend_pk[u][1 + c] = (_Float16)fast_sigmoid(rgb[c]);
        }
      }
    }
    flush_pending();
#undef D0
#undef DO
#undef C0
#undef C1
#undef CO
}

// ---- encode + both MLPs in ONE kernel (tiled layout, inference) -------------------------------------------------------------------------
// The encoder is bound by the texture-address path and loses nothing down to 5 waves per SIMD; its MFMA units idle.  Here a wave encodes one
// ROW of the tiled layout (64 samples, lane = pixel of the tile, all 16 levels) into registers, the two lane halves trade half of their feature
// dwords with 8 v_permlane32_swap -- after which the four 16-byte feature groups ARE the first-layer B fragments of two 32-sample MFMA tiles
// (tile A = pixels 0-31: groups 0 and 2, tile B = pixels 32-63: groups 1 and 3; the same fragment-major order k_grid_encode writes to memory)
// -- and runs the MLP chain of k_ngp_mlp on them, one tile after the other.  No feature buffer (128 B per sample of HBM traffic), no second
// kernel; the matrix work of one wave runs in the shadow of the other waves' gathers.
template <int SRC>
__global__ void __launch_bounds__(256, 4) k_encode_mlp(QueryIn in, int64_t base, int64_t n, const __half2* __restrict__ table, GridCfg g, int narrow_levels,
                                                       const h8* __restrict__ ray_sh, const __half* __restrict__ Wd, const __half* __restrict__ Wc,
                                                       __half* __restrict__ packed) {
    static_assert(SRC == SRC_TILED, "tiled layout only");
    enum { F_D0 = 0, F_DO = 4, F_C0 = 8, F_C1 = 12, F_CO = 20, N_FRAG = 24 };
    if constexpr (SRC == SRC_TILED) {
        if (in.n_rows_dev) {   // fixed row capacity: only the rows the march produced (uniform over the launch)
            const int64_t have = (int64_t)in.n_rows_dev[0] * 64 - base;
            n = have < n ? have : n;
            if (n <= 0) return;
        }
    }
    __shared__ h8 wlds[N_FRAG][64];
    const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
    for (int f = threadIdx.x >> 6; f < N_FRAG; f += 4) {
        h8 v;
        if (f < F_DO) v = load_w_frag<false>(Wd, 32, 64, (f - F_D0) >> 1, (f - F_D0) & 1, r, hh);
        else if (f < F_C0) v = load_w_head_frag(Wd + 64 * 32, 64, 16, (f - F_DO) & 1, lane);     // as in k_ngp_mlp: the two kernels paint the same picture bit for bit
        else if (f < F_C1) {
            const int mt = (f - F_C0) >> 1, sk = (f - F_C0) & 1;
            v = load_w_frag<false>(Wc, 32, 64, mt, sk, r, hh);
        } else if (f < F_CO) v = load_w_frag<true>(Wc + 64 * 32, 64, 64, (f - F_C1) >> 2, (f - F_C1) & 3, r, hh);
        else v = load_w_head_frag(Wc + 64 * 32 + 64 * 64, 64, 16, (f - F_CO) & 1, lane);
        wlds[f][lane] = v;
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t trs = make_table_rsrc(table, g.total_entries * 4u);
    const int64_t n_rows = (n + 63) / 64;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
    for (int64_t row = wave0; row < n_rows; row += n_waves) {
        const int64_t j = row * 64 + lane;
        float px = 0.f, py = 0.f, pz = 0.f;
        const bool live = j < n && fetch_pos<SRC>(in, base + j, px, py, pz) > 0;
        const unsigned long long live_mask = __ballot(live);
        if (live_mask == 0ull) continue;  // a row of holes
        uint32_t G0[4] = {0u, 0u, 0u, 0u}, G1[4] = {0u, 0u, 0u, 0u}, G2[4] = {0u, 0u, 0u, 0u}, G3[4] = {0u, 0u, 0u, 0u};
#pragma unroll 1
        for (int grp = 0; grp < 4; grp++) {
            uint32_t v[4] = {0u, 0u, 0u, 0u};
            if (live) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int level = 4 * grp + q;
                    Corner8 c;
                    float f0, f1;
                    if (g.hashed[level]) grid_corners_u<true>(px, py, pz, g.scale[level], g.res[level], g.size[level], g.offset[level], c);
                    else grid_corners_u<false>(px, py, pz, g.scale[level], g.res[level], g.size[level], g.offset[level], c);
                    if (level < narrow_levels) grid_level_features_narrow(trs, c, f0, f1);
                    else if (g.hashed[level]) grid_level_features_hashed(trs, c, f0, f1);
                    else grid_level_features(trs, c, f0, f1);
                    const __half2 h = __floats2half2_rn(f0, f1);
                    v[q] = *reinterpret_cast<const uint32_t*>(&h);
                    asm volatile("" : "+v"(v[q]));
                }
            }
            // wave-uniform group index: the four groups live in named registers
            if (grp == 0) { G0[0] = v[0]; G0[1] = v[1]; G0[2] = v[2]; G0[3] = v[3]; }
            else if (grp == 1) { G1[0] = v[0]; G1[1] = v[1]; G1[2] = v[2]; G1[3] = v[3]; }
            else if (grp == 2) { G2[0] = v[0]; G2[1] = v[1]; G2[2] = v[2]; G2[3] = v[3]; }
            else { G3[0] = v[0]; G3[1] = v[1]; G3[2] = v[2]; G3[3] = v[3]; }
        }
        // lanes 32-63 of G0 / G2 <-> lanes 0-31 of G1 / G3: lane 32 + r receives pixel r's groups 1 and 3 (k-half hh = 1 of tile A), lane r
        // receives pixel 32 + r's groups 0 and 2 (k-half hh = 0 of tile B)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const auto s01 = __builtin_amdgcn_permlane32_swap(G0[q], G1[q], false, false);
            G0[q] = s01[0]; G1[q] = s01[1];
            const auto s23 = __builtin_amdgcn_permlane32_swap(G2[q], G3[q], false, false);
            G2[q] = s23[0]; G3[q] = s23[1];
        }
        const int32_t rt = tile_of(in.row_tile[(base + row * 64) >> 6]);
#pragma unroll 1
        for (int u = 0; u < 2; u++) {
            uint4 b0, b1;
            if (u == 0) { b0 = make_uint4(G0[0], G0[1], G0[2], G0[3]); b1 = make_uint4(G2[0], G2[1], G2[2], G2[3]); }
            else { b0 = make_uint4(G1[0], G1[1], G1[2], G1[3]); b1 = make_uint4(G3[0], G3[1], G3[2], G3[3]); }
            const bool valid = (live_mask >> (32 * u + r)) & 1ull;
            if (__ballot(valid) == 0ull) continue;
            h8 B[2], X[2], H[4], HB[2][2];
            f16v acc[2];
            f4v o[2];
            B[0] = *reinterpret_cast<const h8*>(&b0);
            B[1] = *reinterpret_cast<const h8*>(&b1);
            X[0] = ray_sh[((int64_t)rt * 2 + hh) * 64 + 32 * u + r];
            acc[0] = zero16(); acc[1] = zero16(); o[0] = f4v{0.f, 0.f, 0.f, 0.f}; o[1] = o[0];
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int sk = 0; sk < 2; sk++) acc[mt] = NRC_MFMA(wlds[F_D0 + 2 * mt + sk][lane], B[sk], acc[mt]);
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int gq = 0; gq < 2; gq++) H[2 * mt + gq] = acc_to_frag_relu(acc[mt], gq);
            head_split(H, HB);
#pragma unroll
            for (int sk = 0; sk < 2; sk++)
#pragma unroll
                for (int nb = 0; nb < 2; nb++) o[nb] = NRC_MFMA16(wlds[F_DO + sk][lane], HB[nb][sk], o[nb]);
            X[1] = head_join(o[0], o[1]);
            const _Float16 h0 = X[1][0];
            acc[0] = zero16(); acc[1] = zero16();
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int sk = 0; sk < 2; sk++) acc[mt] = NRC_MFMA(wlds[F_C0 + 2 * mt + sk][lane], X[sk], acc[mt]);
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int gq = 0; gq < 2; gq++) H[2 * mt + gq] = acc_to_frag_relu(acc[mt], gq);
            acc[0] = zero16(); acc[1] = zero16();
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int sk = 0; sk < 4; sk++) acc[mt] = NRC_MFMA(wlds[F_C1 + 4 * mt + sk][lane], H[sk], acc[mt]);
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int gq = 0; gq < 2; gq++) H[2 * mt + gq] = acc_to_frag_relu(acc[mt], gq);
            head_split(H, HB);
            o[0] = f4v{0.f, 0.f, 0.f, 0.f}; o[1] = o[0];
#pragma unroll
            for (int sk = 0; sk < 2; sk++)
#pragma unroll
                for (int nb = 0; nb < 2; nb++) o[nb] = NRC_MFMA16(wlds[F_CO + sk][lane], HB[nb][sk], o[nb]);
            float rgb[3];
            head_join_rgb(o[0], o[1], rgb);
            if (valid && hh == 0) {
                h4 pk;
                pk[0] = h0;
#pragma unroll
                for (int c = 0; c < 3; c++) pk[1 + c] = (_Float16)fast_sigmoid(rgb[c]);
                *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(packed) + 4 * (base + row * 64 + 32 * u + r)) = pk;
            }
        }
    }
}

__global__ void k_f32_to_f16(const float* __restrict__ src, __half* __restrict__ dst, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i + 3 < n) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        h4 o = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(dst) + i) = o;
    } else {
        for (int64_t k = i; k < n; k++) dst[k] = __float2half(src[k]);
    }
}

int make_grid_cfg(int n_levels, int log2_T, int base_res, float pls, GridCfg& g, uint32_t* offsets_out) {
    if (n_levels < 1 || n_levels > NRC_MAX_LEVELS || log2_T < 1 || log2_T > 30 || base_res < 1 || !(pls > 0.f)) return NRC_ERR_INVALID;
    const float log2_pls = log2f(pls);
    uint32_t off = 0;
    for (int l = 0; l < NRC_MAX_LEVELS; l++) { g.offset[l] = 0; g.size[l] = 8; g.res[l] = 2; g.hashed[l] = 0; g.scale[l] = 1.f; }
    for (int l = 0; l < n_levels; l++) {
        const float scale = exp2f(l * log2_pls) * base_res - 1.0f;
        const uint32_t res = (uint32_t)ceilf(scale) + 1;
        const uint32_t max_params = 0xffffffffu / 2;
        uint32_t n = powf((float)res, 3.0f) > (float)max_params ? max_params : res * res * res;
        n = (n + 7u) / 8u * 8u;
        if (n > (1u << log2_T)) n = 1u << log2_T;
        // the dense path is taken while the running stride fits (see oracle/tcnn_oracle.c grid_index)
        uint64_t stride = 1;
        bool hashed = false;
        for (int d = 0; d < 3; d++) { if (stride <= n) stride *= res; }
        hashed = n < stride;
        g.offset[l] = off; g.size[l] = n; g.res[l] = res; g.hashed[l] = hashed ? 1u : 0u; g.scale[l] = scale;
        if (offsets_out) offsets_out[l] = off;
        off += n;
    }
    g.total_entries = off;
    if (offsets_out) offsets_out[n_levels] = off;
    return NRC_OK;
}

int pick_blocks(int64_t M) {
    // every workgroup stages the network's weight fragments in LDS (24 KB) before its first tile: few workgroups with several tiles per wave beat
    // one tile per wave (NRC_MLP_BLOCKS: cap on the number of workgroups, default 2 048)
    static const int64_t cap_env = [] { const char* e = getenv("NRC_MLP_BLOCKS"); return e ? atoll(e) : (int64_t)0; }();
    // measured on a 264 K-sample batch (us, density + colour forward): 2 048 workgroups 25.7 + 47.6, 1 024: 23.8 + 40.5, 512: 22.9 + 36.8, 256: 22.9 + 37.4
    const int64_t cap = cap_env > 0 ? cap_env : (M < (int64_t(1) << 20) ? 512 : 2048);
    const int64_t need = nrc_cdiv(nrc_cdiv(M, 32), 4);
    return (int)(need < cap ? (need > 0 ? need : 1) : cap);
}

}  // namespace

#ifndef NRC_QUERY_CHUNK_LOG2
#define NRC_QUERY_CHUNK_LOG2 23
#endif
#define NRC_QUERY_CHUNK (int64_t(1) << NRC_QUERY_CHUNK_LOG2)  // samples per encode/MLP round (8 Mi: 512 MB of features).  Measured per 800x800 image,
                                            // round 1: 2 Mi 11.0 ms, 4 Mi 10.5, 8 Mi 10.15, 16 Mi 10.17; round 2 (tools/exp_image.py): 2 Mi 9.93, 4 Mi 9.83, 8 Mi 9.56 --
                                            // the gaps between 2 x 38 launches cost more than features that spill past the Infinity Cache

// workspace: [features of one chunk: roundup32(chunk) x 64 B][ray_sh: n_ray_tiles x 2 KB (tiled layout only)]
static int64_t query_feat_bytes(int64_t M) {
    const int64_t c = M < NRC_QUERY_CHUNK ? M : NRC_QUERY_CHUNK;
    return ((c > 0 ? c : 1) + 31) / 32 * 32 * 64 + 256;
}

static thread_local int g_enc_shape_override = 0;   // nrc_ngp_set_encoder_shape: (log2 pixels along x) << 4 | (log2 pixels along y) of a wave's brick, 0 = the default;
                                                    // per host thread: the renderer sets it and enqueues the frame from the same thread
template <int SRC>
static void launch_encode(const QueryIn& in, int64_t base, int64_t n, const void* table, const GridCfg& g, uint4* feat, hipStream_t s) {
    // levels whose cells are larger than a wave's footprint: narrow gathers (see grid_level_features_narrow)
    static const int narrow_env = [] { const char* e = getenv("NRC_ENC_NARROW"); return e ? atoi(e) : -1; }();
    int narrow = 0;
    if (narrow_env >= 0) narrow = narrow_env;
    else while (narrow < NRC_MAX_LEVELS && !g.hashed[narrow] && g.size[narrow] > 8) narrow++;
    static const int hashed_mode = [] { const char* e = getenv("NRC_ENC_HASHED"); return e ? atoi(e) : 1; }();
    // measurement switch (profiles/: L1 lookups per level range): NRC_ENC_LEVELS=lo-hi encodes only levels lo <= l < hi, the others read nothing and
    // come out as zeros -- WRONG pictures by design, never set outside a counter run
    static const int lvl_range = [] {
        const char* e = getenv("NRC_ENC_LEVELS");
        int lo = 0, hi = NRC_MAX_LEVELS;
        if (e && sscanf(e, "%d-%d", &lo, &hi) == 2 && lo >= 0 && hi <= NRC_MAX_LEVELS && lo <= hi) return lo | (hi << 8);
        return NRC_MAX_LEVELS << 8;
    }();
    // below ~2 M samples the chip is not full with one lane per sample: split the four level groups over workgroup rows
    static const int64_t split_below = [] { const char* e = getenv("NRC_ENC_SPLIT_BELOW"); return e ? atoll(e) : (int64_t)2 << 20; }();
    if constexpr (SRC == SRC_ARRAYS) {
        static const int pairs_mode = [] { const char* e = getenv("NRC_ENC_PAIRS"); return e ? atoi(e) : 1; }();
        if (pairs_mode && n < split_below && base == 0 && lvl_range == (NRC_MAX_LEVELS << 8)) {
            hipLaunchKernelGGL(k_grid_encode_pairs, dim3((unsigned)(8 * nrc_cdiv(n, 256))), dim3(256), 0, s, in, n, (const __half2*)table, g, (uint2*)feat, narrow,
                               hashed_mode, in.m_live);
            return;
        }
    }
    const unsigned rows = (SRC == SRC_ARRAYS && n < split_below) ? 4u : 1u;
    // coarse levels of the tiled layout: wave-uniform cells through the scalar cache (NRC_ENC_UNIFORM = number of levels that try).  Measured
    // (round 3, profiles/): it removes the vector lookups of those levels and changes NOTHING in time (0.643 ms off, 0.650-0.652 ms with 5, 7
    // or 9 levels): the kernel is bound by the four finest levels' L1 misses (52 % of its time, 8.8 of 9.3 L2 requests per sample), not by
    // the lookups of the coarse ones.  Default off; bit-identical features either way.
    static const int uniform_levels = [] { const char* e = getenv("NRC_ENC_UNIFORM"); const int v = e ? atoi(e) : 0; return v < 0 ? 0 : (v > NRC_MAX_LEVELS ? NRC_MAX_LEVELS : v); }();
    // NRC_ENC_SHAPE=uv: a wave encodes 2^u x 2^v pixels x 2^(6-u-v) steps of the tiled layout (default 31 = 8 x 2 x 4; 33 = a row, the round-2 form)
    static const int lane_shape = [] {
        const char* e = getenv("NRC_ENC_SHAPE");
        const int v = e ? atoi(e) : 31, lu = v / 10, lv = v % 10;
        return (lu < 0 || lu > NRC_TILE_W_LOG2 || lv < 0 || lv > 6 - NRC_TILE_W_LOG2 || lu + lv < 2 || lu + lv == 6) ? 0 : (lu << 4) | lv;   // 2 <= u + v: at most 16 steps
    }();
    const int lanes = (SRC == SRC_TILED && (base & 1023) == 0) ? (g_enc_shape_override ? g_enc_shape_override : lane_shape) : 0;
    static const int xcd_ranges = [] { const char* e = getenv("NRC_ENC_XCD"); return e ? (atoi(e) != 0) : 1; }();   // measured: 612 -> 591 us per launch (mean of 24 poses)
    // (the remapping permutes whole blocks of 1024 slots: the launch covers whole blocks)
    // NRC_ENC_FINE_SPLIT=1 (experiment, see k_grid_encode_fine): levels 12-15 in a second launch, one level per XCD
    static const int fine_split = [] { const char* e = getenv("NRC_ENC_FINE_SPLIT"); return e ? atoi(e) : 0; }();
    bool split = false;
    if constexpr (SRC == SRC_TILED)
        split = fine_split && lanes && rows == 1u && hashed_mode == 1 && lvl_range == (NRC_MAX_LEVELS << 8) && g.hashed[NRC_MAX_LEVELS - 4] && narrow <= NRC_MAX_LEVELS - 4;
    hipLaunchKernelGGL(k_grid_encode<SRC>, dim3((unsigned)(lanes ? 4 * nrc_cdiv(n, 1024) : nrc_cdiv(n, 256)), rows), dim3(256), 0, s, in, base, n, (const __half2*)table, g, feat,
                       narrow, hashed_mode | (lvl_range << 4) | (uniform_levels << 20) | (xcd_ranges << 28) | ((split ? 1 : 0) << 29), lanes);
    if constexpr (SRC == SRC_TILED) {
        if (split) {
            const int64_t per_half = 2 * nrc_cdiv(n, 1024);   // 256-slot blocks per half of the launch's slots
            hipLaunchKernelGGL(k_grid_encode_fine<SRC>, dim3((unsigned)(8 * per_half)), dim3(256), 0, s, in, base, n, (const __half2*)table, g, feat, NRC_MAX_LEVELS - 4, lanes);
        }
    }
}

template <int SRC>
static void launch_encode_mlp(const QueryIn& in, int64_t base, int64_t n, const void* table, const GridCfg& g, const void* ray_sh, const void* wd,
                              const void* wc, void* packed, hipStream_t s) {
    int narrow = 0;
    while (narrow < NRC_MAX_LEVELS && !g.hashed[narrow] && g.size[narrow] > 8) narrow++;
    // persistent waves: a workgroup stages the 24 KB of weight fragments once and walks rows with a stride
    const int64_t rows = nrc_cdiv(n, 64);
    const int64_t want = nrc_cdiv(rows, 4);
    const unsigned blocks = (unsigned)(want < 256 * 6 ? (want > 0 ? want : 1) : 256 * 6);
    hipLaunchKernelGGL(k_encode_mlp<SRC>, dim3(blocks), dim3(256), 0, s, in, base, n, (const __half2*)table, g, narrow, (const h8*)ray_sh,
                       (const __half*)wd, (const __half*)wc, (__half*)packed);
}
// NRC_QUERY_FUSED=1 selects k_encode_mlp.  Measured on the 800x800 bench: 65.1-65.4 Mrays/s against 66.9-67.0 for the two kernels through the
// feature buffer (77 VGPRs, 6 waves per SIMD; prefetching the SH fragments across the encoding: 100 VGPRs, 63.3; staggered workgroup
// starts: 64.1-65.0) -- the fused kernel costs the SUM of the two kernels, the matrix phase does not hide behind the other waves' gathers.
// Kept as an experiment switch (and covered by a parity test), not the default.
static bool query_fused_kernel() {
    static const bool v = [] { const char* e = getenv("NRC_QUERY_FUSED"); return e && e[0] == '1'; }();
    return v;
}
static int mlp_tiles_per_wave() {
    static const int v = [] { const char* e = getenv("NRC_MLP_NT"); return (e && e[0] == '1') ? 1 : 2; }();
    return v;
}
template <int SRC>
static void launch_mlp(const QueryIn& in, int64_t base, int64_t n, const void* feat, const void* ray_sh, const void* wd, const void* wc,
                       float* sigmas, float* rgbs, void* packed, hipStream_t s) {
    if (mlp_tiles_per_wave() == 2)
        hipLaunchKernelGGL((k_ngp_mlp<SRC, 2>), dim3(pick_blocks(nrc_cdiv(n, 2))), dim3(256), 0, s, in, base, n, (const uint4*)feat, (const h8*)ray_sh,
                           (const __half*)wd, (const __half*)wc, sigmas, rgbs, (__half*)packed);
    else
        hipLaunchKernelGGL((k_ngp_mlp<SRC, 1>), dim3(pick_blocks(n)), dim3(256), 0, s, in, base, n, (const uint4*)feat, (const h8*)ray_sh,
                           (const __half*)wd, (const __half*)wc, sigmas, rgbs, (__half*)packed);
}

// (Measured and dropped: running the MLP kernel of chunk c on a second stream next to the encode kernel of chunk c+1, with two
// feature buffers -- 11.27 ms per image against 10.99 ms in one stream; the encode grid fills every CU, the MLP blocks only queue.)
template <int SRC>
static int run_query(const QueryIn& in, int64_t M, int64_t n_ray_tiles, const void* wd, const void* wc, const void* table, const GridCfg& g,
                     float* sigmas, float* rgbs, void* packed, void* workspace, hipStream_t s) {
    uint4* feat = (uint4*)workspace;
    h8* ray_sh = (h8*)((char*)workspace + query_feat_bytes(M));
    if constexpr (SRC == SRC_TILED)
        hipLaunchKernelGGL(k_ray_sh, dim3((unsigned)nrc_cdiv(n_ray_tiles * 64, 256)), dim3(256), 0, s, in.ray_od, n_ray_tiles, ray_sh, in.tile_off,
                           in.tile_off ? const_cast<int32_t*>(in.row_tile) : (int32_t*)nullptr, M / 64);
    NRC_STAGE(s, nullptr);      // (armed stage timer: the two kernels of every chunk IN the frame's own sequence -- bench.py's in-frame ruler)
    for (int64_t base = 0; base < M; base += NRC_QUERY_CHUNK) {
        const int64_t n = (M - base) < NRC_QUERY_CHUNK ? (M - base) : NRC_QUERY_CHUNK;
        if constexpr (SRC == SRC_TILED) {
            if (query_fused_kernel()) { launch_encode_mlp<SRC>(in, base, n, table, g, ray_sh, wd, wc, packed, s); NRC_STAGE(s, "k_encode_mlp"); continue; }
        }
        launch_encode<SRC>(in, base, n, table, g, feat, s);
        NRC_STAGE(s, "k_grid_encode");
        launch_mlp<SRC>(in, base, n, feat, ray_sh, wd, wc, sigmas, rgbs, packed, s);
        NRC_STAGE(s, "k_ngp_mlp");
    }
    return NRC_OK;
}


extern "C" {

int nrc_grid_layout(int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale, uint32_t* offsets_host) {
    GridCfg g;
    return make_grid_cfg(n_levels, log2_hashmap_size, base_resolution, per_level_scale, g, offsets_host);
}

int nrc_f32_to_f16(const float* src, void* dst, int64_t n, nrc_stream_t stream) {
    NRC_ENTER();
    if (n < 0 || (n > 0 && (!src || !dst))) return NRC_ERR_INVALID;
    if (n == 0) return NRC_OK;
    hipLaunchKernelGGL(k_f32_to_f16, dim3(nrc_cdiv(nrc_cdiv(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, src, (__half*)dst, n);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_nwie_forward(int32_t encoding, const void* input, int32_t input_ld, int64_t M, const void* weights_f16, const void* table_f16,
                     int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale,
                     int32_t n_hidden, int32_t out_act, int32_t n_out_rows, void* out_f16, int32_t out_ld, int32_t n_store,
                     void* save_in, void* save_acts, void* workspace, nrc_stream_t stream) {
    NRC_ENTER();
    if (M < 0 || !weights_f16 || n_out_rows < 1 || n_out_rows > 16 || out_ld < n_store || n_store < 4 || n_store > 16 || (n_store & 3)) return NRC_ERR_INVALID;
    // encoding: low byte = kind; kind 0: bits 8..15 = features per table entry (0 = 2)
    const int n_features = (encoding >> 8) & 0xff ? (encoding >> 8) & 0xff : 2;
    encoding &= 0xff;
    if (encoding != ENC_GRID && encoding != ENC_SH_ID) return NRC_ERR_UNSUPPORTED;
    if (n_hidden < 1 || n_hidden > 2 || (out_act != ACT_NONE && out_act != ACT_SIGMOID)) return NRC_ERR_UNSUPPORTED;
    if (M == 0) return NRC_OK;
    if (!input || !out_f16) return NRC_ERR_INVALID;
    GridCfg g;
    bool general = false;
    if (encoding == ENC_GRID) {
        if (!table_f16) return NRC_ERR_INVALID;
        if ((n_features != 2 && n_features != 4) || n_levels < 1 || n_levels * n_features > 32) return NRC_ERR_UNSUPPORTED;  // the encoding feeds 32 first-layer inputs
        general = !(n_levels == 16 && n_features == 2);
        const int rc = make_grid_cfg(n_levels, log2_hashmap_size, base_resolution, per_level_scale, g, nullptr);
        if (rc != NRC_OK) return rc;
    } else {
        make_grid_cfg(1, 4, 2, 2.f, g, nullptr);
        if (input_ld < 19) return NRC_ERR_INVALID;
    }
    const bool save = save_in && save_acts;
    if ((save_in == nullptr) != (save_acts == nullptr)) return NRC_ERR_INVALID;
    const dim3 grid(pick_blocks(M)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define NRC_FWD(E, H, A, S)                                                                                                   \
    hipLaunchKernelGGL((k_nwie_fwd<E, H, A, S>), grid, block, 0, s, input, (int)input_ld, M, (const __half*)weights_f16,       \
                       (const __half2*)table_f16, g, (int)n_out_rows, (__half*)out_f16, (int)out_ld, (int)n_store,             \
                       (__half*)save_in, (__half*)save_acts, (float*)nullptr, (float*)nullptr, (const int32_t*)nullptr, (int)n_levels)
#define NRC_FWD_S(E, H, A) do { if (save) NRC_FWD(E, H, A, true); else NRC_FWD(E, H, A, false); } while (0)
#define NRC_FWD_A(E, H) do { if (out_act == ACT_SIGMOID) NRC_FWD_S(E, H, ACT_SIGMOID); else NRC_FWD_S(E, H, ACT_NONE); } while (0)
    if (encoding == ENC_GRID && general) {
        // any other (levels, features) the yaml asks for: one kernel, plain gathers (the drop-in's general path)
        if (n_features == 4) { if (n_hidden == 1) NRC_FWD_A(ENC_GRID_F4, 1); else NRC_FWD_A(ENC_GRID_F4, 2); }
        else { if (n_hidden == 1) NRC_FWD_A(ENC_GRID_F2, 1); else NRC_FWD_A(ENC_GRID_F2, 2); }
    } else if (encoding == ENC_GRID && workspace) {
        // split form (what the image pipeline uses): all 128 gathers of a sample by a low-register, 8-waves-per-SIMD kernel, then the
        // MFMA chain from the fragment-major features; the single-kernel form below is 3x slower on a 264 K-sample batch (130 us)
        QueryIn qin = {};
        qin.xyz01 = (const float*)input;
        launch_encode<SRC_ARRAYS>(qin, 0, M, table_f16, g, (uint4*)workspace, s);
        input = workspace;
        if (n_hidden == 1) NRC_FWD_A(ENC_FEAT, 1); else NRC_FWD_A(ENC_FEAT, 2);
    } else if (encoding == ENC_GRID) { if (n_hidden == 1) NRC_FWD_A(ENC_GRID, 1); else NRC_FWD_A(ENC_GRID, 2); }
    else { if (n_hidden == 1) NRC_FWD_A(ENC_SH_ID, 1); else NRC_FWD_A(ENC_SH_ID, 2); }
#undef NRC_FWD_A
#undef NRC_FWD_S
#undef NRC_FWD
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}


int64_t nrc_nwie_forward_ws_bytes(int64_t M) {
    if (M < 0) return NRC_ERR_INVALID;
    return ((M > 0 ? M : 1) + 31) / 32 * 32 * 64 + 256;
}

int64_t nrc_ngp_query_ws_bytes(int64_t M) {
    if (M < 0) return NRC_ERR_INVALID;
    return query_feat_bytes(M);
}

int64_t nrc_ngp_query_samples_ws_bytes(int64_t n_rows, int64_t n_ray_tiles) {
    if (n_rows < 0 || n_ray_tiles < 0) return NRC_ERR_INVALID;
    return query_feat_bytes(n_rows * 64) + n_ray_tiles * 2048 + 256;
}

int nrc_ngp_query_fused(const float* xyz01, const float* dirs, int64_t M, const void* density_weights_f16,
                        const void* color_weights_f16, const void* table_f16, int32_t n_levels, int32_t log2_hashmap_size,
                        int32_t base_resolution, float per_level_scale, float* sigmas, float* rgbs, void* workspace, nrc_stream_t stream) {
    NRC_ENTER();
    if (M < 0 || !density_weights_f16 || !color_weights_f16 || !table_f16) return NRC_ERR_INVALID;
    if (n_levels != 16) return NRC_ERR_UNSUPPORTED;
    if (M == 0) return NRC_OK;
    if (!xyz01 || !dirs || !sigmas || !rgbs || !workspace) return NRC_ERR_INVALID;
    GridCfg g;
    const int rc = make_grid_cfg(n_levels, log2_hashmap_size, base_resolution, per_level_scale, g, nullptr);
    if (rc != NRC_OK) return rc;
    QueryIn in = {};
    in.xyz01 = xyz01; in.dirs = dirs;
    run_query<SRC_ARRAYS>(in, M, 0, density_weights_f16, color_weights_f16, table_f16, g, sigmas, rgbs, nullptr, workspace, (hipStream_t)stream);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_ngp_set_encoder_shape(int32_t log2_x, int32_t log2_y) {
    if (log2_x < 0 && log2_y < 0) { g_enc_shape_override = 0; return NRC_OK; }
    if (log2_x < 0 || log2_x > NRC_TILE_W_LOG2 || log2_y < 0 || log2_y > 6 - NRC_TILE_W_LOG2 || log2_x + log2_y < 2 || log2_x + log2_y == 6) return NRC_ERR_INVALID;
    g_enc_shape_override = (log2_x << 4) | log2_y;
    return NRC_OK;
}
int nrc_ngp_encode_samples(const float* ts, const int32_t* row_tile, const float* ray_od, int64_t first_row, int64_t n_rows, const float* xyz_min3,
                           const float* xyz_size3, const void* table_f16, int32_t n_levels, int32_t log2_hashmap_size,
                           int32_t base_resolution, float per_level_scale, void* features_f16, const int32_t* arena_tile_off, int32_t arena_rows,
                           nrc_stream_t stream) {
    NRC_ENTER();
    const int64_t n = n_rows * 64;
    if (n_rows < 0 || first_row < 0 || n > NRC_QUERY_CHUNK || !table_f16 || !xyz_min3 || !xyz_size3) return NRC_ERR_INVALID;
    if ((arena_tile_off != nullptr) != (arena_rows > 0) || arena_rows < 0) return NRC_ERR_INVALID;
    if (n_levels != 16) return NRC_ERR_UNSUPPORTED;
    if (n == 0) return NRC_OK;
    if (!ts || !row_tile || !ray_od || !features_f16) return NRC_ERR_INVALID;
    GridCfg g;
    const int rc = make_grid_cfg(n_levels, log2_hashmap_size, base_resolution, per_level_scale, g, nullptr);
    if (rc != NRC_OK) return rc;
    QueryIn in = {};
    in.ts = ts; in.row_tile = row_tile; in.ray_od = ray_od; in.tile_off = arena_tile_off; in.arena_rows = arena_rows;
    for (int k = 0; k < 3; k++) { in.mn[k] = xyz_min3[k]; in.sz[k] = xyz_size3[k]; }
    launch_encode<SRC_TILED>(in, first_row * 64, n, table_f16, g, (uint4*)features_f16, (hipStream_t)stream);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_ngp_mlp_samples(const float* ts, const int32_t* row_tile, const float* ray_od, int64_t first_row, int64_t n_rows, int64_t n_ray_tiles,
                        const void* features_f16, const void* density_weights_f16, const void* color_weights_f16, void* packed_f16,
                        void* ray_sh_workspace, const int32_t* arena_tile_off, int32_t arena_rows, nrc_stream_t stream) {
    NRC_ENTER();
    const int64_t n = n_rows * 64;
    if (n_rows < 0 || first_row < 0 || n > NRC_QUERY_CHUNK || n_ray_tiles < 0 || !density_weights_f16 || !color_weights_f16) return NRC_ERR_INVALID;
    if ((arena_tile_off != nullptr) != (arena_rows > 0) || arena_rows < 0) return NRC_ERR_INVALID;
    if (n == 0) return NRC_OK;
    if (!ts || !row_tile || !ray_od || !features_f16 || !packed_f16 || !ray_sh_workspace) return NRC_ERR_INVALID;
    QueryIn in = {};
    in.ts = ts; in.row_tile = row_tile; in.ray_od = ray_od; in.tile_off = arena_tile_off; in.arena_rows = arena_rows;
    hipStream_t s = (hipStream_t)stream;
    if (n_ray_tiles > 0)  // 0: the workspace already holds the rays' SH coefficients (they are per image, not per chunk)
        hipLaunchKernelGGL(k_ray_sh, dim3((unsigned)nrc_cdiv(n_ray_tiles * 64, 256)), dim3(256), 0, s, ray_od, n_ray_tiles, (h8*)ray_sh_workspace);
    launch_mlp<SRC_TILED>(in, first_row * 64, n, features_f16, ray_sh_workspace, density_weights_f16, color_weights_f16, nullptr, nullptr, packed_f16, s);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_ngp_query_samples(const float* ts, int32_t* row_tile, const float* ray_od, int64_t n_rows, int64_t n_ray_tiles, const float* xyz_min3,
                          const float* xyz_size3, const void* density_weights_f16, const void* color_weights_f16,
                          const void* table_f16, int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution,
                          float per_level_scale, void* packed_f16, void* workspace, const int32_t* n_rows_dev, const int32_t* arena_tile_off,
                          int32_t arena_rows, nrc_stream_t stream) {
    NRC_ENTER();
    const int64_t M = n_rows * 64;
    if ((arena_tile_off != nullptr) != (arena_rows > 0) || arena_rows < 0) return NRC_ERR_INVALID;
    if (n_rows < 0 || n_ray_tiles < 0 || !density_weights_f16 || !color_weights_f16 || !table_f16 || !xyz_min3 || !xyz_size3) return NRC_ERR_INVALID;
    if (n_levels != 16) return NRC_ERR_UNSUPPORTED;
    if (M == 0) return NRC_OK;
    if (!ts || !row_tile || !ray_od || !packed_f16 || !workspace || n_ray_tiles == 0) return NRC_ERR_INVALID;
    GridCfg g;
    const int rc = make_grid_cfg(n_levels, log2_hashmap_size, base_resolution, per_level_scale, g, nullptr);
    if (rc != NRC_OK) return rc;
    QueryIn in = {};
    in.ts = ts; in.row_tile = row_tile; in.ray_od = ray_od; in.n_rows_dev = n_rows_dev;
    in.tile_off = arena_tile_off; in.arena_rows = arena_rows;
    for (int k = 0; k < 3; k++) { in.mn[k] = xyz_min3[k]; in.sz[k] = xyz_size3[k]; }
    run_query<SRC_TILED>(in, M, n_ray_tiles, density_weights_f16, color_weights_f16, table_f16, g, nullptr, nullptr, packed_f16, workspace, (hipStream_t)stream);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

/* workspace of the layer-major pipeline: [features of one chunk][ray_sh][state 6 x n f32][ray_alive n u8][next_k nt i32][tile_alive nt u8] */
static int64_t layers_state_bytes(int64_t n_ray_tiles) {
    const int64_t n = n_ray_tiles * 64;
    return (n * 6 * 4 + n + n_ray_tiles * 4 + n_ray_tiles + 1023) / 256 * 256;
}
int64_t nrc_ngp_render_layers_ws_bytes(int64_t n_rows, int64_t n_ray_tiles) {
    if (n_rows < 0 || n_ray_tiles < 0) return NRC_ERR_INVALID;
    return query_feat_bytes(n_rows * 64) + n_ray_tiles * 2048 + 256 + layers_state_bytes(n_ray_tiles);
}

int nrc_ngp_render_layers(const float* ts, int32_t* row_tile, const float* ray_od, int64_t n_rows, int64_t n_ray_tiles, const float* xyz_min3,
                          const float* xyz_size3, const void* density_weights_f16, const void* color_weights_f16, const void* table_f16,
                          int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale, const int32_t* ray_cnt,
                          const int32_t* tile_rows, const int32_t* tile_off, const int32_t* row_of, int32_t width, int32_t height, int64_t tile_begin,
                          int32_t cascades, float exp_step_factor, int32_t grid_size, int32_t max_samples, float T_threshold, const float* bg3_host,
                          void* packed_f16, float* rgb, float* alpha, float* depth, int32_t* skipped_rows, void* workspace, const int32_t* arena_row_k,
                          int32_t arena_rows, nrc_stream_t stream) {
    NRC_ENTER();
    const int64_t M = n_rows * 64;
    if ((arena_row_k != nullptr) != (arena_rows > 0) || arena_rows < 0) return NRC_ERR_INVALID;
    if (n_rows < 0 || n_ray_tiles < 1 || !density_weights_f16 || !color_weights_f16 || !table_f16 || !xyz_min3 || !xyz_size3 || !bg3_host ||
        width < 1 || height < 1 || tile_begin < 0 || cascades < 1 || grid_size < 1 || max_samples < 1)
        return NRC_ERR_INVALID;
    if (n_levels != 16) return NRC_ERR_UNSUPPORTED;
    if (!ray_od || !ray_cnt || !tile_rows || !tile_off || !rgb || !alpha || !depth || !workspace) return NRC_ERR_INVALID;
    if (M > 0 && (!ts || !row_tile || !row_of || !packed_f16)) return NRC_ERR_INVALID;
    GridCfg g;
    const int rc = make_grid_cfg(n_levels, log2_hashmap_size, base_resolution, per_level_scale, g, nullptr);
    if (rc != NRC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    char* wsp = (char*)workspace;
    uint4* feat = (uint4*)wsp;
    h8* ray_sh = (h8*)(wsp + query_feat_bytes(M));
    char* st = wsp + query_feat_bytes(M) + n_ray_tiles * 2048 + 256;
    const int64_t n = n_ray_tiles * 64;
    float* state = (float*)st;
    uint8_t* ray_alive = (uint8_t*)(st + n * 6 * 4);
    int32_t* next_k = (int32_t*)(st + (n * 6 * 4 + n + 3) / 4 * 4);
    uint8_t* tile_alive = (uint8_t*)(next_k + n_ray_tiles);
    QueryIn in = {};
    in.ts = ts; in.row_tile = row_tile; in.ray_od = ray_od;
    in.row_k = arena_row_k; in.arena_rows = arena_rows;
    for (int k = 0; k < 3; k++) { in.mn[k] = xyz_min3[k]; in.sz[k] = xyz_size3[k]; }
    nrc_launch_layers_init(n_ray_tiles, ray_cnt, state, ray_alive, next_k, tile_alive, skipped_rows, s);
    hipLaunchKernelGGL(k_ray_sh, dim3((unsigned)nrc_cdiv(n_ray_tiles * 64, 256)), dim3(256), 0, s, ray_od, n_ray_tiles, ray_sh);
    // front to back: encode + MLP on a slab of rows, composite it, finished tiles drop out of the following slabs.  No host
    // round trip: slabs behind the last live tile still launch, but their waves return on the alive flag (tens of microseconds).
    int64_t base = 0;
    do {
        const int64_t cn = (M - base) < NRC_QUERY_CHUNK ? (M - base) : NRC_QUERY_CHUNK;
        if (cn > 0) {
            if (query_fused_kernel()) {
                launch_encode_mlp<SRC_TILED>(in, base, cn, table_f16, g, ray_sh, density_weights_f16, color_weights_f16, packed_f16, s);
            } else {
                launch_encode<SRC_TILED>(in, base, cn, table_f16, g, feat, s);
                launch_mlp<SRC_TILED>(in, base, cn, feat, ray_sh, density_weights_f16, color_weights_f16, nullptr, nullptr, packed_f16, s);
            }
        }
        nrc_launch_composite_layers(packed_f16, ts, ray_cnt, tile_rows, tile_off, row_of, row_tile, (base + cn) / 64, width, height, tile_begin, n_ray_tiles, cascades,
                                    exp_step_factor, grid_size, max_samples, T_threshold, bg3_host, state, ray_alive, next_k, tile_alive, rgb, alpha, depth,
                                    skipped_rows, (int)arena_rows, s);
        base += NRC_QUERY_CHUNK;
    } while (base < M);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

}  // extern "C"

// =====================================================================================================================
// Backward
// =====================================================================================================================
namespace {

#define LDP 40  // the wave's two LDS images are 64 * LDP halves each (5 120 B); the transposing image below uses 4 096 of them

// dW = dZ^T . A sums over the SAMPLES, which sit in the lanes of every fragment this kernel holds: both operands go through an LDS image once.
// The image is [sample 0..31][neuron 0..63] fp16 in the dual-use layout of cdna_hip_programming.md T10 (a) -- 8-row x 32-column subtiles of 512 B,
// the 16-byte chunks of a row XOR-swizzled -- so that a lane WRITES its fragment as one 16-byte or two 8-byte pieces of its own row, and the
// fragments whose k index is the sample are READ back with gfx950's transposing ds_read_b64_tr_b16 (two reads of four samples each).  Round 2 wrote
// the transposed image element by element (eight 2-byte LDS writes per fragment, 150 per tile): 3.4 K of the tile's 6.1 K cycles.
__device__ __forceinline__ int img_off(int row, int ch) {   // byte offset of 16-byte chunk ch (8 neurons) of sample row `row`
    return 1024 * (row >> 3) + 512 * (ch >> 2) + 64 * (row & 7) + 16 * ((ch & 3) ^ ((row >> 2) & 3));
}
// B-style fragment (this lane = sample c, lane half hh) -> its row of the image
template <bool ACC_ORDER>
__device__ __forceinline__ void stage_frag_T(_Float16* T, const h8& f, int s, int c, int hh) {
    char* base = reinterpret_cast<char*>(T);
    if constexpr (ACC_ORDER) {   // elements 0..3 = neurons 16 s + 4 hh + (0..3), elements 4..7 = neurons 16 s + 8 + 4 hh + (0..3)
        h4 lo, hi;
#pragma unroll
        for (int j = 0; j < 4; j++) { lo[j] = f[j]; hi[j] = f[4 + j]; }
        *reinterpret_cast<h4*>(base + img_off(c, 2 * s) + 8 * hh) = lo;
        *reinterpret_cast<h4*>(base + img_off(c, 2 * s + 1) + 8 * hh) = hi;
    } else {                     // elements 0..7 = neurons 16 s + 8 hh + (0..7)
        *reinterpret_cast<h8*>(base + img_off(c, 2 * s + hh)) = f;
    }
}
// fragment whose k index is the SAMPLE: neuron 32 mt + (lane & 31), samples 16 s + 8 (lane >> 5) .. + 7.  EXEC must be all ones (the read gathers
// across lanes): call it from wave-uniform code only.
typedef __fp16 tr_f4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ h8 read_T_frag(const _Float16* T, int mt, int s, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const int row0 = 16 * s + 8 * (g >> 1), ch = 4 * mt + 2 * (g & 1) + (p >> 1);
    const char* base = reinterpret_cast<const char*>(T);
    const tr_f4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) tr_f4*)(base + img_off(row0 + q, ch) + 8 * (p & 1)));
    const tr_f4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) tr_f4*)(base + img_off(row0 + 4 + q, ch) + 8 * (p & 1)));
    h8 f;
#pragma unroll
    for (int j = 0; j < 4; j++) { f[j] = (_Float16)v0[j]; f[4 + j] = (_Float16)v1[j]; }
    return f;
}
__device__ __forceinline__ h8 zero_h8() {
    h8 f;
#pragma unroll
    for (int j = 0; j < 8; j++) f[j] = (_Float16)0.f;
    return f;
}
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// dZ = (H > 0) ? dH : 0 as next B fragments; H fragments (ACC order) coincide element-for-element with the D layout
// Packed: H is a post-ReLU activation (fp16 max(x, +0): never negative), so H > 0 <=> its 16 bits are not all zero; min_u16(bits, 1) is 0 / 1 per
// half, its negation 0x0000 / 0xffff, and the mask is ANDed onto the packed conversion of the accumulators -- 4 v_cvt_pk_f16_f32 + 3 x 4 packed integer
// instructions per fragment where the element-wise select compiled to 8 x (convert, compare, select) + 4 packs (28; it was 224 of the 402 vector
// instructions the colour network's backward issued per 32-sample tile).  A NaN activation (forward overflow) lets its gradient through instead of
// zeroing it; the scaler's check sees either.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ h8 masked_grad_frag(const f16v& acc, int g, const h8& H) {
    const u32x4 f = __builtin_bit_cast(u32x4, acc_to_frag(acc, g)), hb = __builtin_bit_cast(u32x4, H);
    u32x4 out;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t m;   // (inline assembly: written with vector types the optimiser turns the mask back into eight compares and selects)
        asm("v_pk_min_u16 %0, %1, %2\n\tv_pk_sub_u16 %0, 0, %0" : "=v"(m) : "v"(hb[q]), "v"(0x00010001u));
        out[q] = f[q] & m;
    }
    return __builtin_bit_cast(h8, out);
}
__device__ __forceinline__ h8 load_acc_order_frag(const _Float16* row64, int s, int hh) {
    const h4 a = *reinterpret_cast<const h4*>(row64 + 16 * s + 4 * hh);
    const h4 b = *reinterpret_cast<const h4*>(row64 + 16 * s + 8 + 4 * hh);
    h8 f;
#pragma unroll
    for (int j = 0; j < 4; j++) { f[j] = a[j]; f[4 + j] = b[j]; }
    return f;
}
// atomically accumulate a 32x32 D tile (rows = out neuron, cols = in neuron) into a row-major f32 gradient matrix
__device__ __forceinline__ void atomic_add_tile(float* __restrict__ G, int ld, int n_rows, int mt, int nt, const f16v& acc, float mul, int c, int hh) {
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
        const int row = 32 * mt + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
        if (row < n_rows) atomicAdd(G + (size_t)row * ld + 32 * nt + c, acc[reg] * mul);
    }
}

// The fused training query's colour network (nrc_ngp_train_query_backward): the two element-wise kernels around its backward live in the kernel.
//   d_rgbs   dL/drgb (M,3) f32, read INSTEAD of d_out: the fp16 output-gradient row is (half)d_rgbs[0..2], 0
//   d_sigmas, h   dL/dsigma (M) f32 and the density network's output rows (M,16) fp16: dh0 = d_sigmas * exp(clamp(h[.,0], +-15)) (TruncExp backward)
//   d_h16    WRITTEN instead of d_in: the density network's output-gradient rows (M,16) fp16 = (half)(d_in[., 16..31] (+ dh0 on column 0))
struct TrainQ { const float* d_rgbs; const float* d_sigmas; const __half* h; __half* d_h16; };
#if defined(NRC_BWD_PROBE)   // developer build (tools/build_variant.sh): cycle stamps of wave 0 of workgroup 0 at the phases of the first tiles
__device__ unsigned long long g_bwd_probe[2][128];
#define NRC_PROBE(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); g_bwd_probe[N_HIDDEN - 1][(k)] = __builtin_readcyclecounter(); } } while (0)
#define NRC_PROBE_NW(k) do { if (blockIdx.x == 300 && threadIdx.x == 0) g_bwd_probe[0][(k)] = __builtin_readcyclecounter(); } while (0)
#else
#define NRC_PROBE(k) do { } while (0)
#define NRC_PROBE_NW(k) do { } while (0)
#endif
template <int N_HIDDEN, int OUT_ACT>
__global__ void __launch_bounds__(256) k_nwie_bwd(int64_t M, const __half* __restrict__ W, int n_out_rows, const __half* __restrict__ d_out,
                                                  const __half* __restrict__ out, int out_ld, const __half* __restrict__ save_in,
                                                  const __half* __restrict__ save_acts, float loss_scale, float* __restrict__ dW,
                                                  float* __restrict__ d_in, int d_in_pair_major, TrainQ tq = TrainQ{nullptr, nullptr, nullptr, nullptr},
                                                  const int32_t* __restrict__ m_live = nullptr, float* __restrict__ bad_flag = nullptr) {
    // bad_flag (optional, DEVICE f32): set to 1 when a value this kernel hands on -- an input gradient or a weight-gradient sum -- is inf / NaN.
    // Everything downstream (the hash-grid gradient) is a finite-weighted sum of those, so the flags of the two backward launches ARE the
    // GradScaler's found_inf, known before the grid backward starts (nrc_ngp_train_backward_step applies the step inside that backward)
    bool bad = false;
    __shared__ __attribute__((aligned(16))) _Float16 lds[4][2][64 * LDP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    _Float16* T_act = lds[wv][0];
    _Float16* T_dz = lds[wv][1];
    const int64_t M_cap = M, n_tiles_cap = (M + 31) / 32;   // the layouts (saved state, pair-major input gradients) follow the capacity
    if (m_live) M = min(M, (int64_t)max(*m_live, 0));       // ... the work the rows that hold samples (see k_nwie_fwd)
    const int64_t n_tiles = (M + 31) / 32;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + wv, n_waves = (int64_t)gridDim.x * 4;
    const float inv_scale = 1.0f / loss_scale;
    NRC_PROBE(0);

    const __half* W0 = W;
    const __half* W1 = W + 64 * 32;                                   // only when N_HIDDEN == 2
    const __half* Wo = W + 64 * 32 + (N_HIDDEN - 1) * 64 * 64;
    // transposed weights as A operands of the back-propagation products
    h8 WoT[2], W1T[N_HIDDEN > 1 ? 2 : 1][4], W0T[4];
#pragma unroll
    for (int mt = 0; mt < 2; mt++) WoT[mt] = load_wT_frag<false>(Wo, 64, n_out_rows, 64, mt, 0, r, hh);
    if constexpr (N_HIDDEN > 1) {
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int s = 0; s < 4; s++) W1T[mt][s] = load_wT_frag<true>(W1, 64, 64, 64, mt, s, r, hh);
    }
#pragma unroll
    for (int s = 0; s < 4; s++) W0T[s] = load_wT_frag<true>(W0, 32, 64, 32, 0, s, r, hh);
    // the second hidden layer forward again (k_nwie_fwd's own fragments and instruction order: the same bits it had there)
    h8 A1[(N_HIDDEN > 1 && NRC_BWD_RECOMPUTE) ? 2 : 1][4];
    if constexpr (N_HIDDEN > 1 && NRC_BWD_RECOMPUTE) {
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int s = 0; s < 4; s++) A1[mt][s] = load_w_frag<true>(W1, 64, 64, mt, s, r, hh);
    }

    f16v gWo[2] = {zero16(), zero16()};
    f16v gW1[N_HIDDEN > 1 ? 2 : 1][2];
    f16v gW0[2] = {zero16(), zero16()};
#pragma unroll
    for (int a = 0; a < (N_HIDDEN > 1 ? 2 : 1); a++) { gW1[a][0] = zero16(); gW1[a][1] = zero16(); }

    // The saved forward state of a tile (input, hidden activations, output and its gradient: 13 or 21 loads per lane) is requested one tile ahead:
    // with one wave per SIMD nothing else covers the ~2 us a dependent global load takes under load.
    struct TileState { h8 X[2], H0[4], H1[(N_HIDDEN > 1 && !NRC_BWD_RECOMPUTE) ? 4 : 1], g, y; };
    auto fetch = [&](int64_t tile, TileState& t) {
        const int64_t i = tile * 32 + r;
        const int64_t ic = i < M ? i : M - 1;
        // fragment-major saved state (see k_nwie_fwd): one 16-byte load per fragment, 512 contiguous bytes per lane half
        const _Float16* xin = reinterpret_cast<const _Float16*>(save_in) + ((tile * 4 + hh) * 32 + r) * 8;
        t.X[0] = *reinterpret_cast<const h8*>(xin); t.X[1] = *reinterpret_cast<const h8*>(xin + 2 * 32 * 8);
        const _Float16* a0 = reinterpret_cast<const _Float16*>(save_acts) + ((tile * 8 + hh) * 32 + r) * 8;
#pragma unroll
        for (int s = 0; s < 4; s++) t.H0[s] = *reinterpret_cast<const h8*>(a0 + s * 2 * 32 * 8);
        if constexpr (N_HIDDEN > 1 && !NRC_BWD_RECOMPUTE) {
            const _Float16* a1 = a0 + n_tiles_cap * 32 * 64;
#pragma unroll
            for (int s = 0; s < 4; s++) t.H1[s] = *reinterpret_cast<const h8*>(a1 + s * 2 * 32 * 8);
        }
        // rows 8 hh .. 8 hh + 7 of the (padded) output row: 16 bytes when the row has 16 entries, the first 4 (8 bytes, hh = 0) when it has 4
        const _Float16* gp = reinterpret_cast<const _Float16*>(d_out) + ic * out_ld;
        const _Float16* yp = reinterpret_cast<const _Float16*>(out) + ic * out_ld;
        t.g = zero_h8(); t.y = zero_h8();
        if (tq.d_rgbs) {   // uniform: colour network of the fused training query
            if (hh == 0) {
#pragma unroll
                for (int j = 0; j < 3; j++) t.g[j] = (_Float16)tq.d_rgbs[3 * ic + j];
#pragma unroll
                for (int j = 0; j < 4; j++) if constexpr (OUT_ACT == ACT_SIGMOID) t.y[j] = yp[j];
            }
        } else if (out_ld == 16) {
            t.g = *reinterpret_cast<const h8*>(gp + 8 * hh);
            if constexpr (OUT_ACT == ACT_SIGMOID) t.y = *reinterpret_cast<const h8*>(yp + 8 * hh);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int row = 8 * hh + j;
                if (row < out_ld) { t.g[j] = gp[row]; if constexpr (OUT_ACT == ACT_SIGMOID) t.y[j] = yp[row]; }
            }
        }
    };
    NRC_PROBE(1);
    TileState nxt;
    if (wave0 < n_tiles) fetch(wave0, nxt);
    int probe_t = 0;
    for (int64_t tile = wave0; tile < n_tiles; tile += n_waves, probe_t++) {
        const int pb = 10 + 10 * (probe_t < 10 ? probe_t : 10);
        (void)pb;
        const int64_t i = tile * 32 + r;
        const bool valid = i < M;
        const TileState cur = nxt;
        if (tile + n_waves < n_tiles) fetch(tile + n_waves, nxt);
        NRC_PROBE(pb + 0);
        const h8 (&X)[2] = cur.X;
        const h8 (&H0)[4] = cur.H0;
        h8 H1r[4];
        if constexpr (N_HIDDEN > 1 && NRC_BWD_RECOMPUTE) {
            f16v a2[2] = {zero16(), zero16()};
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int s = 0; s < 4; s++) a2[mt] = NRC_MFMA(A1[mt][s], H0[s], a2[mt]);
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int gq = 0; gq < 2; gq++) H1r[2 * mt + gq] = acc_to_frag_relu(a2[mt], gq);
        } else if constexpr (N_HIDDEN > 1) {
#pragma unroll
            for (int s = 0; s < 4; s++) H1r[s] = cur.H1[s];
        }
        const h8 (&HL)[4] = N_HIDDEN > 1 ? H1r : cur.H0;  // last hidden layer
        // ---- dZ of the output layer (natural order over the 16 padded output rows)
        h8 dZo = zero_h8();
        if (valid) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int row = 8 * hh + j;
                if (row < out_ld) {
                    float v = (float)cur.g[j];
                    if constexpr (OUT_ACT == ACT_SIGMOID) { const float yy = (float)cur.y[j]; v = v * yy * (1.f - yy); }
                    dZo[j] = (_Float16)(v * loss_scale);
                }
            }
        }
        NRC_PROBE(pb + 1);
        // ---- dWout += dZo^T . HL   (k = samples, through the transposed LDS images)
        stage_frag_T<false>(T_dz, dZo, 0, r, hh);
#pragma unroll
        for (int s = 0; s < 4; s++) stage_frag_T<true>(T_act, HL[s], s, r, hh);
        wave_lds_sync();
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const h8 a_all = read_T_frag(T_dz, 0, s, lane);   // (all lanes take part in the read)
            const h8 a = r < 16 ? a_all : zero_h8();
#pragma unroll
            for (int nt = 0; nt < 2; nt++) gWo[nt] = NRC_MFMA(a, read_T_frag(T_act, nt, s, lane), gWo[nt]);
        }
        wave_lds_sync();
        NRC_PROBE(pb + 2);
        // ---- dHL^T = Wout^T . dZo^T ; dZL = relu'(HL) * dHL
        f16v acc[2] = {zero16(), zero16()};
#pragma unroll
        for (int mt = 0; mt < 2; mt++) acc[mt] = NRC_MFMA(WoT[mt], dZo, acc[mt]);
        h8 dZ[4];
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int gq = 0; gq < 2; gq++) dZ[2 * mt + gq] = masked_grad_frag(acc[mt], gq, HL[2 * mt + gq]);
        NRC_PROBE(pb + 3);
        if constexpr (N_HIDDEN > 1) {
            // ---- dW1 += dZ1^T . H0
#pragma unroll
            for (int s = 0; s < 4; s++) { stage_frag_T<true>(T_dz, dZ[s], s, r, hh); stage_frag_T<true>(T_act, H0[s], s, r, hh); }
            wave_lds_sync();
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const h8 b0 = read_T_frag(T_act, 0, s, lane), b1 = read_T_frag(T_act, 1, s, lane);
#pragma unroll
                for (int mt = 0; mt < 2; mt++) {
                    const h8 a = read_T_frag(T_dz, mt, s, lane);
                    gW1[mt][0] = NRC_MFMA(a, b0, gW1[mt][0]);
                    gW1[mt][1] = NRC_MFMA(a, b1, gW1[mt][1]);
                }
            }
            wave_lds_sync();
            NRC_PROBE(pb + 4);
            // ---- dH0^T = W1^T . dZ1^T ; dZ0 = relu'(H0) * dH0
            acc[0] = zero16(); acc[1] = zero16();
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int s = 0; s < 4; s++) acc[mt] = NRC_MFMA(W1T[mt][s], dZ[s], acc[mt]);
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int gq = 0; gq < 2; gq++) dZ[2 * mt + gq] = masked_grad_frag(acc[mt], gq, H0[2 * mt + gq]);
        }
        NRC_PROBE(pb + 5);
        // ---- dW0 += dZ0^T . X
#pragma unroll
        for (int s = 0; s < 4; s++) stage_frag_T<true>(T_dz, dZ[s], s, r, hh);
        stage_frag_T<false>(T_act, X[0], 0, r, hh);
        stage_frag_T<false>(T_act, X[1], 1, r, hh);
        wave_lds_sync();
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const h8 b = read_T_frag(T_act, 0, s, lane);
#pragma unroll
            for (int mt = 0; mt < 2; mt++) gW0[mt] = NRC_MFMA(read_T_frag(T_dz, mt, s, lane), b, gW0[mt]);
        }
        wave_lds_sync();
        NRC_PROBE(pb + 6);
        // ---- d_in^T = W0^T . dZ0^T  (32 input features x 32 samples), unscaled f32
        f16v din = zero16();
#pragma unroll
        for (int s = 0; s < 4; s++) din = NRC_MFMA(W0T[s], dZ[s], din);
        if (valid && tq.d_h16) {
            // accumulator registers 8..15 of lane half hh are input features 16 + 4 hh + (0..3) and 24 + 4 hh + (0..3): the density outputs
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = din[8 + j] * inv_scale;
            if (hh == 0) {
                const float h0 = fminf(15.f, fmaxf(-15.f, __half2float(tq.h[i * 16])));
                v[0] += tq.d_sigmas[i] * expf(h0);
            }
            h4 lo, hi;
#pragma unroll
            for (int j = 0; j < 4; j++) { lo[j] = (_Float16)v[j]; hi[j] = (_Float16)v[4 + j]; }
            _Float16* p = reinterpret_cast<_Float16*>(tq.d_h16) + i * 16 + 4 * hh;
            *reinterpret_cast<h4*>(p) = lo;
            *reinterpret_cast<h4*>(p + 8) = hi;
        }
        if (valid && d_in) {
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                float4 v = make_float4(din[4 * gq] * inv_scale, din[4 * gq + 1] * inv_scale, din[4 * gq + 2] * inv_scale, din[4 * gq + 3] * inv_scale);
                bad = bad || !(fabsf(v.x) < __builtin_inff()) || !(fabsf(v.y) < __builtin_inff()) || !(fabsf(v.z) < __builtin_inff()) || !(fabsf(v.w) < __builtin_inff());
                if (d_in_pair_major) {  // [16 pairs][M][2]: what the grid backward reads, coalesced over the samples
                    const int pair = 4 * gq + 2 * hh;
                    *reinterpret_cast<float2*>(d_in + ((int64_t)pair * M_cap + i) * 2) = make_float2(v.x, v.y);
                    *reinterpret_cast<float2*>(d_in + ((int64_t)(pair + 1) * M_cap + i) * 2) = make_float2(v.z, v.w);
                } else {
                    *reinterpret_cast<float4*>(d_in + i * 32 + 8 * gq + 4 * hh) = v;
                }
            }
        }
        NRC_PROBE(pb + 7);
    }
    NRC_PROBE(120);
    // ---- flush the weight-gradient accumulators (layout of W).  The four waves of the workgroup first add their tiles up through LDS (the
    // staging area is free now), tile after tile with two alternating slots, and ONE wave per tile issues the global atomics: with one flush
    // per wave the 1 024 waves of a launch queued 1 024 float atomics on each of the 3 072 / 7 168 addresses (54 of the 160 us of the two
    // backward launches of a training iteration), now 256.
    float* gW0p = dW;
    float* gW1p = dW + 64 * 32;
    float* gWop = dW + 64 * 32 + (N_HIDDEN - 1) * 64 * 64;
    float* red = reinterpret_cast<float*>(&lds[0][0][0]);   // [2 slots][4 waves][16 registers][64 lanes] f32 = 32 KB of the 40 KB
    int tile_no = 0;
    auto reduce_flush = [&](float* G, int ld, int n_rows, int mt, int nt, f16v acc) {
        float* slot = red + (tile_no & 1) * 4096;
        __syncthreads();   // the slot's previous tile (two tiles ago) has been read; first tile: every wave has left the sample loop
#pragma unroll
        for (int reg = 0; reg < 16; reg++) slot[(wv * 16 + reg) * 64 + lane] = acc[reg];
        __syncthreads();
        if (wv == (tile_no & 3)) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 4; w++) v += slot[(w * 16 + reg) * 64 + lane];   // the same order on every run
                acc[reg] = v;
                bad |= !(fabsf(v) < __builtin_inff());
            }
            atomic_add_tile(G, ld, n_rows, mt, nt, acc, inv_scale, r, hh);
        }
        tile_no++;
    };
#pragma unroll
    for (int mt = 0; mt < 2; mt++) reduce_flush(gW0p, 32, 64, mt, 0, gW0[mt]);
    if constexpr (N_HIDDEN > 1) {
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++) reduce_flush(gW1p, 64, 64, mt, nt, gW1[mt][nt]);
    }
#pragma unroll
    for (int nt = 0; nt < 2; nt++) reduce_flush(gWop, 64, n_out_rows, 0, nt, gWo[nt]);
    if (bad_flag && __ballot(bad) != 0ull && lane == 0) __hip_atomic_store(bad_flag, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    NRC_PROBE(121);
}

// hash-grid backward, small / dense levels: one lane per (sample, level); scatter-add of w_corner * dL/dfeature into the f32 table
// gradient.  Neighbouring lanes are consecutive samples of a ray and, on all but the finest levels, fall into the same cell: the
// sixteen contributions (8 corners x 2 features) are combined over RUNS of equal cells inside the wave (segmented scan) and only the
// last lane of a run has anything to add.  Those sums leave the wave COOPERATIVELY: the run's last lane parks its sixteen values and
// eight entry indices in a wave-private LDS slot, and sixteen lanes per run issue the atomics, lane q the value q -- the two features
// of an entry are 8 contiguous bytes and, on a dense level, the x-neighbour is the next entry, so one wave instruction adds 16-byte
// groups (8-byte on a hashed level) instead of 64 unrelated floats twice.  Float atomics execute at the memory side per 64-byte
// request: the request count, not the lane count, is what they cost (the same finding as k_render_bw's records, gs_raster.hip).
// d_feat: pair-major [n_levels][M][2] or sample-major (M, 2 n_levels).
struct LevelList { int n; int level[NRC_MAX_LEVELS]; };
__global__ void __launch_bounds__(256) k_grid_bwd(const float* __restrict__ x, int64_t M, const float* __restrict__ d_feat, int pair_major, GridCfg g,
                                                  int n_levels, LevelList ll, float* __restrict__ grad_table, const int32_t* __restrict__ m_live = nullptr) {
    __shared__ float s_val[4][64][16];
    __shared__ uint32_t s_key[4][64][8];
    const int level = ll.level[blockIdx.y];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t M_cap = M;
    if (m_live) M = min(M, (int64_t)max(*m_live, 0));
    const bool in_range = i < M;
    const int64_t ic = in_range ? i : max(M - 1, (int64_t)0);
    const float2 gf = pair_major ? *reinterpret_cast<const float2*>(d_feat + ((int64_t)level * M_cap + ic) * 2)
                                 : *reinterpret_cast<const float2*>(d_feat + ic * 2 * n_levels + 2 * level);
    const bool live = in_range && !(gf.x == 0.f && gf.y == 0.f);
    if (__ballot(live) == 0ull) return;
    Corner8 c;
    if (g.hashed[level]) grid_corners_u<true>(x[3 * ic], x[3 * ic + 1], x[3 * ic + 2], g.scale[level], g.res[level], g.size[level], g.offset[level], c);
    else grid_corners_u<false>(x[3 * ic], x[3 * ic + 1], x[3 * ic + 2], g.scale[level], g.res[level], g.size[level], g.offset[level], c);
    // a run: consecutive live lanes whose eight entries are all the same (dead lanes split runs)
    // (every shuffle executed by the whole wave: no short-circuit in front of a cross-lane read)
    const int prev_live = __shfl_up((int)live, 1, 64);
    uint32_t differ = 0u;
#pragma unroll
    for (int k = 0; k < 8; k++) differ |= __shfl_up(c.e[k], 1, 64) ^ c.e[k];
    const bool same = lane > 0 && live && prev_live != 0 && differ == 0u;
    const bool head = !same;
    const int next_head = __shfl_down((int)head, 1, 64);
    const bool tail = lane == 63 || next_head != 0;
    float v[16];
#pragma unroll
    for (int k = 0; k < 8; k++) { v[2 * k] = live ? c.w[k] * gf.x : 0.f; v[2 * k + 1] = live ? c.w[k] * gf.y : 0.f; }
    if (__ballot(!head) != 0ull) {  // some run is longer than one lane
        int start = head ? lane : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(start, d, 64);
            if (lane >= d) start = max(start, o);
        }
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const bool take = lane - d >= start;
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const float o = __shfl_up(v[q], d, 64);
                if (take) v[q] += o;
            }
        }
    }
    const bool flush = live && tail;
    const uint64_t tails = __ballot(flush);
    const int n_tails = __popcll(tails);
    if (flush) {
        const int slot = __popcll(tails & ((1ull << lane) - 1ull));
        float4* sv = reinterpret_cast<float4*>(s_val[wv][slot]);
#pragma unroll
        for (int q = 0; q < 4; q++) sv[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        uint4* sk = reinterpret_cast<uint4*>(s_key[wv][slot]);
        sk[0] = make_uint4(c.e[0], c.e[1], c.e[2], c.e[3]);
        sk[1] = make_uint4(c.e[4], c.e[5], c.e[6], c.e[7]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int q = lane & 15;
#ifndef NRC_GBW_TAILS_PER_INSTR
#define NRC_GBW_TAILS_PER_INSTR 4
#endif
    for (int s0 = 0; s0 < n_tails; s0 += NRC_GBW_TAILS_PER_INSTR) {
        const int sl = s0 + (lane >> 4);
        if ((lane >> 4) < NRC_GBW_TAILS_PER_INSTR && sl < n_tails) {
            atomicAdd(grad_table + 2 * (size_t)s_key[wv][sl][q >> 1] + (q & 1), s_val[wv][sl][q]);
        }
    }
}

// hash-grid backward, hashed levels of a large batch: OWNERSHIP instead of atomics.  Scattered f32 atomics run at ~20 G/s on this
// chip whatever their locality (measured per level: 4.2 M atomics = 0.2 ms, 1.3 of the 1.5 ms of the whole backward), and a
// sample's corners on these levels are pseudo-random, so nothing can be combined.  Instead every workgroup OWNS a 16 K-entry
// slice of one level's table, keeps it in LDS (128 KB), walks ALL samples, recomputes their eight corner indices and keeps the
// (general grids: see k_grid_bwd_general below)
// ones that fall into its slice (LDS atomics).  32x redundant index arithmetic, zero global atomics; the slice is added to the
// gradient table with plain coalesced read-modify-writes (the workgroup is its only writer).
template <int F>
__global__ void __launch_bounds__(256) k_grid_bwd_general(const float* __restrict__ x, int64_t M, const float* __restrict__ d_in, GridCfg g, float* __restrict__ grad_table) {
    const int level = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    float gl[F];
    bool any = false;
#pragma unroll
    for (int f = 0; f < F; f++) { gl[f] = d_in[i * 32 + level * F + f]; any = any || gl[f] != 0.f; }
    if (!any) return;
    Corner8 c;
    if (g.hashed[level]) grid_corners_u<true>(x[3 * i], x[3 * i + 1], x[3 * i + 2], g.scale[level], g.res[level], g.size[level], g.offset[level], c);
    else grid_corners_u<false>(x[3 * i], x[3 * i + 1], x[3 * i + 2], g.scale[level], g.res[level], g.size[level], g.offset[level], c);
#pragma unroll
    for (int k = 0; k < 8; k++)
#pragma unroll
        for (int f = 0; f < F; f++) atomicAdd(grad_table + (size_t)c.e[k] * F + f, c.w[k] * gl[f]);
}

#define OWN_ENTRIES 16384
#define OWN_THREADS 1024
struct OwnedCfg { int n_levels; int level[NRC_MAX_LEVELS]; int unit0[NRC_MAX_LEVELS + 1]; };
__global__ void __launch_bounds__(OWN_THREADS) k_grid_bwd_owned(const float* __restrict__ x, int64_t M, const float* __restrict__ d_feat, GridCfg g,
                                                                OwnedCfg oc, float* __restrict__ grad_table, const int32_t* __restrict__ m_live) {
    extern __shared__ float own_acc[];  // [OWN_ENTRIES][2]
    const int64_t M_cap = M;            // d_feat is pair-major over the row CAPACITY; only the live rows hold samples
    if (m_live) M = min(M, (int64_t)max(*m_live, 0));
    int li = 0;
    while (li + 1 < oc.n_levels && (int)blockIdx.x >= oc.unit0[li + 1]) li++;
    const int level = oc.level[li];
    const uint32_t chunk = (uint32_t)((int)blockIdx.x - oc.unit0[li]);
    const uint32_t lo = g.offset[level] + chunk * OWN_ENTRIES;
    const uint32_t n_own = min((uint32_t)OWN_ENTRIES, g.size[level] - chunk * OWN_ENTRIES);
    for (uint32_t j = threadIdx.x; j < 2 * n_own; j += OWN_THREADS) own_acc[j] = 0.f;
    __syncthreads();
    const float scale = g.scale[level];
    const uint32_t res = g.res[level], size = g.size[level], off = g.offset[level];
    const float2* gfp = reinterpret_cast<const float2*>(d_feat) + (int64_t)level * M_cap;
    for (int64_t i = threadIdx.x; i < M; i += OWN_THREADS) {
        const float2 gf = gfp[i];
        if (gf.x == 0.f && gf.y == 0.f) continue;
        Corner8 c;
        grid_corners_u<true>(x[3 * i], x[3 * i + 1], x[3 * i + 2], scale, res, size, off, c);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t rel = c.e[k] - lo;
            if (rel < n_own) {
                atomicAdd(&own_acc[2 * rel], c.w[k] * gf.x);
                atomicAdd(&own_acc[2 * rel + 1], c.w[k] * gf.y);
            }
        }
    }
    __syncthreads();
    float2* out = reinterpret_cast<float2*>(grad_table) + lo;
    for (uint32_t j = threadIdx.x; j < n_own; j += OWN_THREADS) {
        const float a0 = own_acc[2 * j], a1 = own_acc[2 * j + 1];
        if (a0 != 0.f || a1 != 0.f) {
            float2 v = out[j];
            v.x += a0; v.y += a1;
            out[j] = v;
        }
    }
}

// Same ownership scheme with a sparse inner loop.  On a hashed level the slice of a corner, (index >> 14), depends only on its
// (y, z) part: index = (x ^ y P1 ^ z P2) & mask and x < 2^14, so x cannot touch the slice bits.  A sample therefore has four
// candidate slices, and seven samples out of eight have none in the slice this workgroup owns.  The main loop only computes the four
// (y, z) hashes and queues the few samples that do hit (per-wave LDS queue); whenever 64 are queued the wave processes them
// densely (all lanes busy) with the full corner arithmetic.  ~70 instead of ~140 instructions per sample and slice.
#define OWN_Q 128  // queue entries per wave
__global__ void __launch_bounds__(OWN_THREADS) k_grid_bwd_owned_q(const float* __restrict__ x, int64_t M, const float* __restrict__ d_feat, GridCfg g,
                                                                  OwnedCfg oc, float* __restrict__ grad_table, const int32_t* __restrict__ m_live) {
    extern __shared__ float own_acc[];  // [OWN_ENTRIES][2] then the queues
    const int64_t M_cap = M;            // d_feat is pair-major over the row CAPACITY; only the live rows hold samples
    if (m_live) M = min(M, (int64_t)max(*m_live, 0));
    uint32_t* queue = reinterpret_cast<uint32_t*>(own_acc + 2 * OWN_ENTRIES) + (threadIdx.x >> 6) * OWN_Q;
    int li = 0;
    while (li + 1 < oc.n_levels && (int)blockIdx.x >= oc.unit0[li + 1]) li++;
    const int level = oc.level[li];
    const uint32_t chunk = (uint32_t)((int)blockIdx.x - oc.unit0[li]);
    const uint32_t lo = g.offset[level] + chunk * OWN_ENTRIES;
    const uint32_t n_own = min((uint32_t)OWN_ENTRIES, g.size[level] - chunk * OWN_ENTRIES);
    for (uint32_t j = threadIdx.x; j < 2 * n_own; j += OWN_THREADS) own_acc[j] = 0.f;
    __syncthreads();
    const float scale = g.scale[level];
    const uint32_t res = g.res[level], size = g.size[level], off = g.offset[level], mask = size - 1u;
    const float2* gfp = reinterpret_cast<const float2*>(d_feat) + (int64_t)level * M_cap;
    const int lane = threadIdx.x & 63;
    auto scatter = [&](int64_t i) {  // all eight corners of sample i, the ones inside the slice are accumulated
        const float2 gf = gfp[i];
        Corner8 c;
        grid_corners_u<true>(x[3 * i], x[3 * i + 1], x[3 * i + 2], scale, res, size, off, c);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t rel = c.e[k] - lo;
            if (rel < n_own) {
                atomicAdd(&own_acc[2 * rel], c.w[k] * gf.x);
                atomicAdd(&own_acc[2 * rel + 1], c.w[k] * gf.y);
            }
        }
    };
    int qn = 0;  // wave-uniform
    // four samples per lane and turn, their loads issued together: with one sample per turn the loop was bound by the latency of its
    // two dependent loads at 4 waves per SIMD (480 us), not by instructions
    enum { UNR = 4 };
    const int64_t M_pad = (M + OWN_THREADS * UNR - 1) / (OWN_THREADS * UNR) * (OWN_THREADS * UNR);
    for (int64_t i0 = threadIdx.x; i0 < M_pad; i0 += OWN_THREADS * UNR) {
        float2 gfv[UNR];
        float yv[UNR], zv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int64_t i = i0 + (int64_t)u * OWN_THREADS;
            const int64_t ic = i < M ? i : max(M - 1, (int64_t)0);
            gfv[u] = gfp[ic];
            yv[u] = x[3 * ic + 1]; zv[u] = x[3 * ic + 2];
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int64_t i = i0 + (int64_t)u * OWN_THREADS;
            bool hit = false;
            if (i < M && !(gfv[u].x == 0.f && gfv[u].y == 0.f)) {
                const float fy = fmaf(scale, yv[u], 0.5f), fz = fmaf(scale, zv[u], 0.5f);
                const uint32_t gy = (uint32_t)(int32_t)floorf(fy), gz = (uint32_t)(int32_t)floorf(fz);
                const uint32_t ty0 = gy * 2654435761u, tz0 = gz * 805459861u;
                const uint32_t ty1 = ty0 + 2654435761u, tz1 = tz0 + 805459861u;
                hit = (((ty0 ^ tz0) & mask) >> 14) == chunk || (((ty1 ^ tz0) & mask) >> 14) == chunk || (((ty0 ^ tz1) & mask) >> 14) == chunk ||
                      (((ty1 ^ tz1) & mask) >> 14) == chunk;
            }
            const unsigned long long m = __ballot(hit);
            if (m) {
                if (hit) queue[qn + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
                qn += __popcll(m);
                if (qn >= 64) {
                    qn -= 64;
                    scatter((int64_t)queue[qn + lane]);
                }
            }
        }
    }
    if (lane < qn) scatter((int64_t)queue[lane]);
    __syncthreads();
    float2* out = reinterpret_cast<float2*>(grad_table) + lo;
    for (uint32_t j = threadIdx.x; j < n_own; j += OWN_THREADS) {
        const float a0 = own_acc[2 * j], a1 = own_acc[2 * j + 1];
        if (a0 != 0.f || a1 != 0.f) {
            float2 v = out[j];
            v.x += a0; v.y += a1;
            out[j] = v;
        }
    }
}

// ---- bucketed ownership ("multisplit"): split once, accumulate once, in fixed point ------------------------------------------------
// The ownership kernels above make every slice owner scan ALL samples (32 x redundant index arithmetic, and every owner streams all
// positions and gradients through L2: 1.3 GB per backward).  Here the (sample, (y,z) corner pair) items are first split by the slice
// that owns them -- on a hashed level the slice, index >> 13, depends only on the (y, z) hash, so both x-neighbours of a pair live in
// the same slice -- in TWO launches and without any global counting:
//   k_gb_split       one workgroup per 1024 samples.  It ranks its items inside their buckets (one LDS atomic each, the rank kept in a
//                    register), scans its <= 1024 bucket counts and writes 16-byte self-contained records {two slice-local entries,
//                    wy wz g (2), wx} bucket after bucket into ITS OWN region of the workspace; a table [bucket][workgroup] holds
//                    where each of its runs starts and how long it is, a second one its largest |gradient| per level.  Nothing is
//                    shared between workgroups: no counters to zero, no count pass, no scan kernel (round 2 had all three: 36 us of a
//                    training iteration and three launches).
//   k_gb_accumulate  one workgroup per bucket: walks the table row of its bucket, reads every workgroup's run (contiguous, ~64 records
//                    on a training batch) into an 8 K-entry LDS slice, then a plain read-modify-write flush.  The slice accumulates
//                    64-bit FIXED POINT: LDS float atomics run at 0.32 lane-operations per clock and CU on this chip, integer ones
//                    (32 or 64 bit) at 1.33 (tools/micro/lds_atomics.hip) -- with f32 atomics this kernel took 380 us, 320 of them in
//                    ds_add_f32.  scale = 2^(62 - ceil log2 max|g| - ceil log2 (2 M + 1)): no overflow whatever the collisions, LSB <=
//                    max|g| 2^-41 for M = 264 K, and the sums no longer depend on the order of the atomics: the hashed levels' gradient
//                    is bit-reproducible.
// (A first version with 4-byte (sample, pair) records still gathered positions / gradients per record -- spread over all samples, i.e.
// the same lines again.)
#define GB_MAX_BUCKETS 1024
#define GB_ENTRIES 8192
#define GB_SHIFT 13
#define GBA_THREADS 1024
struct BucketCfg { int n_levels; int level[NRC_MAX_LEVELS]; int bucket0[NRC_MAX_LEVELS + 1]; };
// workspace: [seg table nb x n_wg u32: start | count << 16][level maxima n_wg x NRC_MAX_LEVELS u32][records n_wg x 1024 x 4 x n_levels x 16 B]
struct GbLayout { int64_t n_wg, seg_bytes, max_bytes, rec_per_wg, total; };
static GbLayout gb_layout(int64_t M, int n_bucket_levels, int nb) {
    GbLayout L;
    L.n_wg = nrc_cdiv(M, OWN_THREADS);
    L.seg_bytes = ((int64_t)nb * L.n_wg * 4 + 255) / 256 * 256;
    L.max_bytes = (L.n_wg * NRC_MAX_LEVELS * 4 + 255) / 256 * 256;
    L.rec_per_wg = (int64_t)OWN_THREADS * 4 * n_bucket_levels;
    L.total = L.seg_bytes + L.max_bytes + L.n_wg * L.rec_per_wg * 16 + 256;
    return L;
}

#define GB_GROUP 16   // levels ranked together: 2 x GB_GROUP rank registers + 2 x GB_GROUP gradient registers per thread (16 = no grouping)
__global__ void __launch_bounds__(OWN_THREADS) k_gb_split(const float* __restrict__ x, int64_t M, const float* __restrict__ d_feat, GridCfg g, BucketCfg bc,
                                                             uint32_t* __restrict__ seg, uint32_t* __restrict__ wg_max, uint4* __restrict__ records,
                                                             int64_t rec_per_wg, const int32_t* __restrict__ m_live = nullptr) {
    // (measured and dropped, round 5: dealing a batch of more than 256 x 1 024 samples evenly over one workgroup per CU in several passes instead of a
    // second, almost empty round of workgroups -- 0.398 against 0.392 ms per training iteration: the few workgroups of the second round are cheap)
    // The levels are ranked in groups of GB_GROUP (rank and gradient registers per thread grow with the group).  16 = all levels at once is the
    // measured best: groups of 6 fit 64 VGPRs, i.e. two workgroups per CU, but spill and repeat the barriers (90 us against 56).
    __shared__ uint32_t hist[GB_MAX_BUCKETS], gbase[GB_MAX_BUCKETS], lmax[NRC_MAX_LEVELS], wave_tot[OWN_THREADS / 64];
    extern __shared__ uint4 stage[];   // [2][4096] records: two level images
    const int64_t M_cap = M;
    if (m_live) M = min(M, (int64_t)max(*m_live, 0));   // workgroups behind the live rows still publish their (empty) runs
    const int n_wg = (int)gridDim.x;
    const int v = (int)blockIdx.x;
    if (threadIdx.x < NRC_MAX_LEVELS) lmax[threadIdx.x] = 0u;
    const int64_t i = (int64_t)blockIdx.x * OWN_THREADS + threadIdx.x;
    const bool in_range = i < M;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (in_range) { px = x[3 * i]; py = x[3 * i + 1]; pz = x[3 * i + 2]; }
    // the four (y, z) pairs of sample i on bucketed level k: bucket and slice-local (y, z) hash
    auto pairs_of = [&](int k, uint32_t (&bucket)[4], uint32_t (&yz)[4], float& wy1, float& wz1) {
        const int level = bc.level[k];
        const float scale = g.scale[level];
        const uint32_t mask = g.size[level] - 1u;
        const float fy = fmaf(scale, py, 0.5f), fz = fmaf(scale, pz, 0.5f);
        const float fly = floorf(fy), flz = floorf(fz);
        const uint32_t gy = (uint32_t)(int32_t)fly, gz = (uint32_t)(int32_t)flz;
        const uint32_t ty[2] = {gy * 2654435761u, gy * 2654435761u + 2654435761u}, tz[2] = {gz * 805459861u, gz * 805459861u + 805459861u};
        wy1 = fy - fly; wz1 = fz - flz;
#pragma unroll
        for (uint32_t pair = 0; pair < 4; pair++) {
            yz[pair] = (ty[pair & 1] ^ tz[pair >> 1]) & mask;
            bucket[pair] = (uint32_t)bc.bucket0[k] + (yz[pair] >> GB_SHIFT);
        }
    };
    uint4* mine = records + (int64_t)v * rec_per_wg;
    uint32_t region = 0u;   // records of the groups before this one
    for (int k0 = 0; k0 < bc.n_levels; k0 += GB_GROUP) {
        const int b_lo = bc.bucket0[k0], b_hi = bc.bucket0[min(k0 + GB_GROUP, bc.n_levels)];
        for (int b = b_lo + threadIdx.x; b < b_hi; b += OWN_THREADS) hist[b] = 0u;
        __syncthreads();   // (also: the previous group's gbase has been read)
        // all gradients of the group requested at once and kept for the second pass: level after level, every one of these loads was a full
        // memory latency in front of the LDS atomics that depend on it
        float2 gf_all[GB_GROUP];
#pragma unroll
        for (int kk = 0; kk < GB_GROUP; kk++) {
            gf_all[kk] = make_float2(0.f, 0.f);
            if (k0 + kk < bc.n_levels && in_range) gf_all[kk] = reinterpret_cast<const float2*>(d_feat)[(int64_t)bc.level[k0 + kk] * M_cap + i];
        }
        // pass 1: rank of every item inside its bucket (kept in registers: 4 x 16 bit per level), largest |gradient| per level
        uint32_t rank01[GB_GROUP], rank23[GB_GROUP];
        uint32_t live_mask = 0u;
#pragma unroll
        for (int kk = 0; kk < GB_GROUP; kk++) {
            rank01[kk] = 0u; rank23[kk] = 0u;
            const int k = k0 + kk;
            if (k < bc.n_levels) {   // uniform
                const float2 gf = gf_all[kk];
                float m = fmaxf(fabsf(gf.x), fabsf(gf.y));
                if (!(fabsf(gf.x) < __builtin_inff()) || !(fabsf(gf.y) < __builtin_inff())) m = __builtin_inff();  // inf / NaN (AMP overflow): poisons the level
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));   // wave maximum first: one LDS atomic per wave and level
                if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(&lmax[k], __float_as_uint(m));
                if (in_range && !(gf.x == 0.f && gf.y == 0.f)) {  // masked / terminated samples have no items
                    live_mask |= 1u << kk;
                    uint32_t bucket[4], yz[4]; float wy1, wz1;
                    pairs_of(k, bucket, yz, wy1, wz1);
                    const uint32_t r0 = atomicAdd(&hist[bucket[0]], 1u), r1 = atomicAdd(&hist[bucket[1]], 1u);
                    const uint32_t r2 = atomicAdd(&hist[bucket[2]], 1u), r3 = atomicAdd(&hist[bucket[3]], 1u);
                    rank01[kk] = r0 | (r1 << 16); rank23[kk] = r2 | (r3 << 16);   // a bucket holds at most 4096 items of this workgroup
                }
            }
        }
        __syncthreads();
        // exclusive scan of the group's bucket counts (two per thread): where each bucket's run starts inside this workgroup's region
        {
            const int t = threadIdx.x, b = b_lo + 2 * t;
            const uint32_t c0 = b < b_hi ? hist[b] : 0u, c1 = b + 1 < b_hi ? hist[b + 1] : 0u;
            const uint32_t cnt = c0 + c1;
            uint32_t incl = cnt;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d, 64);
                if ((t & 63) >= d) incl += o;
            }
            if ((t & 63) == 63) wave_tot[t >> 6] = incl;
            __syncthreads();
            uint32_t before = region, all = 0u;
#pragma unroll
            for (int w = 0; w < OWN_THREADS / 64; w++) { before += w < (t >> 6) ? wave_tot[w] : 0u; all += wave_tot[w]; }
            const uint32_t start = before + incl - cnt;
            if (b < b_hi) {
                gbase[b] = start;
                seg[(int64_t)b * n_wg + v] = start | (c0 << 16);   // start < 1024 * 4 * 16 = 65536, count <= 4096
            }
            if (b + 1 < b_hi) {
                gbase[b + 1] = start + c0;
                seg[(int64_t)(b + 1) * n_wg + v] = (start + c0) | (c1 << 16);
            }
            region += all;
        }
        __syncthreads();
        // pass 2: the records, level by level through an LDS image of the level's part of the region (its buckets are contiguous there), then
        // out in whole lines.  Written straight from the lanes, every store instruction touched 64 different lines (16 bytes each): 700 such
        // instructions per workgroup were most of this kernel's time.  Two images: one barrier per level.
#pragma unroll
        for (int kk = 0; kk < GB_GROUP; kk++) {
            const int k = k0 + kk;
            if (k < bc.n_levels) {   // uniform
                uint4* img = stage + (kk & 1) * (OWN_THREADS * 4);
                const uint32_t lvl_lo = gbase[bc.bucket0[k]];
                if (live_mask >> kk & 1u) {
                    const int level = bc.level[k];
                    const float2 gf = gf_all[kk];
                    uint32_t bucket[4], yz[4]; float wy1, wz1;
                    pairs_of(k, bucket, yz, wy1, wz1);
                    const float fx = fmaf(g.scale[level], px, 0.5f), flx = floorf(fx);
                    const uint32_t gx = (uint32_t)(int32_t)flx;
                    const uint32_t wxb = __float_as_uint(fx - flx);
#pragma unroll
                    for (uint32_t pair = 0; pair < 4; pair++) {
                        const uint32_t rank = (pair < 2 ? rank01[kk] : rank23[kk]) >> (16 * (pair & 1u)) & 0xffffu;
                        const float wyz = ((pair & 1u) ? wy1 : 1.f - wy1) * ((pair >> 1) ? wz1 : 1.f - wz1);
                        const uint32_t e0 = (gx ^ yz[pair]) & (GB_ENTRIES - 1u), e1 = ((gx + 1u) ^ yz[pair]) & (GB_ENTRIES - 1u);
                        img[gbase[bucket[pair]] - lvl_lo + rank] = make_uint4(e0 | (e1 << 16), __float_as_uint(wyz * gf.x), __float_as_uint(wyz * gf.y), wxb);
                    }
                }
                __syncthreads();
                const uint32_t last = (uint32_t)bc.bucket0[k + 1] - 1u;
                const uint32_t lvl_n = gbase[last] + hist[last] - lvl_lo;   // records of this workgroup on the level (<= 4096)
                for (uint32_t j = threadIdx.x; j < lvl_n; j += OWN_THREADS) mine[lvl_lo + j] = img[j];
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < NRC_MAX_LEVELS) wg_max[(int64_t)v * NRC_MAX_LEVELS + threadIdx.x] = lmax[threadIdx.x];
}

__global__ void __launch_bounds__(GBA_THREADS) k_gb_accumulate(GridCfg g, BucketCfg bc, int64_t M, int n_wg, const uint32_t* __restrict__ seg,
                                                               const uint32_t* __restrict__ wg_max, const uint4* __restrict__ records, int64_t rec_per_wg,
                                                               float* __restrict__ grad_table, int assign) {
    const int bid = (int)blockIdx.x;
    extern __shared__ long long fix_acc[];  // [GB_ENTRIES][2]
    __shared__ uint32_t s_max;
    NRC_PROBE_NW(0);
    int li = 0;
    while (li + 1 < bc.n_levels && bid >= bc.bucket0[li + 1]) li++;
    const int level = bc.level[li];
    const uint32_t chunk = (uint32_t)(bid - bc.bucket0[li]);
    const uint32_t lo = g.offset[level] + chunk * GB_ENTRIES;
    float scale = 0.f;   // set below, before the first record is added
    const uint32_t* row = seg + (int64_t)bid * n_wg;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    enum { UNR = 8, WAVES = GBA_THREADS / 64, PASSES = 8 };
    // Wave wv takes the runs of workgroups w = UNR (wv + WAVES m) + u, batch m = 0, 1, ...; the table entries of 64 of its runs come in
    // with ONE vector load (lane 8 j + u).  The UNR runs of a batch are ONE flat item range (a run is ~64 records, not exactly: run by
    // run, the items past the 64th cost a second, nearly empty round of LDS atomics each): item p belongs to run u = #{t : p >= pre[t]}.
    // The records of batch m + 1 are requested before the atomics of batch m are issued (loads and LDS atomics of a wave overlap).
    struct Batch { uint32_t start[UNR], pre[UNR + 1]; int w0; uint4 r[PASSES]; };
    // float -> 64-bit fixed point, round to nearest even, |v| < 2^62: six VALU instructions instead of the ~25 of the generic f32 -> i64
    // conversion (there is no such instruction; four conversions per record made this kernel VALU-bound).  rndne is exact, hi = floor(v / 2^32)
    // fits an i32, v - hi 2^32 is an exact integer in [0, 2^32).
    auto to_fixed = [](float v) -> unsigned long long {
        const float vr = __builtin_rintf(v);
        const float hf = __builtin_floorf(vr * 2.3283064365386963e-10f);
        const int hi = (int)hf;
        const uint32_t lo = (uint32_t)__builtin_fmaf(-hf, 4294967296.f, vr);
        return ((unsigned long long)(uint32_t)hi << 32) | lo;
    };
    auto add_record = [&](const uint4& rr) {
        const uint32_t e0 = rr.x & 0xffffu, e1 = rr.x >> 16;
        const float a = __uint_as_float(rr.y), b = __uint_as_float(rr.z), wx1 = __uint_as_float(rr.w), wx0 = 1.f - wx1;
        // power-of-two scale: the product is exact, the only rounding is to the fixed-point grid
        atomicAdd(reinterpret_cast<unsigned long long*>(&fix_acc[2 * e0]), to_fixed(wx0 * a * scale));
        atomicAdd(reinterpret_cast<unsigned long long*>(&fix_acc[2 * e0 + 1]), to_fixed(wx0 * b * scale));
        atomicAdd(reinterpret_cast<unsigned long long*>(&fix_acc[2 * e1]), to_fixed(wx1 * a * scale));
        atomicAdd(reinterpret_cast<unsigned long long*>(&fix_acc[2 * e1 + 1]), to_fixed(wx1 * b * scale));
    };
    auto item = [&](const Batch& B, uint32_t pp) -> uint4 {   // record of flat item pp (< B.pre[UNR]) of the batch
        uint32_t u = 0u;
#pragma unroll
        for (int t = 1; t < UNR; t++) u += pp >= B.pre[t] ? 1u : 0u;
        uint32_t base = B.start[0], first = 0u;
#pragma unroll
        for (int t = 1; t < UNR; t++) { base = u == (uint32_t)t ? B.start[t] : base; first = u == (uint32_t)t ? B.pre[t] : first; }
        return records[(int64_t)(B.w0 + (int)u) * rec_per_wg + base + (pp - first)];
    };
    const int n_batches = (n_wg + UNR * WAVES - 1) / (UNR * WAVES);   // per wave (the last ones may be empty for the higher waves)
    uint32_t my_desc = 0u;
    auto open_batch = [&](int m, Batch& B) {   // m uniform
        if ((m & (64 / UNR - 1)) == 0) {       // a new block of 64 table entries
            const int my_w = UNR * (wv + WAVES * (m + lane / UNR)) + (lane % UNR);
            my_desc = my_w < n_wg ? row[my_w] : 0u;
        }
        B.w0 = UNR * (wv + WAVES * m);
        B.pre[0] = 0u;
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const uint32_t d = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)my_desc, (m & (64 / UNR - 1)) * UNR + u, 64));
            B.start[u] = d & 0xffffu; B.pre[u + 1] = B.pre[u] + (d >> 16);
        }
#pragma unroll
        for (int q = 0; q < PASSES; q++) {
            const uint32_t pp = 64 * q + lane;
            B.r[q] = make_uint4(0u, 0u, 0u, 0u);
            if (pp < B.pre[UNR]) B.r[q] = item(B, pp);
        }
    };
    // requested first, used last: the level maxima of all split workgroups (one 64-byte line each), the table entries, the first records
    uint32_t m_local = 0u;
    for (int w = threadIdx.x; w < n_wg; w += GBA_THREADS) m_local = max(m_local, wg_max[(int64_t)w * NRC_MAX_LEVELS + li]);
    Batch cur, nxt;
    if (n_batches > 0) open_batch(0, cur);
    NRC_PROBE_NW(1);
    // (the first batch's records are on their way while the slice is cleared and the scale is worked out)
    for (uint32_t j = threadIdx.x; j < 2 * GB_ENTRIES; j += GBA_THREADS) fix_acc[j] = 0ll;
    if (threadIdx.x == 0) s_max = 0u;
    __syncthreads();
    NRC_PROBE_NW(2);
    // the level's largest |gradient| over all workgroups of the split (non-negative floats order like their bit patterns) -> the power-of-two
    // fixed-point scale; every bucket of a level derives the same one
    {
        uint32_t m = m_local;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
        if ((threadIdx.x & 63) == 0 && m) atomicMax(&s_max, m);
    }
    __syncthreads();
    NRC_PROBE_NW(3);
    const float mx = __uint_as_float(s_max);
    if (!(mx < __builtin_inff())) {
        // a non-finite gradient on the level (GradScaler overflow): the fixed-point path cannot carry it; mark the slice NaN, the way f32
        // atomics would have propagated it, so that the scaler's found-inf check still sees it
        if (assign) {   // the slice has no other writer and was not initialised: all of it
            for (uint32_t j = threadIdx.x; j < GB_ENTRIES; j += GBA_THREADS) reinterpret_cast<float2*>(grad_table)[lo + j] = make_float2(__builtin_nanf(""), __builtin_nanf(""));
        } else if (threadIdx.x == 0) reinterpret_cast<float2*>(grad_table)[lo] = make_float2(__builtin_nanf(""), __builtin_nanf(""));
        return;
    }
    int e_max = 0, e_cnt = 0;
    frexpf(mx > 0.f ? mx : 1.f, &e_max);      // mx < 2^e_max
    frexpf((float)(2 * M + 1), &e_cnt);        // 2 M + 1 <= 2^e_cnt (an entry receives at most 2 M values of <= mx)
    scale = ldexpf(1.f, 62 - e_max - e_cnt);
    for (int m = 0; m < n_batches; m++) {
        if (m + 1 < n_batches) open_batch(m + 1, nxt);
        NRC_PROBE_NW(10 + 2 * m);
        const uint32_t total = cur.pre[UNR];
#pragma unroll
        for (int q = 0; q < PASSES; q++)
            if (64 * q + lane < total) add_record(cur.r[q]);
        for (uint32_t pp = 64 * PASSES + lane; pp < total; pp += 64) add_record(item(cur, pp));   // a batch of more than 512 items (rare)
        NRC_PROBE_NW(11 + 2 * m);
        cur = nxt;
    }
    __syncthreads();
    NRC_PROBE_NW(40);
    const double inv = 1.0 / (double)scale;
    float2* out = reinterpret_cast<float2*>(grad_table) + lo;
    // read-modify-write of the slice: all of a thread's table reads first (entry by entry, each one was a full memory latency in front of its store)
    enum { PER_THREAD = GB_ENTRIES / GBA_THREADS };
    if (assign) {   // sole writer of an uninitialised slice (nrc_ngp_train_query_backward_set): every entry is written, nothing is read
#pragma unroll
        for (int t = 0; t < PER_THREAD; t++) {
            const uint32_t j = threadIdx.x + t * GBA_THREADS;
            out[j] = make_float2((float)((double)fix_acc[2 * j] * inv), (float)((double)fix_acc[2 * j + 1] * inv));
        }
        return;
    }
    float2 old_v[PER_THREAD];
#pragma unroll
    for (int t = 0; t < PER_THREAD; t++) old_v[t] = out[threadIdx.x + t * GBA_THREADS];
#pragma unroll
    for (int t = 0; t < PER_THREAD; t++) {
        const uint32_t j = threadIdx.x + t * GBA_THREADS;
        const long long a0 = fix_acc[2 * j], a1 = fix_acc[2 * j + 1];
        if (a0 != 0ll || a1 != 0ll) {
            float2 v = old_v[t];
            v.x += (float)((double)a0 * inv); v.y += (float)((double)a1 * inv);
            out[j] = v;
        }
    }
    NRC_PROBE_NW(41);
}

// which levels take the bucketed path: hashed, whole 8 K slices, x corners below the slice bits; the finest first, <= GB_MAX_BUCKETS
static void pick_bucket_levels(const GridCfg& g, int n_levels, BucketCfg& bc, bool* is_bucketed) {
    bc.n_levels = 0; bc.bucket0[0] = 0;
    for (int l = 0; l < NRC_MAX_LEVELS; l++) is_bucketed[l] = false;
    static const int max_lv = [] { const char* e = getenv("NRC_GB_LEVELS"); return e ? atoi(e) : NRC_MAX_LEVELS; }();
    int first = n_levels, units = 0;
    for (int l = n_levels - 1; l >= 0 && n_levels - l <= max_lv; l--) {
        if (!g.hashed[l] || g.size[l] < (uint32_t)GB_ENTRIES || (g.size[l] % GB_ENTRIES) != 0u || g.res[l] + 1u >= (uint32_t)GB_ENTRIES) break;
        const int c = (int)(g.size[l] / GB_ENTRIES);
        if (units + c > GB_MAX_BUCKETS) break;
        units += c;
        first = l;
    }
    for (int l = first; l < n_levels; l++) {
        is_bucketed[l] = true;
        bc.level[bc.n_levels] = l;
        bc.bucket0[bc.n_levels + 1] = bc.bucket0[bc.n_levels] + (int)(g.size[l] / GB_ENTRIES);
        bc.n_levels++;
    }
}

// ---- fused training query (InstantNGPRayRenderingComponent.query_model, Renderer.py:48-53, as one autograd node) ----------------
// (the f32 outputs the compositor consumes -- sigma = exp(h0), rgb -- are written by the colour kernel's epilogue, see k_nwie_fwd)
// (the two element-wise kernels that used to sit around the colour network's backward -- upstream gradients -> fp16 d_out rows + TruncExp backward,
// d_in of the colour network -> d_out rows of the density network -- are part of k_nwie_bwd now: TrainQ)

}  // namespace

static int nwie_backward_impl(int64_t M, const void* weights_f16, int32_t n_hidden, int32_t out_act, int32_t n_out_rows, const void* d_out_f16,
                              const void* out_f16, int32_t out_ld, const void* save_in, const void* save_acts, float loss_scale,
                              float* grad_weights, float* d_in, int32_t d_in_pair_major, nrc_stream_t stream, TrainQ tq, const int32_t* m_live = nullptr,
                              float* bad_flag = nullptr) {
    if (M < 0 || !weights_f16 || !grad_weights || n_out_rows < 1 || n_out_rows > 16 || out_ld < 4 || out_ld > 16 || !(loss_scale > 0.f)) return NRC_ERR_INVALID;
    if (n_hidden < 1 || n_hidden > 2 || (out_act != ACT_NONE && out_act != ACT_SIGMOID)) return NRC_ERR_UNSUPPORTED;
    if (M == 0) return NRC_OK;
    if ((!d_out_f16 && !tq.d_rgbs) || !out_f16 || !save_in || !save_acts) return NRC_ERR_INVALID;
    if (tq.d_rgbs && (!tq.d_sigmas || !tq.h || !tq.d_h16 || n_hidden != 2 || out_ld != 4)) return NRC_ERR_INVALID;
    // every wave ends with an atomic flush of its weight-gradient accumulators (3 072 / 7 168 values): the number of waves, not the
    // batch, sets that cost -- one workgroup per CU (NRC_BWD_BLOCKS env override for experiments)
    static const int max_blocks = [] { const char* e = getenv("NRC_BWD_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
    const int64_t need = nrc_cdiv(nrc_cdiv(M, 32), 4);
    const dim3 grid((unsigned)(need < max_blocks ? need : max_blocks)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define NRC_BWD(H, A)                                                                                                          \
    hipLaunchKernelGGL((k_nwie_bwd<H, A>), grid, block, 0, s, M, (const __half*)weights_f16, (int)n_out_rows, (const __half*)d_out_f16, \
                       (const __half*)out_f16, (int)out_ld, (const __half*)save_in, (const __half*)save_acts, loss_scale, grad_weights, d_in, (int)d_in_pair_major, tq, m_live, bad_flag)
    if (n_hidden == 1) { if (out_act == ACT_SIGMOID) NRC_BWD(1, ACT_SIGMOID); else NRC_BWD(1, ACT_NONE); }
    else { if (out_act == ACT_SIGMOID) NRC_BWD(2, ACT_SIGMOID); else NRC_BWD(2, ACT_NONE); }
#undef NRC_BWD
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

extern "C" {

int nrc_nwie_backward(int64_t M, const void* weights_f16, int32_t n_hidden, int32_t out_act, int32_t n_out_rows, const void* d_out_f16,
                      const void* out_f16, int32_t out_ld, const void* save_in, const void* save_acts, float loss_scale,
                      float* grad_weights, float* d_in, int32_t d_in_pair_major, nrc_stream_t stream) {
    NRC_ENTER();
    return nwie_backward_impl(M, weights_f16, n_hidden, out_act, n_out_rows, d_out_f16, out_f16, out_ld, save_in, save_acts, loss_scale, grad_weights, d_in,
                              d_in_pair_major, stream, TrainQ{nullptr, nullptr, nullptr, nullptr});
}

#if defined(NRC_BWD_PROBE)
extern "C" int nrc_debug_bwd_probe(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_probe), sizeof(g_bwd_probe)); }
#endif

int64_t nrc_nwie_save_rows(int64_t M) { return M < 0 ? NRC_ERR_INVALID : (M + 31) / 32 * 32; }

int64_t nrc_grid_backward_ws_bytes(int64_t M, int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale) {
    if (M < 0) return NRC_ERR_INVALID;
    GridCfg g;
    if (make_grid_cfg(n_levels, log2_hashmap_size, base_resolution, per_level_scale, g, nullptr) != NRC_OK) return NRC_ERR_INVALID;
    BucketCfg bc; bool isb[NRC_MAX_LEVELS];
    pick_bucket_levels(g, n_levels, bc, isb);
    return gb_layout(M, bc.n_levels, bc.bucket0[bc.n_levels]).total;
}

}  // extern "C"

// the optimizer step behind a training query's backward pass (nrc_ngp_train_backward_step)
struct TrainStep {
    float *param_d, *m_d, *v_d; void* h_d; float l2c_d; int64_t l2n_d;
    float *param_c, *m_c, *v_c; void* h_c; float l2c_c; int64_t l2n_c;
    float lr; const float* lr_dev; float beta1, beta2, eps, weight_decay; int adam_w_mode;
    int32_t* device_step; float* bc; float* scale; int32_t* growth_tracker; float growth_factor, backoff_factor; int32_t growth_interval;
    float* state4; hipStream_t fork_stream;
    float *grad_d, *grad_c; int64_t n_mlp_d, n_d, n_c;   // gradient vectors and sizes (filled in by train_query_backward_impl)
};
// Adam on elements [from, to) of the density vector (l2 slice relative to the vector's start) and, with_colour, on the whole colour vector
static void step_adam_range(const TrainStep& st, int64_t from, int64_t to, bool with_colour, hipStream_t s) {
    const int64_t l2n = st.l2n_d > from ? st.l2n_d - from : 0;
    nrc_launch_amp_adam(st.param_d + from, st.grad_d + from, st.m_d + from, st.v_d + from, st.h_d ? (void*)((__half*)st.h_d + from) : nullptr, to - from, st.l2c_d, l2n,
                        st.param_c, st.grad_c, st.m_c, st.v_c, st.h_c, with_colour ? st.n_c : 0, st.l2c_c, st.l2n_c, st.state4, st.bc, st.lr_dev, st.lr, st.beta1,
                        st.beta2, st.eps, st.weight_decay, st.adam_w_mode, s);
}

// does this call take the bucketed path (all conditions of grid_backward_impl in one place: nrc_ngp_train_query_backward_set needs the answer first)
static bool gb_will_bucket(int64_t M, int pair_major, const void* workspace, const GridCfg& g, int n_levels, BucketCfg& bc, bool* isb) {
    static const bool allow_owned = [] { const char* e = getenv("NRC_GRID_BWD_OWNED"); return !(e && e[0] == '0'); }();
    static const bool allow_buckets = [] { const char* e = getenv("NRC_GRID_BWD_BUCKETS"); return !(e && e[0] == '0'); }();
    if (!(allow_owned && allow_buckets && pair_major && M >= 16384 && workspace && M < (int64_t(1) << 30))) return false;
    pick_bucket_levels(g, n_levels, bc, isb);
    return bc.n_levels > 0;
}

// assign != 0: the bucketed levels' slices are WRITTEN (uninitialised memory, no other writer), only valid when gb_will_bucket() holds
static int grid_backward_impl(const float* x01, int64_t M, const float* d_features, int32_t d_features_pair_major, int32_t n_levels,
                              int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale, float* grad_table, void* workspace,
                              nrc_stream_t stream, int assign, const int32_t* m_live = nullptr, hipStream_t fork_stream = nullptr) {
    if (M < 0 || !grad_table) return NRC_ERR_INVALID;
    if (M == 0) return NRC_OK;
    if (!x01 || !d_features) return NRC_ERR_INVALID;
    GridCfg g;
    const int rc = make_grid_cfg(n_levels, log2_hashmap_size, base_resolution, per_level_scale, g, nullptr);
    if (rc != NRC_OK) return rc;
    // ownership path: hashed levels of a large, pair-major batch (its cost is ~32 x M index computations per level whatever M is
    // worth in atomics, plus a 128 KB slice flush per workgroup: below ~16 K samples the atomics are cheaper)
    static const bool allow_owned = [] { const char* e = getenv("NRC_GRID_BWD_OWNED"); return !(e && e[0] == '0'); }();
    const bool owned = allow_owned && d_features_pair_major && M >= 16384;
    hipStream_t s = (hipStream_t)stream;
    // bucketed ownership (needs the workspace): all hashed levels in two launches, the dense ones through the run-aggregated atomics
    {
        BucketCfg bc; bool isb[NRC_MAX_LEVELS];
        if (gb_will_bucket(M, d_features_pair_major, workspace, g, n_levels, bc, isb)) {
            LevelList rest; rest.n = 0;
            for (int l = 0; l < n_levels; l++) if (!isb[l]) rest.level[rest.n++] = l;
            NRC_STAGE(s, nullptr);
            // fork_stream (optional): the dense levels' atomics run there, next to the bucketed levels' split / accumulate on `stream` (memory-side
            // atomics against LDS atomics: 12-17 us of a training iteration); the two events belong to the library
            // (per host thread and device: a record / wait pair is issued back to back by one thread, so two threads driving two streams never see
            // each other's records; an event belongs to the device it was created on)
            enum { MAX_EV = 2, MAX_DEV = 64 };
            static thread_local hipEvent_t ev_of[MAX_DEV][MAX_EV] = {};
            hipStream_t side = fork_stream ? fork_stream : s;
            int dev = 0;
            if (side != s && (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV)) side = s;   // no events for this device: one stream
            const bool forked = side != s;
            hipEvent_t* ev = ev_of[dev];
            if (forked) {
                if (!ev[0]) {
                    for (int k = 0; k < MAX_EV; k++)
                        if (hipEventCreateWithFlags(&ev[k], hipEventDisableTiming) != hipSuccess) { ev[0] = nullptr; return NRC_ERR_LAUNCH; }
                }
                if (hipEventRecord(ev[0], s) != hipSuccess || hipStreamWaitEvent(side, ev[0], 0) != hipSuccess) return NRC_ERR_LAUNCH;
            }
            if (rest.n > 0)
                hipLaunchKernelGGL(k_grid_bwd, dim3((unsigned)nrc_cdiv(M, 256), rest.n), dim3(256), 0, side, x01, M, d_features, (int)d_features_pair_major, g,
                                   (int)n_levels, rest, grad_table, m_live);
            NRC_STAGE(s, "k_grid_bwd");
            const int nb = bc.bucket0[bc.n_levels];
            const GbLayout L = gb_layout(M, bc.n_levels, nb);
            if (L.n_wg > 0x7fffffff / nb) return NRC_ERR_INVALID;
            uint32_t* seg = reinterpret_cast<uint32_t*>(workspace);
            uint32_t* wg_max = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(workspace) + L.seg_bytes);
            uint4* records = reinterpret_cast<uint4*>(reinterpret_cast<char*>(workspace) + L.seg_bytes + L.max_bytes);
            static const hipError_t attr_s = hipFuncSetAttribute((const void*)k_gb_split, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * OWN_THREADS * 4 * 16);
            (void)attr_s;
            hipLaunchKernelGGL(k_gb_split, dim3((unsigned)L.n_wg), dim3(OWN_THREADS), 2 * OWN_THREADS * 4 * 16, s, x01, M, d_features, g, bc, seg, wg_max, records, L.rec_per_wg, m_live);
            NRC_STAGE(s, "k_gb_split");
            static const hipError_t attr_b = hipFuncSetAttribute((const void*)k_gb_accumulate, hipFuncAttributeMaxDynamicSharedMemorySize, GB_ENTRIES * 16);
            (void)attr_b;
            hipLaunchKernelGGL(k_gb_accumulate, dim3((unsigned)nb), dim3(GBA_THREADS), GB_ENTRIES * 16, s, g, bc, M, (int)L.n_wg, (const uint32_t*)seg,
                               (const uint32_t*)wg_max, (const uint4*)records, L.rec_per_wg, grad_table, assign);
            NRC_STAGE(s, "k_gb_accumulate");
            if (forked && (hipEventRecord(ev[1], side) != hipSuccess || hipStreamWaitEvent(s, ev[1], 0) != hipSuccess)) return NRC_ERR_LAUNCH;
            NRC_LAUNCH_CHECK();
            return NRC_OK;
        }
    }
    LevelList ll; ll.n = 0;
    OwnedCfg oc; oc.n_levels = 0; oc.unit0[0] = 0;
    // Which hashed levels are owned: the FINEST ones, as many as give one workgroup per CU (256 slices = 8 levels at T = 2^19).  A
    // ninth level would add a second, mostly idle round of workgroups (352 slices: 0.44 ms instead of 0.23), while the coarser hashed
    // levels still have long runs of equal entries along a ray and cost 10-60 us each through the run-aggregated atomics.
    int first_owned = n_levels;
    if (owned) {
        int units = 0;
        for (int l = n_levels - 1; l >= 0 && g.hashed[l]; l--) {
            const int c = (int)nrc_cdiv(g.size[l], OWN_ENTRIES);
            if (units + c > 256) break;
            units += c;
            first_owned = l;
        }
    }
    for (int l = 0; l < n_levels; l++) {
        if (l >= first_owned) {
            oc.level[oc.n_levels] = l;
            oc.unit0[oc.n_levels + 1] = oc.unit0[oc.n_levels] + (int)nrc_cdiv(g.size[l], OWN_ENTRIES);
            oc.n_levels++;
        } else {
            ll.level[ll.n++] = l;
        }
    }
    if (ll.n > 0)
        hipLaunchKernelGGL(k_grid_bwd, dim3((unsigned)nrc_cdiv(M, 256), ll.n), dim3(256), 0, s, x01, M, d_features, (int)d_features_pair_major, g,
                           (int)n_levels, ll, grad_table, m_live);   // every path walks the LIVE rows only (rows behind them are uninitialised)
    if (oc.n_levels > 0) {
        static const hipError_t attr = hipFuncSetAttribute((const void*)k_grid_bwd_owned, hipFuncAttributeMaxDynamicSharedMemorySize, OWN_ENTRIES * 8);
        static const hipError_t attr_q = hipFuncSetAttribute((const void*)k_grid_bwd_owned_q, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                             OWN_ENTRIES * 8 + (OWN_THREADS / 64) * OWN_Q * 4);
        (void)attr; (void)attr_q;
        bool sparse_ok = M < (int64_t(1) << 32);  // queue entries are 32-bit sample indices; x corners must stay below the slice bits
        for (int k = 0; k < oc.n_levels; k++) sparse_ok = sparse_ok && g.res[oc.level[k]] + 1u < (uint32_t)OWN_ENTRIES && g.size[oc.level[k]] >= (uint32_t)OWN_ENTRIES;
        static const bool allow_q = [] { const char* e = getenv("NRC_GRID_BWD_SPARSE"); return !(e && e[0] == '0'); }();
        if (sparse_ok && allow_q)
            hipLaunchKernelGGL(k_grid_bwd_owned_q, dim3((unsigned)oc.unit0[oc.n_levels]), dim3(OWN_THREADS), OWN_ENTRIES * 8 + (OWN_THREADS / 64) * OWN_Q * 4,
                               s, x01, M, d_features, g, oc, grad_table, m_live);
        else
            hipLaunchKernelGGL(k_grid_bwd_owned, dim3((unsigned)oc.unit0[oc.n_levels]), dim3(OWN_THREADS), OWN_ENTRIES * 8, s, x01, M, d_features, g, oc,
                               grad_table, m_live);
    }
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

extern "C" {

/* the general grid's backward (see encode_grid_general): d_in (M, 32) f32 sample-major, input k = level * F + c; plain f32 atomics, one lane per (sample, level) */
int nrc_grid_backward_general(const float* x01, int64_t M, const float* d_in, int32_t n_levels, int32_t n_features, int32_t log2_hashmap_size,
                              int32_t base_resolution, float per_level_scale, float* grad_table, nrc_stream_t stream) {
    NRC_ENTER();
    if (M < 0 || !grad_table) return NRC_ERR_INVALID;
    if ((n_features != 2 && n_features != 4) || n_levels < 1 || n_levels * n_features > 32) return NRC_ERR_UNSUPPORTED;
    if (M == 0) return NRC_OK;
    if (!x01 || !d_in) return NRC_ERR_INVALID;
    GridCfg g;
    const int rc = make_grid_cfg(n_levels, log2_hashmap_size, base_resolution, per_level_scale, g, nullptr);
    if (rc != NRC_OK) return rc;
    const dim3 grid((unsigned)nrc_cdiv(M, 256), (unsigned)n_levels);
    if (n_features == 4) hipLaunchKernelGGL(k_grid_bwd_general<4>, grid, dim3(256), 0, (hipStream_t)stream, x01, M, d_in, g, grad_table);
    else hipLaunchKernelGGL(k_grid_bwd_general<2>, grid, dim3(256), 0, (hipStream_t)stream, x01, M, d_in, g, grad_table);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_grid_backward(const float* x01, int64_t M, const float* d_features, int32_t d_features_pair_major, int32_t n_levels,
                      int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale, float* grad_table, void* workspace,
                      nrc_stream_t stream) {
    NRC_ENTER();
    return grid_backward_impl(x01, M, d_features, d_features_pair_major, n_levels, log2_hashmap_size, base_resolution, per_level_scale, grad_table,
                              workspace, stream, 0);
}

/* nrc_grid_backward over the first n_samples_dev[0] of M rows (a launch sized for a row capacity: fixed-capacity training batches) */
int nrc_grid_backward_live(const float* x01, int64_t M, const float* d_features, int32_t d_features_pair_major, int32_t n_levels, int32_t log2_hashmap_size,
                           int32_t base_resolution, float per_level_scale, float* grad_table, void* workspace, const int32_t* n_samples_dev,
                           nrc_stream_t stream) {
    NRC_ENTER();
    return grid_backward_impl(x01, M, d_features, d_features_pair_major, n_levels, log2_hashmap_size, base_resolution, per_level_scale, grad_table,
                              workspace, stream, 0, n_samples_dev);
}

/* bytes of the scratch the two calls below need: features of all M samples */
int64_t nrc_ngp_train_query_ws_bytes(int64_t M) { return nrc_nwie_forward_ws_bytes(M); }

int nrc_ngp_train_query_forward(const float* xyzs, const float* dirs, int64_t M, const float* xyz_min3, const float* xyz_size3,
                                const void* density_weights_f16, const void* color_weights_f16, const void* table_f16, int32_t n_levels,
                                int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale, float* x01, void* h_f16, void* rgb_f16,
                                float* sigmas, float* rgbs, void* save_in_d, void* save_acts_d, void* save_in_c, void* save_acts_c, void* workspace,
                                const int32_t* n_samples_dev, nrc_stream_t stream) {
    NRC_ENTER();
    if (M < 0 || !density_weights_f16 || !color_weights_f16 || !table_f16 || !xyz_min3 || !xyz_size3) return NRC_ERR_INVALID;
    if (n_levels != 16) return NRC_ERR_UNSUPPORTED;
    if (M == 0) return NRC_OK;
    if (!xyzs || !dirs || !x01 || !h_f16 || !rgb_f16 || !sigmas || !rgbs || !save_in_d || !save_acts_d || !save_in_c || !save_acts_c || !workspace)
        return NRC_ERR_INVALID;
    GridCfg g;
    const int rc = make_grid_cfg(n_levels, log2_hashmap_size, base_resolution, per_level_scale, g, nullptr);
    if (rc != NRC_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    QueryIn qin = {};
    qin.xyz01 = xyzs; qin.normalise = 1; qin.x01_out = x01; qin.m_live = n_samples_dev;
    for (int k = 0; k < 3; k++) { qin.mn[k] = xyz_min3[k]; qin.sz[k] = xyz_size3[k]; }
    NRC_STAGE(s, nullptr);
    launch_encode<SRC_ARRAYS>(qin, 0, M, table_f16, g, (uint4*)workspace, s);
    NRC_STAGE(s, "k_grid_encode<train>");
    const dim3 grid(pick_blocks(M)), block(256);
    // density net: 32 -> 64 -> 16, linear output, all 16 columns stored (the colour net reads them back)
    hipLaunchKernelGGL((k_nwie_fwd<ENC_FEAT, 1, ACT_NONE, true>), grid, block, 0, s, (const void*)workspace, 0, M, (const __half*)density_weights_f16,
                       (const __half2*)table_f16, g, 16, (__half*)h_f16, 16, 16, (__half*)save_in_d, (__half*)save_acts_d, (float*)nullptr, (float*)nullptr, n_samples_dev);
    NRC_STAGE(s, "k_nwie_fwd<density>");
    // colour net: [SH(d) | h] -> 64 -> 64 -> 3 (+1 pad), sigmoid
    hipLaunchKernelGGL((k_nwie_fwd<ENC_DIR_H, 2, ACT_SIGMOID, true>), grid, block, 0, s, (const void*)dirs, 0, M, (const __half*)color_weights_f16,
                       (const __half2*)h_f16, g, 3, (__half*)rgb_f16, 4, 4, (__half*)save_in_c, (__half*)save_acts_c, sigmas, rgbs, n_samples_dev);
    NRC_STAGE(s, "k_nwie_fwd<colour>");
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

}  // extern "C"

// zeroes two float ranges in one launch (the parts of the gradient buffers nobody writes whole)
__global__ void __launch_bounds__(256) k_zero_two(float* __restrict__ a, int64_t na, float* __restrict__ b, int64_t nb) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < na; i += stride) a[i] = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += stride) b[i] = 0.f;
}

// ---- Model.weight_decay_mlp (Model.py:38-44) as one launch each way ---------------------------------------------------------------------------------
// forward: (sum a[0, na)^2 + sum b[0, nb)^2) * inv_n, ONE workgroup, fixed summation order (reproducible)
__global__ void __launch_bounds__(1024) k_sumsq_two(const float* __restrict__ a, int64_t na, const float* __restrict__ b, int64_t nb, float inv_n, float* __restrict__ out) {
    __shared__ float part[16];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < na; i += 1024) acc += a[i] * a[i];
    for (int64_t i = threadIdx.x; i < nb; i += 1024) acc += b[i] * b[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 16; w++) tot += part[w];
        out[0] = tot * inv_n;
    }
}
// backward, folded into the clearing of the gradient buffers in front of a training query's backward pass: ga[0, za) and gb[0, zb) are cleared and
// their leading seed_a / seed_b elements start at coeff * up[0] * w instead of zero (the gradient of up * mean-square: 2 / n * w * up)
__global__ void __launch_bounds__(256) k_zero_seed_two(float* __restrict__ ga, int64_t za, const float* __restrict__ wa, int64_t seed_a, float* __restrict__ gb, int64_t zb,
                                                       const float* __restrict__ wb, int64_t seed_b, const float* __restrict__ up, float coeff) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const float c = coeff * up[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < za; i += stride) ga[i] = i < seed_a ? c * wa[i] : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < zb; i += stride) gb[i] = i < seed_b ? c * wb[i] : 0.f;
}

static int train_query_backward_impl(const float* dL_dsigmas, const float* dL_drgbs, int64_t M, const float* x01, const void* density_weights_f16,
                                 const void* color_weights_f16, int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution,
                                 float per_level_scale, const void* h_f16, const void* rgb_f16, const void* save_in_d, const void* save_acts_d,
                                 const void* save_in_c, const void* save_acts_c, float loss_scale, float* grad_density_params, float* grad_color_params,
                                 int64_t n_density_mlp_params, void* scratch, nrc_stream_t stream, int64_t n_density_params, int64_t n_color_params,
                                 bool precleared = false, const int32_t* m_live = nullptr, const TrainStep* step = nullptr, hipStream_t fork_stream = nullptr,
                                 int phases = 3, float* nonfinite_flag = nullptr) {
    // phases: 1 = the two network backward launches, 2 = the hash-grid backward behind them (group 14 issues them as two calls so that a data-parallel
    // rank can start its small collective in between); nonfinite_flag: where the network launches flag inf / NaN when there is no TrainStep
    if (step) nonfinite_flag = step->state4;
    if (M < 0 || !density_weights_f16 || !color_weights_f16 || !grad_density_params || !grad_color_params || n_density_mlp_params < 0) return NRC_ERR_INVALID;
    if (M > 0 && (phases & 1) && (!dL_dsigmas || !dL_drgbs || !h_f16 || !rgb_f16 || !save_in_d || !save_acts_d || !save_in_c || !save_acts_c)) return NRC_ERR_INVALID;
    if (M > 0 && (!x01 || !scratch)) return NRC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    // _set variant: uninitialised gradient buffers.  Bucketed levels are written by their owners; everything else is zeroed here, in one launch.
    int assign = 0;
    if (n_density_params > 0 || n_color_params > 0) {
        GridCfg g; BucketCfg bc; bool isb[NRC_MAX_LEVELS];
        const int rc0 = make_grid_cfg(n_levels, log2_hashmap_size, base_resolution, per_level_scale, g, nullptr);
        if (rc0 != NRC_OK) return rc0;
        int64_t zero_d = n_density_params;
        if (M > 0) {
            char* q = (char*)scratch + (M * 8 + 255) / 256 * 256 + (M * 4 + 255) / 256 * 256 + M * 128 + M * 32 + M * 128;   // grid_ws, as laid out below
            if (gb_will_bucket(M, 1, q, g, n_levels, bc, isb)) {
                assign = 1;
                zero_d = n_density_mlp_params + 2 * (int64_t)g.offset[bc.level[0]];   // bucketed levels are the finest ones: everything in front of the first
                if (zero_d > n_density_params) return NRC_ERR_INVALID;
            }
        }
        if (!precleared && (phases & 1))   // (nrc_ngp_train_query_backward_cleared: the caller cleared exactly these ranges, nrc_ngp_train_query_clear_floats)
            hipLaunchKernelGGL(k_zero_two, dim3(512), dim3(256), 0, s, grad_density_params, zero_d, grad_color_params, n_color_params);
    }
    if (M == 0) { NRC_LAUNCH_CHECK(); return NRC_OK; }
    // scratch: [(M x 4 f16)(M f32)(M x 32 f32): unused since the element-wise kernels moved into k_nwie_bwd][d_h16 M x 16 f16][d_in_density 16 x M x 2 f32]
    char* p = (char*)scratch;
    p += (M * 8 + 255) / 256 * 256; p += (M * 4 + 255) / 256 * 256; p += M * 128;
    __half* d_h16 = (__half*)p; p += M * 32;
    float* d_in_d = (float*)p;
    NRC_STAGE(s, nullptr);
    int rc = NRC_OK;
    if (phases & 1) {
    // colour network: dL/drgb read as it is, the TruncExp backward and the density network's output gradient written by its epilogue (TrainQ)
    rc = nwie_backward_impl(M, color_weights_f16, 2, ACT_SIGMOID, 3, nullptr, rgb_f16, 4, save_in_c, save_acts_c, loss_scale, grad_color_params, nullptr, 0,
                            stream, TrainQ{dL_drgbs, dL_dsigmas, (const __half*)h_f16, d_h16}, m_live, nonfinite_flag);
    if (rc != NRC_OK) return rc;
    NRC_STAGE(s, "k_nwie_bwd<colour>");
    rc = nwie_backward_impl(M, density_weights_f16, 1, ACT_NONE, 16, d_h16, h_f16, 16, save_in_d, save_acts_d, loss_scale, grad_density_params, d_in_d, 1, stream,
                            TrainQ{nullptr, nullptr, nullptr, nullptr}, m_live, nonfinite_flag);
    if (rc != NRC_OK) return rc;
    NRC_STAGE(s, "k_nwie_bwd<density>");
    }
    if (!(phases & 2)) { NRC_LAUNCH_CHECK(); return NRC_OK; }
    void* grid_ws = (char*)d_in_d + M * 128;  // nrc_grid_backward_ws_bytes (all levels at most) behind the pair-major gradients
    TrainStep st;
    if (step) {
        // found_inf is known: the two launches above flagged every non-finite value they handed on.  Step counter, bias corrections, 1 / scale
        // and the scale update in one thread in front of the grid backward (measured: Adam level group by level group on the fork stream next to the
        // accumulation of the next group -- 0.413 against 0.380 ms per iteration; the step inside the slice owners -- 0.386; both dropped, LABBOOK.md)
        nrc_launch_amp_prepare(step->device_step, step->bc, step->scale, step->growth_tracker, step->growth_factor, step->backoff_factor, step->growth_interval,
                               step->beta1, step->beta2, step->state4, s);
        NRC_STAGE(s, "k_amp_prepare");
        st = *step;
        st.grad_d = grad_density_params; st.grad_c = grad_color_params; st.n_mlp_d = n_density_mlp_params; st.n_d = n_density_params; st.n_c = n_color_params;
    }
    rc = grid_backward_impl(x01, M, d_in_d, 1, n_levels, log2_hashmap_size, base_resolution, per_level_scale, grad_density_params + n_density_mlp_params,
                            grid_ws, stream, assign, m_live, fork_stream);
    if (rc != NRC_OK) return rc;
    if (step) {   // one Adam launch over everything, no pass over the gradients in front of it
        NRC_STAGE(s, nullptr);
        step_adam_range(st, 0, n_density_params, true, s);
        NRC_STAGE(s, "k_amp_adam");
    }
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

extern "C" {

int nrc_ngp_train_query_backward(const float* dL_dsigmas, const float* dL_drgbs, int64_t M, const float* x01, const void* density_weights_f16,
                                 const void* color_weights_f16, int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale,
                                 const void* h_f16, const void* rgb_f16, const void* save_in_d, const void* save_acts_d, const void* save_in_c,
                                 const void* save_acts_c, float loss_scale, float* grad_density_params, float* grad_color_params,
                                 int64_t n_density_mlp_params, void* scratch, nrc_stream_t stream) {
    NRC_ENTER();
    return train_query_backward_impl(dL_dsigmas, dL_drgbs, M, x01, density_weights_f16, color_weights_f16, n_levels, log2_hashmap_size, base_resolution,
                                     per_level_scale, h_f16, rgb_f16, save_in_d, save_acts_d, save_in_c, save_acts_c, loss_scale, grad_density_params,
                                     grad_color_params, n_density_mlp_params, scratch, stream, 0, 0);
}
int nrc_ngp_train_query_backward_set(const float* dL_dsigmas, const float* dL_drgbs, int64_t M, const float* x01, const void* density_weights_f16,
                                     const void* color_weights_f16, int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution,
                                     float per_level_scale, const void* h_f16, const void* rgb_f16, const void* save_in_d, const void* save_acts_d,
                                     const void* save_in_c, const void* save_acts_c, float loss_scale, float* grad_density_params,
                                     float* grad_color_params, int64_t n_density_mlp_params, int64_t n_density_params, int64_t n_color_params,
                                     void* scratch, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_density_params <= 0 || n_color_params <= 0 || !grad_density_params || !grad_color_params || (M > 0 && !scratch)) return NRC_ERR_INVALID;
    return train_query_backward_impl(dL_dsigmas, dL_drgbs, M, x01, density_weights_f16, color_weights_f16, n_levels, log2_hashmap_size, base_resolution,
                                     per_level_scale, h_f16, rgb_f16, save_in_d, save_acts_d, save_in_c, save_acts_c, loss_scale, grad_density_params,
                                     grad_color_params, n_density_mlp_params, scratch, stream, n_density_params, n_color_params);
}

/* leading floats of grad_density_params that nrc_ngp_train_query_backward_cleared expects cleared (all of grad_color_params as well) */
int64_t nrc_ngp_train_query_clear_floats(int64_t M, int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale,
                                         int64_t n_density_mlp_params, int64_t n_density_params) {
    if (M < 0 || n_density_mlp_params < 0 || n_density_params < n_density_mlp_params) return NRC_ERR_INVALID;
    GridCfg g; BucketCfg bc; bool isb[NRC_MAX_LEVELS];
    if (make_grid_cfg(n_levels, log2_hashmap_size, base_resolution, per_level_scale, g, nullptr) != NRC_OK) return NRC_ERR_INVALID;
    if (M > 0 && gb_will_bucket(M, 1, (const void*)16, g, n_levels, bc, isb)) {
        const int64_t z = n_density_mlp_params + 2 * (int64_t)g.offset[bc.level[0]];
        return z > n_density_params ? NRC_ERR_INVALID : z;
    }
    return n_density_params;
}
int nrc_ngp_train_query_backward_cleared(const float* dL_dsigmas, const float* dL_drgbs, int64_t M, const float* x01, const void* density_weights_f16,
                                         const void* color_weights_f16, int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution,
                                         float per_level_scale, const void* h_f16, const void* rgb_f16, const void* save_in_d, const void* save_acts_d,
                                         const void* save_in_c, const void* save_acts_c, float loss_scale, float* grad_density_params,
                                         float* grad_color_params, int64_t n_density_mlp_params, int64_t n_density_params, int64_t n_color_params,
                                         void* scratch, const int32_t* n_samples_dev, nrc_stream_t fork_stream, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_density_params <= 0 || n_color_params <= 0 || !grad_density_params || !grad_color_params || (M > 0 && !scratch)) return NRC_ERR_INVALID;
    return train_query_backward_impl(dL_dsigmas, dL_drgbs, M, x01, density_weights_f16, color_weights_f16, n_levels, log2_hashmap_size, base_resolution,
                                     per_level_scale, h_f16, rgb_f16, save_in_d, save_acts_d, save_in_c, save_acts_c, loss_scale, grad_density_params,
                                     grad_color_params, n_density_mlp_params, scratch, stream, n_density_params, n_color_params, true, n_samples_dev, nullptr, (hipStream_t)fork_stream);
}

/* group 14: the backward pass of nrc_ngp_train_query_backward_cleared as two calls */
int nrc_ngp_train_networks_backward(const float* dL_dsigmas, const float* dL_drgbs, int64_t M, const float* x01, const void* density_weights_f16,
                                    const void* color_weights_f16, int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution,
                                    float per_level_scale, const void* h_f16, const void* rgb_f16, const void* save_in_d, const void* save_acts_d,
                                    const void* save_in_c, const void* save_acts_c, float loss_scale, float* grad_density_params,
                                    float* grad_color_params, int64_t n_density_mlp_params, int64_t n_density_params, int64_t n_color_params,
                                    void* scratch, const int32_t* n_samples_dev, float* nonfinite_flag, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_density_params <= 0 || n_color_params <= 0 || !grad_density_params || !grad_color_params || (M > 0 && !scratch)) return NRC_ERR_INVALID;
    return train_query_backward_impl(dL_dsigmas, dL_drgbs, M, x01, density_weights_f16, color_weights_f16, n_levels, log2_hashmap_size, base_resolution,
                                     per_level_scale, h_f16, rgb_f16, save_in_d, save_acts_d, save_in_c, save_acts_c, loss_scale, grad_density_params,
                                     grad_color_params, n_density_mlp_params, scratch, stream, n_density_params, n_color_params, true, n_samples_dev, nullptr, nullptr,
                                     1, nonfinite_flag);
}
int nrc_ngp_train_grid_backward(int64_t M, const float* x01, const void* density_weights_f16, const void* color_weights_f16, int32_t n_levels,
                                int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale, float* grad_density_params,
                                float* grad_color_params, int64_t n_density_mlp_params, int64_t n_density_params, int64_t n_color_params, void* scratch,
                                const int32_t* n_samples_dev, nrc_stream_t fork_stream, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_density_params <= 0 || n_color_params <= 0 || !grad_density_params || !grad_color_params || (M > 0 && !scratch)) return NRC_ERR_INVALID;
    return train_query_backward_impl(nullptr, nullptr, M, x01, density_weights_f16, color_weights_f16, n_levels, log2_hashmap_size, base_resolution,
                                     per_level_scale, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1.f, grad_density_params, grad_color_params,
                                     n_density_mlp_params, scratch, stream, n_density_params, n_color_params, true, n_samples_dev, nullptr, (hipStream_t)fork_stream, 2);
}

int nrc_ngp_train_backward_step(const float* dL_dsigmas, const float* dL_drgbs, int64_t M, const float* x01, const void* density_weights_f16,
                                const void* color_weights_f16, int32_t n_levels, int32_t log2_hashmap_size, int32_t base_resolution, float per_level_scale,
                                const void* h_f16, const void* rgb_f16, const void* save_in_d, const void* save_acts_d, const void* save_in_c,
                                const void* save_acts_c, float loss_scale, float* grad_density_params, float* grad_color_params, int64_t n_density_mlp_params,
                                int64_t n_density_params, int64_t n_color_params, void* scratch, const int32_t* n_samples_dev, float* param_d, float* exp_avg_d,
                                float* exp_avg_sq_d, void* param_f16_d, float l2_coeff_d, int64_t l2_count_d, float* param_c, float* exp_avg_c,
                                float* exp_avg_sq_c, void* param_f16_c, float l2_coeff_c, int64_t l2_count_c, float lr, const float* lr_dev, float beta1,
                                float beta2, float eps, float weight_decay, int32_t adam_w_mode, int32_t* device_step, float* bias_corrections, float* scale,
                                int32_t* growth_tracker, float growth_factor, float backoff_factor, int32_t growth_interval, float* state4,
                                nrc_stream_t fork_stream, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_density_params <= 0 || n_color_params <= 0 || !grad_density_params || !grad_color_params || (M > 0 && !scratch) || !param_d || !exp_avg_d ||
        !exp_avg_sq_d || !param_c || !exp_avg_c || !exp_avg_sq_c || !device_step || !bias_corrections || !state4 || (scale && (!growth_tracker || growth_interval < 1)) ||
        l2_count_d > n_density_mlp_params)
        return NRC_ERR_INVALID;
    TrainStep st;
    st.param_d = param_d; st.m_d = exp_avg_d; st.v_d = exp_avg_sq_d; st.h_d = param_f16_d; st.l2c_d = l2_coeff_d; st.l2n_d = l2_count_d;
    st.param_c = param_c; st.m_c = exp_avg_c; st.v_c = exp_avg_sq_c; st.h_c = param_f16_c; st.l2c_c = l2_coeff_c; st.l2n_c = l2_count_c;
    st.lr = lr; st.lr_dev = lr_dev; st.beta1 = beta1; st.beta2 = beta2; st.eps = eps; st.weight_decay = weight_decay; st.adam_w_mode = adam_w_mode;
    st.device_step = device_step; st.bc = bias_corrections; st.scale = scale; st.growth_tracker = growth_tracker; st.growth_factor = growth_factor;
    st.backoff_factor = backoff_factor; st.growth_interval = growth_interval; st.state4 = state4; st.fork_stream = (hipStream_t)fork_stream;
    return train_query_backward_impl(dL_dsigmas, dL_drgbs, M, x01, density_weights_f16, color_weights_f16, n_levels, log2_hashmap_size, base_resolution,
                                     per_level_scale, h_f16, rgb_f16, save_in_d, save_acts_d, save_in_c, save_acts_c, loss_scale, grad_density_params,
                                     grad_color_params, n_density_mlp_params, scratch, stream, n_density_params, n_color_params, true, n_samples_dev, &st, (hipStream_t)fork_stream);
}

int nrc_sum_squares_two(const float* a, int64_t n_a, const float* b, int64_t n_b, float inv_n, float* out, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_a < 0 || n_b < 0 || (n_a && !a) || (n_b && !b) || !out) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_sumsq_two, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, n_a, b, n_b, inv_n, out);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_clear_seed_two(float* grad_a, int64_t clear_a, const float* w_a, int64_t seed_a, float* grad_b, int64_t clear_b, const float* w_b, int64_t seed_b,
                       const float* upstream_dev, float coeff, nrc_stream_t stream) {
    NRC_ENTER();
    if (clear_a < 0 || clear_b < 0 || seed_a < 0 || seed_b < 0 || seed_a > clear_a || seed_b > clear_b || (clear_a && !grad_a) || (clear_b && !grad_b) ||
        (seed_a && !w_a) || (seed_b && !w_b) || ((seed_a || seed_b) && !upstream_dev))
        return NRC_ERR_INVALID;
    const int64_t big = clear_a > clear_b ? clear_a : clear_b;
    if (big == 0) return NRC_OK;
    static const float one = 1.f; (void)one;
    const int64_t blocks = nrc_cdiv(big, 256 * 4);
    hipLaunchKernelGGL(k_zero_seed_two, dim3((unsigned)(blocks < 1024 ? (blocks > 0 ? blocks : 1) : 1024)), dim3(256), 0, (hipStream_t)stream, grad_a, clear_a, w_a, seed_a, grad_b,
                       clear_b, w_b, seed_b, upstream_dev ? upstream_dev : grad_a, coeff);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int64_t nrc_ngp_train_query_scratch_bytes(int64_t M) {
    if (M < 0) return NRC_ERR_INVALID;
    return (M * 8 + 255) / 256 * 256 + (M * 4 + 255) / 256 * 256 + M * 128 + M * 32 + M * 128 + 256 + gb_layout(M, NRC_MAX_LEVELS, GB_MAX_BUCKETS).total;
}

}  // extern "C"
