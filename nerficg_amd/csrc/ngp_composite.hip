// ngp_composite.hip -- front-to-back alpha compositing (train fw/bw, test fw) and the distortion loss for gfx950.
//
// Reference semantics: volumerendering.cu:6-45 (train fw), :87-151 (train bw), :205-249 (test fw);
// losses.cu:9-61 (fw), :112-142 (bw).  The reference walks one ray per THREAD, serially, so neighbouring lanes read
// addresses one ray-length apart.  Here one wave (or a G-lane group of a wave) owns a ray and walks it 64 (G) samples at
// a time: every load/store is a contiguous 256-byte (sigmas/deltas/ts/ws) or 768-byte (rgbs) segment, the running
// transmittance is a wave-level multiplicative scan, the per-ray sums are wave reductions.  HBM-bound by design:
//   fw : 28 B read + 4 B written per sample ; bw : 44 B read + 16 B written per sample.
// f32 sums are therefore tree-ordered instead of serial; the tolerance against the oracle is stated in the tests.
#include "common.h"
#include <hip/hip_fp16.h>

namespace {

__device__ __forceinline__ float alpha_of(float sigma, float delta) { return 1.0f - __expf(-sigma * delta); }

// Transmittance bookkeeping for one G-lane chunk.  `a` = alpha of this lane's sample (0 for lanes past the ray end).
// carry = T before the chunk.  Returns T before / after this lane's sample and updates carry to T after the chunk.
template <int G>
__device__ __forceinline__ void chunk_transmittance(float a, int gl, float& carry, float& T_before, float& T_after) {
    const float incl = nrc_group_incl_prod<G>(1.0f - a, gl);
    float excl = __shfl_up(incl, 1, G);
    if (gl == 0) excl = 1.0f;
    T_before = carry * excl;
    T_after = carry * incl;
    carry = __shfl(T_after, G - 1, G);
}
// index (within the G-lane group) of the first lane with valid && T_after <= thr, or G if none
template <int G>
__device__ __forceinline__ int first_saturated(bool sat, int lane) {
    const unsigned long long m = __ballot(sat);
    unsigned long long gm = m;
    if constexpr (G < 64) gm = (m >> (lane & ~(G - 1))) & ((1ull << G) - 1ull);
    return gm ? __ffsll((long long)gm) - 1 : G;
}

// ------------------------------------------------------------------------------------------------ train forward
__global__ void __launch_bounds__(256) k_composite_train_fw(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                            const float* __restrict__ deltas, const float* __restrict__ ts,
                                                            const int64_t* __restrict__ rays_a, int64_t n_rays, float thr,
                                                            int64_t* __restrict__ total_samples, float* __restrict__ opacity,
                                                            float* __restrict__ depth, float* __restrict__ rgb,
                                                            float* __restrict__ ws) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_rays) return;
    const int64_t ray_idx = rays_a[3 * n], start = rays_a[3 * n + 1];
    const int N = (int)rays_a[3 * n + 2];
    float carry = 1.0f, accR = 0.f, accG = 0.f, accB = 0.f, accD = 0.f, accO = 0.f;
    int counted = N;
    for (int c = 0; c < N; c += 64) {
        const int i = c + lane;
        const bool valid = i < N;
        const int64_t s = start + i;
        float a = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, tt = 0.f;
        if (valid) {
            a = alpha_of(sigmas[s], deltas[s]);
            cr = rgbs[3 * s]; cg = rgbs[3 * s + 1]; cb = rgbs[3 * s + 2];
            tt = ts[s];
        }
        float Tb, Ta;
        chunk_transmittance<64>(a, lane, carry, Tb, Ta);
        const int fs = first_saturated<64>(valid && Ta <= thr, lane);
        if (valid && lane <= fs) {
            const float w = a * Tb;
            accR += w * cr; accG += w * cg; accB += w * cb; accD += w * tt; accO += w;
            ws[s] = w;
        }
        if (fs < 64) { counted = c + fs; break; }  // the saturating sample is composited but not counted (:41-44)
    }
    accR = nrc_group_sum<64>(accR); accG = nrc_group_sum<64>(accG); accB = nrc_group_sum<64>(accB);
    accD = nrc_group_sum<64>(accD); accO = nrc_group_sum<64>(accO);
    if (lane == 0) {
        rgb[3 * ray_idx] = accR; rgb[3 * ray_idx + 1] = accG; rgb[3 * ray_idx + 2] = accB;
        depth[ray_idx] = accD; opacity[ray_idx] = accO;
        total_samples[ray_idx] = counted;
    }
}

// ------------------------------------------------------------------------------------------------ ray gradients of the training march
// x_i = o + t_i d  and  dir_i = d  for the samples i of a ray: dL/do = sum_i dL/dx_i,  dL/dd = sum_i (t_i dL/dx_i + dL/ddir_i)
// (custom_functions.py:122-137 builds the same sums with torch_scatter.segment_csr).  One wave per row of rays_a, coalesced 768-byte reads.
__global__ void __launch_bounds__(256) k_march_train_bw(const float* __restrict__ g_xyzs, const float* __restrict__ g_dirs, const float* __restrict__ ts,
                                                        const int64_t* __restrict__ rays_a, int64_t n_rays, float* __restrict__ g_o, float* __restrict__ g_d) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_rays) return;
    const int64_t start = rays_a[3 * n + 1];
    const int N = (int)rays_a[3 * n + 2];
    float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 0.f;
    for (int i = lane; i < N; i += 64) {
        const int64_t s = start + i;
        const float t = ts[s];
        const float gx = g_xyzs[3 * s], gy = g_xyzs[3 * s + 1], gz = g_xyzs[3 * s + 2];
        ox += gx; oy += gy; oz += gz;
        dx += gx * t; dy += gy * t; dz += gz * t;
        if (g_dirs) { dx += g_dirs[3 * s]; dy += g_dirs[3 * s + 1]; dz += g_dirs[3 * s + 2]; }
    }
    ox = nrc_group_sum<64>(ox); oy = nrc_group_sum<64>(oy); oz = nrc_group_sum<64>(oz);
    dx = nrc_group_sum<64>(dx); dy = nrc_group_sum<64>(dy); dz = nrc_group_sum<64>(dz);
    if (lane == 0) {
        g_o[3 * n] = ox; g_o[3 * n + 1] = oy; g_o[3 * n + 2] = oz;
        g_d[3 * n] = dx; g_d[3 * n + 1] = dy; g_d[3 * n + 2] = dz;
    }
}

// ------------------------------------------------------------------------------------------------ train backward
__global__ void __launch_bounds__(256) k_composite_train_bw(
    const float* __restrict__ dL_dopacity, const float* __restrict__ dL_ddepth, const float* __restrict__ dL_drgb,
    const float* __restrict__ dL_dws, const float* __restrict__ sigmas, const float* __restrict__ rgbs,
    const float* __restrict__ ws, const float* __restrict__ deltas, const float* __restrict__ ts,
    const int64_t* __restrict__ rays_a, const float* __restrict__ opacity, const float* __restrict__ depth,
    const float* __restrict__ rgb, int64_t n_rays, float thr, float* __restrict__ dL_dsigmas, float* __restrict__ dL_drgbs) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_rays) return;
    const int64_t ray_idx = rays_a[3 * n], start = rays_a[3 * n + 1];
    const int N = (int)rays_a[3 * n + 2];
    if (N <= 0) return;
    const float R = rgb[3 * ray_idx], Gc = rgb[3 * ray_idx + 1], B = rgb[3 * ray_idx + 2];
    const float O = opacity[ray_idx], D = depth[ray_idx];
    const float gR = dL_drgb[3 * ray_idx], gG = dL_drgb[3 * ray_idx + 1], gB = dL_drgb[3 * ray_idx + 2];
    const float gO = dL_dopacity ? dL_dopacity[ray_idx] : 0.f, gD = dL_ddepth ? dL_ddepth[ray_idx] : 0.f;  // NULL: no gradient reached that output
    // pass 1: sum over the whole ray of dL_dws * ws  (volumerendering.cu:119-123)
    float part = 0.f;
    if (dL_dws)
        for (int c = lane; c < N; c += 64) part += dL_dws[start + c] * ws[start + c];
    const float dws_sum = nrc_group_sum<64>(part);
    // pass 2
    float carry = 1.0f, cr_ = 0.f, cg_ = 0.f, cb_ = 0.f, cd_ = 0.f, cs_ = 0.f;  // running prefix carries
    for (int c = 0; c < N; c += 64) {
        const int i = c + lane;
        const bool valid = i < N;
        const int64_t s = start + i;
        float a = 0.f, sr = 0.f, sg = 0.f, sb = 0.f, tt = 0.f, dl = 0.f, gw = 0.f, wsv = 0.f;
        if (valid) {
            dl = deltas[s];
            a = alpha_of(sigmas[s], dl);
            sr = rgbs[3 * s]; sg = rgbs[3 * s + 1]; sb = rgbs[3 * s + 2];
            tt = ts[s]; gw = dL_dws ? dL_dws[s] : 0.f; wsv = ws[s];
        }
        float Tb, Ta;
        chunk_transmittance<64>(a, lane, carry, Tb, Ta);
        const float w = a * Tb;
        const float r = cr_ + nrc_group_incl_sum<64>(w * sr, lane);
        const float g = cg_ + nrc_group_incl_sum<64>(w * sg, lane);
        const float b = cb_ + nrc_group_incl_sum<64>(w * sb, lane);
        const float d = cd_ + nrc_group_incl_sum<64>(w * tt, lane);
        const float sc = cs_ + nrc_group_incl_sum<64>(gw * wsv, lane);
        cr_ = __shfl(r, 63, 64); cg_ = __shfl(g, 63, 64); cb_ = __shfl(b, 63, 64); cd_ = __shfl(d, 63, 64); cs_ = __shfl(sc, 63, 64);
        const int fs = first_saturated<64>(valid && Ta <= thr, lane);
        if (valid && lane <= fs) {
            dL_drgbs[3 * s] = gR * w; dL_drgbs[3 * s + 1] = gG * w; dL_drgbs[3 * s + 2] = gB * w;
            dL_dsigmas[s] = dl * (gR * (sr * Ta - (R - r)) + gG * (sg * Ta - (Gc - g)) + gB * (sb * Ta - (B - b)) +
                                  gO * (1 - O) + gD * (tt * Ta - (D - d)) + Ta * gw - (dws_sum - sc));
        }
        if (fs < 64) break;
    }
}

// ------------------------------------------------------------------------------------------------ test forward
// G lanes per alive ray (G = smallest power of two >= N_samples, capped at 64): rows of N_samples samples.
template <int G>
__global__ void __launch_bounds__(256) k_composite_test_fw(const float* __restrict__ sigmas, const float* __restrict__ rgbs,
                                                           const float* __restrict__ deltas, const float* __restrict__ ts,
                                                           int64_t* __restrict__ alive, int64_t n_alive, int N_samples,
                                                           float thr, const int32_t* __restrict__ n_eff,
                                                           float* __restrict__ opacity, float* __restrict__ depth,
                                                           float* __restrict__ rgb) {
    const int lane = threadIdx.x & 63, gl = lane & (G - 1);
    const int64_t n = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
    // groups past the end still take part in the wave-wide ballots/shuffles below with N = 0
    const bool live = n < n_alive;
    const int N = live ? n_eff[n] : 0;
    const int64_t r = live ? alive[n] : 0;
    float carry = (live && N > 0) ? 1.0f - opacity[r] : 1.0f;
    float accR = 0.f, accG = 0.f, accB = 0.f, accD = 0.f, accO = 0.f;
    bool dead = false, done = N <= 0;
    // all groups of a wave iterate together until every group is done (uniform trip count for the cross-lane ops)
    const int maxN = [&] { int m = N; for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d, 64)); return m; }();
    for (int c = 0; c < maxN; c += G) {
        const int i = c + gl;
        const bool valid = !done && i < N;
        const int64_t s = n * N_samples + i;
        float a = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, tt = 0.f;
        if (valid) {
            a = alpha_of(sigmas[s], deltas[s]);
            cr = rgbs[3 * s]; cg = rgbs[3 * s + 1]; cb = rgbs[3 * s + 2];
            tt = ts[s];
        }
        float Tb, Ta;
        chunk_transmittance<G>(a, gl, carry, Tb, Ta);
        const int fs = first_saturated<G>(valid && Ta <= thr, lane);
        if (valid && gl <= fs) {
            const float w = a * Tb;
            accR += w * cr; accG += w * cg; accB += w * cb; accD += w * tt; accO += w;
        }
        if (fs < G && !done) { dead = true; done = true; }
        if (c + G >= N) done = true;
    }
    accR = nrc_group_sum<G>(accR); accG = nrc_group_sum<G>(accG); accB = nrc_group_sum<G>(accB);
    accD = nrc_group_sum<G>(accD); accO = nrc_group_sum<G>(accO);
    if (live && gl == 0) {
        if (N == 0) { alive[n] = -1; return; }  // no hit (:222-225)
        rgb[3 * r] += accR; rgb[3 * r + 1] += accG; rgb[3 * r + 2] += accB;
        depth[r] += accD; opacity[r] += accO;
        if (dead) alive[n] = -1;  // saturated (:243-246)
    }
}

// ------------------------------------------------------------------------------------------------ distortion loss
__global__ void __launch_bounds__(256) k_distortion_fw(const float* __restrict__ ws, const float* __restrict__ deltas,
                                                       const float* __restrict__ ts, const int64_t* __restrict__ rays_a,
                                                       int64_t n_rays, float* __restrict__ loss, float* __restrict__ ws_incl,
                                                       float* __restrict__ wts_incl) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_rays) return;
    const int64_t ray_idx = rays_a[3 * n], start = rays_a[3 * n + 1];
    const int N = (int)rays_a[3 * n + 2];
    float cw = 0.f, cwt = 0.f, acc = 0.f;
    for (int c = 0; c < N; c += 64) {
        const int i = c + lane;
        const bool valid = i < N;
        const int64_t s = start + i;
        const float w = valid ? ws[s] : 0.f, dl = valid ? deltas[s] : 0.f, wt = valid ? w * ts[s] : 0.f;
        const float wi = cw + nrc_group_incl_sum<64>(w, lane), wti = cwt + nrc_group_incl_sum<64>(wt, lane);
        float we = __shfl_up(wi, 1, 64), wte = __shfl_up(wti, 1, 64);
        if (lane == 0) { we = cw; wte = cwt; }
        if (valid) {
            ws_incl[s] = wi; wts_incl[s] = wti;
            acc += 2 * (wti * we - wi * wte) + 1.0f / 3 * w * w * dl;
        }
        cw = __shfl(wi, 63, 64); cwt = __shfl(wti, 63, 64);
    }
    acc = nrc_group_sum<64>(acc);
    if (lane == 0) loss[ray_idx] = acc;
}
__global__ void __launch_bounds__(256) k_distortion_bw(const float* __restrict__ dL_dloss, const float* __restrict__ ws_incl,
                                                       const float* __restrict__ wts_incl, const float* __restrict__ ws,
                                                       const float* __restrict__ deltas, const float* __restrict__ ts,
                                                       const int64_t* __restrict__ rays_a, int64_t n_rays,
                                                       float* __restrict__ dL_dws) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_rays) return;
    const int64_t ray_idx = rays_a[3 * n], start = rays_a[3 * n + 1];
    const int N = (int)rays_a[3 * n + 2];
    if (N <= 0) return;
    const int64_t end = start + N - 1;
    const float ws_sum = ws_incl[end], wts_sum = wts_incl[end], g = dL_dloss[ray_idx];
    for (int64_t s = start + lane; s <= end; s += 64) {
        const float t = ts[s];
        const float prev = s == start ? 0.0f : (t * ws_incl[s - 1] - wts_incl[s - 1]);
        float v = g * 2 * (prev + (wts_sum - wts_incl[s] - t * (ws_sum - ws_incl[s])));
        v += g * 2.0f / 3 * ws[s] * deltas[s];
        dL_dws[s] = v;
    }
}


// ------------------------------------------------------------------------------------------------ image compositor (fused pipeline)
// Tile-interleaved layout (see ngp_march.hip): one wave per 8x8 pixel tile, lane = ray.  Row k holds the k-th sample of the
// 64 rays: 512-byte (packed values) + 256-byte (t) coalesced reads per row.  Each lane composites ITS ray serially, in the
// reference's order, with the inference early-out (volumerendering.cu:205-249: a ray dies after compositing the sample
// that brings T <= threshold) and the finalisation of render_rays_inference (Renderer.py:133-138).  sigma = exp(h0)
// (TruncExp), dt re-derived from t with the test kernel's step rule (raymarching.cu:370).
__global__ void __launch_bounds__(256) k_composite_image(const __half* __restrict__ packed, const float* __restrict__ ts,
                                                         const int32_t* __restrict__ ray_cnt, const int32_t* __restrict__ tile_off,
                                                         int width, int height, int tiles_x, int64_t tile_begin, int64_t n_tiles,
                                                         float esf, float dt_min, float dt_max, float thr, float bg_r, float bg_g,
                                                         float bg_b, float* __restrict__ rgb, float* __restrict__ alpha_out,
                                                         float* __restrict__ depth_out, int64_t row_cap, int arena_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t lt = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lt >= n_tiles) return;
    const int64_t tile = tile_begin + lt;
    const int px = (int)(tile % tiles_x) * NRC_TILE_W + (lane & (NRC_TILE_W - 1)), py = (int)(tile / tiles_x) * NRC_TILE_H + (lane >> NRC_TILE_W_LOG2);
    const bool inside = px < width && py < height;
    const int64_t row0 = tile_off[lt];
    // row_cap (fixed-capacity frames): rows behind it were never written -- an overflowed frame reads nothing out of bounds (its picture is discarded)
    const int N = inside ? (int)min((int64_t)ray_cnt[lt * 64 + lane], max(row_cap - row0, (int64_t)0)) : 0;
    float T = 1.0f, accR = 0.f, accG = 0.f, accB = 0.f, accD = 0.f, accO = 0.f;
    bool alive = N > 0;
    // several rows per turn, their loads issued together (a row's loads depend on nothing but k): the serial per-lane loop was bound by
    // one exposed load latency per sample
#ifndef NRC_COMPOSITE_ROWS
#define NRC_COMPOSITE_ROWS 8   // measured per 800x800 image: 4 rows 218 us, 8 rows 202, 16 rows 210
#endif
    enum { CU = NRC_COMPOSITE_ROWS };
    for (int k = 0; __any(alive); k += CU) {
        uint2 raw[CU];
        float tv[CU];
#pragma unroll
        for (int u = 0; u < CU; u++) {
            const bool in = alive && k + u < N;
            const int64_t s = (row0 + (in ? k + u : 0)) * 64 + lane;
            raw[u] = in ? *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(packed) + 4 * s) : make_uint2(0u, 0u);
            // arena_rows > 0: `ts` is the count pass's arena, sample k of this tile in its row lt * arena_rows + k
            tv[u] = in ? ts[arena_rows > 0 ? ((lt * arena_rows + k + u) << 6) + lane : s] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < CU; u++) {
            if (alive && k + u < N) {
                const float2 f01 = __half22float2(*reinterpret_cast<const __half2*>(&raw[u].x));
                const float2 f23 = __half22float2(*reinterpret_cast<const __half2*>(&raw[u].y));
                const float t = tv[u];
                const float dt = fmaxf(dt_min, fminf(t * esf, dt_max));
                const float a = alpha_of(expf(f01.x), dt);
                const float w = a * T;
                accR += w * f01.y; accG += w * f23.x; accB += w * f23.y; accD += w * t; accO += w;
                T *= 1.0f - a;
                if (T <= thr || k + u + 1 >= N) alive = false;
            }
        }
    }
    if (inside) {
        const int64_t n = (int64_t)py * width + px;
        const float al = fminf(fmaxf(accO, 0.f), 1.f);
        const float Tr = 1.f - al;
        rgb[3 * n] = fminf(fmaxf(accR + Tr * bg_r, 0.f), 1.f);
        rgb[3 * n + 1] = fminf(fmaxf(accG + Tr * bg_g, 0.f), 1.f);
        rgb[3 * n + 2] = fminf(fmaxf(accB + Tr * bg_b, 0.f), 1.f);
        alpha_out[n] = al;
        depth_out[n] = Tr < 1.0f ? accD / al : 0.0f;
    }
}

// up to five output arrays cleared by ONE launch (five hipMemsetAsync calls are five launches of host time in a ~90-launch training step)
struct ZeroList { uint32_t* p[5]; int64_t words[5]; };
__global__ void __launch_bounds__(256) k_zero_multi(ZeroList z) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int a = 0; a < 5; a++)
        if (i < z.words[a]) z.p[a][i] = 0u;
}

// ---- layer-major image compositing: one call per chunk of rows, front to back ------------------------------------------------
// State per ray (T and the five running sums), per tile the next layer k to consume and an alive flag.  A tile is finished when
// none of its rays is alive (saturated or out of samples) or its rows are exhausted; it then writes its pixels once and its flag
// drops, which turns its remaining rows into holes for the encode / MLP kernels of the following chunks.
__global__ void __launch_bounds__(256) k_layers_init(int64_t n_tiles, const int32_t* __restrict__ ray_cnt, float* __restrict__ state,
                                                     uint8_t* __restrict__ ray_alive, int32_t* __restrict__ next_k, uint8_t* __restrict__ tile_alive,
                                                     int32_t* __restrict__ skipped_rows) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q == 0 && skipped_rows) *skipped_rows = 0;
    if (q >= n_tiles * 64) return;
    state[q] = 1.0f;  // T; the sums follow as planes of n_tiles * 64 floats
#pragma unroll
    for (int c = 1; c < 6; c++) state[c * n_tiles * 64 + q] = 0.f;
    ray_alive[q] = ray_cnt[q] > 0;
    if ((q & 63) == 0) { next_k[q >> 6] = 0; tile_alive[q >> 6] = 1; }
}
__global__ void __launch_bounds__(256) k_composite_layers(const __half* __restrict__ packed, const float* __restrict__ ts,
                                                          const int32_t* __restrict__ ray_cnt, const int32_t* __restrict__ tile_rows,
                                                          const int32_t* __restrict__ tile_off, const int32_t* __restrict__ row_of, int32_t* __restrict__ row_tile,
                                                          int64_t row_end,
                                                          int width, int height, int tiles_x, int64_t tile_begin, int64_t n_tiles, float esf,
                                                          float dt_min, float dt_max, float thr, float bg_r, float bg_g, float bg_b,
                                                          float* __restrict__ state, uint8_t* __restrict__ ray_alive, int32_t* __restrict__ next_k,
                                                          uint8_t* __restrict__ tile_alive, float* __restrict__ rgb, float* __restrict__ alpha_out,
                                                          float* __restrict__ depth_out, int32_t* __restrict__ skipped_rows, int arena_rows) {
    const int lane = threadIdx.x & 63;
    const int64_t lt = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lt >= n_tiles || !tile_alive[lt]) return;
    const int R = tile_rows[lt];
    const int64_t base = tile_off[lt], q = lt * 64 + lane, plane = n_tiles * 64;
    int k = next_k[lt];
    if (k < R && (int64_t)row_of[base + k] >= row_end) return;  // nothing of this tile in the chunk
    const int N = ray_cnt[q];
    float T = state[q], accR = state[plane + q], accG = state[2 * plane + q], accB = state[3 * plane + q], accD = state[4 * plane + q],
          accO = state[5 * plane + q];
    bool alive = ray_alive[q] != 0;
    // The rows of a slab, EIGHT at a time with their loads issued together (round 4: one row per turn was a chain of up to sixteen dependent round
    // trips per slab, 62-75 us per launch); the arithmetic walks them in order, stops where the one-row loop stopped (row behind the chunk, all
    // lanes finished, last row of the tile) and leaves the same k.  arena_rows > 0: `ts` is the count pass's arena (sample k of this tile in its
    // row lt * arena_rows + k); `packed` stays indexed by the slab-major rows.
    enum { CU = 8 };
    bool behind = false;
    while (k < R && !behind && __any(alive)) {
        int64_t rows[CU];
        uint2 raw[CU];
        float tv[CU];
#pragma unroll
        for (int u = 0; u < CU; u++) rows[u] = k + u < R ? (int64_t)row_of[base + k + u] : INT64_MAX;
#pragma unroll
        for (int u = 0; u < CU; u++) {
            const bool in = alive && k + u < N && rows[u] < row_end;
            const int64_t s = rows[u] * 64 + lane;
            raw[u] = in ? *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(packed) + 4 * s) : make_uint2(0u, 0u);
            tv[u] = in ? ts[arena_rows > 0 ? ((lt * arena_rows + k + u) << 6) + lane : s] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < CU; u++) {
            if (k >= R || !__any(alive)) break;
            if (rows[u] >= row_end) { behind = true; break; }
            if (alive) {
                const float2 f01 = __half22float2(*reinterpret_cast<const __half2*>(&raw[u].x));
                const float2 f23 = __half22float2(*reinterpret_cast<const __half2*>(&raw[u].y));
                const float t = tv[u];
                const float dt = fmaxf(dt_min, fminf(t * esf, dt_max));
                const float a = alpha_of(expf(f01.x), dt);
                const float w = a * T;
                accR += w * f01.y; accG += w * f23.x; accB += w * f23.y; accD += w * t; accO += w;
                T *= 1.0f - a;
                if (T <= thr || k + 1 >= N) alive = false;
            }
            k++;
        }
    }
    const bool finished = !__any(alive) || k >= R;
    if (!finished) {
        state[q] = T; state[plane + q] = accR; state[2 * plane + q] = accG; state[3 * plane + q] = accB; state[4 * plane + q] = accD;
        state[5 * plane + q] = accO;
        ray_alive[q] = alive;
        if (lane == 0) next_k[lt] = k;
        return;
    }
    if (lane == 0) {
        tile_alive[lt] = 0;
        if (skipped_rows && k < R) atomicAdd(skipped_rows, R - k);  // rows no kernel will touch any more (statistics for the caller)
    }
    // the tile's remaining rows (all in later slabs) are marked as finished where the query kernels look first: row_tile = -1 - tile
    if (row_tile)
        for (int kk = k + lane; kk < R; kk += 64) row_tile[row_of[base + kk]] = -1 - (int32_t)lt;
    const int64_t tile = tile_begin + lt;
    const int px = (int)(tile % tiles_x) * NRC_TILE_W + (lane & (NRC_TILE_W - 1)), py = (int)(tile / tiles_x) * NRC_TILE_H + (lane >> NRC_TILE_W_LOG2);
    if (px < width && py < height) {
        const int64_t n = (int64_t)py * width + px;
        const float al = fminf(fmaxf(accO, 0.f), 1.f);
        const float Tr = 1.f - al;
        rgb[3 * n] = fminf(fmaxf(accR + Tr * bg_r, 0.f), 1.f);
        rgb[3 * n + 1] = fminf(fmaxf(accG + Tr * bg_g, 0.f), 1.f);
        rgb[3 * n + 2] = fminf(fmaxf(accB + Tr * bg_b, 0.f), 1.f);
        alpha_out[n] = al;
        depth_out[n] = Tr < 1.0f ? accD / al : 0.0f;
    }
}

}  // namespace

// ---- training pixels: what InstantNGPRenderer.render_rays_training does with the composited sums (Renderer.py:80-84) as one launch each way ----
// rgb_out = rgb + (1 - opacity) * bg,  depth_out = depth / (opacity + 1e-6)  (same operation order as the torch expressions they replace)
__global__ void __launch_bounds__(256) k_train_pixels_fw(int64_t n, const float* __restrict__ opacity, const float* __restrict__ depth,
                                                         const float* __restrict__ rgb, const float* __restrict__ bg, float* __restrict__ rgb_out,
                                                         float* __restrict__ depth_out) {
#pragma clang fp contract(off)   // rgb + (1 - a) * bg with two roundings, like the torch expression (this file is otherwise built with contraction on)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float see_through = 1 - opacity[i];
#pragma unroll
    for (int c = 0; c < 3; c++) rgb_out[3 * i + c] = rgb[3 * i + c] + see_through * bg[c];
    depth_out[i] = depth[i] / (opacity[i] + 1e-6f);
}
__global__ void __launch_bounds__(256) k_train_pixels_bw(int64_t n, const float* __restrict__ g_rgb, const float* __restrict__ g_alpha,
                                                         const float* __restrict__ g_depth, const float* __restrict__ opacity,
                                                         const float* __restrict__ depth, const float* __restrict__ bg, float* __restrict__ dL_dopacity,
                                                         float* __restrict__ dL_ddepth) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float gO = g_alpha ? g_alpha[i] : 0.f, gD = 0.f;
    if (g_rgb) gO -= g_rgb[3 * i] * bg[0] + g_rgb[3 * i + 1] * bg[1] + g_rgb[3 * i + 2] * bg[2];
    if (g_depth) {
        const float den = opacity[i] + 1e-6f;
        gD = g_depth[i] / den;
        gO -= g_depth[i] * depth[i] / (den * den);
    }
    dL_dopacity[i] = gO;
    dL_ddepth[i] = gD;
}

// colour loss of a training batch with the GradScaler's multiplication folded in (see the header): ONE workgroup, fixed order
__global__ void __launch_bounds__(1024) k_mse_scaled_fw(int64_t n, const float* __restrict__ pred, const float* __restrict__ target,
                                                        const float* __restrict__ scale, float* __restrict__ out2) {
    __shared__ float part[16];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) { const float d = pred[i] - target[i]; acc += d * d; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 16; w++) tot += part[w];
        const float mean = tot / (float)n;
        out2[0] = mean;
        out2[1] = mean * (scale ? scale[0] : 1.f);
    }
}
// InstantNGPLoss.forward (Loss.py:18-26: MSE of the colours + 0.5e-6 * mean squared MLP weight) as ONE launch: a workgroup, both sums in a fixed order.
// out3 = (mse + wd_weight * wd, mse, wd)
__global__ void __launch_bounds__(1024) k_ngp_loss_fw(int64_t n, const float* __restrict__ pred, const float* __restrict__ target, const float* __restrict__ wa,
                                                      int64_t na, const float* __restrict__ wb, int64_t nb, float inv_n_weights, float wd_weight,
                                                      float* __restrict__ out3) {
    __shared__ float part[2][16];
    float acc = 0.f, sq = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) { const float d = pred[i] - target[i]; acc += d * d; }
    for (int64_t i = threadIdx.x; i < na; i += 1024) sq += wa[i] * wa[i];
    for (int64_t i = threadIdx.x; i < nb; i += 1024) sq += wb[i] * wb[i];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { acc += __shfl_xor(acc, d, 64); sq += __shfl_xor(sq, d, 64); }
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = acc; part[1][threadIdx.x >> 6] = sq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int w = 0; w < 16; w++) { t0 += part[0][w]; t1 += part[1][w]; }
        const float mse = t0 / (float)n, wd = t1 * inv_n_weights;
        out3[0] = mse + wd_weight * wd; out3[1] = mse; out3[2] = wd;
    }
}
__global__ void __launch_bounds__(256) k_mse_scaled_bw(int64_t n, const float* __restrict__ pred, const float* __restrict__ target,
                                                       const float* __restrict__ scale, const float* __restrict__ g_loss,
                                                       const float* __restrict__ g_scaled, float* __restrict__ grad_pred) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float up = (g_loss ? g_loss[0] : 0.f) + (g_scaled ? g_scaled[0] * (scale ? scale[0] : 1.f) : 0.f);
    grad_pred[i] = 2.f / (float)n * (pred[i] - target[i]) * up;
}

// ---- compositing + colour loss of a fused training iteration, forward AND backward, one launch (include/nerficg_hip.h group 13) -------------------
// Renderer.py:78-84 (compositing, rgb + (1 - alpha) bg), Loss.py:15-22 (mean squared error against the ray colours), Trainer.py:88 (times the
// GradScaler's scale) and the way back to dL/dsigma, dL/drgb of every sample: a wave owns a ray, walks it front to back for the sums, turns them into
// the pixel, the squared error and the pixel's gradient in registers, and walks the ray again for the sample gradients (the second walk hits L2: a
// ray is ~120 samples x 28 B).  Same expressions as k_composite_train_fw -> k_train_pixels_fw -> k_mse_scaled_fw / _bw -> k_train_pixels_bw ->
// k_composite_train_bw, which it replaces together with their three clearing launches (nine launches of the recorded iteration of round 4).
// The loss is summed in a fixed order by the last workgroup to finish (per-ray partials, one ticket), so it is reproducible run to run.
// Extra workgroups behind the rays' clear what the backward pass of the networks expects cleared: the sample gradients of the rows no ray owns
// and two caller-given float ranges (the parts of the parameter gradients that are accumulated with atomics).
struct TrainLoss {
    const float *sigmas, *rgbs, *deltas, *ts;
    const int64_t* rays_a;
    const int32_t* counter;   // [0] marched samples (uncut), [1] live rays
    const float *bg, *target, *scale;
    float thr;
    int64_t n_rays, n_samples;   // capacities
    float *ray_rgb, *ray_alpha, *ray_depth, *dL_dsigmas, *dL_drgbs, *partial, *loss2;
    uint32_t* ticket;
    float* zero0; int64_t n_zero0; float* zero1; int64_t n_zero1;
};
__global__ void __launch_bounds__(256) k_train_composite_loss(TrainLoss p) {
    __shared__ float part_s[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ray_blocks = (p.n_rays + 3) / 4;
    if ((int64_t)blockIdx.x >= ray_blocks) {   // clearing workgroups (grid-stride over the three ranges)
        const int64_t stride = ((int64_t)gridDim.x - ray_blocks) * 256;
        const int64_t i0 = ((int64_t)blockIdx.x - ray_blocks) * 256 + threadIdx.x;
        const int64_t used = min((int64_t)p.counter[0], p.n_samples);
        for (int64_t i = used + i0; i < p.n_samples; i += stride) { p.dL_dsigmas[i] = 0.f; p.dL_drgbs[3 * i] = 0.f; p.dL_drgbs[3 * i + 1] = 0.f; p.dL_drgbs[3 * i + 2] = 0.f; }
        for (int64_t i = i0; i < p.n_zero0; i += stride) p.zero0[i] = 0.f;
        for (int64_t i = i0; i < p.n_zero1; i += stride) p.zero1[i] = 0.f;
        return;
    }
    const int64_t n = (int64_t)blockIdx.x * 4 + wave;
    const int64_t n_live = p.counter[1];
    if (n < p.n_rays) {
        const int64_t ray_idx = p.rays_a[3 * n], start = p.rays_a[3 * n + 1];
        const int N = (int)p.rays_a[3 * n + 2];
        // ---- forward walk (k_composite_train_fw)
        float carry = 1.0f, accR = 0.f, accG = 0.f, accB = 0.f, accD = 0.f, accO = 0.f;
        for (int c = 0; c < N; c += 64) {
            const int i = c + lane;
            const bool valid = i < N;
            const int64_t s = start + i;
            float a = 0.f, cr = 0.f, cg = 0.f, cb = 0.f, tt = 0.f;
            if (valid) {
                a = alpha_of(p.sigmas[s], p.deltas[s]);
                cr = p.rgbs[3 * s]; cg = p.rgbs[3 * s + 1]; cb = p.rgbs[3 * s + 2];
                tt = p.ts[s];
            }
            float Tb, Ta;
            chunk_transmittance<64>(a, lane, carry, Tb, Ta);
            const int fs = first_saturated<64>(valid && Ta <= p.thr, lane);
            if (valid && lane <= fs) {
                const float w = a * Tb;
                accR += w * cr; accG += w * cg; accB += w * cb; accD += w * tt; accO += w;
            }
            if (fs < 64) break;
        }
        const float R = nrc_group_sum<64>(accR), Gc = nrc_group_sum<64>(accG), B = nrc_group_sum<64>(accB);
        const float D = nrc_group_sum<64>(accD), O = nrc_group_sum<64>(accO);
        // ---- the pixel, its squared error and its gradient (k_train_pixels_fw, k_mse_scaled_fw / _bw, k_train_pixels_bw)
        const bool live = ray_idx < n_live;
        const float bg0 = p.bg[0], bg1 = p.bg[1], bg2 = p.bg[2];
        float pr, pg, pb;
        {
#pragma clang fp contract(off)   // rgb + (1 - a) * bg with two roundings, like the torch expression
            const float see_through = 1 - O;
            pr = R + see_through * bg0; pg = Gc + see_through * bg1; pb = B + see_through * bg2;
        }
        const float n_el = (float)max((int64_t)1, 3 * n_live);
        const float up = p.scale ? p.scale[0] : 1.f;
        float gR = 0.f, gG = 0.f, gB = 0.f, sq = 0.f;
        if (live) {
            const float dr = pr - p.target[3 * ray_idx], dg = pg - p.target[3 * ray_idx + 1], db = pb - p.target[3 * ray_idx + 2];
            sq = dr * dr + dg * dg + db * db;
            gR = 2.f / n_el * dr * up; gG = 2.f / n_el * dg * up; gB = 2.f / n_el * db * up;
        }
        const float gO = 0.f - (gR * bg0 + gG * bg1 + gB * bg2);
        if (lane == 0) {
            if (p.ray_rgb) { p.ray_rgb[3 * ray_idx] = pr; p.ray_rgb[3 * ray_idx + 1] = pg; p.ray_rgb[3 * ray_idx + 2] = pb; }
            if (p.ray_alpha) p.ray_alpha[ray_idx] = O;
            if (p.ray_depth) p.ray_depth[ray_idx] = D / (O + 1e-6f);
            __hip_atomic_store(&p.partial[ray_idx], sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- backward walk (k_composite_train_bw without dL/dws and dL/ddepth); rows behind the saturating sample get zeros
        float cr_ = 0.f, cg_ = 0.f, cb_ = 0.f;
        carry = 1.0f;
        bool open = true;
        for (int c = 0; c < N; c += 64) {
            const int i = c + lane;
            const bool valid = i < N;
            const int64_t s = start + i;
            if (!open) {
                if (valid) { p.dL_dsigmas[s] = 0.f; p.dL_drgbs[3 * s] = 0.f; p.dL_drgbs[3 * s + 1] = 0.f; p.dL_drgbs[3 * s + 2] = 0.f; }
                continue;
            }
            float a = 0.f, sr = 0.f, sg = 0.f, sb = 0.f, dl = 0.f;
            if (valid) {
                dl = p.deltas[s];
                a = alpha_of(p.sigmas[s], dl);
                sr = p.rgbs[3 * s]; sg = p.rgbs[3 * s + 1]; sb = p.rgbs[3 * s + 2];
            }
            float Tb, Ta;
            chunk_transmittance<64>(a, lane, carry, Tb, Ta);
            const float w = a * Tb;
            const float r = cr_ + nrc_group_incl_sum<64>(w * sr, lane);
            const float g = cg_ + nrc_group_incl_sum<64>(w * sg, lane);
            const float b = cb_ + nrc_group_incl_sum<64>(w * sb, lane);
            cr_ = __shfl(r, 63, 64); cg_ = __shfl(g, 63, 64); cb_ = __shfl(b, 63, 64);
            const int fs = first_saturated<64>(valid && Ta <= p.thr, lane);
            if (valid) {
                const bool in = lane <= fs;
                p.dL_drgbs[3 * s] = in ? gR * w : 0.f; p.dL_drgbs[3 * s + 1] = in ? gG * w : 0.f; p.dL_drgbs[3 * s + 2] = in ? gB * w : 0.f;
                p.dL_dsigmas[s] = in ? dl * (gR * (sr * Ta - (R - r)) + gG * (sg * Ta - (Gc - g)) + gB * (sb * Ta - (B - b)) + gO * (1 - O)) : 0.f;
            }
            if (fs < 64) open = false;
        }
    }
    // ---- the last workgroup of the rays' sums the per-ray squared errors in a fixed order
    if (!nrc_last_workgroup(p.ticket, blockIdx.x, (uint32_t)ray_blocks)) return;
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < p.n_rays; i += 256) acc += __hip_atomic_load(&p.partial[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    acc = nrc_group_sum<64>(acc);
    if (lane == 0) part_s[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = part_s[0] + part_s[1] + part_s[2] + part_s[3];
        const float mean = tot / (float)max((int64_t)1, 3 * n_live);
        p.loss2[0] = mean;
        p.loss2[1] = mean * (p.scale ? p.scale[0] : 1.f);
    }
}

static void zero_arrays(hipStream_t s, void* p0, int64_t b0, void* p1 = nullptr, int64_t b1 = 0, void* p2 = nullptr, int64_t b2 = 0, void* p3 = nullptr,
                        int64_t b3 = 0, void* p4 = nullptr, int64_t b4 = 0) {
    ZeroList z;
    void* ps[5] = {p0, p1, p2, p3, p4};
    const int64_t bs[5] = {b0, b1, b2, b3, b4};
    int64_t mx = 0;
    for (int a = 0; a < 5; a++) { z.p[a] = (uint32_t*)ps[a]; z.words[a] = ps[a] ? bs[a] / 4 : 0; mx = z.words[a] > mx ? z.words[a] : mx; }
    if (mx > 0) hipLaunchKernelGGL(k_zero_multi, dim3((unsigned)nrc_cdiv(mx, 256)), dim3(256), 0, s, z);
}

// internal launchers used by nrc_ngp_render_layers (ngp_net.hip)
void nrc_launch_layers_init(int64_t n_tiles, const int32_t* ray_cnt, float* state, uint8_t* ray_alive, int32_t* next_k, uint8_t* tile_alive,
                            int32_t* skipped_rows, hipStream_t s) {
    hipLaunchKernelGGL(k_layers_init, dim3((unsigned)nrc_cdiv(n_tiles * 64, 256)), dim3(256), 0, s, n_tiles, ray_cnt, state, ray_alive, next_k, tile_alive,
                       skipped_rows);
}
void nrc_launch_composite_layers(const void* packed, const float* ts, const int32_t* ray_cnt, const int32_t* tile_rows, const int32_t* tile_off,
                                 const int32_t* row_of, int32_t* row_tile, int64_t row_end, int width, int height, int64_t tile_begin, int64_t n_tiles, int cascades,
                                 float esf, int grid_size, int max_samples, float thr, const float* bg3, float* state, uint8_t* ray_alive, int32_t* next_k,
                                 uint8_t* tile_alive, float* rgb, float* alpha, float* depth, int32_t* skipped_rows, int arena_rows, hipStream_t s) {
    const int tiles_x = (width + NRC_TILE_W - 1) / NRC_TILE_W;
    const float dt_min = 1.73205080757f / max_samples, dt_max = 1.73205080757f * 2 * (float)cascades / grid_size;
    hipLaunchKernelGGL(k_composite_layers, dim3((unsigned)nrc_cdiv(n_tiles, 4)), dim3(256), 0, s, (const __half*)packed, ts, ray_cnt, tile_rows, tile_off,
                       row_of, row_tile, row_end, width, height, tiles_x, tile_begin, n_tiles, esf, dt_min, dt_max, thr, bg3[0], bg3[1], bg3[2], state, ray_alive,
                       next_k, tile_alive, rgb, alpha, depth, skipped_rows, arena_rows);
}

extern "C" {

int nrc_raymarching_train_bw(const float* dL_dxyzs, const float* dL_ddirs, const float* ts, const int64_t* rays_a, int64_t n_rays, int64_t n_samples,
                             float* dL_drays_o, float* dL_drays_d, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_rays < 0 || n_samples < 0) return NRC_ERR_INVALID;
    if (n_rays == 0) return NRC_OK;
    if (!rays_a || !dL_drays_o || !dL_drays_d || (n_samples > 0 && (!dL_dxyzs || !ts))) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_march_train_bw, dim3(nrc_cdiv(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, dL_dxyzs, dL_ddirs, ts, rays_a, n_rays, dL_drays_o,
                       dL_drays_d);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_composite_train_fw(const float* sigmas, const float* rgbs, const float* deltas, const float* ts, const int64_t* rays_a,
                           int64_t n_rays, int64_t n_samples, float T_threshold, int64_t* total_samples, float* opacity,
                           float* depth, float* rgb, float* ws, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_rays < 0 || n_samples < 0) return NRC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (n_samples > 0 && !ws) return NRC_ERR_INVALID;
    if (n_rays == 0) { if (n_samples > 0) zero_arrays(s, ws, n_samples * 4); return NRC_OK; }
    if (!rays_a || !total_samples || !opacity || !depth || !rgb || (n_samples > 0 && (!sigmas || !rgbs || !deltas || !ts))) return NRC_ERR_INVALID;
    // rows of rays_a name their output slot (ray_idx); slots never named keep the reference's zero initialisation
    zero_arrays(s, n_samples > 0 ? ws : nullptr, n_samples * 4, total_samples, n_rays * 8, opacity, n_rays * 4, depth, n_rays * 4, rgb, n_rays * 12);
    hipLaunchKernelGGL(k_composite_train_fw, dim3(nrc_cdiv(n_rays, 4)), dim3(256), 0, s, sigmas, rgbs, deltas, ts, rays_a, n_rays,
                       T_threshold, total_samples, opacity, depth, rgb, ws);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_composite_train_bw(const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb, const float* dL_dws,
                           const float* sigmas, const float* rgbs, const float* ws, const float* deltas, const float* ts,
                           const int64_t* rays_a, const float* opacity, const float* depth, const float* rgb, int64_t n_rays,
                           int64_t n_samples, float T_threshold, float* dL_dsigmas, float* dL_drgbs, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_rays < 0 || n_samples < 0) return NRC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (n_samples > 0) {
        if (!dL_dsigmas || !dL_drgbs) return NRC_ERR_INVALID;
        zero_arrays(s, dL_dsigmas, n_samples * 4, dL_drgbs, n_samples * 12);
    }
    if (n_rays == 0 || n_samples == 0) return NRC_OK;
    if (!dL_drgb || !sigmas || !rgbs || !ws || !deltas || !ts || !rays_a || !opacity || !depth || !rgb) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_composite_train_bw, dim3(nrc_cdiv(n_rays, 4)), dim3(256), 0, s, dL_dopacity, dL_ddepth, dL_drgb, dL_dws,
                       sigmas, rgbs, ws, deltas, ts, rays_a, opacity, depth, rgb, n_rays, T_threshold, dL_dsigmas, dL_drgbs);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_ngp_train_pixels_fw(int64_t n_rays, const float* opacity, const float* depth, const float* rgb, const float* bg_dev, float* rgb_out,
                            float* depth_out, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_rays < 0) return NRC_ERR_INVALID;
    if (n_rays == 0) return NRC_OK;
    if (!opacity || !depth || !rgb || !bg_dev || !rgb_out || !depth_out) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_train_pixels_fw, dim3(nrc_cdiv(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, n_rays, opacity, depth, rgb, bg_dev, rgb_out,
                       depth_out);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_ngp_train_pixels_bw(int64_t n_rays, const float* g_rgb, const float* g_alpha, const float* g_depth, const float* opacity, const float* depth,
                            const float* bg_dev, float* dL_dopacity, float* dL_ddepth, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_rays < 0) return NRC_ERR_INVALID;
    if (n_rays == 0) return NRC_OK;
    if (!opacity || !depth || !bg_dev || !dL_dopacity || !dL_ddepth) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_train_pixels_bw, dim3(nrc_cdiv(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, n_rays, g_rgb, g_alpha, g_depth, opacity, depth,
                       bg_dev, dL_dopacity, dL_ddepth);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_mse_scaled_forward(int64_t n, const float* pred, const float* target, const float* scale, float* out2, nrc_stream_t stream) {
    NRC_ENTER();
    if (n <= 0 || n > (int64_t(1) << 24) || !pred || !target || !out2) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_mse_scaled_fw, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, pred, target, scale, out2);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_ngp_loss_forward(int64_t n, const float* pred, const float* target, const float* weights_a, int64_t n_a, const float* weights_b, int64_t n_b,
                         float inv_n_weights, float weight_decay_weight, float* out3, nrc_stream_t stream) {
    NRC_ENTER();
    if (n <= 0 || n > (int64_t(1) << 24) || n_a < 0 || n_b < 0 || !pred || !target || (n_a && !weights_a) || (n_b && !weights_b) || !out3) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_ngp_loss_fw, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, pred, target, weights_a, n_a, weights_b, n_b, inv_n_weights, weight_decay_weight, out3);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_mse_scaled_backward(int64_t n, const float* pred, const float* target, const float* scale, const float* g_loss, const float* g_scaled,
                            float* grad_pred, nrc_stream_t stream) {
    NRC_ENTER();
    if (n <= 0 || n > (int64_t(1) << 24) || !pred || !target || !grad_pred) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_mse_scaled_bw, dim3((unsigned)nrc_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n, pred, target, scale, g_loss, g_scaled,
                       grad_pred);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int64_t nrc_ngp_train_loss_ws_bytes(int64_t ray_capacity) { return ray_capacity < 0 ? NRC_ERR_INVALID : NRC_TICKET_WORDS * 4 + ray_capacity * 4; }
int nrc_ngp_train_loss(const float* sigmas, const float* rgbs, const float* deltas, const float* ts, const int64_t* rays_a, const int32_t* counter,
                       int64_t ray_capacity, int64_t sample_capacity, float T_threshold, const float* bg_dev, const float* target_rgb,
                       const float* loss_scale_dev, float* ray_rgb, float* ray_alpha, float* ray_depth, float* loss2, float* dL_dsigmas, float* dL_drgbs,
                       float* zero_a, int64_t n_zero_a, float* zero_b, int64_t n_zero_b, void* workspace, nrc_stream_t stream) {
    NRC_ENTER();
    if (ray_capacity < 1 || sample_capacity < 1 || n_zero_a < 0 || n_zero_b < 0) return NRC_ERR_INVALID;
    if (!sigmas || !rgbs || !deltas || !ts || !rays_a || !counter || !bg_dev || !target_rgb || !loss2 || !dL_dsigmas || !dL_drgbs || !workspace ||
        (n_zero_a && !zero_a) || (n_zero_b && !zero_b))
        return NRC_ERR_INVALID;
    TrainLoss p;
    p.sigmas = sigmas; p.rgbs = rgbs; p.deltas = deltas; p.ts = ts; p.rays_a = rays_a; p.counter = counter; p.bg = bg_dev; p.target = target_rgb;
    p.scale = loss_scale_dev; p.thr = T_threshold; p.n_rays = ray_capacity; p.n_samples = sample_capacity; p.ray_rgb = ray_rgb; p.ray_alpha = ray_alpha;
    p.ray_depth = ray_depth; p.dL_dsigmas = dL_dsigmas; p.dL_drgbs = dL_drgbs; p.ticket = (uint32_t*)workspace; p.partial = (float*)((char*)workspace + NRC_TICKET_WORDS * 4);
    p.loss2 = loss2; p.zero0 = zero_a; p.n_zero0 = n_zero_a; p.zero1 = zero_b; p.n_zero1 = n_zero_b;
    const int64_t big = n_zero_a > n_zero_b ? n_zero_a : n_zero_b;
    int64_t clear_blocks = nrc_cdiv(big > sample_capacity ? big : sample_capacity, 256 * 4);
    if (clear_blocks < 1) clear_blocks = 1;
    if (clear_blocks > 1024) clear_blocks = 1024;
    hipStream_t s = (hipStream_t)stream;
    NRC_STAGE(s, nullptr);
    hipLaunchKernelGGL(k_train_composite_loss, dim3((unsigned)(nrc_cdiv(ray_capacity, 4) + clear_blocks)), dim3(256), 0, s, p);
    NRC_STAGE(s, "k_train_composite_loss");
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_composite_test_fw(const float* sigmas, const float* rgbs, const float* deltas, const float* ts, int64_t* alive,
                          int64_t n_alive, int32_t N_samples, float T_threshold, const int32_t* n_eff, float* opacity,
                          float* depth, float* rgb, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_alive < 0 || N_samples < 1) return NRC_ERR_INVALID;
    if (n_alive == 0) return NRC_OK;
    if (!sigmas || !rgbs || !deltas || !ts || !alive || !n_eff || !opacity || !depth || !rgb) return NRC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
#define NRC_CT(G)                                                                                                          \
    hipLaunchKernelGGL(k_composite_test_fw<G>, dim3(nrc_cdiv(n_alive * G, 256)), dim3(256), 0, s, sigmas, rgbs, deltas, ts, \
                       alive, n_alive, N_samples, T_threshold, n_eff, opacity, depth, rgb)
    if (N_samples <= 1) NRC_CT(1);
    else if (N_samples <= 2) NRC_CT(2);
    else if (N_samples <= 4) NRC_CT(4);
    else if (N_samples <= 8) NRC_CT(8);
    else if (N_samples <= 16) NRC_CT(16);
    else if (N_samples <= 32) NRC_CT(32);
    else NRC_CT(64);
#undef NRC_CT
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_distortion_loss_fw(const float* ws, const float* deltas, const float* ts, const int64_t* rays_a, int64_t n_rays,
                           int64_t n_samples, float* loss, float* ws_incl, float* wts_incl, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_rays < 0 || n_samples < 0) return NRC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (n_samples > 0) {
        if (!ws_incl || !wts_incl) return NRC_ERR_INVALID;
        nrc_zero_async(ws_incl, n_samples * sizeof(float), s);
        nrc_zero_async(wts_incl, n_samples * sizeof(float), s);
    }
    if (n_rays == 0) return NRC_OK;
    if (!loss || !rays_a || (n_samples > 0 && (!ws || !deltas || !ts))) return NRC_ERR_INVALID;
    nrc_zero_async(loss, n_rays * sizeof(float), s);
    hipLaunchKernelGGL(k_distortion_fw, dim3(nrc_cdiv(n_rays, 4)), dim3(256), 0, s, ws, deltas, ts, rays_a, n_rays, loss, ws_incl, wts_incl);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_distortion_loss_bw(const float* dL_dloss, const float* ws_incl, const float* wts_incl, const float* ws,
                           const float* deltas, const float* ts, const int64_t* rays_a, int64_t n_rays, int64_t n_samples,
                           float* dL_dws, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_rays < 0 || n_samples < 0) return NRC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (n_samples > 0) { if (!dL_dws) return NRC_ERR_INVALID; nrc_zero_async(dL_dws, n_samples * sizeof(float), s); }
    if (n_rays == 0 || n_samples == 0) return NRC_OK;
    if (!dL_dloss || !ws_incl || !wts_incl || !ws || !deltas || !ts || !rays_a) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_distortion_bw, dim3(nrc_cdiv(n_rays, 4)), dim3(256), 0, s, dL_dloss, ws_incl, wts_incl, ws, deltas, ts,
                       rays_a, n_rays, dL_dws);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_ngp_composite_image(const void* packed_f16, const float* ts, const int32_t* ray_cnt, const int32_t* tile_off, int32_t width,
                            int32_t height, int64_t tile_begin, int64_t n_tiles, int32_t cascades, float exp_step_factor,
                            int32_t grid_size, int32_t max_samples, float T_threshold, const float* bg3_host, float* rgb, float* alpha,
                            float* depth, int64_t row_capacity, int32_t arena_rows, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_tiles < 0 || tile_begin < 0 || width < 1 || height < 1 || !bg3_host || grid_size < 1 || max_samples < 1 || row_capacity < 0 || arena_rows < 0) return NRC_ERR_INVALID;
    if (n_tiles == 0) return NRC_OK;
    if (!ray_cnt || !tile_off || !rgb || !alpha || !depth) return NRC_ERR_INVALID;
    const int tiles_x = (width + NRC_TILE_W - 1) / NRC_TILE_W;
    // calc_dt of the test kernel (raymarching.cu:11-13 with `cascades` in place of `scale`, :370)
    const float dt_min = 1.73205080757f / max_samples, dt_max = 1.73205080757f * 2 * (float)cascades / grid_size;
    hipLaunchKernelGGL(k_composite_image, dim3(nrc_cdiv(n_tiles, 4)), dim3(256), 0, (hipStream_t)stream, (const __half*)packed_f16, ts, ray_cnt,
                       tile_off, (int)width, (int)height, tiles_x, tile_begin, n_tiles, exp_step_factor, dt_min, dt_max, T_threshold,
                       bg3_host[0], bg3_host[1], bg3_host[2], rgb, alpha, depth, row_capacity > 0 ? row_capacity : INT64_MAX, (int)arena_rows);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

}  // extern "C"
