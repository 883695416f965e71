// gs_raster.hip -- 3D Gaussian Splatting tile rasterizer for gfx950, forward and backward.
//
// Replaces diff-gaussian-rasterization @ 59f5f77e (src/Thirdparty/DiffGaussianRasterization.py:9) behind the call sites of
// src/Methods/GaussianSplatting/Renderer.py:60-81,94-153,163-183.  Algorithm = Kerbl et al. 2023; conventions and the
// statement the kernels are checked against: oracle/gs_oracle_impl.h.
//
// BUILD NOTE: compiled with -ffp-contract=off.  Radii, tile rectangles and the depth keys that order the blend are integer
// decisions taken from f32 arithmetic and must be bit-exact against the oracle (which is built the same way).
//
// MI355X-native structure (vs the reference's pipeline: inclusive scan -> key duplication -> DEVICE-WIDE 64-bit radix sort ->
// range search, ~100 B of HBM traffic per (tile, Gaussian) instance):
//   1. k_preprocess   one lane per Gaussian: cull, cov3D, EWA cov2D, conic, radius, SH -> RGB; counts instances per TILE.
//   2. k_scan_tiles   exclusive scan of the per-tile counts = the tile ranges (no key search pass).
//   3. k_scatter      every Gaussian drops (depth bits << 32 | id) into its tiles' segments (8 B written per instance).
//   4. k_sort_tiles   one workgroup per tile sorts ITS segment by (depth, id) inside LDS (bitonic, 64-bit keys; segments
//                     larger than the LDS budget fall back to the same network in global memory) and emits the id list.
//                     The order (depth, then index) is exactly what a stable radix sort over (tile | depth) keys produces.
//   5. k_render       16x16-pixel tile per workgroup (4 waves), 256-Gaussian batches staged in LDS, front-to-back blend.
//   => 20 B of traffic per instance instead of ~100, and no global sort.
// Backward:
//   6. k_render_bw    same tiles, back-to-front; per-Gaussian gradients are reduced ACROSS THE WAVE with DPP shuffles and
//                     leave as one atomic per wave and quantity (the reference issues one atomic per PIXEL and quantity).
//   7. k_preprocess_bw one lane per Gaussian: conic -> cov2D -> cov3D -> scale/rotation, mean2D -> mean3D, colour -> SH.
#include <hip/hip_runtime.h>

#include "common.h"

#define TILE 16
#define BATCH 256
#define SORT_LDS_CAP 8192  // 64-bit keys sorted in LDS per tile (64 KB)

namespace {

struct GsCam {
    float view[16], proj[16], campos[3];
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
    int W, H, gx, gy, D, M;
};

__device__ __forceinline__ void xform43(const float* p, const float* m, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
__device__ __forceinline__ void xform44(const float* p, const float* m, float* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}
__device__ __forceinline__ void quat_R(const float* q, float* R) {
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - r * z);     R[2] = 2 * (x * z + r * y);
    R[3] = 2 * (x * y + r * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - r * x);
    R[6] = 2 * (x * z - r * y);     R[7] = 2 * (y * z + r * x);     R[8] = 1 - 2 * (x * x + y * y);
}
__device__ __forceinline__ void cov3d(const float* scale, float mod, const float* q, float* c) {
    float R[9], A[9];
    quat_R(q, R);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) A[3 * i + k] = R[3 * i + k] * (mod * scale[k]);
    c[0] = A[0] * A[0] + A[1] * A[1] + A[2] * A[2];
    c[1] = A[0] * A[3] + A[1] * A[4] + A[2] * A[5];
    c[2] = A[0] * A[6] + A[1] * A[7] + A[2] * A[8];
    c[3] = A[3] * A[3] + A[4] * A[4] + A[5] * A[5];
    c[4] = A[3] * A[6] + A[4] * A[7] + A[5] * A[8];
    c[5] = A[6] * A[6] + A[7] * A[7] + A[8] * A[8];
}
__device__ __forceinline__ void proj_jac(const float* t_in, float fx, float fy, float tanx, float tany, const float* vm, float* Mx,
                                         float* My, float* t_cl, int* gx, int* gy) {
    const float limx = 1.3f * tanx, limy = 1.3f * tany;
    const float txtz = t_in[0] / t_in[2], tytz = t_in[1] / t_in[2];
    t_cl[0] = fminf(limx, fmaxf(-limx, txtz)) * t_in[2];
    t_cl[1] = fminf(limy, fmaxf(-limy, tytz)) * t_in[2];
    t_cl[2] = t_in[2];
    *gx = (txtz < -limx || txtz > limx) ? 0 : 1;
    *gy = (tytz < -limy || tytz > limy) ? 0 : 1;
    const float j00 = fx / t_cl[2], j02 = -(fx * t_cl[0]) / (t_cl[2] * t_cl[2]);
    const float j11 = fy / t_cl[2], j12 = -(fy * t_cl[1]) / (t_cl[2] * t_cl[2]);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        Mx[k] = j00 * vm[0 + 4 * k] + j02 * vm[2 + 4 * k];
        My[k] = j11 * vm[1 + 4 * k] + j12 * vm[2 + 4 * k];
    }
}
__device__ __forceinline__ void sym_mul(const float* c, const float* v, float* o) {
    o[0] = c[0] * v[0] + c[1] * v[1] + c[2] * v[2];
    o[1] = c[1] * v[0] + c[3] * v[1] + c[4] * v[2];
    o[2] = c[2] * v[0] + c[4] * v[1] + c[5] * v[2];
}

#define SH_C0 0.28209479177387814f
#define SH_C1 0.4886025119029199f
__constant__ float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
__constant__ float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                               -0.4570457994644658f, 1.445305721320277f,  -0.5900435899266435f};

__device__ __forceinline__ void sh_color(int deg, const float* pos, const float* campos, const float* sh, float* rgb, uint8_t* clamped) {
    float d[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
    const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    d[0] /= len; d[1] /= len; d[2] /= len;
    const float x = d[0], y = d[1], z = d[2];
    uint8_t mask = 0;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float r = SH_C0 * sh[c];
        if (deg > 0) {
            r = r - SH_C1 * y * sh[3 + c] + SH_C1 * z * sh[6 + c] - SH_C1 * x * sh[9 + c];
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * sh[12 + c] + SH_C2[1] * yz * sh[15 + c] + SH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + c] +
                    SH_C2[3] * xz * sh[21 + c] + SH_C2[4] * (xx - yy) * sh[24 + c];
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * sh[27 + c] + SH_C3[1] * xy * z * sh[30 + c] +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + c] + SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + c] +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + c] + SH_C3[5] * z * (xx - yy) * sh[42 + c] +
                        SH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + c];
                }
            }
        }
        r += 0.5f;
        if (r < 0) mask |= (uint8_t)(1u << c);
        rgb[c] = r < 0 ? 0 : r;
    }
    *clamped = mask;
}
__device__ __forceinline__ void tile_rect(const float* p, int radius, int gx, int gy, int* rmin, int* rmax) {
    rmin[0] = min(gx, max(0, (int)((p[0] - radius) / TILE)));
    rmin[1] = min(gy, max(0, (int)((p[1] - radius) / TILE)));
    rmax[0] = min(gx, max(0, (int)((p[0] + radius + TILE - 1) / TILE)));
    rmax[1] = min(gy, max(0, (int)((p[1] + radius + TILE - 1) / TILE)));
}

// ------------------------------------------------------------------------------------------------ 1. preprocess
// one Gaussian; sh_row = its SH coefficients (LDS copy, see k_preprocess)
__device__ __forceinline__ void preprocess_one(int i, const GsCam& cam, const float* __restrict__ means3D, const float* sh_row,
                                               const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                                               const float* __restrict__ scales, const float* __restrict__ rotations,
                                               const float* __restrict__ cov3D_precomp, int32_t* __restrict__ radii,
                                               float* __restrict__ depths, float* __restrict__ points_xy,
                                               float* __restrict__ conic_opacity, float* __restrict__ rgb, uint8_t* __restrict__ clamped,
                                               float* __restrict__ cov3D, uint32_t* __restrict__ tiles_touched,
                                               uint32_t* __restrict__ tile_counts) {
    radii[i] = 0; tiles_touched[i] = 0; depths[i] = 0.f;
    points_xy[2 * i] = 0.f; points_xy[2 * i + 1] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) conic_opacity[4 * i + k] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) rgb[3 * i + k] = 0.f;
    clamped[i] = 0;
    const float p[3] = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
    float pv[3];
    xform43(p, cam.view, pv);
    float c3[6];
    if (cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) c3[k] = cov3D_precomp[6 * i + k];
    } else {
        const float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
        const float q[4] = {rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2], rotations[4 * i + 3]};
        cov3d(s, cam.scale_modifier, q, c3);
    }
#pragma unroll
    for (int k = 0; k < 6; k++) cov3D[6 * i + k] = c3[k];
    if (pv[2] <= 0.2f) return;
    float ph[4];
    xform44(p, cam.proj, ph);
    const float pw = 1.0f / (ph[3] + 0.0000001f);
    const float ndc[2] = {ph[0] * pw, ph[1] * pw};
    float tc[3], Mx[3], My[3], sx[3], sy[3];
    int gmx, gmy;
    proj_jac(pv, cam.focal_x, cam.focal_y, cam.tan_fovx, cam.tan_fovy, cam.view, Mx, My, tc, &gmx, &gmy);
    sym_mul(c3, Mx, sx); sym_mul(c3, My, sy);
    const float cov[3] = {Mx[0] * sx[0] + Mx[1] * sx[1] + Mx[2] * sx[2] + 0.3f, Mx[0] * sy[0] + Mx[1] * sy[1] + Mx[2] * sy[2],
                          My[0] * sy[0] + My[1] * sy[1] + My[2] * sy[2] + 0.3f};
    const float det = cov[0] * cov[2] - cov[1] * cov[1];
    if (det == 0.0f) return;
    const float det_inv = 1.0f / det;
    const float conic[3] = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
    const float mid = 0.5f * (cov[0] + cov[2]);
    const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    const int my_radius = (int)ceilf(3.0f * sqrtf(fmaxf(lambda1, lambda2)));
    const float pix[2] = {((ndc[0] + 1.0f) * cam.W - 1.0f) * 0.5f, ((ndc[1] + 1.0f) * cam.H - 1.0f) * 0.5f};
    int rmin[2], rmax[2];
    tile_rect(pix, my_radius, cam.gx, cam.gy, rmin, rmax);
    if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) return;
    if (colors_precomp) {
#pragma unroll
        for (int k = 0; k < 3; k++) rgb[3 * i + k] = colors_precomp[3 * i + k];
    } else {
        float col[3];
        uint8_t cl;
        sh_color(cam.D, p, cam.campos, sh_row, col, &cl);
#pragma unroll
        for (int k = 0; k < 3; k++) rgb[3 * i + k] = col[k];
        clamped[i] = cl;
    }
    depths[i] = pv[2]; radii[i] = my_radius;
    points_xy[2 * i] = pix[0]; points_xy[2 * i + 1] = pix[1];
    conic_opacity[4 * i] = conic[0]; conic_opacity[4 * i + 1] = conic[1]; conic_opacity[4 * i + 2] = conic[2];
    conic_opacity[4 * i + 3] = opacities[i];
    tiles_touched[i] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
    if (tile_counts) {  // fallback binning for tile grids too large for the LDS histograms
        for (int y = rmin[1]; y < rmax[1]; y++)
            for (int x = rmin[0]; x < rmax[0]; x++) atomicAdd(&tile_counts[y * cam.gx + x], 1u);
    }
}
// Workgroup = 128 Gaussians; their SH coefficients (one contiguous 128 x 3M float block) are staged through LDS with coalesced
// loads -- a lane walking its own 192-byte row touches 48 cache lines per wave instruction (same staging as k_preprocess_bw).
#define PRE_BLOCK 128
#define PRE_MAXM 16
__global__ void __launch_bounds__(PRE_BLOCK) k_preprocess(int P, GsCam cam, const float* __restrict__ means3D, const float* __restrict__ shs,
                                                          const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                                                          const float* __restrict__ scales, const float* __restrict__ rotations,
                                                          const float* __restrict__ cov3D_precomp, int32_t* __restrict__ radii,
                                                          float* __restrict__ depths, float* __restrict__ points_xy,
                                                          float* __restrict__ conic_opacity, float* __restrict__ rgb, uint8_t* __restrict__ clamped,
                                                          float* __restrict__ cov3D, uint32_t* __restrict__ tiles_touched,
                                                          uint32_t* __restrict__ tile_counts) {
    __shared__ float s_sh[PRE_BLOCK * (3 * PRE_MAXM + 1)];
    const int first = blockIdx.x * PRE_BLOCK, i = first + threadIdx.x;
    const int row_len = 3 * cam.M, pitch = row_len + 1;
    if (shs) {
        const int count = min(PRE_BLOCK, P - first);
        const float* src = shs + (size_t)first * row_len;
        for (int k = threadIdx.x; k < count * row_len; k += PRE_BLOCK) {
            const int r = k / row_len;
            s_sh[r * pitch + (k - r * row_len)] = src[k];
        }
        __syncthreads();
    }
    if (i < P)
        preprocess_one(i, cam, means3D, s_sh + threadIdx.x * pitch, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii, depths,
                       points_xy, conic_opacity, rgb, clamped, cov3D, tiles_touched, tile_counts);
}

// ---- binning without global atomics and without a per-tile sort ----------------------------------------------------------
// History (profiles/): v1 global atomics onto 4 346 tile counters = 2.4 ms of a 4 ms forward; v2 LDS histograms per
// workgroup slice + (tile|depth) keys + per-tile bitonic sorts; v3 depth pre-sort + LDS-atomic scatter + segmented finishing
// sort (0.39 ms); v4 (this one) depth pre-sort + STABLE wave walk, which makes the scatter itself produce the sorted lists.
// ---- depth pre-sort of the P Gaussians (LSD radix, 4 x 8 bits, stable: equal depths keep ascending index) ---------------
// Sorting the P Gaussians once by depth (8 B x P x 4 passes) replaces sorting the D >> P instances per tile: the binning
// passes below take their slices from the depth-ordered list and keep that order inside every tile.
#define RS_ITEMS 16                  // keys per thread per block
#define RS_TILE (256 * RS_ITEMS)     // keys per block
__global__ void __launch_bounds__(256) k_depth_keys(int P, const int32_t* __restrict__ radii, const float* __restrict__ depths,
                                                    uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    keys[i] = radii[i] > 0 ? __float_as_uint(depths[i]) : 0xffffffffu;  // invisible Gaussians go last
    vals[i] = (uint32_t)i;
}
// counts: digit-major [256][nblk4] (rows padded to a multiple of 4 for 16-byte loads); tot[256] = digit totals (atomics)
__global__ void __launch_bounds__(256) k_radix_count(int n, int shift, int nblk4, const uint32_t* __restrict__ keys, uint32_t* __restrict__ counts,
                                                     uint32_t* __restrict__ tot) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int r = 0; r < RS_ITEMS; r++) {
        const int i = base + r * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    const uint32_t c = h[threadIdx.x];
    counts[(size_t)threadIdx.x * nblk4 + blockIdx.x] = c;  // digit-major: the scan order of a stable LSD pass
    if (c) atomicAdd(&tot[threadIdx.x], c);
}
// The global offset of (digit d, block b) = sum of tot[d' < d] + sum of counts[d][b' < b]: every scatter workgroup derives its
// 256 offsets itself (thread d reads a prefix of row d with 16-byte loads) instead of waiting for a scan kernel over the whole
// 256 x nblk matrix, which is latency-bound in a single workgroup (37 us per pass measured, more than count + scatter together).
__global__ void __launch_bounds__(256) k_radix_scatter(int n, int shift, int nblk4, const uint32_t* __restrict__ keys_in,
                                                       const uint32_t* __restrict__ vals_in, const uint32_t* __restrict__ counts,
                                                       const uint32_t* __restrict__ tot, uint32_t* __restrict__ keys_out,
                                                       uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t base_s[256];
    __shared__ uint32_t whist[4][256];
    __shared__ uint32_t wtot[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        const uint32_t mine = tot[threadIdx.x];
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wtot[wave] = incl;
        const uint32_t* row = counts + (size_t)threadIdx.x * nblk4;
        const int nb = (int)blockIdx.x;
        uint32_t pre = 0;
        int b = 0;
#pragma unroll 4
        for (; b + 4 <= nb; b += 4) {
            const uint4 v = *reinterpret_cast<const uint4*>(row + b);
            pre += v.x + v.y + v.z + v.w;
        }
        for (; b < nb; b++) pre += row[b];
        __syncthreads();
        uint32_t off = incl - mine + pre;
        for (int w = 0; w < wave; w++) off += wtot[w];
        base_s[threadIdx.x] = off;
    }
#pragma unroll
    for (int w = 0; w < 4; w++) whist[w][threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE;
    for (int r = 0; r < RS_ITEMS; r++) {
        const int i = base + r * 256 + threadIdx.x;
        const bool valid = i < n;
        const uint32_t key = valid ? keys_in[i] : 0u, val = valid ? vals_in[i] : 0u;
        const uint32_t d = (key >> shift) & 255u;
        // lanes of this wave holding the same digit (8 ballots), rank among them in lane order = stable
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank_w = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (valid && rank_w == 0) whist[wave][d] = (uint32_t)__popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t off = base_s[d];
            for (int w = 0; w < wave; w++) off += whist[w][d];
            keys_out[off + rank_w] = key;
            vals_out[off + rank_w] = val;
        }
        __syncthreads();
        {
            uint32_t t = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) { t += whist[w][threadIdx.x]; whist[w][threadIdx.x] = 0; }
            base_s[threadIdx.x] += t;
        }
        __syncthreads();
    }
}

#define GS_MAX_LDS_TILES 16384
#define GS_SLICE_MAX 2048  // slices = waves of the binning passes
#define GS_GROUPS 32       // slice groups of the two-level column scan over the slices x tiles matrix
// ---- stable binning: one WAVE per slice of the depth-ordered Gaussians ----------------------------------------------------
// Counting pass: lane = Gaussian, LDS atomics (order is irrelevant for counts).
// slice = blockIdx.  (An XCD-contiguous mapping -- slice = (blockIdx % 8) * nb/8 + blockIdx / 8, so that the id writes falling
// into one cache line come from one XCD's L2 -- was measured slower, 172 vs 138 us: the invisible tail of the depth order then
// sits on one XCD and the other seven carry 8/7 of the work.)
__device__ __forceinline__ int xcd_slice(int nb) { return (int)blockIdx.x < nb ? (int)blockIdx.x : -1; }
__global__ void __launch_bounds__(64) k_bin_count(int P, int nb, int chunk, int gx, int gy, const uint32_t* __restrict__ order,
                                                  const int32_t* __restrict__ radii, const float* __restrict__ points_xy,
                                                  uint32_t* __restrict__ hist) {
    extern __shared__ uint32_t lt[];
    const int n_tiles = gx * gy, lane = threadIdx.x;
    const int slice = xcd_slice(nb);
    if (slice < 0) return;
    for (int t = lane; t < n_tiles; t += 64) lt[t] = 0u;
    __syncthreads();
    const int begin = slice * chunk, end = min(P, begin + chunk);
    for (int j = begin + lane; j < end; j += 64) {
        const int id = (int)order[j];
        const int r = radii[id];
        if (r <= 0) continue;
        const float pxy[2] = {points_xy[2 * id], points_xy[2 * id + 1]};
        int rmin[2], rmax[2];
        tile_rect(pxy, r, gx, gy, rmin, rmax);
        for (int y = rmin[1]; y < rmax[1]; y++)
            for (int x = rmin[0]; x < rmax[0]; x++) atomicAdd(&lt[y * gx + x], 1u);
    }
    __syncthreads();
    uint32_t* row = hist + (size_t)slice * n_tiles;
    for (int t = lane; t < n_tiles; t += 64) row[t] = lt[t];
}
// Scatter pass: the wave walks its slice in depth order, one Gaussian per step, lanes = the tiles of that Gaussian's rectangle.
// The tiles of one rectangle are distinct, so the per-tile cursors in LDS are plain read-modify-writes (no atomics), and because
// steps run in program order every tile receives its Gaussians in depth order: the ids land directly in their final, sorted
// positions -- no (tile|depth) keys, no per-tile sort.  (A variant with four 16-lane Gaussians per step and a claim byte per tile
// to detect shared tiles was measured at 2x the time: rectangles are heavy-tailed, large ones need 4x the rounds at 16 lanes.)
__global__ void __launch_bounds__(64) k_bin_scatter(int P, int nb, int chunk, int gx, int gy, const uint32_t* __restrict__ order,
                                                    const int32_t* __restrict__ radii, const float* __restrict__ points_xy,
                                                    const uint32_t* __restrict__ bases, int32_t* __restrict__ point_list) {
    extern __shared__ uint32_t lt[];
    const int n_tiles = gx * gy, lane = threadIdx.x;
    const int slice = xcd_slice(nb);
    if (slice < 0) return;
    const uint32_t* row = bases + (size_t)slice * n_tiles;
    for (int t = lane; t < n_tiles; t += 64) lt[t] = row[t];
    __syncthreads();
    const int begin = slice * chunk, end = min(P, begin + chunk);
    for (int j0 = begin; j0 < end; j0 += 64) {
        const int j = j0 + lane;
        int id = 0;
        uint32_t org = 0u, ext = 0u;  // x0 | y0 << 16,  w | h << 16
        if (j < end) {
            id = (int)order[j];
            const int r = radii[id];
            if (r > 0) {
                const float pxy[2] = {points_xy[2 * id], points_xy[2 * id + 1]};
                int rmin[2], rmax[2];
                tile_rect(pxy, r, gx, gy, rmin, rmax);
                org = (uint32_t)rmin[0] | (uint32_t)rmin[1] << 16;
                ext = (uint32_t)(rmax[0] - rmin[0]) | (uint32_t)(rmax[1] - rmin[1]) << 16;
                if ((ext & 0xffffu) == 0u || (ext >> 16) == 0u) ext = 0u;
            }
        }
        const uint32_t wq = ext & 0xffffu;
        const uint32_t magic = wq > 1u ? 0xffffffffu / wq + 1u : 0u;  // c / w == umulhi(c, magic) for c < 2^16, w >= 2
        unsigned long long todo = __ballot(ext != 0u);
        while (todo) {
            const int u = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const uint32_t uorg = (uint32_t)__builtin_amdgcn_readlane((int)org, u), uext = (uint32_t)__builtin_amdgcn_readlane((int)ext, u);
            const int uid = __builtin_amdgcn_readlane(id, u);
            const uint32_t um = (uint32_t)__builtin_amdgcn_readlane((int)magic, u);
            const int ux0 = uorg & 0xffffu, uy0 = uorg >> 16, uw = uext & 0xffffu, uh = uext >> 16;
            const int un = uw * uh;
            for (int c0 = 0; c0 < un; c0 += 64) {
                const int c = c0 + lane;
                if (c < un) {
                    const int yy = uw == 1 ? c : (int)__umulhi((uint32_t)c, um);
                    const int t = (uy0 + yy) * gx + ux0 + (c - yy * uw);
                    const uint32_t slot = lt[t];
                    lt[t] = slot + 1u;
                    point_list[slot] = uid;
                }
            }
        }
    }
}
// column sums per slice group: part[g][t] = sum of hist[b][t] over the slices b of group g   (64 tiles x 4 groups per workgroup)
__global__ void __launch_bounds__(256) k_tile_totals(int nb, int n_tiles, const uint32_t* __restrict__ hist, uint32_t* __restrict__ part) {
    const int tx = threadIdx.x & 63, g = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int t = blockIdx.x * 64 + tx;
    const int per = (nb + GS_GROUPS - 1) / GS_GROUPS, b0 = min(nb, g * per), b1 = min(nb, b0 + per);
    if (t >= n_tiles) return;
    uint32_t s = 0;
#pragma unroll 8
    for (int b = b0; b < b1; b++) s += hist[(size_t)b * n_tiles + t];
    part[(size_t)g * n_tiles + t] = s;
}
// hist[b][t] <- first output position of slice b in tile t (exclusive column scan inside the group, seeded with the group base)
__global__ void __launch_bounds__(256) k_tile_bases(int nb, int n_tiles, const uint32_t* __restrict__ part, uint32_t* __restrict__ hist) {
    const int tx = threadIdx.x & 63, g = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int t = blockIdx.x * 64 + tx;
    const int per = (nb + GS_GROUPS - 1) / GS_GROUPS, b0 = min(nb, g * per), b1 = min(nb, b0 + per);
    if (t >= n_tiles) return;
    uint32_t run = part[(size_t)g * n_tiles + t];
#pragma unroll 8
    for (int b = b0; b < b1; b++) {
        const uint32_t c = hist[(size_t)b * n_tiles + t];
        hist[(size_t)b * n_tiles + t] = run;
        run += c;
    }
}
// tile ranges from the group sums; part[g][t] becomes the first output position of group g in tile t
__global__ void __launch_bounds__(1024) k_scan_tiles_grouped(uint32_t* __restrict__ part, int n, uint32_t* __restrict__ ranges,
                                                             uint32_t* __restrict__ fill, int64_t* __restrict__ num_rendered) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        uint32_t pv[GS_GROUPS];
        uint32_t v = 0;
#pragma unroll
        for (int g = 0; g < GS_GROUPS; g++) { pv[g] = i < n ? part[(size_t)g * n + i] : 0u; v += pv[g]; }
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t off = carry_s;
        for (int w = 0; w < wave; w++) off += wave_tot[w];
        if (i < n) {
            uint32_t run = off + incl - v;
            ranges[2 * i] = run; ranges[2 * i + 1] = off + incl; fill[i] = 0u;
#pragma unroll
            for (int g = 0; g < GS_GROUPS; g++) { part[(size_t)g * n + i] = run; run += pv[g]; }
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *num_rendered = (int64_t)carry_s;
}

// ------------------------------------------------------------------------------------------------ 2. tile ranges
__global__ void __launch_bounds__(1024) k_scan_tiles(const uint32_t* __restrict__ counts, int n, uint32_t* __restrict__ ranges,
                                                     uint32_t* __restrict__ fill, int64_t* __restrict__ num_rendered) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t v = i < n ? counts[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t off = carry_s;
        for (int w = 0; w < wave; w++) off += wave_tot[w];
        if (i < n) { ranges[2 * i] = off + incl - v; ranges[2 * i + 1] = off + incl; fill[i] = 0u; }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *num_rendered = (int64_t)carry_s;
}

// ------------------------------------------------------------------------------------------------ 3. scatter instances
__global__ void __launch_bounds__(256) k_scatter(int P, int gx, int gy, const int32_t* __restrict__ radii, const float* __restrict__ depths,
                                                 const float* __restrict__ points_xy, const uint32_t* __restrict__ ranges,
                                                 uint32_t* __restrict__ fill, uint64_t* __restrict__ keys) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P || radii[i] <= 0) return;
    const float pxy[2] = {points_xy[2 * i], points_xy[2 * i + 1]};
    int rmin[2], rmax[2];
    tile_rect(pxy, radii[i], gx, gy, rmin, rmax);
    const uint64_t key = ((uint64_t)__float_as_uint(depths[i]) << 32) | (uint32_t)i;
    for (int y = rmin[1]; y < rmax[1]; y++)
        for (int x = rmin[0]; x < rmax[0]; x++) {
            const int t = y * gx + x;
            const uint32_t slot = atomicAdd(&fill[t], 1u);
            keys[ranges[2 * t] + slot] = key;
        }
}

// ------------------------------------------------------------------------------------------------ 4. per-tile sort
// Normalised bitonic network on 64-bit keys: every compare-exchange puts the smaller key at the lower index, the first
// step of each merge pairs i with its mirror i ^ (k-1).  Partners beyond n are skipped (they behave like +inf, which
// never has to move), so any n is sorted in place without padding.  `a` may be LDS or global memory.
__device__ __forceinline__ void bitonic_sort_any(uint64_t* a, int n) {
    for (int k = 2; (k >> 1) < n; k <<= 1) {
        for (int i = threadIdx.x; i < n; i += 256) {
            const int l = i ^ (k - 1);
            if (l > i && l < n) {
                const uint64_t x = a[i], y = a[l];
                if (x > y) { a[i] = y; a[l] = x; }
            }
        }
        __syncthreads();
        for (int j = k >> 2; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n; i += 256) {
                const int l = i ^ j;
                if (l > i && l < n) {
                    const uint64_t x = a[i], y = a[l];
                    if (x > y) { a[i] = y; a[l] = x; }
                }
            }
            __syncthreads();
        }
    }
}
// capacity classes: a workgroup only handles tiles whose segment length lies in (LO, CAP] so that small tiles do not reserve
// the LDS of the largest ones (LDS, not registers, limits how many tiles a CU sorts concurrently)
template <int LO, int CAP>
__global__ void __launch_bounds__(256) k_sort_tiles(const uint32_t* __restrict__ ranges, uint64_t* __restrict__ keys,
                                                    int32_t* __restrict__ point_list) {
    __shared__ uint64_t lds[CAP];
    const int t = blockIdx.x;
    const uint32_t r0 = ranges[2 * t], r1 = ranges[2 * t + 1];
    const int n = (int)(r1 - r0);
    if (n <= LO || (n > CAP && CAP < SORT_LDS_CAP)) return;
    if (n <= CAP) {
        for (int i = threadIdx.x; i < n; i += 256) lds[i] = keys[r0 + i];
        __syncthreads();
        bitonic_sort_any(lds, n);
        for (int i = threadIdx.x; i < n; i += 256) point_list[r0 + i] = (int32_t)(uint32_t)lds[i];
    } else {
        // oversized segment (> 8192 Gaussians on one tile): the same network directly on the global segment; the workgroup is
        // the only reader/writer of it and __syncthreads() orders its own global accesses
        bitonic_sort_any(keys + r0, n);
        for (int i = threadIdx.x; i < n; i += 256) point_list[r0 + i] = (int32_t)(uint32_t)keys[r0 + i];
    }
}

// ------------------------------------------------------------------------------------------------ 5. render
// thread -> pixel of the 16x16 tile: every wave owns an 8x8 quadrant (not a 16x4 strip), so that fewer Gaussians of the tile list
// touch a given wave (smaller perimeter) and more steps are skipped with the whole wave inactive
__device__ __forceinline__ int tile_px(unsigned t) { return (int)((t & 7u) + ((t >> 6) & 1u) * 8u); }
__device__ __forceinline__ int tile_py(unsigned t) { return (int)(((t >> 3) & 7u) + (t >> 7) * 8u); }
// The tile lists come from the square 3-sigma bound of preprocess (reference semantics, kept bit-exact), but a pixel only blends a
// Gaussian where alpha = o exp(power) >= 1/255, i.e. inside the ellipse A dx^2 + 2 B dx dy + C dy^2 <= 2 ln(255 o).  On the bench
// scene 52 % of the (tile, Gaussian) pairs of a list have no such pixel in the tile, and a pair that has touches 2.5 of the 4 wave
// quadrants (tools/gs_stats.py).  Both render kernels therefore (1) drop the pairs whose ellipse box misses the tile while a batch
// is staged into LDS (order-preserving compaction, the list position travels with the entry), and (2) test the box against the
// wave's 8x8 quadrant with wave-uniform arithmetic before any per-pixel work.  The box is conservative (margins below), so
// exactly the same pixels blend exactly the same Gaussians.
__device__ __forceinline__ float2 splat_extent(const float4& co) {
    const float inf = __builtin_inff();
    if (!(co.w > 0.f)) return make_float2(-1.f, -1.f);              // alpha <= 0 everywhere
    const float tau = 2.f * (logf(255.f * co.w) + 1e-3f);
    if (!(tau > 0.f)) return make_float2(-1.f, -1.f);               // o < 1/255: never reaches the threshold
    const float det = co.x * co.z - co.y * co.y;
    if (!(det > 0.f) || !(co.x > 0.f) || !(co.z > 0.f)) return make_float2(inf, inf);  // not an ellipse: no culling
    return make_float2(sqrtf(tau * co.z / det) * 1.001f + 0.01f, sqrtf(tau * co.x / det) * 1.001f + 0.01f);
}
__device__ __forceinline__ bool box_hits(const float2& xy, const float2& ext, float x0, float y0, float span) {
    return xy.x + ext.x >= x0 && xy.x - ext.x <= x0 + span && xy.y + ext.y >= y0 && xy.y - ext.y <= y0 + span;
}
// order-preserving compaction of the workgroup's flagged threads: returns this thread's slot, total in *n_out
__device__ __forceinline__ int block_compact(bool flag, int* s_wcnt, int* n_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    if (lane == 0) s_wcnt[wave] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const int c = s_wcnt[w]; tot += c; if (w < wave) off += c; }
    *n_out = tot;
    return off + __popcll(m & ((1ull << lane) - 1ull));
}

// Staging of one batch for both render kernels: thread t holds entry t of the batch (flags = bit q set when the alpha >= 1/255 ellipse box
// of its Gaussian reaches quadrant q of the tile).  Builds, per quadrant, the ascending list of batch entries that reach it --
// every wave then walks ONLY its own list: no per-step box test, no skipped steps (a listed pair used to cost a wave-uniform box test,
// and 37 % of the steps of a wave were pairs that miss its quadrant).  Returns the length of the calling wave's list.
__device__ __forceinline__ int quadrant_lists(unsigned flags, uint8_t (*s_list)[BATCH], int (*s_qcnt)[4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long m[4];
#pragma unroll
    for (int q = 0; q < 4; q++) m[q] = __ballot((flags >> q) & 1u);
    if (lane < 4) s_qcnt[wave][lane] = __popcll(lane == 0 ? m[0] : lane == 1 ? m[1] : lane == 2 ? m[2] : m[3]);
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    int mine = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        int off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) { const int c = s_qcnt[w][q]; tot += c; if (w < wave) off += c; }
        if ((flags >> q) & 1u) s_list[q][off + __popcll(m[q] & below)] = (uint8_t)threadIdx.x;
        if (q == wave) mine = tot;
    }
    __syncthreads();
    return mine;
}
__device__ __forceinline__ unsigned quadrant_flags(const float2& xy, const float2& ext, float tx0, float ty0) {
    unsigned f = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) f |= (unsigned)box_hits(xy, ext, tx0 + (float)((q & 1) * 8), ty0 + (float)((q >> 1) * 8), 7.f) << q;
    return f;
}

__global__ void __launch_bounds__(256) k_render(GsCam cam, const uint32_t* __restrict__ ranges, const int32_t* __restrict__ point_list,
                                                const float* __restrict__ points_xy, const float* __restrict__ conic_opacity,
                                                const float* __restrict__ rgb, float bg0, float bg1, float bg2, float* __restrict__ out_color,
                                                uint32_t* __restrict__ n_contrib, float* __restrict__ final_T) {
    __shared__ float2 s_xy[BATCH];
    __shared__ float4 s_co[BATCH];
    __shared__ float s_rgb[BATCH * 3];
    __shared__ __attribute__((aligned(4))) uint8_t s_list[4][BATCH];
    __shared__ int s_qcnt[4][4];
    const int tile = blockIdx.y * cam.gx + blockIdx.x;
    const int px = blockIdx.x * TILE + tile_px(threadIdx.x), py = blockIdx.y * TILE + tile_py(threadIdx.x);
    const bool inside = px < cam.W && py < cam.H;
    const float fx = (float)px, fy = (float)py;
    const float tx0 = (float)(blockIdx.x * TILE), ty0 = (float)(blockIdx.y * TILE);
    const int wave = threadIdx.x >> 6;  // = quadrant (tile_px / tile_py)
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    bool done = !inside;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last = 0;
    for (uint32_t base = r0; base < r1; base += BATCH) {
        if (__syncthreads_count(done) == 256) break;
        const uint32_t k = base + threadIdx.x;
        unsigned flags = 0;
        if (k < r1) {
            const int id = point_list[k];
            const float2 xy = make_float2(points_xy[2 * id], points_xy[2 * id + 1]);
            const float4 co = *reinterpret_cast<const float4*>(conic_opacity + 4 * id);
            flags = quadrant_flags(xy, splat_extent(co), tx0, ty0);
            if (flags) {
                s_xy[threadIdx.x] = xy; s_co[threadIdx.x] = co;
                s_rgb[3 * threadIdx.x] = rgb[3 * id]; s_rgb[3 * threadIdx.x + 1] = rgb[3 * id + 1]; s_rgb[3 * threadIdx.x + 2] = rgb[3 * id + 2];
            }
        }
        const int n_mine = quadrant_lists(flags, s_list, s_qcnt);
        const uint32_t pos0 = base - r0 + 1u;  // contributor number of batch entry 0 = its position in the tile list + 1
        // the list is read four entries (one dword, wave-uniform) at a time: one dependent LDS round trip per four Gaussians
        const uint32_t* list4 = reinterpret_cast<const uint32_t*>(s_list[wave]);
        for (int jj = 0; jj < n_mine; jj += 4) {
            if (__ballot(!done) == 0ull) break;  // wave-uniform: every pixel of the quadrant is saturated
            const uint32_t pack = (uint32_t)__builtin_amdgcn_readfirstlane((int)list4[jj >> 2]);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (jj + u >= n_mine) break;
                const int j = (int)((pack >> (8 * u)) & 0xffu);
                // Branch-free body: the reference's four early-outs (done / power > 0 / alpha < 1/255 / saturation) become lane masks.
                // All lanes of a wave execute the same instructions anyway; the nested `continue`s cost ~30 scalar exec-mask instructions
                // per step next to ~45 vector ones.  Same arithmetic, same order: bit-identical pixels.
                const float2 xy_j = s_xy[j];
                const float dx = xy_j.x - fx, dy = xy_j.y - fy;
                const float4 co_j = s_co[j];
                const float power = -0.5f * (co_j.x * dx * dx + co_j.z * dy * dy) - co_j.y * dx * dy;
                const float alpha = fminf(0.99f, co_j.w * expf(power));
                const float test_T = T * (1 - alpha);
                const bool valid = !done & !(power > 0.0f) & !(alpha < 1.0f / 255.0f);
                const bool sat = valid & (test_T < 0.0001f);
                const bool upd = valid & !sat;
                done = done | sat;
                const float a = upd ? alpha : 0.f;
                C0 += s_rgb[3 * j] * a * T; C1 += s_rgb[3 * j + 1] * a * T; C2 += s_rgb[3 * j + 2] * a * T;
                T = upd ? test_T : T;
                last = upd ? pos0 + (uint32_t)j : last;
            }
        }
        __syncthreads();
    }
    if (inside) {
        const size_t pix = (size_t)py * cam.W + px, hw = (size_t)cam.H * cam.W;
        final_T[pix] = T;
        n_contrib[pix] = last;
        out_color[pix] = C0 + T * bg0; out_color[hw + pix] = C1 + T * bg1; out_color[2 * hw + pix] = C2 + T * bg2;
    }
}

// ------------------------------------------------------------------------------------------------ 6. render backward
// wave64 sum with DPP row operations (no LDS crossbar traffic): the total lands in lane 63.  All steps run UNMASKED
// (row_mask = bank_mask = 0xf, out-of-row sources read 0): only lane 63 has to be right, and without masks every step is a
// single v_add_f32_dpp -- with the textbook masks the compiler needs v_mov 0 + v_mov_dpp + v_add for four of the six steps
// (126 instead of 54 instructions for the nine sums of a Gaussian).
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));  // row_shr:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));  // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));  // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));  // row_shr:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xf, 0xf, true));  // row_bcast:15
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xf, 0xf, true));  // row_bcast:31
    return v;
}
// the same over the 16-lane DPP rows only (4 steps): lanes 15, 31, 47, 63 hold their row's sum.  The two row_bcast steps that carry the
// sums on to lane 63 cost 18 VALU instructions per Gaussian for the nine quantities; four single-lane LDS atomics instead of one do not
// show in the kernel time (the kernel issues VALU instructions 89 % of the time): fwd+bwd 1.93 -> 1.88 ms.  Stopping after three steps
// (eight lanes per wave on the same LDS address) does: 2.30 ms.
__device__ __forceinline__ float row_sum_to_lane15(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));  // row_shr:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));  // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));  // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));  // row_shr:8
    return v;
}
__global__ void __launch_bounds__(256) k_render_bw(GsCam cam, const uint32_t* __restrict__ ranges, const int32_t* __restrict__ point_list,
                                                   const float* __restrict__ points_xy, const float* __restrict__ conic_opacity,
                                                   const float* __restrict__ rgb, float bg0, float bg1, float bg2,
                                                   const uint32_t* __restrict__ n_contrib, const float* __restrict__ final_T,
                                                   const float* __restrict__ dL_dpix, float* __restrict__ dL_dmean2D,
                                                   float* __restrict__ dL_dconic, float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolor) {
    __shared__ float2 s_xy[BATCH];
    __shared__ float4 s_co[BATCH];
    __shared__ float s_rgb[BATCH * 3];
    __shared__ float s_acc[BATCH][9];  // per-Gaussian gradient sums of the tile's 4 waves, flushed once per batch
    __shared__ __attribute__((aligned(4))) uint8_t s_list[4][BATCH];
    __shared__ int s_qcnt[4][4];
    const int tile = blockIdx.y * cam.gx + blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;  // wave = quadrant
    const float tx0 = (float)(blockIdx.x * TILE), ty0 = (float)(blockIdx.y * TILE);
    const int px = blockIdx.x * TILE + tile_px(threadIdx.x), py = blockIdx.y * TILE + tile_py(threadIdx.x);
    const bool inside = px < cam.W && py < cam.H;
    const float fx = (float)px, fy = (float)py;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    const int n_tile = (int)(r1 - r0);
    const size_t pix = (size_t)py * cam.W + px, hw = (size_t)cam.H * cam.W;
    const float T_final = inside ? final_T[pix] : 0.f;
    float T = T_final;
    const int last = inside ? (int)n_contrib[pix] : 0;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, last_alpha = 0.f;
    const float g0 = inside ? dL_dpix[pix] : 0.f, g1 = inside ? dL_dpix[hw + pix] : 0.f, g2 = inside ? dL_dpix[2 * hw + pix] : 0.f;
    const float bg_dot = bg0 * g0 + bg1 * g1 + bg2 * g2;
    const float ddelx_dx = 0.5f * cam.W, ddely_dy = 0.5f * cam.H;
    // Only the first max(last) entries of the tile list were blended by any pixel of the tile (k_render stops at saturation,
    // typically after a tenth of the list): the backward walk starts there, not at the end of the list, and each wave skips
    // the entries beyond its own maximum with a scalar compare (these skips were 60 % of the kernel's time before).
    __shared__ int s_maxlast;
    if (threadIdx.x == 0) s_maxlast = 0;
    __syncthreads();
    int wave_last = last;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) wave_last = max(wave_last, __shfl_xor(wave_last, d, 64));
    if (lane == 0) atomicMax(&s_maxlast, wave_last);
    __syncthreads();
    const int n_eff = min(n_tile, s_maxlast);
    // batches are taken from the END of the blended prefix: position p (0-based from the front) has contributor number p + 1;
    // batch entry t sits at position n_eff - 1 - done_cnt - t
    for (int done_cnt = 0; done_cnt < n_eff; done_cnt += BATCH) {
        __syncthreads();
        const int nb_raw = min(BATCH, n_eff - done_cnt);
#pragma unroll
        for (int q = 0; q < 9; q++) s_acc[threadIdx.x][q] = 0.f;
        // stage the batch; per quadrant the list of entries whose alpha >= 1/255 ellipse box reaches it (see k_render)
        int id_l = 0;
        unsigned flags = 0;
        const int pos_top = n_eff - 1 - done_cnt;  // list position of batch entry 0
        if ((int)threadIdx.x < nb_raw) {
            id_l = point_list[r0 + pos_top - threadIdx.x];
            const float2 xy_l = make_float2(points_xy[2 * id_l], points_xy[2 * id_l + 1]);
            const float4 co_l = *reinterpret_cast<const float4*>(conic_opacity + 4 * id_l);
            flags = quadrant_flags(xy_l, splat_extent(co_l), tx0, ty0);
            if (flags) {
                s_xy[threadIdx.x] = xy_l; s_co[threadIdx.x] = co_l;
                s_rgb[3 * threadIdx.x] = rgb[3 * id_l]; s_rgb[3 * threadIdx.x + 1] = rgb[3 * id_l + 1]; s_rgb[3 * threadIdx.x + 2] = rgb[3 * id_l + 2];
            }
        }
        const int n_mine = quadrant_lists(flags, s_list, s_qcnt);
        const uint32_t* list4 = reinterpret_cast<const uint32_t*>(s_list[wave]);
        for (int jj = 0; jj < n_mine; jj++) {
            // four list entries per (wave-uniform) dword read
            const uint32_t pack = (uint32_t)__builtin_amdgcn_readfirstlane((int)list4[jj >> 2]);
            const int j = (int)((pack >> (8 * (jj & 3))) & 0xffu);
            const int pos = pos_top - j;
            if (pos >= wave_last) continue;  // wave-uniform
            // branch-free like the forward loop: lanes that do not blend the Gaussian carry zeros into the sums
            const float4 co = s_co[j];
            const float2 xy = s_xy[j];
            const float dx = xy.x - fx, dy = xy.y - fy;
            const float power = -0.5f * (co.x * dx * dx + co.z * dy * dy) - co.y * dx * dy;
            const float G = expf(power);
            const float alpha = fminf(0.99f, co.w * G);
            const bool active = inside & (pos < last) & !(power > 0.0f) & !(alpha < 1.0f / 255.0f);
            if (__ballot(active) == 0ull) continue;  // nobody in this wave sees the Gaussian
            const float one_minus = 1 - alpha;
            const float T_new = T / one_minus;
            const float dch = alpha * T_new;
            const float c0 = s_rgb[3 * j], c1 = s_rgb[3 * j + 1], c2 = s_rgb[3 * j + 2];
            const float n_acc0 = last_alpha * lc0 + (1 - last_alpha) * acc0;
            const float n_acc1 = last_alpha * lc1 + (1 - last_alpha) * acc1;
            const float n_acc2 = last_alpha * lc2 + (1 - last_alpha) * acc2;
            float dL_dalpha = (c0 - n_acc0) * g0;
            dL_dalpha += (c1 - n_acc1) * g1;
            dL_dalpha += (c2 - n_acc2) * g2;
            dL_dalpha *= T_new;
            dL_dalpha += (-T_final / one_minus) * bg_dot;
            const float dL_dG = co.w * dL_dalpha;
            const float gdx = G * dx, gdy = G * dy;
            const float dG_ddelx = -gdx * co.x - gdy * co.y;
            const float dG_ddely = -gdy * co.z - gdx * co.y;
            float d_c0 = active ? dch * g0 : 0.f, d_c1 = active ? dch * g1 : 0.f, d_c2 = active ? dch * g2 : 0.f;
            float d_mx = active ? dL_dG * dG_ddelx * ddelx_dx : 0.f, d_my = active ? dL_dG * dG_ddely * ddely_dy : 0.f;
            float d_cx = active ? -0.5f * gdx * dx * dL_dG : 0.f, d_cy = active ? -0.5f * gdx * dy * dL_dG : 0.f;
            float d_cw = active ? -0.5f * gdy * dy * dL_dG : 0.f, d_op = active ? G * dL_dalpha : 0.f;
            T = active ? T_new : T;
            acc0 = active ? n_acc0 : acc0; acc1 = active ? n_acc1 : acc1; acc2 = active ? n_acc2 : acc2;
            lc0 = active ? c0 : lc0; lc1 = active ? c1 : lc1; lc2 = active ? c2 : lc2;
            last_alpha = active ? alpha : last_alpha;
            // wave-level reduction, then ONE atomic per wave and quantity
            d_c0 = row_sum_to_lane15(d_c0); d_c1 = row_sum_to_lane15(d_c1); d_c2 = row_sum_to_lane15(d_c2);
            d_mx = row_sum_to_lane15(d_mx); d_my = row_sum_to_lane15(d_my);
            d_cx = row_sum_to_lane15(d_cx); d_cy = row_sum_to_lane15(d_cy); d_cw = row_sum_to_lane15(d_cw); d_op = row_sum_to_lane15(d_op);
            // keep the last DPP add of each sum out of the one-lane branch below (sunk into it, it splits into v_mov_dpp + v_add)
            asm volatile("" : "+v"(d_c0), "+v"(d_c1), "+v"(d_c2), "+v"(d_mx), "+v"(d_my), "+v"(d_cx), "+v"(d_cy), "+v"(d_cw), "+v"(d_op));
            if ((lane & 15) == 15) {  // LDS atomics: the 4 rows of each of the 4 waves meet here, the global atomics happen once per (tile, Gaussian)
                float* a = s_acc[j];
                atomicAdd(a + 0, d_c0); atomicAdd(a + 1, d_c1); atomicAdd(a + 2, d_c2); atomicAdd(a + 3, d_mx); atomicAdd(a + 4, d_my);
                atomicAdd(a + 5, d_cx); atomicAdd(a + 6, d_cy); atomicAdd(a + 7, d_cw); atomicAdd(a + 8, d_op);
            }
        }
        __syncthreads();
        if (flags) {  // the thread that staged the entry flushes it
            const float* a = s_acc[threadIdx.x];
            bool any = false;
#pragma unroll
            for (int q = 0; q < 9; q++) any = any || (a[q] != 0.f);
            if (any) {
                const int id = id_l;
                atomicAdd(dL_dcolor + 3 * id, a[0]); atomicAdd(dL_dcolor + 3 * id + 1, a[1]); atomicAdd(dL_dcolor + 3 * id + 2, a[2]);
                atomicAdd(dL_dmean2D + 3 * id, a[3]); atomicAdd(dL_dmean2D + 3 * id + 1, a[4]);
                atomicAdd(dL_dconic + 4 * id, a[5]); atomicAdd(dL_dconic + 4 * id + 1, a[6]); atomicAdd(dL_dconic + 4 * id + 3, a[7]);
                atomicAdd(dL_dopacity + id, a[8]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ 7. preprocess backward
// One visible Gaussian.  sh_row: this Gaussian's SH coefficients in LDS (3*M floats); each is replaced in place by its gradient.
__device__ __forceinline__ void preprocess_bw_one(int i, const GsCam& cam, const float* __restrict__ means3D, float* sh_row, int use_sh,
                                                  const float* __restrict__ scales, const float* __restrict__ rotations, int use_scale_rot,
                                                  const uint8_t* __restrict__ clamped, const float* __restrict__ cov3D,
                                                  const float* __restrict__ dL_dmean2D, const float* __restrict__ dL_dconic,
                                                  const float* __restrict__ dL_dcolor, float* __restrict__ dL_dmean3D,
                                                  float* __restrict__ dL_dcov3D, float* __restrict__ dL_dscale, float* __restrict__ dL_drot) {
    const float mean[3] = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
    float c3[6];
#pragma unroll
    for (int k = 0; k < 6; k++) c3[k] = cov3D[6 * i + k];
    const float fx = cam.focal_x, fy = cam.focal_y;
    const float* vm = cam.view;
    float t[3], tc[3], Mx[3], My[3], sx[3], sy[3];
    int gmx, gmy;
    xform43(mean, vm, t);
    proj_jac(t, fx, fy, cam.tan_fovx, cam.tan_fovy, vm, Mx, My, tc, &gmx, &gmy);
    sym_mul(c3, Mx, sx); sym_mul(c3, My, sy);
    const float a = Mx[0] * sx[0] + Mx[1] * sx[1] + Mx[2] * sx[2] + 0.3f;
    const float b = Mx[0] * sy[0] + Mx[1] * sy[1] + Mx[2] * sy[2];
    const float c = My[0] * sy[0] + My[1] * sy[1] + My[2] * sy[2] + 0.3f;
    const float gcx = dL_dconic[4 * i], gcy = dL_dconic[4 * i + 1], gcz = dL_dconic[4 * i + 3];
    const float denom = a * c - b * b;
    const float denom2inv = 1.0f / (denom * denom + 0.0000001f);
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    float o[6] = {0, 0, 0, 0, 0, 0};
    if (denom2inv != 0) {
        dL_da = denom2inv * (-c * c * gcx + 2 * b * c * gcy + (denom - a * c) * gcz);
        dL_dc = denom2inv * (-a * a * gcz + 2 * a * b * gcy + (denom - a * c) * gcx);
        dL_db = denom2inv * 2 * (b * c * gcx - (denom + 2 * b * b) * gcy + a * b * gcz);
        o[0] = Mx[0] * Mx[0] * dL_da + Mx[0] * My[0] * dL_db + My[0] * My[0] * dL_dc;
        o[3] = Mx[1] * Mx[1] * dL_da + Mx[1] * My[1] * dL_db + My[1] * My[1] * dL_dc;
        o[5] = Mx[2] * Mx[2] * dL_da + Mx[2] * My[2] * dL_db + My[2] * My[2] * dL_dc;
        o[1] = 2 * Mx[0] * Mx[1] * dL_da + (Mx[0] * My[1] + Mx[1] * My[0]) * dL_db + 2 * My[0] * My[1] * dL_dc;
        o[2] = 2 * Mx[0] * Mx[2] * dL_da + (Mx[0] * My[2] + Mx[2] * My[0]) * dL_db + 2 * My[0] * My[2] * dL_dc;
        o[4] = 2 * Mx[2] * Mx[1] * dL_da + (Mx[1] * My[2] + Mx[2] * My[1]) * dL_db + 2 * My[1] * My[2] * dL_dc;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = o[k];
    float dMx[3], dMy[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { dMx[k] = 2 * sx[k] * dL_da + sy[k] * dL_db; dMy[k] = 2 * sy[k] * dL_dc + sx[k] * dL_db; }
    float dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        dJ00 += vm[0 + 4 * k] * dMx[k]; dJ02 += vm[2 + 4 * k] * dMx[k];
        dJ11 += vm[1 + 4 * k] * dMy[k]; dJ12 += vm[2 + 4 * k] * dMy[k];
    }
    const float tz = 1.0f / tc[2], tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = (float)gmx * -fx * tz2 * dJ02;
    const float dty = (float)gmy * -fy * tz2 * dJ12;
    const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * tc[0]) * tz3 * dJ02 + (2 * fy * tc[1]) * tz3 * dJ12;
    float dmean[3];
#pragma unroll
    for (int k = 0; k < 3; k++) dmean[k] = vm[0 + 4 * k] * dtx + vm[1 + 4 * k] * dty + vm[2 + 4 * k] * dtz;
    const float* pm = cam.proj;
    float mh[4];
    xform44(mean, pm, mh);
    const float mw = 1.0f / (mh[3] + 0.0000001f);
    const float mul1 = mh[0] * mw * mw, mul2 = mh[1] * mw * mw;
    const float g2x = dL_dmean2D[3 * i], g2y = dL_dmean2D[3 * i + 1];
    dmean[0] += (pm[0] * mw - pm[3] * mul1) * g2x + (pm[1] * mw - pm[3] * mul2) * g2y;
    dmean[1] += (pm[4] * mw - pm[7] * mul1) * g2x + (pm[5] * mw - pm[7] * mul2) * g2y;
    dmean[2] += (pm[8] * mw - pm[11] * mul1) * g2x + (pm[9] * mw - pm[11] * mul2) * g2y;
    if (use_sh) {
        const float dir0[3] = {mean[0] - cam.campos[0], mean[1] - cam.campos[1], mean[2] - cam.campos[2]};
        const float sum2 = dir0[0] * dir0[0] + dir0[1] * dir0[1] + dir0[2] * dir0[2];
        const float len = sqrtf(sum2);
        const float x = dir0[0] / len, y = dir0[1] / len, z = dir0[2] / len;
        const uint8_t cl = clamped[i];
        float dRGB[3], ddir[3] = {0, 0, 0};
#pragma unroll
        for (int ch = 0; ch < 3; ch++) dRGB[ch] = ((cl >> ch) & 1) ? 0.f : dL_dcolor[3 * i + ch];
        float Bv[16], Bx[16], By[16], Bz[16];
#pragma unroll
        for (int k = 0; k < 16; k++) { Bv[k] = 0; Bx[k] = 0; By[k] = 0; Bz[k] = 0; }
        const int D = cam.D;
        Bv[0] = SH_C0;
        if (D > 0) {
            Bv[1] = -SH_C1 * y; By[1] = -SH_C1;
            Bv[2] = SH_C1 * z;  Bz[2] = SH_C1;
            Bv[3] = -SH_C1 * x; Bx[3] = -SH_C1;
            if (D > 1) {
                const float xx = x * x, yy = y * y, zz = z * z;
                const float c0 = SH_C2[0], c1 = SH_C2[1], c2 = SH_C2[2], c3_ = SH_C2[3], c4 = SH_C2[4];
                Bv[4] = c0 * x * y; Bx[4] = c0 * y; By[4] = c0 * x;
                Bv[5] = c1 * y * z; By[5] = c1 * z; Bz[5] = c1 * y;
                Bv[6] = c2 * (2 * zz - xx - yy); Bx[6] = c2 * -2 * x; By[6] = c2 * -2 * y; Bz[6] = c2 * 4 * z;
                Bv[7] = c3_ * x * z; Bx[7] = c3_ * z; Bz[7] = c3_ * x;
                Bv[8] = c4 * (xx - yy); Bx[8] = c4 * 2 * x; By[8] = c4 * -2 * y;
                if (D > 2) {
                    const float e0 = SH_C3[0], e1 = SH_C3[1], e2 = SH_C3[2], e3 = SH_C3[3], e4 = SH_C3[4], e5 = SH_C3[5], e6 = SH_C3[6];
                    Bv[9] = e0 * y * (3 * xx - yy); Bx[9] = e0 * 6 * x * y; By[9] = e0 * (3 * xx - 3 * yy);
                    Bv[10] = e1 * x * y * z; Bx[10] = e1 * y * z; By[10] = e1 * x * z; Bz[10] = e1 * x * y;
                    Bv[11] = e2 * y * (4 * zz - xx - yy); Bx[11] = e2 * -2 * x * y; By[11] = e2 * (4 * zz - xx - 3 * yy); Bz[11] = e2 * 8 * y * z;
                    Bv[12] = e3 * z * (2 * zz - 3 * xx - 3 * yy); Bx[12] = e3 * -6 * x * z; By[12] = e3 * -6 * y * z; Bz[12] = e3 * (6 * zz - 3 * xx - 3 * yy);
                    Bv[13] = e4 * x * (4 * zz - xx - yy); Bx[13] = e4 * (4 * zz - 3 * xx - yy); By[13] = e4 * -2 * x * y; Bz[13] = e4 * 8 * x * z;
                    Bv[14] = e5 * z * (xx - yy); Bx[14] = e5 * 2 * x * z; By[14] = e5 * -2 * y * z; Bz[14] = e5 * (xx - yy);
                    Bv[15] = e6 * x * (xx - 3 * yy); Bx[15] = e6 * (3 * xx - 3 * yy); By[15] = e6 * -6 * x * y;
                }
            }
        }
        const int nb = (D + 1) * (D + 1);
        for (int k = 0; k < nb; k++)
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                const float shv = sh_row[3 * k + ch];
                sh_row[3 * k + ch] = Bv[k] * dRGB[ch];
                ddir[0] += Bx[k] * shv * dRGB[ch];
                ddir[1] += By[k] * shv * dRGB[ch];
                ddir[2] += Bz[k] * shv * dRGB[ch];
            }
        for (int k = 3 * nb; k < 3 * cam.M; k++) sh_row[k] = 0.f;  // coefficients above the active degree
        const float inv32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        dmean[0] += ((sum2 - dir0[0] * dir0[0]) * ddir[0] - dir0[1] * dir0[0] * ddir[1] - dir0[2] * dir0[0] * ddir[2]) * inv32;
        dmean[1] += (-dir0[0] * dir0[1] * ddir[0] + (sum2 - dir0[1] * dir0[1]) * ddir[1] - dir0[2] * dir0[1] * ddir[2]) * inv32;
        dmean[2] += (-dir0[0] * dir0[2] * ddir[0] - dir0[1] * dir0[2] * ddir[1] + (sum2 - dir0[2] * dir0[2]) * ddir[2]) * inv32;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) dL_dmean3D[3 * i + k] = dmean[k];
    if (use_scale_rot) {
        const float q[4] = {rotations[4 * i], rotations[4 * i + 1], rotations[4 * i + 2], rotations[4 * i + 3]};
        const float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
        const float mod = cam.scale_modifier;
        float R[9], A[9], Gs[9], dA[9], dR[9];
        quat_R(q, R);
#pragma unroll
        for (int r_ = 0; r_ < 3; r_++)
#pragma unroll
            for (int k = 0; k < 3; k++) A[3 * r_ + k] = R[3 * r_ + k] * (mod * s[k]);
        Gs[0] = o[0]; Gs[1] = 0.5f * o[1]; Gs[2] = 0.5f * o[2];
        Gs[3] = 0.5f * o[1]; Gs[4] = o[3]; Gs[5] = 0.5f * o[4];
        Gs[6] = 0.5f * o[2]; Gs[7] = 0.5f * o[4]; Gs[8] = o[5];
#pragma unroll
        for (int r_ = 0; r_ < 3; r_++)
#pragma unroll
            for (int k = 0; k < 3; k++) dA[3 * r_ + k] = 2 * (Gs[3 * r_] * A[k] + Gs[3 * r_ + 1] * A[3 + k] + Gs[3 * r_ + 2] * A[6 + k]);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            dL_dscale[3 * i + k] = mod * (dA[k] * R[k] + dA[3 + k] * R[3 + k] + dA[6 + k] * R[6 + k]);
#pragma unroll
            for (int r_ = 0; r_ < 3; r_++) dR[3 * r_ + k] = dA[3 * r_ + k] * (mod * s[k]);
        }
        const float r = q[0], x = q[1], y = q[2], z = q[3];
        dL_drot[4 * i] = 2 * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
        dL_drot[4 * i + 1] = 2 * (y * dR[1] + z * dR[2] + y * dR[3] - 2 * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2 * x * dR[8]);
        dL_drot[4 * i + 2] = 2 * (-2 * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2 * y * dR[8]);
        dL_drot[4 * i + 3] = 2 * (-2 * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2 * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
    }
}

// Workgroup = 128 Gaussians.  Their SH coefficients (and, on the way out, the SH gradients) are one contiguous 128 x 3M float
// block of shs / dL_dsh: it is staged through LDS with coalesced transfers (a lane reading or writing its own 192-byte row
// touches 48 different cache lines per wave instruction; measured 1.86 GB of HBM traffic for 0.5 GB of data).  Every output
// row is written by this kernel, zeros for culled Gaussians, so the caller does not clear 300 B per Gaussian beforehand.
#define PBW_BLOCK 128
#define PBW_MAXM 16
__global__ void __launch_bounds__(PBW_BLOCK) k_preprocess_bw(int P, GsCam cam, const float* __restrict__ means3D, const float* __restrict__ shs,
                                                             int use_sh, const float* __restrict__ scales, const float* __restrict__ rotations,
                                                             int use_scale_rot, const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped,
                                                             const float* __restrict__ cov3D, const float* __restrict__ dL_dmean2D,
                                                             const float* __restrict__ dL_dconic, const float* __restrict__ dL_dcolor,
                                                             float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
                                                             float* __restrict__ dL_dscale, float* __restrict__ dL_drot) {
    __shared__ float s_sh[PBW_BLOCK * (3 * PBW_MAXM + 1)];
    const int first = blockIdx.x * PBW_BLOCK, i = first + threadIdx.x;
    const int row_len = 3 * cam.M, pitch = row_len + 1;  // +1: rows start in different LDS banks
    const int count = min(PBW_BLOCK, P - first);
    const bool visible = i < P && radii[i] > 0;
    if (use_sh) {
        const float* src = shs + (size_t)first * row_len;
        for (int k = threadIdx.x; k < count * row_len; k += PBW_BLOCK) {
            const int r = k / row_len;
            s_sh[r * pitch + (k - r * row_len)] = src[k];
        }
        __syncthreads();
    }
    float* sh_row = s_sh + threadIdx.x * pitch;
    if (visible) {
        preprocess_bw_one(i, cam, means3D, sh_row, use_sh, scales, rotations, use_scale_rot, clamped, cov3D, dL_dmean2D, dL_dconic, dL_dcolor,
                          dL_dmean3D, dL_dcov3D, dL_dscale, dL_drot);
    } else if (i < P) {
        if (use_sh) for (int k = 0; k < row_len; k++) sh_row[k] = 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) dL_dmean3D[3 * i + k] = 0.f;
#pragma unroll
        for (int k = 0; k < 6; k++) dL_dcov3D[6 * i + k] = 0.f;
        if (use_scale_rot) {
#pragma unroll
            for (int k = 0; k < 3; k++) dL_dscale[3 * i + k] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) dL_drot[4 * i + k] = 0.f;
        }
    }
    if (use_sh) {
        __syncthreads();
        float* dst = dL_dsh + (size_t)first * row_len;
        for (int k = threadIdx.x; k < count * row_len; k += PBW_BLOCK) {
            const int r = k / row_len;
            dst[k] = s_sh[r * pitch + (k - r * row_len)];
        }
    }
}

// slices = waves; all of them should be resident at once: 160 KB of LDS per CU, (4 B x n_tiles) per wave, 256 CUs
int gs_bin_blocks(int P, int n_tiles) {
    int per_cu = (int)((160 * 1024) / ((int64_t)n_tiles * 4 + 512));
    per_cu = per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu);
    const int cap = per_cu * 256 < GS_SLICE_MAX ? per_cu * 256 : GS_SLICE_MAX;
    const int nb = (int)nrc_cdiv(P, 256);
    return nb < cap ? (nb > 0 ? nb : 1) : cap;
}
int gs_bin_chunk(int P, int n_tiles) { const int nb = gs_bin_blocks(P, n_tiles); return (int)(nrc_cdiv(nrc_cdiv(P > 0 ? P : 1, nb), 64) * 64); }
// binning workspace: [hist NB x n_tiles][part GS_GROUPS x n_tiles][keyA P][valA P][keyB P][valB P][pad][radix counts 256 x nblk4][tot 4 x 256]  (u32)
struct BinWs { uint32_t *hist, *part, *keyA, *valA, *keyB, *valB, *counts, *tot; int nblk, nblk4; };
int64_t gs_bin_ws_words(int P, int n_tiles) {
    const int64_t nblk = nrc_cdiv(P > 0 ? P : 1, RS_TILE);
    return (int64_t)(gs_bin_blocks(P, n_tiles) + GS_GROUPS) * n_tiles + 4 * (int64_t)(P > 0 ? P : 1) + 256 * (nblk + 4) + 1024 + 64;
}
BinWs gs_bin_ws(uint32_t* base, int P, int n_tiles) {
    BinWs w;
    const int64_t p1 = P > 0 ? P : 1;
    w.nblk = (int)nrc_cdiv(p1, RS_TILE);
    w.hist = base;
    w.part = w.hist + (int64_t)gs_bin_blocks(P, n_tiles) * n_tiles;
    w.keyA = w.part + (int64_t)GS_GROUPS * n_tiles;
    w.valA = w.keyA + p1; w.keyB = w.valA + p1; w.valB = w.keyB + p1;
    w.nblk4 = (w.nblk + 3) / 4 * 4;
    w.counts = base + ((w.valB + p1 - base) + 3) / 4 * 4;  // 16-byte aligned rows (base itself comes 256-byte aligned from the caller)
    w.tot = w.counts + (int64_t)256 * w.nblk4;
    return w;
}

int make_cam(GsCam& cam, int W, int H, int D, int M, const float* view, const float* proj, const float* campos, float tanx, float tany,
             float scale_modifier) {
    if (W < 1 || H < 1 || D < 0 || D > 3 || !view || !proj || !campos || !(tanx > 0.f) || !(tany > 0.f)) return NRC_ERR_INVALID;
    for (int k = 0; k < 16; k++) { cam.view[k] = view[k]; cam.proj[k] = proj[k]; }
    for (int k = 0; k < 3; k++) cam.campos[k] = campos[k];
    cam.tan_fovx = tanx; cam.tan_fovy = tany;
    cam.focal_x = W / (2.0f * tanx); cam.focal_y = H / (2.0f * tany);
    cam.scale_modifier = scale_modifier;
    cam.W = W; cam.H = H; cam.gx = (W + TILE - 1) / TILE; cam.gy = (H + TILE - 1) / TILE; cam.D = D; cam.M = M;
    return NRC_OK;
}

}  // namespace

extern "C" {

int64_t nrc_gs_bin_hist_bytes(int32_t P, int32_t W, int32_t H) {
    if (P < 0 || W < 1 || H < 1) return NRC_ERR_INVALID;
    const int64_t n_tiles = (int64_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    if (n_tiles > GS_MAX_LDS_TILES) return 0;  // global-atomic fallback: no histogram matrix
    return gs_bin_ws_words(P, (int)n_tiles) * (int64_t)sizeof(uint32_t);
}

int nrc_gs_preprocess(int32_t P, int32_t D, int32_t M, int32_t W, int32_t H, const float* means3D, const float* shs,
                      const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                      const float* rotations, const float* cov3D_precomp, const float* viewmatrix_host, const float* projmatrix_host,
                      const float* campos_host, float tan_fovx, float tan_fovy, int32_t* radii, float* depths, float* points_xy,
                      float* conic_opacity, float* rgb, uint8_t* clamped, float* cov3D, uint32_t* tiles_touched, uint32_t* tile_counts,
                      uint32_t* ranges, uint32_t* tile_fill, uint32_t* bin_hist, int64_t* num_rendered, nrc_stream_t stream) {
    NRC_ENTER();
    GsCam cam;
    const int rc = make_cam(cam, W, H, D, M, viewmatrix_host, projmatrix_host, campos_host, tan_fovx, tan_fovy, scale_modifier);
    if (rc != NRC_OK) return rc;
    if (P < 0 || !tile_counts || !ranges || !tile_fill || !num_rendered) return NRC_ERR_INVALID;
    if (P > 0) {
        if ((shs == nullptr) == (colors_precomp == nullptr)) return NRC_ERR_INVALID;                      // exactly one colour source
        if (((scales != nullptr) && (rotations != nullptr)) == (cov3D_precomp != nullptr)) return NRC_ERR_INVALID;  // exactly one covariance source
        if (shs && M < (D + 1) * (D + 1)) return NRC_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    const int n_tiles = cam.gx * cam.gy;
    const bool lds_path = n_tiles <= GS_MAX_LDS_TILES && bin_hist != nullptr;
    hipMemsetAsync(tile_counts, 0, sizeof(uint32_t) * n_tiles, s);
    if (P > 0) {
        if (!means3D || !opacities || !radii || !depths || !points_xy || !conic_opacity || !rgb || !clamped || !cov3D || !tiles_touched) return NRC_ERR_INVALID;
        if (shs && M > PRE_MAXM) return NRC_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(k_preprocess, dim3(nrc_cdiv(P, PRE_BLOCK)), dim3(PRE_BLOCK), 0, s, P, cam, means3D, shs, colors_precomp, opacities, scales, rotations,
                           cov3D_precomp, radii, depths, points_xy, conic_opacity, rgb, clamped, cov3D, tiles_touched,
                           lds_path ? (uint32_t*)nullptr : tile_counts);
        if (lds_path) {
            const int nb = gs_bin_blocks(P, n_tiles), chunk = gs_bin_chunk(P, n_tiles);
            const BinWs w = gs_bin_ws(bin_hist, P, n_tiles);
            // depth pre-sort of the Gaussians: 4 stable 8-bit passes, (keyA,valA) -> ... -> (keyA,valA); valA = depth order
            hipLaunchKernelGGL(k_depth_keys, dim3(nrc_cdiv(P, 256)), dim3(256), 0, s, P, radii, depths, w.keyA, w.valA);
            hipMemsetAsync(w.tot, 0, sizeof(uint32_t) * 4 * 256, s);
            for (int pass = 0; pass < 4; pass++) {
                const uint32_t *ki = (pass & 1) ? w.keyB : w.keyA, *vi = (pass & 1) ? w.valB : w.valA;
                uint32_t *ko = (pass & 1) ? w.keyA : w.keyB, *vo = (pass & 1) ? w.valA : w.valB;
                hipLaunchKernelGGL(k_radix_count, dim3(w.nblk), dim3(256), 0, s, P, 8 * pass, w.nblk4, ki, w.counts, w.tot + 256 * pass);
                hipLaunchKernelGGL(k_radix_scatter, dim3(w.nblk), dim3(256), 0, s, P, 8 * pass, w.nblk4, ki, vi, w.counts, w.tot + 256 * pass, ko, vo);
            }
            const dim3 cgrid((unsigned)nrc_cdiv(n_tiles, 64), GS_GROUPS / 4);
            hipLaunchKernelGGL(k_bin_count, dim3((nb + 7) / 8 * 8), dim3(64), n_tiles * sizeof(uint32_t), s, P, nb, chunk, cam.gx, cam.gy, w.valA, radii, points_xy, w.hist);
            hipLaunchKernelGGL(k_tile_totals, cgrid, dim3(256), 0, s, nb, n_tiles, w.hist, w.part);
            hipLaunchKernelGGL(k_scan_tiles_grouped, dim3(1), dim3(1024), 0, s, w.part, n_tiles, ranges, tile_fill, num_rendered);
            hipLaunchKernelGGL(k_tile_bases, cgrid, dim3(256), 0, s, nb, n_tiles, w.part, w.hist);
        }
    }
    if (!(P > 0 && lds_path)) hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, tile_counts, n_tiles, ranges, tile_fill, num_rendered);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_gs_bin_render(int32_t P, int32_t W, int32_t H, const float* bg_host, const int32_t* radii, const float* depths, const float* points_xy,
                      const float* conic_opacity, const float* rgb, const uint32_t* ranges, uint32_t* tile_fill, const uint32_t* bin_hist, uint64_t* keys,
                      int32_t* point_list, float* out_color, uint32_t* n_contrib, float* final_T, nrc_stream_t stream) {
    NRC_ENTER();
    if (P < 0 || W < 1 || H < 1 || !bg_host || !ranges || !tile_fill || !out_color || !n_contrib || !final_T) return NRC_ERR_INVALID;
    GsCam cam = {};
    cam.W = W; cam.H = H; cam.gx = (W + TILE - 1) / TILE; cam.gy = (H + TILE - 1) / TILE;
    hipStream_t s = (hipStream_t)stream;
    if (P > 0) {
        if (!radii || !depths || !points_xy || !conic_opacity || !rgb || !keys || !point_list) return NRC_ERR_INVALID;
        const int n_tiles = cam.gx * cam.gy;
        if (n_tiles <= GS_MAX_LDS_TILES && bin_hist) {
            const BinWs w = gs_bin_ws(const_cast<uint32_t*>(bin_hist), P, n_tiles);
            // the same depth-ordered walk as the counting pass, now with the scanned matrix rows as LDS cursors: ids land sorted
            static const hipError_t lds_attr = hipFuncSetAttribute((const void*)k_bin_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
            (void)lds_attr;  // tile grids above 12 K tiles need more than the default 64 KB of dynamic LDS
            const int nb = gs_bin_blocks(P, n_tiles);
            hipLaunchKernelGGL(k_bin_scatter, dim3((nb + 7) / 8 * 8), dim3(64), n_tiles * 4, s, P, nb, gs_bin_chunk(P, n_tiles), cam.gx, cam.gy,
                               w.valA, radii, points_xy, w.hist, point_list);
        } else {
            hipLaunchKernelGGL(k_scatter, dim3(nrc_cdiv(P, 256)), dim3(256), 0, s, P, cam.gx, cam.gy, radii, depths, points_xy, ranges, tile_fill, keys);
            hipLaunchKernelGGL((k_sort_tiles<0, 1024>), dim3(n_tiles), dim3(256), 0, s, ranges, keys, point_list);
            hipLaunchKernelGGL((k_sort_tiles<1024, 4096>), dim3(n_tiles), dim3(256), 0, s, ranges, keys, point_list);
            hipLaunchKernelGGL((k_sort_tiles<4096, SORT_LDS_CAP>), dim3(n_tiles), dim3(256), 0, s, ranges, keys, point_list);
        }
    }
    hipLaunchKernelGGL(k_render, dim3(cam.gx, cam.gy), dim3(256), 0, s, cam, ranges, point_list, points_xy, conic_opacity, rgb, bg_host[0],
                       bg_host[1], bg_host[2], out_color, n_contrib, final_T);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_gs_backward(int32_t P, int32_t D, int32_t M, int32_t W, int32_t H, const float* bg_host, const float* means3D, const float* shs,
                    const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                    const float* cov3D_precomp, const float* viewmatrix_host, const float* projmatrix_host, const float* campos_host,
                    float tan_fovx, float tan_fovy, const int32_t* radii, const float* points_xy, const float* conic_opacity,
                    const float* rgb, const uint8_t* clamped, const float* cov3D, const int32_t* point_list, const uint32_t* ranges,
                    const uint32_t* n_contrib, const float* final_T, const float* dL_dpix, float* dL_dmean2D, float* dL_dconic,
                    float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                    float* dL_drot, nrc_stream_t stream) {
    NRC_ENTER();
    GsCam cam;
    const int rc = make_cam(cam, W, H, D, M, viewmatrix_host, projmatrix_host, campos_host, tan_fovx, tan_fovy, scale_modifier);
    if (rc != NRC_OK) return rc;
    if (P < 0 || !bg_host) return NRC_ERR_INVALID;
    if (P == 0) return NRC_OK;
    if (!means3D || !radii || !points_xy || !conic_opacity || !rgb || !clamped || !cov3D || !point_list || !ranges || !n_contrib || !final_T ||
        !dL_dpix || !dL_dmean2D || !dL_dconic || !dL_dopacity || !dL_dcolor || !dL_dmean3D || !dL_dcov3D)
        return NRC_ERR_INVALID;
    const int use_sh = colors_precomp == nullptr, use_sr = cov3D_precomp == nullptr;
    if ((use_sh && (!shs || !dL_dsh)) || (use_sr && (!scales || !rotations || !dL_dscale || !dL_drot))) return NRC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    hipMemsetAsync(dL_dmean2D, 0, sizeof(float) * 3 * P, s);
    hipMemsetAsync(dL_dconic, 0, sizeof(float) * 4 * P, s);
    hipMemsetAsync(dL_dopacity, 0, sizeof(float) * P, s);
    hipMemsetAsync(dL_dcolor, 0, sizeof(float) * 3 * P, s);
    if (use_sh && M > PBW_MAXM) return NRC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_render_bw, dim3(cam.gx, cam.gy), dim3(256), 0, s, cam, ranges, point_list, points_xy, conic_opacity, rgb, bg_host[0],
                       bg_host[1], bg_host[2], n_contrib, final_T, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor);
    hipLaunchKernelGGL(k_preprocess_bw, dim3(nrc_cdiv(P, PBW_BLOCK)), dim3(PBW_BLOCK), 0, s, P, cam, means3D, shs, use_sh, scales, rotations, use_sr, radii,
                       clamped, cov3D, dL_dmean2D, dL_dconic, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

}  // extern "C"
