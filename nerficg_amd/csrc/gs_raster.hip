// gs_raster.hip -- 3D Gaussian Splatting tile rasterizer for gfx950, forward and backward.
//
// Replaces diff-gaussian-rasterization @ 59f5f77e (src/Thirdparty/DiffGaussianRasterization.py:9) behind the call sites of
// src/Methods/GaussianSplatting/Renderer.py:60-81,94-153,163-183.  Algorithm = Kerbl et al. 2023; conventions and the
// statement the kernels are checked against: oracle/gs_oracle_impl.h.
//
// BUILD NOTE: compiled with -ffp-contract=off.  Radii, tile rectangles and the depth keys that order the blend are integer
// decisions taken from f32 arithmetic and must be bit-exact against the oracle (which is built the same way).
//
// MI355X-native structure (vs the reference's pipeline: inclusive scan -> key duplication -> DEVICE-WIDE 64-bit radix sort over the D >> P
// (tile, Gaussian) instances -> range search, ~100 B of HBM traffic per instance).  DESIGN.md 3.3 has the history and the numbers.
//   1. k_preprocess     128 Gaussians per workgroup, SH rows staged through LDS: cull, cov3D, EWA cov2D, conic, radius, SH -> RGB, tile rectangle;
//                       writes the geometry buffers of the API and one 64-byte splat record per Gaussian (what the blend kernels read).
//   2. depth pre-sort   of the P Gaussians (not of the instances): k_depth_keys (keys, rectangles, the digit totals of all four passes, the span
//                       count of every tile row), then 4 x k_radix_pass -- ONE launch per 8-bit digit (round 4; rounds 1-3: count / offsets /
//                       scatter): a tile publishes its digit counts (value + flag in one sc1 word) and adds up what the earlier tiles published
//                       through a two-level prefix.  LSD, stable, carrying the Gaussian's index and its packed tile rectangle.
//   3. span binning     two-level MSD scatter over the tile id, ranked with LDS bit matrices (no serial walk, no sort):
//                       level 1 (k_span_sweep, one launch, the same published-count prefix): Gaussian -> one 8-byte span record per tile ROW;
//                       level 2 (k_item_count / k_item_scan / k_item_scatter): a row's spans -> ids appended to the row's tiles; the last workgroup
//                       of k_item_scan also scans the tile totals into `ranges` and orders the tiles by list length (longest lists launch first).
//                       Stable in depth order, so the per-tile lists equal what a stable radix sort over (tile | depth) keys produces.
//                       Tile grids above SPAN_DIM_MAX x SPAN_DIM_MAX fall back to k_scatter (global atomics) + k_sort_tiles (per-tile bitonic sort)
//   4.                  + k_scan_tiles + k_tile_order.  Forward: 21 launches in round 3, 12 now (preprocess, keys, 4 passes, sweep, 3 x item, render).
//   5. k_render         16x16-pixel tile per workgroup (4 waves = 8x8 quadrants), DPP row = 4x4 pixel block with its OWN culled work list,
//                       256-entry batches staged in LDS, branch-free front-to-back blend.
// Backward:
//   6. k_render_bw      same tiles and lists, back-to-front; per-Gaussian sums reduced by a transposing DPP butterfly, combined across the
//                       tile's blocks with ONE 64-bit fixed-point LDS atomic per row, then one global f32 atomic per (tile, Gaussian) and quantity.
//   7. k_preprocess_bw  128 Gaussians per workgroup: conic -> cov2D -> cov3D -> scale / rotation, mean2D -> mean3D, colour -> SH.
#include <hip/hip_runtime.h>

#include "common.h"
#include "adam_math.h"

#define TILE 16
#define BATCH 256
#define SORT_LDS_CAP 8192  // 64-bit keys sorted in LDS per tile (64 KB)

namespace {

struct GsCam {
    float view[16], proj[16], campos[3];
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
    int W, H, gx, gy, D, M;
    int raw;  // 1: opacities are logits, scales log-scales, rotations unnormalised (the activations of Gaussians.get_* run inside the kernels)
};

// Pose-dependent part of the camera from device memory (graph capture, cameras that never visit the host): 38 floats = view (16, column-major
// as in the settings = w2c^T row-major), proj (16), campos (3), background (3).  NULL: the values of the kernel argument stand.  The address
// is wave-uniform and the block read-only, so these are scalar loads like the kernel-argument loads they replace.
#define GS_POSE_FLOATS 38
__device__ __forceinline__ GsCam cam_with_pose(const GsCam& arg, const float* __restrict__ pose) {
    GsCam c = arg;
    if (pose) {
#pragma unroll
        for (int k = 0; k < 16; k++) { c.view[k] = pose[k]; c.proj[k] = pose[16 + k]; }
#pragma unroll
        for (int k = 0; k < 3; k++) c.campos[k] = pose[32 + k];
    }
    return c;
}

// activations of GaussianSplatting/Model.py:45-87 for the raw-parameter mode: exp (scales), x / max(|x|, 1e-12) (rotations), sigmoid (opacities)
__device__ __forceinline__ float act_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ void act_scales(const float* raw, int on, float* s) {
#pragma unroll
    for (int k = 0; k < 3; k++) s[k] = on ? expf(raw[k]) : raw[k];
}
__device__ __forceinline__ float act_rotation(const float* raw, int on, float* q) {  // returns the norm used (1 when off)
    float n = 1.f;
    if (on) n = fmaxf(sqrtf(raw[0] * raw[0] + raw[1] * raw[1] + raw[2] * raw[2] + raw[3] * raw[3]), 1e-12f);
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = on ? raw[k] / n : raw[k];
    return n;
}
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
// Splat record: what the two render kernels need of a Gaussian, in ONE 64-byte line (they used to gather xy, conic + opacity and rgb from
// three arrays = three cache lines per list entry), plus the per-Gaussian part of their block tests precomputed once instead of once per
// (tile, Gaussian) pair:  q0 = {X, Y, A, B}  q1 = {C, o, ext_x, ext_y}  q2 = {tau / A, det / A^2, B / A, y_off}  q3 = {r, g, b, -}
// (round 4: the colour has the last vector to itself -- the one part of the record that the SH evaluation produces; the colour pass of the
// preprocessing (k_preprocess<2>, side stream) writes it with one 16-byte store, the geometry pass (k_preprocess<1>) the other three vectors)
//   alpha >= 1/255  <=>  A dx^2 + 2 B dx dy + C dy^2 <= tau = 2 ln(255 o)   (the ellipse a pixel must be inside to blend the Gaussian)
//   ext_x, ext_y = half extents of that ellipse's bounding box (margins: x 1.001 + 0.01 px);  ext_x = -1: never blends;  +inf: not an ellipse
//   y_off = (B / C) sqrt(tau C / det): the ellipse's leftmost / rightmost points lie at Y +- y_off
__device__ __forceinline__ void write_splat_record(float4 (&rec)[4], float X, float Y, float A, float B, float C, float o, const float* col) {
    const float inf = __builtin_inff();
    float ex = -1.f, ey = -1.f, ta = 0.f, da = 0.f, det = 0.f, ba = 0.f, yoff = 0.f;
    if (o > 0.f) {
        const float tau = 2.f * (logf(255.f * o) + 1e-3f);
        if (tau > 0.f) {
            det = A * C - B * B;
            if (!(det > 0.f) || !(A > 0.f) || !(C > 0.f)) { ex = inf; ey = inf; }
            else {
                ex = sqrtf(tau * C / det) * 1.001f + 0.01f; ey = sqrtf(tau * A / det) * 1.001f + 0.01f;
                ta = tau / A; da = det / (A * A); ba = B / A; yoff = (B / C) * sqrtf(tau * C / det);
            }
        }
    }
    rec[0] = make_float4(X, Y, A, B); rec[1] = make_float4(C, o, ex, ey);
    rec[2] = make_float4(ta, da, ba, yoff); rec[3] = make_float4(col[0], col[1], col[2], 0.f);
}
// one Gaussian; sh_row = its SH coefficients (LDS copy, see k_preprocess).  The same computation whatever the template flags; GEOM: the geometry
// outputs are stored (radii, depths, screen position, conic + opacity, cov3D, tiles_touched, vectors 0..2 of the record), COLOR: the colour outputs
// (rgb, clamped, vector 3 of the record).  The two passes of a split preprocessing decide a Gaussian's visibility identically (same arithmetic).
template <bool GEOM, bool COLOR>
__device__ __forceinline__ void preprocess_one(int i, const GsCam& cam, const float* __restrict__ means3D, const float* sh_row,
                                               const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                                               const float* __restrict__ scales, const float* __restrict__ rotations,
                                               const float* __restrict__ cov3D_precomp, int32_t* __restrict__ radii,
                                               float* __restrict__ depths, float* __restrict__ points_xy,
                                               float* __restrict__ conic_opacity, float* __restrict__ rgb, uint8_t* __restrict__ clamped,
                                               float* __restrict__ cov3D, uint32_t* __restrict__ tiles_touched,
                                               uint32_t* __restrict__ tile_counts, bool want_record, float4 (&splat)[4]) {
    if (GEOM) {
        radii[i] = 0; tiles_touched[i] = 0; depths[i] = 0.f;
        points_xy[2 * i] = 0.f; points_xy[2 * i + 1] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) conic_opacity[4 * i + k] = 0.f;
    }
    if (COLOR) {
#pragma unroll
        for (int k = 0; k < 3; k++) rgb[3 * i + k] = 0.f;
        clamped[i] = 0;
    }
    const float p[3] = {means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2]};
    float pv[3];
    xform43(p, cam.view, pv);
    float c3[6];
    if (cov3D_precomp) {
#pragma unroll
        for (int k = 0; k < 6; k++) c3[k] = cov3D_precomp[6 * i + k];
    } else {
        float s[3], q[4];
        act_scales(scales + 3 * i, cam.raw, s);
        act_rotation(rotations + 4 * i, cam.raw, q);
        cov3d(s, cam.scale_modifier, q, c3);
    }
    if (GEOM) {
#pragma unroll
        for (int k = 0; k < 6; k++) cov3D[6 * i + k] = c3[k];
    }
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
    float col[3] = {0.f, 0.f, 0.f};
    if (COLOR) {
        if (colors_precomp) {
#pragma unroll
            for (int k = 0; k < 3; k++) col[k] = colors_precomp[3 * i + k];
        } else {
            uint8_t cl;
            sh_color(cam.D, p, cam.campos, sh_row, col, &cl);
            clamped[i] = cl;
        }
#pragma unroll
        for (int k = 0; k < 3; k++) rgb[3 * i + k] = col[k];
    }
    const float opacity = cam.raw ? act_sigmoid(opacities[i]) : opacities[i];
    if (want_record) write_splat_record(splat, pix[0], pix[1], conic[0], conic[1], conic[2], opacity, col);  // this thread's row of the LDS image
    if (!GEOM) return;
    depths[i] = pv[2]; radii[i] = my_radius;
    points_xy[2 * i] = pix[0]; points_xy[2 * i + 1] = pix[1];
    conic_opacity[4 * i] = conic[0]; conic_opacity[4 * i + 1] = conic[1]; conic_opacity[4 * i + 2] = conic[2];
    conic_opacity[4 * i + 3] = opacity;
    tiles_touched[i] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
    if (tile_counts) {  // fallback binning for tile grids too large for the LDS histograms
        for (int y = rmin[1]; y < rmax[1]; y++)
            for (int x = rmin[0]; x < rmax[0]; x++) atomicAdd(&tile_counts[y * cam.gx + x], 1u);
    }
}
// SH rows of `count` Gaussians starting at `first` <-> LDS rows of `pitch` floats.  One array (P, M, 3), or split like the model stores them
// (Model.py:28-31: dc (P, 1, 3) + rest (P, M-1, 3)) -- the (P, M, 3) concatenation Gaussians.get_features materialises per render
// (192 B read + written per Gaussian, and its backward split) is then never built.
// (Round 4, measured and dropped: the global side as 16-byte pieces -- four scalar LDS stores per piece, lanes four words apart: 4- to 8-way
// bank conflicts -- k_preprocess 109 us against 99 on the same box; as 8-byte pieces 144 us.  The 4-byte walk, whose lanes write consecutive LDS
// words, stays, although a stream of 4-byte loads only reaches 4.1 TB/s against 6.2 for wider ones: profiles/r04_fetch_calibration.md.
// Also built: the (P, 16, 3) block by LDS-DMA (global_load_lds_dwordx4) into a linear image, rows read with twelve 16-byte LDS reads -- colours
// bit-identical, k_preprocess 91.9 us against 92.9: the width of the SH loads is not what the kernel waits for.)
// (Round 5: the row / column of element k used to be k / row_len and the remainder -- an integer division by a runtime value is ~25 vector
// instructions, 48 of them per thread and direction: more than the whole per-Gaussian arithmetic of these kernels.  k advances by the block size, so
// (row, column) advance by its quotient and remainder: one division per call, four instructions per element.)
#ifndef SHB_U
#define SHB_U 16   // loads per trip: 8 / 12 / 16 -> k_preprocess 101 / 99 / 95-98 us at 1 M Gaussians (k_preprocess_bw indifferent)
#endif
template <bool TO_LDS, int SHB_U_ = SHB_U, bool BATCH_TAIL = true>
__device__ __forceinline__ void sh_block_copy(float* s_sh, int pitch, int col0, int len, int count, float* g, int nthreads) {
    constexpr int U = SHB_U_;
    const int dq = nthreads / len, dr = nthreads - dq * len;
    int r = (int)threadIdx.x / len, c = (int)threadIdx.x - r * len;
    const int n = count * len;
    int k = threadIdx.x;
    // U elements per trip, their loads issued together (the compiler does not batch them across the carried (row, column) by itself:
    // one load per trip and a wait behind it ran at half the speed of the divisions it replaced)
    for (; k + (U - 1) * nthreads < n; k += U * nthreads) {
        float v[U];
        int off[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            off[u] = r * pitch + col0 + c;
            v[u] = TO_LDS ? g[k + u * nthreads] : s_sh[off[u]];
            r += dq; c += dr;
            if (c >= len) { c -= len; r++; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) { if (TO_LDS) s_sh[off[u]] = v[u]; else g[k + u * nthreads] = v[u]; }
    }
    // the rest (a block of 45-float rows -- the model's split `rest` tensor -- leaves 13 of 45 elements per thread, the 3-float dc rows all 3): ONE more
    // batched trip with predicated accesses.  Round 6: this used to be a one-load-one-wait loop, 13 + 3 dependent round trips per thread -- the split
    // form of the training step cost k_preprocess +24 us and k_preprocess_bw +31 us over the concatenated (P, 16, 3) form (tools/exp_gs_pre.py)
    if constexpr (!BATCH_TAIL) {   // whole (P, M, 3) rows: 48 floats per thread are three full trips of 16; only the last workgroup of a launch has a remainder
        for (; k < n; k += nthreads) {
            if (TO_LDS) s_sh[r * pitch + col0 + c] = g[k]; else g[k] = s_sh[r * pitch + col0 + c];
            r += dq; c += dr;
            if (c >= len) { c -= len; r++; }
        }
    } else if (k < n) {
        float v[U];
        int off[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool in = k + u * nthreads < n;
            off[u] = r * pitch + col0 + c;
            v[u] = 0.f;
            if (in) v[u] = TO_LDS ? g[k + u * nthreads] : s_sh[off[u]];
            r += dq; c += dr;
            if (c >= len) { c -= len; r++; }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (k + u * nthreads < n) { if (TO_LDS) s_sh[off[u]] = v[u]; else g[k + u * nthreads] = v[u]; }
        }
    }
}
// SPLIT: the form is a compile-time property of the kernel instance (round 6: one kernel carrying both forms spilled scalar registers in the path of the
// other -- the concatenated form of the bench frame lost 10 % to code it never runs)
template <bool TO_LDS, int SHB_U_ = SHB_U, bool SPLIT = false>
__device__ __forceinline__ void sh_rows_copy(float* s_sh, int pitch, int row_len, int count, size_t first, float* sh, float* sh_rest, int nthreads) {
    if constexpr (!SPLIT) {
        sh_block_copy<TO_LDS, SHB_U, false>(s_sh, pitch, 0, row_len, count, sh + first * row_len, nthreads);
    } else {
        // the three dc floats per Gaussian ride along with the first trip of the `rest` block: their loads are issued in front of it and land in LDS
        // behind it (a separate one-trip copy was a dependent round trip of its own, per direction, in kernels that run a few waves per CU)
        float* dc = sh + first * 3;
        const int n3 = count * 3;
        float dv[3];
        int doff[3];
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int k = (int)threadIdx.x + u * nthreads;
            const int r = k / 3, c = k - 3 * r;
            doff[u] = r * pitch + c;
            dv[u] = 0.f;
            if (k < n3) dv[u] = TO_LDS ? dc[k] : s_sh[doff[u]];
        }
        if (!TO_LDS) {
#pragma unroll
            for (int u = 0; u < 3; u++) { const int k = (int)threadIdx.x + u * nthreads; if (k < n3) dc[k] = dv[u]; }
        }
        const int rl = row_len - 3;
        if (rl > 0) sh_block_copy<TO_LDS, SHB_U_>(s_sh, pitch, 3, rl, count, sh_rest + first * rl, nthreads);
        if (TO_LDS) {
#pragma unroll
            for (int u = 0; u < 3; u++) { const int k = (int)threadIdx.x + u * nthreads; if (k < n3) s_sh[doff[u]] = dv[u]; }
        }
    }
}
#define SPAN_DIM_MAX 256     // tile rows / columns the lane-private cursors cover (4 registers x 64 lanes): 4096 x 4096 pixels
// Header words of the sort, zeroed by the first workgroup of k_preprocess (the kernel in front): [digit totals 4 x 256][row span totals 256][tile tickets 4 + 4 spare]
#define RS_HDR_TOT 0
#define RS_HDR_ROWS 1024
#define RS_HDR_TICKET (1024 + 256)
#define RS_HDR_WORDS (1024 + 256 + 8)
// Workgroup = 128 Gaussians; their SH coefficients (one contiguous 128 x 3M float block) are staged through LDS with coalesced
// loads -- a lane walking its own 192-byte row touches 48 cache lines per wave instruction (same staging as k_preprocess_bw).
#ifndef PRE_BLOCK
#define PRE_BLOCK 128   // measured 64 / 128 / 256 in round 5: see LABBOOK A.4b
#endif
#ifndef PRE_MAXM
#define PRE_MAXM 16
#endif
// PART 0: everything in one pass (tile grids without the span binning, graph captures, armed stage timer).  PART 1: the geometry outputs -- what
// the depth sort and the binning wait for (44 B read per Gaussian instead of 236).  PART 2: the colour outputs (rgb, clamped, vector 3 of the
// records): the same per-Gaussian computation, the 192-byte SH rows staged through LDS, on a FEW persistent workgroups (grid-stride over the
// 128-Gaussian blocks) of a side stream, so that it runs NEXT TO the geometry pass, the depth sort and the binning, whose kernels are
// latency-bound and leave the memory system idle, without taking their compute units (round 4; joined before nrc_gs_preprocess returns).
template <int PART, bool SPLIT = false>
__global__ void __launch_bounds__(PRE_BLOCK) k_preprocess(int P, GsCam cam_arg, const float* __restrict__ pose, const float* __restrict__ means3D, const float* __restrict__ shs,
                                                          const float* __restrict__ shs_rest,
                                                          const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                                                          const float* __restrict__ scales, const float* __restrict__ rotations,
                                                          const float* __restrict__ cov3D_precomp, int32_t* __restrict__ radii,
                                                          float* __restrict__ depths, float* __restrict__ points_xy,
                                                          float* __restrict__ conic_opacity, float* __restrict__ rgb, uint8_t* __restrict__ clamped,
                                                          float* __restrict__ cov3D, uint32_t* __restrict__ tiles_touched,
                                                          uint32_t* __restrict__ tile_counts, float4* __restrict__ splat, uint32_t* __restrict__ sort_hdr) {
    __shared__ __attribute__((aligned(16))) float s_sh[PART == 1 ? PRE_BLOCK * 12 : PRE_BLOCK * (3 * PRE_MAXM + 1)];
    if (PART != 2 && sort_hdr && blockIdx.x == 0)   // digit totals, row totals and tile tickets of the depth sort behind this kernel start at zero
        for (int k = threadIdx.x; k < RS_HDR_WORDS; k += PRE_BLOCK) sort_hdr[k] = 0u;
    const GsCam cam = cam_with_pose(cam_arg, pose);
    const int row_len = 3 * cam.M, pitch = row_len + 1;
    const int n_blocks = (P + PRE_BLOCK - 1) / PRE_BLOCK;
    for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const int first = blk * PRE_BLOCK, i = first + threadIdx.x;
        if (PART != 1 && shs) {
            sh_rows_copy<true, SHB_U, SPLIT>(s_sh, pitch, row_len, min(PRE_BLOCK, P - first), (size_t)first, const_cast<float*>(shs), const_cast<float*>(shs_rest), PRE_BLOCK);
            __syncthreads();
        }
        // splat records leave through LDS: a lane storing its own record touches 64 cache lines per store instruction; from the LDS image (the SH
        // staging area, free again once every thread has evaluated its colour) the workgroup writes its block with coalesced 16-byte stores.
        // Rows of culled Gaussians hold zeros or stale bytes: no tile list ever names them.
        float4 rec[4] = {};
        if (i < P)
            preprocess_one<PART != 2, PART != 1>(i, cam, means3D, s_sh + threadIdx.x * pitch, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii, depths,
                                                 points_xy, conic_opacity, rgb, clamped, cov3D, tiles_touched, tile_counts, splat != nullptr, rec);
        if (splat) {
            if (PART == 2) {   // vector 3 only: one 16-byte store per Gaussian
                if (i < P) splat[4 * (size_t)i + 3] = rec[3];
            } else {
                constexpr int NV = PART == 0 ? 4 : 3;   // vectors of a record this pass writes
                float4* s_rec = reinterpret_cast<float4*>(s_sh);
                __syncthreads();
#pragma unroll
                for (int k = 0; k < NV; k++) s_rec[NV * threadIdx.x + k] = rec[k];
                __syncthreads();
                const int nv = NV * min(PRE_BLOCK, P - first);
                float4* dst = splat + 4 * (size_t)first;
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    const int e = threadIdx.x + PRE_BLOCK * k;
                    if (e < nv) dst[NV == 4 ? e : 4 * (e / 3) + e % 3] = s_rec[e];
                }
            }
        }
        if (PART == 2) __syncthreads();   // the SH rows of the next block overwrite the staging area
    }
}

// ---- binning without global atomics and without a per-tile sort ----------------------------------------------------------
// History (profiles/): v1 global atomics onto 4 346 tile counters = 2.4 ms of a 4 ms forward; v2 LDS histograms per
// workgroup slice + (tile|depth) keys + per-tile bitonic sorts; v3 depth pre-sort + LDS-atomic scatter + segmented finishing
// sort (0.39 ms); v4 (this one) depth pre-sort + STABLE wave walk, which makes the scatter itself produce the sorted lists.
// ---- depth pre-sort of the P Gaussians (LSD radix, 4 x 8 bits, stable: equal depths keep ascending index) ---------------
// Sorting the P Gaussians once by depth (8 B x P x 4 passes) replaces sorting the D >> P instances per tile: the binning
// passes below take their slices from the depth-ordered list and keep that order inside every tile.
#ifndef RS_THREADS
#define RS_THREADS 512               // threads per tile: 8 waves rank 512 keys each (the rounds of a wave are a serial chain: 8 rounds instead of 16)
#endif
#define RS_WAVES (RS_THREADS / 64)
#define RS_TILE 4096                 // keys per tile (one workgroup)
#define RS_ITEMS (RS_TILE / RS_THREADS)   // keys per thread
#define RS_GROUP 64                  // tiles per group of the two-level prefix over the tiles
#define RS_PUB 0x80000000u           // "published" bit of a status word (counts stay below 2^31)
// sort key (depth bits; invisible Gaussians last), the Gaussian's index, and its tile rectangle packed as x0 | y0 << 8 | (w-1) << 16 | (h-1) << 24
// (tile grids of at most 256 x 256; RECT_NONE = no tile).  The rectangle travels through the sort as a second value, so the binning passes
// stream (id, rectangle) in depth order instead of gathering radii / points_xy by id (246 us at 6 M as a separate gather kernel).
#define RECT_NONE 0xffffffffu
// Round 4: ONE kernel per digit instead of three (count / offsets / scatter), and all 16 keys of a thread requested up front instead of one
// round ahead (the old scatter's 16 rounds were 16 dependent global-load latencies: 19.6 us for 4 096 keys per workgroup).
//   k_depth_keys   keys, rectangles, the digit totals of ALL four passes (a key's digits do not depend on where the sort has put it) and the
//                  number of spans per tile row (LDS histograms, one global atomic per non-empty bin and workgroup);
//   k_radix_pass   tile t = a ticket (tiles wait for LOWER tickets only: whoever holds one is running).  The tile counts its digits in LDS and
//                  PUBLISHES the 256 counts (one sc1 store per thread, value and "published" bit in one word: no fence), ranks its keys into LDS
//                  (stable, as before) while the other tiles publish, then adds up what the earlier tiles of its GROUP of 64 published (<= 63
//                  independent sc1 loads per thread, issued in batches of 16; a word that is not there yet is polled again) and the totals of the
//                  earlier groups (published by each group's last tile as soon as it has its own sum): a dependency chain of depth two,
//                  whatever the number of tiles.  A serial decoupled look-back (the usual one-sweep formulation) would walk through every
//                  concurrent predecessor -- all 245 tiles of a 1 M sort are resident at once on this chip -- at one L2 round trip per tile.
// 1 M Gaussians: 4 x (9.5 + 4.9 + 19.6) + 6.4 us in 13 launches -> 5 launches (numbers in DESIGN.md section 3.3).
#define DK_BLOCK 256
#define DK_ITEMS 16
__global__ void __launch_bounds__(DK_BLOCK) k_depth_keys(int P, int gx, int gy, const int32_t* __restrict__ radii, const float* __restrict__ depths,
                                                         const float* __restrict__ points_xy, uint32_t* __restrict__ keys, uint32_t* __restrict__ rects,
                                                         uint32_t* __restrict__ hdr, uint32_t* __restrict__ status, int64_t n_status) {
    __shared__ uint32_t h[4][256];
    __shared__ uint32_t rows[SPAN_DIM_MAX];
#pragma unroll
    for (int k = 0; k < 4; k++) h[k][threadIdx.x] = 0u;
    rows[threadIdx.x] = 0u;
    // the status words of the four passes start unpublished
    for (int64_t k = (int64_t)blockIdx.x * DK_BLOCK + threadIdx.x; k < n_status; k += (int64_t)gridDim.x * DK_BLOCK) status[k] = 0u;
    __syncthreads();
    const int base = blockIdx.x * (DK_BLOCK * DK_ITEMS);
    int rad[DK_ITEMS]; float dep[DK_ITEMS]; float2 xy[DK_ITEMS];
#pragma unroll
    for (int r = 0; r < DK_ITEMS; r++) {
        const int i = base + r * DK_BLOCK + threadIdx.x;
        rad[r] = i < P ? radii[i] : 0;
        dep[r] = i < P ? depths[i] : 0.f;
        xy[r] = i < P ? reinterpret_cast<const float2*>(points_xy)[i] : make_float2(0.f, 0.f);
    }
    uint32_t top = 0xffffffffu, top_run = 0u;  // the top digit takes a handful of values: runs of equal digits of a thread's keys are added as one
#pragma unroll
    for (int r = 0; r < DK_ITEMS; r++) {
        const int i = base + r * DK_BLOCK + threadIdx.x;
        if (i >= P) break;
        uint32_t rect = RECT_NONE;
        if (rad[r] > 0) {
            const float pxy[2] = {xy[r].x, xy[r].y};
            int rmin[2], rmax[2];
            tile_rect(pxy, rad[r], gx, gy, rmin, rmax);
            if (rmax[0] > rmin[0] && rmax[1] > rmin[1]) {
                rect = (uint32_t)rmin[0] | (uint32_t)rmin[1] << 8 | (uint32_t)(rmax[0] - rmin[0] - 1) << 16 | (uint32_t)(rmax[1] - rmin[1] - 1) << 24;
                for (int y = rmin[1]; y < rmax[1]; y++) atomicAdd(&rows[y], 1u);
            }
        }
        const uint32_t key = rad[r] > 0 ? __float_as_uint(dep[r]) : 0xffffffffu;  // invisible Gaussians go last
        keys[i] = key;
        rects[i] = rect;
        atomicAdd(&h[0][key & 255u], 1u);
        atomicAdd(&h[1][(key >> 8) & 255u], 1u);
        atomicAdd(&h[2][(key >> 16) & 255u], 1u);
        const uint32_t d3 = key >> 24;
        if (d3 != top) {
            if (top_run) atomicAdd(&h[3][top], top_run);
            top = d3; top_run = 0u;
        }
        top_run++;
    }
    if (top_run) atomicAdd(&h[3][top], top_run);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t c = h[k][threadIdx.x];
        if (c) atomicAdd(&hdr[RS_HDR_TOT + 256 * k + threadIdx.x], c);
    }
    if ((int)threadIdx.x < gy && rows[threadIdx.x]) atomicAdd(&hdr[RS_HDR_ROWS + threadIdx.x], rows[threadIdx.x]);
}
__device__ __forceinline__ uint32_t rs_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void rs_publish(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v | RS_PUB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// sum over words [first, first + n) x stride of a published column (this thread's digit); words not yet published are polled again
__device__ __forceinline__ uint32_t rs_sum_published(const uint32_t* col, int n, int stride) {
    uint32_t sum = 0u;
    for (int t0 = 0; t0 < n; t0 += 16) {
        uint32_t v[16];
        bool ok;
        do {
            ok = true;
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = t0 + k < n ? rs_load(col + (size_t)(t0 + k) * stride) : RS_PUB;
#pragma unroll
            for (int k = 0; k < 16; k++) ok = ok && (v[k] & RS_PUB);
            if (!ok) __builtin_amdgcn_s_sleep(4);
        } while (!ok);
#pragma unroll
        for (int k = 0; k < 16; k++) sum += v[k] & ~RS_PUB;
    }
    return sum;
}
// vals_in == nullptr: the values are the positions themselves (first pass); keys_out == nullptr: the keys are not needed any more (last pass).
// A tile is a TICKET (one returning atomic per workgroup): a tile waits for LOWER tickets only, and whoever holds a ticket is running -- no
// assumption about dispatch order or about how many workgroups are resident (other kernels, other processes on the chip).  Measured against
// tile = blockIdx at 1 M Gaussians (245 tiles, all resident): 15.4 against 16.2 us per pass -- the ticket order is the start order, so the tiles
// a tile waits for tend to have published already.
// Ranking: a wave owns 64 * RS_ITEMS CONSECUTIVE keys of the tile (round r = 64 consecutive keys) and ranks them against a wave-private table of
// running digit counts -- 8 ballots find a lane's peers, the first peer advances the count -- so the rounds need no workgroup barrier (the old
// scatter had one per round, 16 of them); afterwards thread d adds up the four waves' counts of digit d (published at once), one scan gives the
// digit runs, the per-wave counts turn into the waves' offsets inside the runs, and every key goes to run start + wave offset + its rank:
// wave-major, round-major, lane-major = ascending position = stable.  The 63 status words of the own group are requested BEFORE the keys are
// staged and looked at afterwards.
#if defined(NRC_SORT_PROBE)   // developer build (tools/build_variant.sh): shader-clock stamps of thread 0 of three tiles at the phases of the LAST pass
__device__ unsigned long long g_sort_probe[3][16];
#define RS_PROBE(k) do { if (pass == 3 && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == 100 || blockIdx.x == gridDim.x - 1)) \
    g_sort_probe[blockIdx.x == 0 ? 0 : (blockIdx.x == 100 ? 1 : 2)][(k)] = __builtin_readcyclecounter(); } while (0)
#else
#define RS_PROBE(k) do { } while (0)
#endif
__global__ void __launch_bounds__(RS_THREADS) k_radix_pass(int n, int shift, int pass, int n_tiles, const uint32_t* __restrict__ keys_in,
                                                           const uint32_t* __restrict__ vals_in, const uint32_t* __restrict__ rects_in,
                                                           uint32_t* __restrict__ hdr, uint32_t* __restrict__ status, uint32_t* __restrict__ gstat,
                                                           uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t* __restrict__ rects_out) {
    __shared__ uint32_t first_g[256];           // global position of the tile's first element with digit d
    __shared__ uint32_t run0[256];              // position of the digit's run inside the staged tile
    __shared__ uint32_t wcnt[RS_WAVES][256];    // per wave: running count of the digit, later the wave's offset inside the tile
    __shared__ int scan_sm[8];
    __shared__ int tile_s;
    __shared__ uint32_t st_key[RS_TILE], st_val[RS_TILE], st_rect[RS_TILE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool digit_thread = threadIdx.x < 256;   // threads 0..255 also look after one digit each
    RS_PROBE(0);
    if (threadIdx.x == 0) tile_s = (int)atomicAdd(&hdr[RS_HDR_TICKET + pass], 1u);
    for (int k = threadIdx.x; k < RS_WAVES * 256; k += RS_THREADS) (&wcnt[0][0])[k] = 0u;
    __syncthreads();
    const int tile = tile_s;
    const int base = tile * RS_TILE, wbase = base + wave * (64 * RS_ITEMS) + lane;
    uint32_t key[RS_ITEMS], val[RS_ITEMS], rect[RS_ITEMS], rk[RS_ITEMS];
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const int i = wbase + r * 64;
        key[r] = i < n ? keys_in[i] : 0u;
        val[r] = i < n ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;
        rect[r] = i < n ? rects_in[i] : 0u;
    }
    const uint32_t digit_total = digit_thread ? hdr[RS_HDR_TOT + 256 * pass + threadIdx.x] : 0u;
    uint32_t* mine = wcnt[wave];
    RS_PROBE(1);
#if defined(NRC_SORT_PROBE)
    if (key[0] == 0x12345678u && key[RS_ITEMS - 1] == 0x9abcdefu) __builtin_amdgcn_s_sleep(1);   // keeps the loads in front of stamp 2
    __builtin_amdgcn_s_waitcnt(0);
#endif
    RS_PROBE(2);
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        const bool valid = wbase + r * 64 < n;
        const uint32_t d = (key[r] >> shift) & 255u;
        unsigned long long peers = __ballot(valid);   // lanes of this wave holding the same digit (8 ballots); rank among them in lane order = stable
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank_w = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        const uint32_t before = mine[d];
        rk[r] = before + rank_w;
        if (valid && rank_w == 0) mine[d] = before + (uint32_t)__popcll(peers);   // LDS operations of one wave execute in order: the next round reads this
    }
    RS_PROBE(3);
    __syncthreads();
    RS_PROBE(4);
    // thread d: the waves' counts of digit d -> the tile's count (published at once), the digit runs (scan) and the waves' offsets inside them
    uint32_t my_count = 0u;
    uint32_t c[RS_WAVES];
    if (digit_thread) {
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) { c[w] = wcnt[w][threadIdx.x]; my_count += c[w]; }
        rs_publish(status + (size_t)tile * 256 + threadIdx.x, my_count);
    }
    // two exclusive scans over the 256 digits at once (tile counts -> run starts, digit totals -> digit bases): waves 0..3 scan, everyone meets at the barriers
    uint32_t r0 = 0u, digit_base = 0u;
    {
        int inc_a = 0, inc_b = 0;
        if (digit_thread) {
            inc_a = nrc_wave_incl_sum_i((int)my_count, lane); inc_b = nrc_wave_incl_sum_i((int)digit_total, lane);
            if (lane == 63) { scan_sm[wave] = inc_a; scan_sm[4 + wave] = inc_b; }
        }
        __syncthreads();
        if (digit_thread) {
            int ba = 0, bb = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) { if (w < wave) { ba += scan_sm[w]; bb += scan_sm[4 + w]; } }
            r0 = (uint32_t)(ba + inc_a) - my_count; digit_base = (uint32_t)(bb + inc_b) - digit_total;
            run0[threadIdx.x] = r0;
            uint32_t o = r0;
#pragma unroll
            for (int w = 0; w < RS_WAVES; w++) { wcnt[w][threadIdx.x] = o; o += c[w]; }
        }
        __syncthreads();
    }
    RS_PROBE(5);
    // What the earlier tiles hold of this thread's digit: the tiles of the own group of 64 one by one, the earlier groups by their totals.  The
    // words of the own group are requested now.  The group's LAST tile sums and publishes the group total at once (every tile of every later
    // group waits for it); the others stage their keys first and look at the words afterwards.
    const int group = tile / RS_GROUP, g0 = group * RS_GROUP, n_in = tile - g0;
    const bool group_leader = tile == g0 + RS_GROUP - 1 || tile == n_tiles - 1;
    const uint32_t* col = status + (size_t)g0 * 256 + threadIdx.x;
    uint32_t pv[RS_GROUP];
    uint32_t in_group = 0u;
    if (digit_thread) {
#pragma unroll
        for (int k = 0; k < RS_GROUP; k++) pv[k] = k < n_in ? rs_load(col + (size_t)k * 256) : RS_PUB;
        if (group_leader) {
            bool all_there = true;
#pragma unroll
            for (int k = 0; k < RS_GROUP; k++) { all_there = all_there && (pv[k] & RS_PUB); in_group += pv[k] & ~RS_PUB; }
            if (!all_there) in_group = rs_sum_published(col, n_in, 256);
            rs_publish(gstat + (size_t)group * 256 + threadIdx.x, in_group + my_count);
        }
    }
#pragma unroll
    for (int r = 0; r < RS_ITEMS; r++) {
        if (wbase + r * 64 < n) {
            const uint32_t off = mine[(key[r] >> shift) & 255u] + rk[r];
            st_key[off] = key[r]; st_val[off] = val[r]; st_rect[off] = rect[r];
        }
    }
    RS_PROBE(6);
    if (digit_thread) {
        if (!group_leader) {
            bool all_there = true;
#pragma unroll
            for (int k = 0; k < RS_GROUP; k++) { all_there = all_there && (pv[k] & RS_PUB); in_group += pv[k] & ~RS_PUB; }
            RS_PROBE(7);
            if (!all_there) in_group = rs_sum_published(col, n_in, 256);
        }
        RS_PROBE(8);
        const uint32_t groups_before = rs_sum_published(gstat + threadIdx.x, group, 256);
        RS_PROBE(9);
        first_g[threadIdx.x] = digit_base + groups_before + in_group;
    }
    __syncthreads();
    RS_PROBE(10);
    const int n_here = min(RS_TILE, n - base);
    for (int k = threadIdx.x; k < n_here; k += RS_THREADS) {
        const uint32_t kk = st_key[k];
        const uint32_t d = (kk >> shift) & 255u;
        const uint32_t pos = first_g[d] + ((uint32_t)k - run0[d]);
        if (keys_out) keys_out[pos] = kk;
        vals_out[pos] = st_val[k]; rects_out[pos] = st_rect[k];
    }
    RS_PROBE(11);
#if defined(NRC_SORT_PROBE)
    __builtin_amdgcn_s_waitcnt(0);
#endif
    RS_PROBE(12);
}

// ---- stable two-level binning of the depth-ordered Gaussians ---------------------------------------------------------------
// Target: per tile, the ids of the Gaussians whose rectangle covers it, in depth order (= the reference's sort by (tile | depth) keys),
// written as full cache lines.  History: v1 global atomics; v2 LDS histograms + per-tile sorts; v3 depth pre-sort + finishing sort;
// v4 depth pre-sort + one wave per slice of the depth order walking its slice serially with per-(slice, tile) cursors -- correct and
// sort-free, but every one of its 4-byte id stores opened its own cache line (slices x tiles = 8.9 M write frontiers: WRITE_SIZE 7.8 x the
// list, 138 us at 1 M / 1.09 ms at 6 M Gaussians, profiles/r01_pmc_summary.md).  v5 (this one) scatters in two levels, like a two-digit
// MSD radix sort over the tile id whose first digit is the TILE ROW and whose records are row SPANS, not instances:
//   level 1  Gaussian -> one 8-byte span record {id, x0 | x1 << 16} per tile row of its rectangle, appended to the row's span list
//            (a rectangle of 3 x 3 tiles is 3 records, not 9);
//   level 2  the span list of a row, cut into items of 256 .. 1024 spans (one wave each; the size is picked on the device from the span total,
//            so that a small frame still fills the chip) -> ids appended to the tiles x0 .. x1 - 1 of that row.
// Both levels rank with a BIT MATRIX instead of a serial walk: 64 sources (lane = Gaussian / span, in depth order) set bit `lane` in the
// LDS word of every destination (row / tile) they cover with ds_or -- the order of the ORs is irrelevant --, and every source then writes its
// record to each destination at (the destination's cursor) + (number of lower lanes whose bit is set in the destination's word): lane order =
// depth order, so the appends are stable and no step waits for another.  A wave has gy (level 1) or gx (level 2) open write frontiers and
// the level-2 grid is capped at 2048 waves, so the lines being appended stay in the XCD's L2 until they are full.
// Counting passes of the same shape (LDS ds_add per covered destination) give the cursors; everything is deterministic.
#define GS_MAX_LDS_TILES 16384
#define SPAN_CH_MIN 256      // spans per level-2 item (one wave): 256, 512 or 1024, picked from the span total (k_span_sweep, workgroup 0)
#define SPAN_CH_MAX 1024
#ifndef SPAN_GRID
#define SPAN_GRID 2048       // waves of the level-2 scatter (grid-stride over the items): few enough that the lines being appended stay in L2
#endif
#ifndef SPAN_GRID_COUNT
#define SPAN_GRID_COUNT 4096 // waves of the level-2 counting pass (no write frontiers to keep: 20 -> 15 us at 1 M Gaussians, 91 -> 55 at 6 M)
#endif

struct SpanRect { int id, y0, y1, x01; bool ok; };
// Gaussian j of the depth order: id and rectangle as the sort delivered them
__device__ __forceinline__ SpanRect span_rect(int j, int end, const uint32_t* __restrict__ order, const uint32_t* __restrict__ rects) {
    SpanRect r = {0, 0, 0, 0, false};
    if (j < end) {
        const uint32_t q = rects[j];
        r.id = (int)order[j];
        if (q != RECT_NONE) {
            const int x0 = (int)(q & 0xffu), y0 = (int)((q >> 8) & 0xffu);
            r.y0 = y0; r.y1 = y0 + (int)(q >> 24) + 1; r.x01 = x0 | ((x0 + (int)((q >> 16) & 0xffu) + 1) << 16);
            r.ok = true;
        }
    }
    return r;
}
// value of entry `idx` (wave-uniform) of a table held as 4 registers x 64 lanes
__device__ __forceinline__ int lanes_get(const int (&v)[4], int idx) {
    const int k = idx >> 6, l = idx & 63;
    return __builtin_amdgcn_readlane(k == 0 ? v[0] : k == 1 ? v[1] : k == 2 ? v[2] : v[3], l);
}
// Level 1 in ONE launch (round 4; rounds 2-3 ran it as count / scan over the slices / row tables / scatter: four launches).  A workgroup takes a tile of 4 096
// consecutive Gaussians of the depth order, its four waves 1 024 each.  (A) every wave counts its spans per tile row (LDS atomics, wave-private
// tables); thread y publishes the tile's count of row y and adds up what the earlier tiles published -- the same two-level prefix as the
// depth sort (k_radix_pass) --; the row starts come from the row totals that k_depth_keys left in the sort header.  (B) every wave scatters its
// spans with the bit-matrix ranking, cursors = row start + earlier tiles + earlier waves of the tile.  Rectangles and ids stay in registers
// between (A) and (B).  Workgroup 0 also writes the row tables of level 2.
#define SP_TILE 4096
#define SP_THREADS 1024                      // 16 waves x 256 Gaussians: the scatter of a wave is a serial chain over its steps of 64, so many short chains
#define SP_WAVES (SP_THREADS / 64)
#define SP_ITEMS (SP_TILE / SP_THREADS)
// exclusive scan over the values of threads 0..255 of a larger workgroup (the others pass 0 and ignore the result); everyone meets at the barriers
__device__ __forceinline__ int scan256_of_big_block(int v, int* sm, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = 0;
    if (wave < 4) { incl = nrc_wave_incl_sum_i(v, lane); if (lane == 63) sm[wave] = incl; }
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const int t = sm[w]; if (w < wave) base += t; }
    if (total) *total = sm[0] + sm[1] + sm[2] + sm[3];
    __syncthreads();
    return base + incl - v;
}
__global__ void __launch_bounds__(SP_THREADS) k_span_sweep(int P, int gy, int n_tiles, const uint32_t* __restrict__ order, const uint32_t* __restrict__ rects,
                                                           uint32_t* __restrict__ hdr, uint32_t* __restrict__ status, uint32_t* __restrict__ gstat, int64_t cap,
                                                           uint2* __restrict__ spans, uint32_t* __restrict__ rowtot_out, uint32_t* __restrict__ roff_out,
                                                           uint32_t* __restrict__ nitems, uint32_t* __restrict__ ioff, int64_t* __restrict__ meta_spans,
                                                           uint32_t* __restrict__ meta) {
    __shared__ uint32_t cnt[SP_WAVES][SPAN_DIM_MAX];        // per wave: spans per row, later the wave's first position in the row's list
    __shared__ uint32_t bits[SP_WAVES][SPAN_DIM_MAX][2];
    __shared__ int sm[8];
    __shared__ int tile_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, y_of = threadIdx.x;
    const bool row_thread = y_of < gy;   // threads 0 .. gy - 1 also look after one tile row each
    if (threadIdx.x == 0) tile_s = (int)atomicAdd(&hdr[RS_HDR_TICKET + 4], 1u);
    for (int k = threadIdx.x; k < SP_WAVES * SPAN_DIM_MAX; k += SP_THREADS) { (&cnt[0][0])[k] = 0u; (&bits[0][0][0])[2 * k] = 0u; (&bits[0][0][0])[2 * k + 1] = 0u; }
    const int row_total = row_thread ? (int)hdr[RS_HDR_ROWS + y_of] : 0;
    __syncthreads();
    const int tile = tile_s;
    const int begin = tile * SP_TILE + wave * (64 * SP_ITEMS), end = min(P, begin + 64 * SP_ITEMS);
    SpanRect r[SP_ITEMS];
#pragma unroll
    for (int k = 0; k < SP_ITEMS; k++) r[k] = span_rect(begin + 64 * k + lane, end, order, rects);
    uint32_t* mine = cnt[wave];
#pragma unroll
    for (int k = 0; k < SP_ITEMS; k++)
        if (r[k].ok)
            for (int y = r[k].y0; y < r[k].y1; y++) atomicAdd(&mine[y], 1u);
    __syncthreads();
    uint32_t my_count = 0u;
    if (row_thread) {
#pragma unroll
        for (int w = 0; w < SP_WAVES; w++) my_count += cnt[w][y_of];
        rs_publish(status + (size_t)tile * 256 + y_of, my_count);
    }
    // row starts (and, for workgroup 0, the level-2 tables): two scans over the rows
    int tot_s, tot_i;
    const int row_start = scan256_of_big_block(row_total, sm, &tot_s);
    const int ch = tot_s < (3 << 20) ? SPAN_CH_MIN : (tot_s < (6 << 20) ? 2 * SPAN_CH_MIN : SPAN_CH_MAX);
    const int it = (row_total + ch - 1) / ch;
    const int item_start = scan256_of_big_block(it, sm, &tot_i);
    if (blockIdx.x == 0) {
        if (row_thread) { rowtot_out[y_of] = (uint32_t)row_total; roff_out[y_of] = (uint32_t)row_start; nitems[y_of] = (uint32_t)it; ioff[y_of] = (uint32_t)item_start; }
        if (threadIdx.x == 0) { *meta_spans = (int64_t)tot_s; meta[0] = (uint32_t)tot_i; meta[1] = (uint32_t)ch; }
    }
    const int group = tile / RS_GROUP, g0 = group * RS_GROUP, n_in = tile - g0;
    const bool group_leader = tile == g0 + RS_GROUP - 1 || tile == n_tiles - 1;
    if (row_thread) {
        const uint32_t* col = status + (size_t)g0 * 256 + y_of;
        const uint32_t in_group = rs_sum_published(col, n_in, 256);
        if (group_leader) rs_publish(gstat + (size_t)group * 256 + y_of, in_group + my_count);
        uint32_t o = (uint32_t)row_start + in_group + rs_sum_published(gstat + y_of, group, 256);
#pragma unroll
        for (int w = 0; w < SP_WAVES; w++) { const uint32_t c = cnt[w][y_of]; cnt[w][y_of] = o; o += c; }
    }
    __syncthreads();
    const int nk = (gy + 63) >> 6;
    uint32_t (*wbits)[2] = bits[wave];
    const uint32_t below_lo = lane < 32 ? (1u << lane) - 1u : 0xffffffffu, below_hi = lane < 32 ? 0u : (1u << (lane - 32)) - 1u;
#pragma unroll
    for (int k = 0; k < SP_ITEMS; k++) {
        if (begin + 64 * k >= end) break;
        // One wave: its LDS operations execute in program order, so the ORs of all lanes are in place when the words are read (no workgroup barrier;
        // the wavefront-scope fences only keep the compiler from moving the accesses).  A lane is a GAUSSIAN: it writes its span record to every row
        // of its rectangle at (the wave's next position in the row) + (number of earlier lanes of the step that cover the row) -- the loop is as
        // long as the tallest rectangle of the step (a few rows), not as long as the most covered row's list, which is what the first form (a lane per
        // ROW popping the bits of its word one by one) paid for: 36 -> 23 us at 1 M Gaussians, 131 -> 96 us at 6 M.  Same positions, same order.
        if (r[k].ok)
            for (int y = r[k].y0; y < r[k].y1; y++) atomicOr(&wbits[y][lane >> 5], 1u << (lane & 31));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (r[k].ok) {
            const uint2 record = make_uint2((uint32_t)r[k].id, (uint32_t)r[k].x01);
            for (int y = r[k].y0; y < r[k].y1; y++) {
                const int64_t pos = (int64_t)mine[y] + __popc(wbits[y][0] & below_lo) + __popc(wbits[y][1] & below_hi);
                if (pos < cap) spans[pos] = record;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (q >= nk) break;
            const int y = 64 * q + lane;
            const uint32_t lo = wbits[y][0], hi = wbits[y][1];
            if (lo | hi) { mine[y] += (uint32_t)(__popc(lo) + __popc(hi)); wbits[y][0] = 0u; wbits[y][1] = 0u; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}
// the level-2 item `item`: its row, its first span and its span count (row tables in 4 registers x 64 lanes)
struct SpanItem { int y; int64_t s0; int n; };
__device__ __forceinline__ SpanItem span_item(int item, int ch, int gy, const int (&ioff_v)[4], const int (&nit_v)[4], const int (&roff_v)[4],
                                              const int (&rtot_v)[4]) {
    int y = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) y += __popcll(__ballot(64 * k + (int)threadIdx.x < gy && ioff_v[k] + nit_v[k] <= item));
    // = the number of rows that end at or before `item` (a row without items ends where it starts): all rows in front of the item's row
    const int c = item - lanes_get(ioff_v, y);
    SpanItem it;
    it.y = y;
    it.s0 = (int64_t)(uint32_t)lanes_get(roff_v, y) + (int64_t)c * ch;
    it.n = min(ch, lanes_get(rtot_v, y) - c * ch);
    return it;
}
#define SPAN_LOAD_ROW_TABLES()                                                                                                   \
    int ioff_v[4], nit_v[4], roff_v[4], rtot_v[4];                                                                               \
    _Pragma("unroll") for (int k = 0; k < 4; k++) {                                                                              \
        const int yy = 64 * k + lane;                                                                                            \
        ioff_v[k] = yy < gy ? (int)ioff[yy] : 0; nit_v[k] = yy < gy ? (int)nitems[yy] : 0;                                        \
        roff_v[k] = yy < gy ? (int)roff[yy] : 0; rtot_v[k] = yy < gy ? (int)rowtot[yy] : 0;                                       \
    }
// level 2, counting: cnt2[item][x] = number of spans of the item that cover tile x of the item's row
__global__ void __launch_bounds__(64) k_item_count(int gx, int gy, const uint32_t* __restrict__ rowtot, const uint32_t* __restrict__ roff,
                                                   const uint32_t* __restrict__ nitems, const uint32_t* __restrict__ ioff, const uint32_t* __restrict__ meta_items,
                                                   int64_t cap, int item_cap, const uint2* __restrict__ spans, uint32_t* __restrict__ cnt2) {
    __shared__ uint32_t cnt[SPAN_DIM_MAX];
    const int lane = threadIdx.x;
    SPAN_LOAD_ROW_TABLES();
    const int n_items = min((int)meta_items[0], item_cap);  // more items than the workspace holds: the caller sees spans > capacity and retries
    const int ch = (int)meta_items[1];
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const SpanItem it = span_item(item, ch, gy, ioff_v, nit_v, roff_v, rtot_v);
        for (int x = lane; x < SPAN_DIM_MAX; x += 64) cnt[x] = 0u;
        __syncthreads();
        for (int j = lane; j < it.n; j += 64) {
            if (it.s0 + j >= cap) break;
            const uint32_t x01 = spans[it.s0 + j].y;
            for (int x = (int)(x01 & 0xffffu); x < (int)(x01 >> 16); x++) atomicAdd(&cnt[x], 1u);
        }
        __syncthreads();
        for (int x = lane; x < gx; x += 64) cnt2[(size_t)item * gx + x] = cnt[x];
        __syncthreads();
    }
}
// per tile (y, x): exclusive scan of cnt2[item][x] over the items of row y (in place) and the tile total.  32 tiles x 32 item groups per workgroup.
// Round 4: the loads of a thread's items are requested eight at a time (one at a time, each was a full L2 latency), and the workgroup that
// finishes LAST runs what used to be two more launches: the scan of the tile totals into `ranges` (k_scan_tiles) and the launch order of the
// tiles, longest list first (k_tile_order).  Hand-over: the totals are written through (sc1) and drained, one lane per workgroup takes a
// ticket with a returning agent-scope atomic, the last ticket holder reads the totals with sc1 loads.
__device__ __forceinline__ void tile_order_of(int n_tiles, const uint32_t* __restrict__ ranges, uint32_t* __restrict__ order, uint32_t* hist, uint32_t* wsum);
#define IS_XL 32     // tiles (columns) per workgroup
#define IS_NG 32     // item groups per workgroup: thread (xl, g) scans the g-th 1/32 of the row's items for column x, sixteen loads at a time
                     // (~250 items per row at 1 M Gaussians: one batch for the sum, one for the prefix)
__global__ void __launch_bounds__(1024) k_item_scan(int gx, int gy, int item_cap, const uint32_t* __restrict__ nitems, const uint32_t* __restrict__ ioff,
                                                    uint32_t* __restrict__ cnt2, uint32_t* __restrict__ tcount, uint32_t* __restrict__ done_counter,
                                                    uint32_t cap, uint32_t* __restrict__ ranges, uint32_t* __restrict__ order, int64_t* __restrict__ num_rendered,
                                                    int64_t* __restrict__ mailbox, int64_t mailbox_ticket) {
    __shared__ uint32_t gsum[IS_NG][IS_XL];
    __shared__ uint32_t hist[2048];
    __shared__ uint32_t wave_tot[16];
    __shared__ int last_s;
    const int xl = threadIdx.x & (IS_XL - 1), g = threadIdx.x / IS_XL, y = blockIdx.y, x = blockIdx.x * IS_XL + xl;
    const int i0 = min((int)ioff[y], item_cap), ni = min((int)nitems[y], item_cap - i0);
    const int per = (ni + IS_NG - 1) / IS_NG, a = min(ni, g * per), b = min(ni, a + per);
    enum { DEPTH = 16 };
    uint32_t s = 0;
    if (x < gx)
        for (int i = a; i < b; i += DEPTH) {
            uint32_t v[DEPTH];
#pragma unroll
            for (int k = 0; k < DEPTH; k++) v[k] = i + k < b ? cnt2[(size_t)(i0 + i + k) * gx + x] : 0u;
#pragma unroll
            for (int k = 0; k < DEPTH; k++) s += v[k];
        }
    gsum[g][xl] = s;
    __syncthreads();
    uint32_t run = 0, tot = 0;
#pragma unroll
    for (int q = 0; q < IS_NG; q++) { const uint32_t v = gsum[q][xl]; tot += v; if (q < g) run += v; }
    if (x < gx) {
        for (int i = a; i < b; i += DEPTH) {
            uint32_t v[DEPTH];
#pragma unroll
            for (int k = 0; k < DEPTH; k++) v[k] = i + k < b ? cnt2[(size_t)(i0 + i + k) * gx + x] : 0u;
#pragma unroll
            for (int k = 0; k < DEPTH; k++) {
                if (i + k < b) cnt2[(size_t)(i0 + i + k) * gx + x] = run;
                run += v[k];
            }
        }
        if (g == 0) __hip_atomic_store(&tcount[(size_t)y * gx + x], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the last workgroup to get here scans the tile totals and orders the tiles
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t ticket = __hip_atomic_fetch_add(done_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = ticket == gridDim.x * gridDim.y - 1u;
    }
    __syncthreads();
    if (!last_s) return;
    // thread t owns the contiguous tiles [t c, (t + 1) c), c = ceil(n / 1024): all of its totals are requested at once and ONE block scan over the
    // chunk sums follows (a scan per 1024 tiles, each behind its own sc1 round trip, was 12 of this launch's 20 us at 4 346 tiles)
    const int n = gx * gy;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = (n + 1023) / 1024, t0 = (int)threadIdx.x * chunk, t1 = min(n, t0 + chunk);
    uint32_t mine = 0, first[16];   // the thread's first 16 totals stay in registers (all of them up to 16 384 tiles)
#pragma unroll
    for (int k = 0; k < 16; k++) first[k] = t0 + k < t1 ? __hip_atomic_load(&tcount[t0 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
    for (int k = 0; k < 16; k++) mine += first[k];
    for (int i = t0 + 16; i < t1; i++) mine += __hip_atomic_load(&tcount[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t trun = incl - mine, total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { const uint32_t t = wave_tot[w]; total += t; if (w < wave) trun += t; }
    if (threadIdx.x == 0) {
        *num_rendered = (int64_t)total;
        if (mailbox) {
            // the two counts of the frame straight into host memory (nrc_host_mailbox_alloc: pinned, mapped, coherent), the ticket LAST: the host polls
            // the ticket and then sizes / checks the lists without an event, a copy or a stream wait (posted writes of one agent arrive in order)
            __hip_atomic_store(mailbox, (int64_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mailbox + 1, num_rendered[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // spans: written by k_span_sweep, two launches ago
            __hip_atomic_store(mailbox + 2, mailbox_ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (t0 + k < t1) { ranges[2 * (t0 + k)] = min(trun, cap); ranges[2 * (t0 + k) + 1] = min(trun + first[k], cap); trun += first[k]; }
    for (int i = t0 + 16; i < t1; i++) {
        const uint32_t v = __hip_atomic_load(&tcount[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ranges[2 * i] = min(trun, cap); ranges[2 * i + 1] = min(trun + v, cap);
        trun += v;
    }
    __syncthreads();
    tile_order_of(n, ranges, order, hist, wave_tot);
}
// level 2, scatter: ids of the item's spans appended to the tiles they cover; tile (y, x) of this item starts at ranges[tile].first + cnt2[item][x]
template <bool CAPPED>  // CAPPED: the caller fixed the length of point_list (graph capture); entries that would land behind it are dropped
__global__ void __launch_bounds__(64) k_item_scatter(int gx, int gy, const uint32_t* __restrict__ rowtot, const uint32_t* __restrict__ roff,
                                                     const uint32_t* __restrict__ nitems, const uint32_t* __restrict__ ioff, const uint32_t* __restrict__ meta_items,
                                                     int64_t cap, int item_cap, const uint2* __restrict__ spans, const uint32_t* __restrict__ cnt2,
                                                     const uint32_t* __restrict__ ranges, uint32_t list_cap, int32_t* __restrict__ point_list) {
    // bits[x]: which of the chunk's 64 spans cover tile x; base[x]: where the item's next id of tile x goes.  A lane is a SPAN: it writes its id to
    // every tile it covers at base[x] + (number of earlier spans of the chunk covering x) -- the loop runs as long as the widest span of the chunk
    // (a few tiles), not as long as the most covered tile's list (tens of ids), which is what the earlier form (a lane per TILE picking the ids
    // out of bits[x] one by one) was paying for.  Same positions, same order.
    __shared__ uint32_t bits[SPAN_DIM_MAX][2];
    __shared__ uint32_t base[SPAN_DIM_MAX];
    const int lane = threadIdx.x;
    const int nk = (gx + 63) >> 6;
    SPAN_LOAD_ROW_TABLES();
    // the same clamps as k_item_count / k_item_scan: items behind item_cap have no cnt2 row, spans behind cap were never written (a frame that
    // outgrew the workspace: the sized call retries, the fixed-capacity call reports spans > capacity and blends what fits)
    const int n_items = min((int)meta_items[0], item_cap), ch = (int)meta_items[1];
#pragma unroll
    for (int k = 0; k < 4; k++) { bits[64 * k + lane][0] = 0u; bits[64 * k + lane][1] = 0u; }
    const uint32_t below_lo = lane < 32 ? (1u << lane) - 1u : 0xffffffffu, below_hi = lane < 32 ? 0u : (1u << (lane - 32)) - 1u;
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const SpanItem it = span_item(item, ch, gy, ioff_v, nit_v, roff_v, rtot_v);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int x = 64 * k + lane;
            if (x < gx) base[x] = ranges[2 * ((size_t)it.y * gx + x)] + cnt2[(size_t)item * gx + x];
        }
        uint2 sp = lane < it.n && it.s0 + lane < cap ? spans[it.s0 + lane] : make_uint2(0u, 0u);
        for (int j0 = 0; j0 < it.n; j0 += 64) {
            const uint2 nxt = j0 + 64 + lane < it.n && it.s0 + j0 + 64 + lane < cap ? spans[it.s0 + j0 + 64 + lane] : make_uint2(0u, 0u);
            const int x0 = (int)(sp.y & 0xffffu), x1 = (int)(sp.y >> 16);
            for (int x = x0; x < x1; x++) atomicOr(&bits[x][lane >> 5], 1u << (lane & 31));
            __syncthreads();
            for (int x = x0; x < x1; x++) {
                const uint32_t pos = base[x] + (uint32_t)__popc(bits[x][0] & below_lo) + (uint32_t)__popc(bits[x][1] & below_hi);
                if (!CAPPED || pos < list_cap) point_list[pos] = (int)sp.x;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (k >= nk) break;
                const int x = 64 * k + lane;
                const uint32_t lo = bits[x][0], hi = bits[x][1];
                if (lo | hi) { base[x] += (uint32_t)(__popc(lo) + __popc(hi)); bits[x][0] = 0u; bits[x][1] = 0u; }
            }
            __syncthreads();
            sp = nxt;
        }
    }
}

// ------------------------------------------------------------------------------------------------ 2. tile ranges
// `cap`: length of the instance list the caller allocated (fixed-capacity mode; 0xffffffff otherwise): ranges are cut there, the scatter drops
// what would land behind it, num_rendered keeps the uncut total (> cap <=> instances were dropped).
__global__ void __launch_bounds__(1024) k_scan_tiles(const uint32_t* __restrict__ counts, int n, uint32_t cap, uint32_t* __restrict__ ranges,
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
        if (i < n) { ranges[2 * i] = min(off + incl - v, cap); ranges[2 * i + 1] = min(off + incl, cap); fill[i] = 0u; }
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

// ------------------------------------------------------------------------------------------------ 4b. launch order of the tiles
// Tile lists are heavy-tailed (mean 1 920, max 7 010 entries on the bench frame) and a frame is only ~2 rounds of workgroups: a long tile
// that starts late finishes alone.  The render kernels therefore take their tile from a table sorted by list length, longest first
// (counting sort over 2 048 length classes, one workgroup).
// one workgroup of 1024 threads; hist: 2048 words, wsum: 16 words of LDS.  `ranges` may have been written by this very workgroup (behind a barrier).
__device__ __forceinline__ void tile_order_of(int n_tiles, const uint32_t* __restrict__ ranges, uint32_t* __restrict__ order, uint32_t* hist, uint32_t* wsum) {
    for (int k = threadIdx.x; k < 2048; k += 1024) hist[k] = 0u;
    __syncthreads();
    auto cls = [&](int t) { const uint32_t len = ranges[2 * t + 1] - ranges[2 * t]; return 2047u - min(len >> 2, 2047u); };  // class 0 = longest
    for (int t = threadIdx.x; t < n_tiles; t += 1024) atomicAdd(&hist[cls(t)], 1u);
    __syncthreads();
    // exclusive scan of the 2048 classes: two per thread
    const uint32_t a = hist[2 * threadIdx.x], b = hist[2 * threadIdx.x + 1];
    uint32_t incl = a + b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t off = incl - (a + b);
    for (int w = 0; w < wave; w++) off += wsum[w];
    __syncthreads();
    hist[2 * threadIdx.x] = off; hist[2 * threadIdx.x + 1] = off + a;
    __syncthreads();
    // placement with an LDS cursor per class: the order inside a class depends on the atomics' arrival, which only permutes the launch order of
    // tiles of (nearly) equal length -- every tile's result is independent of when it runs
    for (int t = threadIdx.x; t < n_tiles; t += 1024) order[atomicAdd(&hist[cls(t)], 1u)] = (uint32_t)t;
}
__global__ void __launch_bounds__(1024) k_tile_order(int n_tiles, const uint32_t* __restrict__ ranges, uint32_t* __restrict__ order) {
    __shared__ uint32_t hist[2048];
    __shared__ uint32_t wsum[16];
    tile_order_of(n_tiles, ranges, order, hist, wsum);
}

// ------------------------------------------------------------------------------------------------ 5. render
// thread -> pixel of the 16x16 tile: wave = 8x8 quadrant, DPP row (16 lanes) = 4x4 block of the quadrant.  block index b = 4 * wave + row,
// block (cx, cy) of the 4x4 block grid of the tile: cx = 2 * (wave & 1) + (row & 1), cy = 2 * (wave >> 1) + (row >> 1).
__device__ __forceinline__ int tile_px(unsigned t) { return (int)((t & 3u) + ((t >> 4) & 1u) * 4u + ((t >> 6) & 1u) * 8u); }
__device__ __forceinline__ int tile_py(unsigned t) { return (int)(((t >> 2) & 3u) + ((t >> 5) & 1u) * 4u + (t >> 7) * 8u); }
// The tile lists come from the square 3-sigma bound of preprocess (reference semantics, kept bit-exact), but a pixel only blends a
// Gaussian where alpha = o exp(power) >= 1/255, i.e. inside the ellipse A dx^2 + 2 B dx dy + C dy^2 <= 2 ln(255 o).  On the bench
// scene 52 % of the (tile, Gaussian) pairs of a list have no such pixel in the tile, and the box of a pair that has is ~5 pixels wide
// (tools/gs_stats.py): it reaches 2.5 of the four 8x8 quadrants (160 lanes) but only ~5 of the sixteen 4x4 blocks (80 lanes).
// Both render kernels therefore keep ONE WORK LIST PER 4x4 BLOCK: while a batch of 256 list entries is staged into LDS every thread tests
// its entry's box against the 16 blocks; 16 ballots and a prefix over the 4 waves turn the flags into 16 ascending index lists; and every
// DPP row of a wave walks ITS OWN list -- the four rows of a wave blend four different Gaussians in the same instruction.  The box is
// conservative (margins below), so exactly the same pixels blend exactly the same Gaussians in the same order: bit-identical pictures.
// bit b set when the alpha >= 1/255 ellipse of the record can reach a pixel centre of block b (block = 4 x 4 pixels of the 16 x 16 tile).
// Two conservative tests: the ellipse's bounding box against the four column / row bands, then, per row band, the x-interval the ellipse
// spans inside the band (chords at the band's two edges -- clamped to the ellipse's y-range --, widened to the box where the leftmost /
// rightmost point of the ellipse lies inside the band) against the column bands.  On the bench scene the box keeps 6.2 blocks per
// (tile, Gaussian) pair of a blended prefix, the band intervals 4.7, an exact per-pixel test 4.5 (tools/gs_stats.py).
__device__ __forceinline__ unsigned block_flags(const float4& q0, const float4& q1, const float4& q2, float tx0, float ty0) {
    const float X = q0.x, Y = q0.y, ex = q1.z, ey = q1.w;
    if (!(ex >= 0.f)) return 0u;  // never reaches the threshold
    unsigned col = 0, row = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float x0 = tx0 + (float)(4 * c), y0 = ty0 + (float)(4 * c);
        col |= (unsigned)(X + ex >= x0 && X - ex <= x0 + 3.f) << c;
        row |= (unsigned)(Y + ey >= y0 && Y - ey <= y0 + 3.f) << c;
    }
    if (col == 0u || row == 0u) return 0u;
    unsigned cols[4] = {col, col, col, col};
    if (ex < __builtin_inff()) {
        const float ta = q2.x, da = q2.y, ba = q2.z, yoff = q2.w;   // half chord at offset d from the centre row: sqrt(tau / A - (det / A^2) d^2)
        float lo[5], hi[5];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const float d = fminf(fmaxf(Y - (ty0 + (float)(4 * k)), -ey), ey);
            const float w = __builtin_amdgcn_sqrtf(fmaxf(ta - da * d * d, 0.f));
            const float mid = X + ba * d;
            lo[k] = mid - w; hi[k] = mid + w;
        }
        const float yl = Y + yoff, yr = Y - yoff;
#pragma unroll
        for (int sl = 0; sl < 4; sl++) {
            const float ca = fmaxf(ty0 + (float)(4 * sl), Y - ey), cb = fminf(ty0 + (float)(4 * sl + 4), Y + ey);
            float l = fminf(lo[sl], lo[sl + 1]), h = fmaxf(hi[sl], hi[sl + 1]);
            if ((yl >= ca && yl <= cb) || (yr >= ca && yr <= cb)) { l = fminf(l, X - ex); h = fmaxf(h, X + ex); }
            l -= 0.02f; h += 0.02f;
            unsigned cb4 = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float x0 = tx0 + (float)(4 * c);
                cb4 |= (unsigned)(h >= x0 && l <= x0 + 3.f) << c;
            }
            cols[sl] = col & cb4;
        }
    }
    // block b = 4 * (2 * (cy >> 1) + (cx >> 1)) + 2 * (cy & 1) + (cx & 1)
    unsigned f = 0;
#pragma unroll
    for (int cy = 0; cy < 4; cy++) {
        const unsigned c4 = ((row >> cy) & 1u) ? cols[cy] : 0u;
        f |= (c4 & 3u) << (8 * (cy >> 1) + 2 * (cy & 1));
        f |= (c4 >> 2) << (8 * (cy >> 1) + 4 + 2 * (cy & 1));
    }
    return f;
}
// The exponent of a Gaussian at a pixel: -1/2 (A dx^2 + C dy^2) - B dx dy, as SIX instructions (one of them packed) with explicit fused multiply-adds where the source-order
// form takes nine (this translation unit is built without contraction: the integer decisions of the per-Gaussian kernels must not move).  The blend
// kernels are bound by VALU issue (profiles/: 0.70-0.78 of the issue slots at 315 M blends per frame), so instructions per blend are what they cost.
// The upstream binary is an nvcc build with -fmad=true, i.e. it fuses in these places too; WHICH pairs it fuses is not recoverable, so the oracle
// (oracle/gs_oracle_impl.h: GS_BLEND_POWER / test_T / the colour sums) states the same fusions and n_contrib / final_T stay bit-exact against it.
typedef float v2f __attribute__((ext_vector_type(2)));
// (A, C) and (dx, dy) travel as register PAIRS: the two first products are one v_pk_mul_f32, the pixel offset one v_pk_add_f32 -- the same
// roundings as the scalar form, two VALU instructions less per blend.  (This translation unit is built with -fno-slp-vectorize: left to itself
// the vectoriser pairs the products of two DIFFERENT list entries and pays the packing back in v_mov, 20 per four blends.)
__device__ __forceinline__ float blend_power(v2f ac, float B, v2f d) {
    const v2f p = ac * d;                                  // (A dx, C dy)
    const float t2 = fmaf(p.y, d.y, p.x * d.x);
    return fmaf(-0.5f, t2, -((B * d.x) * d.y));
}
// expf for the blend loops, WITHOUT the two range guards of the library routine: the same nine instructions (Cody-Waite split of x log2(e), v_exp_f32 on the
// fraction, v_ldexp_f32 by the integer part), hence bit for bit the library's value wherever the result is a normal number, -87.3 <= x <= 88.72
// (tools/micro/exp_blend_check.hip: 2 x 10^8 inputs -- a sweep of [-104, 0], every float in [-1e-3, 0], 2^13 floats around every multiple of ln 2 --
// 0 differences there; below, where the result is a denormal, the two differ in how they flush: a blend needs alpha >= 1/255, i.e. x >= -5.6, so none
// of those values is ever used).  The four instructions dropped -- two compares and two selects forcing 0 / inf outside the range -- were a tenth of
// k_render's VALU instructions per blend (40 -> 36; 168 -> 157 us at 1 M Gaussians).  x > 88.72 (power > 0: the blend discards the entry) overflows to
// inf in v_ldexp_f32 like the library's select.
__device__ __forceinline__ float exp_blend(float x) {
    const float ph = x * 0x1.715476p+0f;
    float pl = fmaf(x, 0x1.715476p+0f, -ph);
    pl = fmaf(x, 0x1.4ae0bep-26f, pl);
    const float e = __builtin_rintf(ph);
    const float a = (ph - e) + pl;
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(a), (int)e);
}
#define N_BLOCKS 16
// One staged batch: thread t read entry t's 64-byte record; what the blend loop needs goes to LDS as two 16-byte vectors and a scalar.
// LT = what a block list holds per entry: uint8_t = the batch index j (k_render_bw: its LDS budget, with the 18 KB of per-Gaussian sums, has no
// room for more), uint16_t = 16 j, the BYTE OFFSET of the entry's records in a[] / b[] / c[] (k_render: the address arrives with the list read, three
// VALU instructions per blend less).  Blue sits in a float4 array of its own so that all three records of an entry live at the same offset.
template <typename LT>
struct StageLdsT {
    float4 a[BATCH];      // X, Y, r, g
    float4 b[BATCH];      // A, C, B, o  (conic xx, yy, xy, opacity: the pair (A, C) is what blend_power multiplies with (dx, dy))
    float4 c[BATCH];      // b, -, -, -
    uint16_t flags[BATCH];
    __attribute__((aligned(16))) LT list[N_BLOCKS][BATCH];
};
template <typename LT>
__device__ __forceinline__ unsigned stage_entry(StageLdsT<LT>& st, const float4* __restrict__ splat, int id, float tx0, float ty0) {
    const float4* rec = splat + 4 * (size_t)id;
    const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
    const unsigned flags = block_flags(q0, q1, q2, tx0, ty0);
    if (flags) {
        st.a[threadIdx.x] = make_float4(q0.x, q0.y, q3.x, q3.y);
        st.b[threadIdx.x] = make_float4(q0.z, q1.x, q0.w, q1.y);
        st.c[threadIdx.x].x = q3.z;
    }
    return flags;
}
// Builds the 16 per-block lists of a staged batch: the 16 lanes of DPP row b (= the threads whose pixels form block b) build list b, lane l
// from the flags of entries 16 l .. 16 l + 15 (two 16-byte LDS reads), ranked with a row prefix sum.  st.list[b] receives the batch indices
// of the entries that reach block b in ascending order.  Returns the length of the calling thread's own list; *n_wave = the longest of the
// calling wave's four lists.
template <typename LT>
__device__ __forceinline__ int block_lists(StageLdsT<LT>& st, unsigned flags, int* n_wave) {
    st.flags[threadIdx.x] = (uint16_t)flags;
    __syncthreads();
    const int b = threadIdx.x >> 4, l16 = threadIdx.x & 15;
    const uint4 fa = reinterpret_cast<const uint4*>(st.flags)[2 * l16], fb = reinterpret_cast<const uint4*>(st.flags)[2 * l16 + 1];
    const uint32_t w[8] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w};
    uint32_t mask = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t t = (w[k] >> b) & 0x00010001u;
        mask |= ((t & 1u) | (t >> 15)) << (2 * k);
    }
    const int cnt = __popc(mask);
    int incl = cnt;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, true);  // row_shr:1, zero fill
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, true);
    const int mine = __shfl(incl, 15, 16);
    LT* dst = st.list[b] + (incl - cnt);
    uint32_t m = mask;
    while (m) {
        const int i = __builtin_ctz(m);
        m &= m - 1u;
        *dst++ = (LT)((16 * l16 + i) * (sizeof(LT) == 2 ? 16 : 1));
    }
    *n_wave = max(max(__builtin_amdgcn_readlane(mine, 0), __builtin_amdgcn_readlane(mine, 16)),
                  max(__builtin_amdgcn_readlane(mine, 32), __builtin_amdgcn_readlane(mine, 48)));
    __syncthreads();
    return mine;
}

__global__ void __launch_bounds__(256) k_render(GsCam cam, const uint32_t* __restrict__ ranges, const int32_t* __restrict__ point_list,
                                                const float4* __restrict__ splat, const uint32_t* __restrict__ tile_order, float bg0, float bg1, float bg2,
                                                const float* __restrict__ pose, float* __restrict__ out_color, uint32_t* __restrict__ n_contrib,
                                                float* __restrict__ final_T) {
    __shared__ StageLdsT<uint16_t> st;
    if (pose) { bg0 = pose[35]; bg1 = pose[36]; bg2 = pose[37]; }
    const int tile = (int)tile_order[blockIdx.x];  // longest lists first
    const int tile_x = tile % cam.gx, tile_y = tile / cam.gx;
    const int px = tile_x * TILE + tile_px(threadIdx.x), py = tile_y * TILE + tile_py(threadIdx.x);
    const bool inside = px < cam.W && py < cam.H;
    const float fx = (float)px, fy = (float)py;
    const float tx0 = (float)(tile_x * TILE), ty0 = (float)(tile_y * TILE);
    const int block = threadIdx.x >> 4;  // = 4 * wave + DPP row
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    bool done = !inside;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last = 0;
    const uint2* list4 = reinterpret_cast<const uint2*>(st.list[block]);   // four 16-bit entries of this row's list per LDS read
    // everything the blend loop reads from LDS must be FINITE also for a row that is past the end of its list and picks up a stale index (see the
    // weight below): thread t clears slot t, which only thread t ever stages into
    st.a[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f); st.b[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f); st.c[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    // ... and every list entry it can pick up must be a valid offset: the lists start as zeros too (later batches leave offsets of earlier entries behind)
    reinterpret_cast<uint4*>(&st.list[0][0])[threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
    reinterpret_cast<uint4*>(&st.list[0][0])[256 + threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
    for (uint32_t base = r0; base < r1; base += BATCH) {
        if (__syncthreads_count(done) == 256) break;
        const uint32_t k = base + threadIdx.x;
        const unsigned flags = k < r1 ? stage_entry(st, splat, point_list[k], tx0, ty0) : 0u;
        int n_wave;
        const int n_mine = block_lists(st, flags, &n_wave);
        const uint32_t pos0 = base - r0 + 1u;  // contributor number of batch entry 0 = its position in the tile list + 1
        int last_off = -1;
        for (int jj = 0; jj < n_wave; jj += 4) {
            if (__ballot(!done) == 0ull) break;  // wave-uniform: every pixel of the quadrant is saturated
            const uint2 pack = list4[jj >> 2];
            // A dependent chain of VALU instructions issues one instruction per ~8 cycles on this chip, and more resident waves do not fill the
            // gaps (tools/micro/valu_rate.hip: 4.5 cycles per instruction at 8 waves per SIMD and one chain, 2.3 with four independent chains per
            // wave).  So the four entries of a pack are taken TOGETHER: (1) all LDS reads, (2) the four alphas -- independent of the pixel's
            // running state, four interleaved chains -- and only then (3) the short serial part (transmittance test, masks, accumulation) in
            // list order.  Entries past the end of the row's list and the reference's four early-outs (done / power > 0 / alpha < 1/255 /
            // saturation) are lane masks; same arithmetic, same order per pixel: bit-identical pixels.
            int off[4];   // 16 j: entries behind the list end are stale offsets of earlier batches (or zeros): readable, masked below
            off[0] = (int)(pack.x & 0xffffu); off[1] = (int)(pack.x >> 16); off[2] = (int)(pack.y & 0xffffu); off[3] = (int)(pack.y >> 16);
            float4 a_j[4], b_j[4];
            float c_j[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a_j[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(st.a) + off[u]);
                b_j[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(st.b) + off[u]);
                c_j[u] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(st.c) + off[u]);
            }
            float alpha[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const v2f d = (v2f){a_j[u].x, a_j[u].y} - (v2f){fx, fy};
                const float power = blend_power((v2f){b_j[u].x, b_j[u].y}, b_j[u].z, d);
                alpha[u] = fminf(0.99f, b_j[u].w * exp_blend(power));
                ok[u] = (jj + u < n_mine) & !(power > 0.0f) & !(alpha[u] < 1.0f / 255.0f);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float test_T = fmaf(-T, alpha[u], T);   // T (1 - alpha), one rounding (blend_power's note)
                const bool valid = ok[u] & !done;
                const bool sat = valid & (test_T < 0.0001f);
                const bool upd = valid & !sat;
                done = done | sat;
                // One weight alpha T per Gaussian -- an exact 0 for a masked lane (a select: alpha of a stale entry behind the list end is finite but
                // arbitrary) -- and three multiply-adds: 5 VALU instructions where the unfused, per-channel-masked form took 15.  The colours of a
                // masked entry are multiplied by that 0: they are finite, because the staging arrays start as zeros and only ever receive records.
                const float w = upd ? alpha[u] * T : 0.f;
                C0 = fmaf(a_j[u].z, w, C0); C1 = fmaf(a_j[u].w, w, C1); C2 = fmaf(c_j[u], w, C2);
                T = upd ? test_T : T;
                last_off = upd ? off[u] : last_off;
            }
        }
        if (last_off >= 0) last = pos0 + ((uint32_t)last_off >> 4);   // contributor number of the last entry this pixel blended (one add per batch, not per blend)
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
// sum over the 16 lanes of a DPP row with row operations (no LDS crossbar traffic): lanes 15, 31, 47, 63 hold their row's sum.  All steps
// run UNMASKED (row_mask = bank_mask = 0xf, out-of-row sources read 0): only the last lane of a row has to be right, and without masks
// every step is a single v_add_f32_dpp.
__device__ __forceinline__ float row_sum_to_lane15(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));  // row_shr:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));  // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));  // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));  // row_shr:8
    return v;
}
// the four accumulators of the blend backward's global atomics, cleared by one launch (3 + 4 + 1 + 3 floats per Gaussian)
// The blend backward's per-Gaussian accumulator: ONE 64-byte record per Gaussian,
//   [0..2] dL/dcolour   [3] dL/dopacity   [4..5] dL/dmean2D (x, y)   [6..8] dL/dconic (xx, xy, yy)   [9..15] unused,
// so that the nine sums a tile flushes for a Gaussian are one contiguous 36-byte group of ONE atomic wave-instruction.  Round 2 kept them in four
// arrays (colour (P,3), opacity (P), mean2D (P,3), conic (P,4)): nine atomic instructions per flushing thread, every lane of each in another
// cache line -- float atomics execute at the memory side, per 64-byte request, and that shape is the slow one (MI355X_MICROARCH.md, Global float
// atomics: "64 lanes in 64 different rows ~17x slower").  Ablation (round 3): without the flush k_render_bw took 367 instead of 485 us.
#define GREC 16
__global__ void __launch_bounds__(256) k_zero_grads(int P, float4* __restrict__ grad_rec) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (int64_t)P * (GREC / 4)) grad_rec[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
// a + (b of the partner lane); CTRL: row_mirror 0x140 (lane ^ 15), row_half_mirror 0x141 (^ 7), quad_perm [3,2,1,0] 0x1b (^ 3), [1,0,3,2] 0xb1 (^ 1)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float keep, float send) {
    return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), CTRL, 0xf, 0xf, true));
}
// (a + partner's a) in all lanes, (b + partner's b) in the lanes of the banks of BANKS.  The masked form has no builtin; the s_nop covers
// the two wait states a DPP read needs after a VALU write of the same register, which the compiler cannot see through the asm.
template <int CTRL, int BANKS>
__device__ __forceinline__ float dpp_add2(float a, float b) {
    float r = dpp_add<CTRL>(a, a);
    if constexpr (CTRL == 0x140)
        asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:%2" : "+v"(r) : "v"(b), "n"(BANKS));
    else
        asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:%2" : "+v"(r) : "v"(b), "n"(BANKS));
    return r;
}
// The nine per-Gaussian sums of a row meet the other blocks of the tile in LDS.  ds_add_f32 retires 0.33 lane-operations per clock per CU on
// this chip however the lanes are spread over instructions (tools/micro/lds_atomic_rate.hip: 27 cycles per (row, entry) update of nine
// values, per CU); ds_add_u64 with the nine values in nine lanes of ONE instruction takes 1.9.  So the sums travel as
// 64-bit fixed point: scaled by a power of two chosen per tile from the largest |dL/dpixel| (every sum is linear in it, so the format is
// scale-free) and per class of quantity from its analytic bound (colour <= 2^8, opacity <= 2^20, mean <= 2^19, conic <= 2^36 times that
// gradient for splats up to sigma ~ 1000 px; beyond, the value saturates instead of wrapping).  Resolution 2^-51 ... 2^-25 of the tile's
// largest pixel gradient: below what the f32 atomics of the next level keep.  Side effect: the per-tile sums no longer depend on the order
// in which the blocks arrive.
#define BW_S_COLOR 51
#define BW_S_OPACITY 39
#define BW_S_MEAN 40
#define BW_S_CONIC 25
#define BW_E_MIN (-60)
#ifndef BW_PAIR
#define BW_PAIR 1
#endif
__global__ void __launch_bounds__(256) k_render_bw(GsCam cam, const uint32_t* __restrict__ ranges, const int32_t* __restrict__ point_list,
                                                   const float4* __restrict__ splat, const uint32_t* __restrict__ tile_order, float bg0, float bg1, float bg2,
                                                   const float* __restrict__ pose, const uint32_t* __restrict__ n_contrib, const float* __restrict__ final_T,
                                                   const float* __restrict__ dL_dpix, float* __restrict__ grad_rec) {
    __shared__ StageLdsT<uint8_t> st;
    __shared__ unsigned long long s_acc[BATCH][9];  // per-Gaussian gradient sums of the tile's 16 blocks (fixed point), flushed once per batch
    __shared__ int s_id[BATCH];                     // Gaussian of batch entry t (-1: nothing staged)
    __shared__ int s_blast[N_BLOCKS];
    __shared__ float s_gmax[4];
    if (pose) { bg0 = pose[35]; bg1 = pose[36]; bg2 = pose[37]; }
    const int tile = (int)tile_order[blockIdx.x];  // longest lists first
    const int tile_x = tile % cam.gx, tile_y = tile / cam.gx;
    const int lane = threadIdx.x & 63;
    const int block = threadIdx.x >> 4;
    const float tx0 = (float)(tile_x * TILE), ty0 = (float)(tile_y * TILE);
    const int px = tile_x * TILE + tile_px(threadIdx.x), py = tile_y * TILE + tile_py(threadIdx.x);
    const bool inside = px < cam.W && py < cam.H;
    const float fx = (float)px, fy = (float)py;
    const uint32_t r0 = ranges[2 * tile], r1 = ranges[2 * tile + 1];
    const int n_tile = (int)(r1 - r0);
    const size_t pix = (size_t)py * cam.W + px, hw = (size_t)cam.H * cam.W;
    const float T_final = inside ? final_T[pix] : 0.f;
    float T = T_final;
    const int last = inside ? (int)n_contrib[pix] : 0;
    float n0 = 0.f, n1 = 0.f, n2 = 0.f;  // colour accumulated behind the current Gaussian
    const float g0 = inside ? dL_dpix[pix] : 0.f, g1 = inside ? dL_dpix[hw + pix] : 0.f, g2 = inside ? dL_dpix[2 * hw + pix] : 0.f;
    const float bg_dot = bg0 * g0 + bg1 * g1 + bg2 * g2;
    const float ddelx_dx = 0.5f * cam.W, ddely_dy = 0.5f * cam.H;
    const float neg_tf_bg = -T_final * bg_dot;
    // Only the first max(last) entries of the tile list were blended by any pixel of the tile (k_render stops at saturation, typically
    // after a tenth of the list): the backward walk starts there, not at the end of the list, and a block's list only receives the entries
    // in front of the block's own maximum.
    int block_last = last;
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) block_last = max(block_last, __shfl_xor(block_last, d, 16));
    if ((threadIdx.x & 15) == 0) s_blast[block] = block_last;
    // The blend loop below masks an inactive lane through three factors only; everything else it reads must be FINITE, also for a row that is
    // past the end of its list and picks up a stale index: the staging arrays start as zeros (later batches leave finite records behind).
    st.a[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f); st.b[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f); st.c[threadIdx.x] = make_float4(0.f, 0.f, 0.f, 0.f);
    float gmax = inside ? fmaxf(fmaxf(fabsf(g0), fabsf(g1)), fabsf(g2)) : 0.f;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, d, 64));
    if (lane == 0) s_gmax[threadIdx.x >> 6] = gmax;
    __syncthreads();
    gmax = fmaxf(fmaxf(s_gmax[0], s_gmax[1]), fmaxf(s_gmax[2], s_gmax[3]));
    if (!(gmax > 0.f)) return;  // no gradient reaches this tile (uniform over the workgroup: nobody is left at a barrier)
    int gexp;
    (void)frexpf(gmax, &gexp);    // every |dL/dpixel| of the tile is < 2^gexp
    gexp = max(gexp, BW_E_MIN);
    // transposing row reduction (see the loop): the lane that ends up with column q of s_acc, its scale, and the partner-selection bits
    const int l16 = lane & 15;
    const bool b3 = l16 & 8, b2 = l16 & 4, b1 = l16 & 2, b0 = l16 & 1;
    const bool q_has = !b0 || l16 == 1;
    const int q_col = b0 ? 4 : (b3 ? 5 : 0) + (b2 ? 2 : 0) + (b1 ? 1 : 0);  // columns: c0 c1 c2 op mx | my cx cy cw
    const int q_s = q_col < 3 ? BW_S_COLOR : q_col == 3 ? BW_S_OPACITY : q_col < 6 ? BW_S_MEAN : BW_S_CONIC;
    const float q_scale = ldexpf(1.0f, q_s - gexp);
    const int blast_v = lane < N_BLOCKS ? s_blast[lane] : 0;  // lanes 0..15 of every wave: the 16 block maxima
    int n_eff = blast_v;
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) n_eff = max(n_eff, __shfl_xor(n_eff, d, 16));
    n_eff = min(n_tile, __builtin_amdgcn_readfirstlane(n_eff));
    const uint32_t* list4 = reinterpret_cast<const uint32_t*>(st.list[block]);
    // batches are taken from the END of the blended prefix: position p (0-based from the front) has contributor number p + 1;
    // batch entry t sits at position n_eff - 1 - done_cnt - t
    for (int done_cnt = 0; done_cnt < n_eff; done_cnt += BATCH) {
        __syncthreads();
        const int nb_raw = min(BATCH, n_eff - done_cnt);
#pragma unroll
        for (int q = 0; q < 9; q++) s_acc[threadIdx.x][q] = 0ull;
        // stage the batch; per block the list of entries whose alpha >= 1/255 ellipse box reaches it (see k_render)
        int id_l = 0;
        unsigned flags = 0;
        const int pos_top = n_eff - 1 - done_cnt;  // list position of batch entry 0
        if ((int)threadIdx.x < nb_raw) {
            id_l = point_list[r0 + pos_top - threadIdx.x];
            const float4* rec = splat + 4 * (size_t)id_l;
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3];
            flags = block_flags(q0, q1, q2, tx0, ty0);
            const int pos_l = pos_top - (int)threadIdx.x;
#pragma unroll
            for (int b = 0; b < N_BLOCKS; b++)  // no pixel of block b blended anything at or behind its maximum
                if (pos_l >= __builtin_amdgcn_readlane(blast_v, b)) flags &= ~(1u << b);
            if (flags) {
                st.a[threadIdx.x] = make_float4(q0.x, q0.y, q3.x, q3.y);
                st.b[threadIdx.x] = make_float4(q0.z, q1.x, q0.w, q1.y);
                st.c[threadIdx.x].x = q3.z;
            }
        }
        s_id[threadIdx.x] = flags ? id_l : -1;
        int n_wave;
        const int n_mine = block_lists(st, flags, &n_wave);
        // One list entry of this row: the record, this pixel's offset, alpha exactly as the forward computed it, and whether the pixel blended it.
        // entry j of the batch sits at list position pos_top - j; this pixel blended the positions < last (0 for a pixel outside the image): j > pos_top - last
        const int j_behind = pos_top - last;
        struct Entry { float4 co; float cx, cy, dx, dy, c0, c1, c2, G, alpha; int j; bool active; };
        auto entry = [&](int jj) {
            Entry e;
            const uint32_t pack = list4[jj >> 2];  // four entries of this row's list per dword
            e.j = (int)__builtin_amdgcn_ubfe(pack, 8u * ((uint32_t)jj & 3u), 8u);   // one v_bfe_u32 (the shift is wave-uniform)
            e.co = st.b[e.j];
            const float4 xyrg = st.a[e.j];
            e.c0 = xyrg.z; e.c1 = xyrg.w; e.c2 = st.c[e.j].x;
            const v2f d = (v2f){xyrg.x, xyrg.y} - (v2f){fx, fy};
            e.dx = d.x; e.dy = d.y;
            const float power = blend_power((v2f){e.co.x, e.co.y}, e.co.z, d);
            e.G = exp_blend(power);
            e.alpha = fminf(0.99f, e.co.w * e.G);
            e.active = (jj < n_mine) & (e.j > j_behind) & !(power > 0.0f) & !(e.alpha < 1.0f / 255.0f);
            return e;
        };
        // Gradient terms of one (pixel, Gaussian) pair and their row sums.  The per-Gaussian constants (0.5 W, 0.5 H, -1/2, signs) wait for the
        // flush and multiply-add pairs are fused: tolerance-checked values, unlike alpha above, which repeats the forward's arithmetic exactly.
        // The colour behind the current Gaussian is carried as ONE running value per channel (the reference keeps last alpha / last colour and
        // rebuilds it every step: same recurrence).  Updates T and n0..n2; returns this lane's column of the row sums.
        auto gradients = [&](const Entry& e) {
            // two masked factors carry `active` through everything below: a lane that does not blend this Gaussian multiplies T by 1 / (1 - 0) = 1
            // exactly, moves its colour behind by 0 and adds 0 to every sum (all other factors are finite: alpha <= 0.99, and only G can overflow,
            // when power > 0)
            const float alpha_m = e.active ? e.alpha : 0.f, Gm = e.active ? e.G : 0.f;
            const float r_om = __builtin_amdgcn_rcpf(1 - alpha_m);  // 1 ulp; feeds gradients only (tolerance, not bit parity); rcp(1) = 1
            const float T_new = T * r_om;
            const float e0 = e.c0 - n0, e1 = e.c1 - n1, e2 = e.c2 - n2;   // colour of this Gaussian minus the colour accumulated behind it
            float dL_dalpha = e0 * g0;
            dL_dalpha = fmaf(e1, g1, dL_dalpha);
            dL_dalpha = fmaf(e2, g2, dL_dalpha);
            dL_dalpha = fmaf(neg_tf_bg, r_om, dL_dalpha * T_new);
            const float dchm = alpha_m * T_new;
            const float d_op = Gm * dL_dalpha;          // sum G dL/dalpha
            const float wgt = e.co.w * d_op;            // G dL/dG
            const float wx = wgt * e.dx, wy = wgt * e.dy;
            const float d_mx = fmaf(wx, e.co.x, wy * e.co.z), d_my = fmaf(wy, e.co.y, wx * e.co.z);   // co = (A, C, B, o); flushed with -0.5 W, -0.5 H
            const float d_cx = wx * e.dx, d_cy = wx * e.dy, d_cw = wy * e.dy;                          // flushed with -1/2
            const float d_c0 = dchm * g0, d_c1 = dchm * g1, d_c2 = dchm * g2;
            T = T_new;
            n0 = fmaf(alpha_m, e0, n0); n1 = fmaf(alpha_m, e1, n1); n2 = fmaf(alpha_m, e2, n2);   // = alpha c + (1 - alpha) n: what lies behind the next one
            // Row-level reduction (row = block = one Gaussian) that also TRANSPOSES: four exchange steps with the partners lane ^ 15, ^ 7, ^ 3, ^ 1
            // (DPP row_mirror, row_half_mirror and two quad permutations); at each step a lane keeps one half of its values and adds what the
            // partner held of that half, so the nine sums end in nine different lanes (21 instructions; nine separate row sums took 36).
            // Steps 1 and 2: the lanes that keep the other half sit in whole DPP banks (lanes 8-15: banks 2, 3; lanes 4-7, 12-15: banks 1, 3),
            // so "pair sum of A, but pair sum of B in those lanes" is one unmasked and one bank-masked v_add_f32_dpp (two selects less per pair).
            const float r0 = dpp_add2<0x140, 0xc>(d_c0, d_my), r1 = dpp_add2<0x140, 0xc>(d_c1, d_cx);
            const float r2 = dpp_add2<0x140, 0xc>(d_c2, d_cy), r3 = dpp_add2<0x140, 0xc>(d_op, d_cw);
            const float r4 = dpp_add<0x140>(d_mx, d_mx);
            const float t0 = dpp_add2<0x141, 0xa>(r0, r2), t1 = dpp_add2<0x141, 0xa>(r1, r3);
            const float t4 = dpp_add<0x141>(r4, r4);
            const float u0 = dpp_add<0x1b>(b1 ? t1 : t0, b1 ? t0 : t1), u4 = dpp_add<0x1b>(t4, t4);
            // last step: a lane keeps ONE of (u0, u4) and its partner lane ^ 1 the other -- select what I keep and what the partner needs first, then
            // one exchange-and-add (selecting afterwards took two exchanges)
            return dpp_add<0xb1>(b0 ? u4 : u0, b0 ? u0 : u4);
        };
        // the nine sums of the row leave with ONE 64-bit integer LDS atomic
        auto deposit = [&](int jj, int j, float mine) {
            if (q_has && jj < n_mine) {
                const float y = __builtin_amdgcn_fmed3f(mine * q_scale, -0x1p61f, 0x1p61f);
                // float -> two's-complement 64-bit, truncating (a bias of half a step of 2^-25 ... 2^-51), in six instructions: the generic
                // (long long) conversion is thirteen -- a seventh of this loop's VALU instructions with the reduction.  t is an integer-valued float,
                // hi = floor(t / 2^32) fits an i32, t - hi 2^32 is an exact integer in [0, 2^32) (the low mantissa bits of t): same bits as the cast.
                const float t = __builtin_truncf(y);
                const float hf = __builtin_floorf(t * 0x1p-32f);
                const uint32_t lo32 = (uint32_t)__builtin_fmaf(-hf, 0x1p32f, t);
                atomicAdd(&s_acc[j][q_col], ((unsigned long long)(uint32_t)(int)hf << 32) | lo32);
            }
        };
        // BW_PAIR entries per trip: their record reads, exponentials and reductions are independent instruction streams (only T and the
        // running colour pass from one to the next), which is what this latency-bound loop lacks
#if BW_PAIR
        for (int jj = 0; jj < n_wave; jj += 2) {
            const Entry ea = entry(jj), eb = entry(jj + 1);
            if (__ballot(ea.active | eb.active) == 0ull) continue;  // nobody in this wave sees either Gaussian
            const float ma = gradients(ea);
            const float mb = gradients(eb);
            deposit(jj, ea.j, ma);
            deposit(jj + 1, eb.j, mb);
        }
#else
        for (int jj = 0; jj < n_wave; jj++) {
            const Entry ea = entry(jj);
            if (__ballot(ea.active) == 0ull) continue;  // nobody in this wave sees its Gaussian
            deposit(jj, ea.j, gradients(ea));
        }
#endif
        __syncthreads();
        // Flush: the 16 lanes of a DPP row take ONE batch entry, lane q its quantity q -- a wave-instruction adds four 36-byte groups, each inside one
        // 64-byte record, instead of 64 lanes in 64 different lines.  Same sums, same one atomic per (tile, Gaussian, quantity).
        {
            const int q = threadIdx.x & 15;
            const float inv = q < 3 ? ldexpf(1.0f, gexp - BW_S_COLOR) : q == 3 ? ldexpf(1.0f, gexp - BW_S_OPACITY)
                            : q == 4 ? -ddelx_dx * ldexpf(1.0f, gexp - BW_S_MEAN) : q == 5 ? -ddely_dy * ldexpf(1.0f, gexp - BW_S_MEAN)
                                                                                           : -0.5f * ldexpf(1.0f, gexp - BW_S_CONIC);
            for (int e = threadIdx.x >> 4; e < nb_raw; e += 16) {
                const int id = s_id[e];
                if (id >= 0 && q < 9) {
                    const long long a = (long long)s_acc[e][q];
                    if (a != 0ll) atomicAdd(grad_rec + (size_t)id * GREC + q, (float)a * inv);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ 7. preprocess backward
// One visible Gaussian.  sh_row: this Gaussian's SH coefficients in LDS (3*M floats); each is replaced in place by its gradient.
__device__ __forceinline__ void preprocess_bw_one(int i, const GsCam& cam, const float* __restrict__ means3D, float* sh_row, int use_sh,
                                                  const float* __restrict__ scales, const float* __restrict__ rotations, int use_scale_rot,
                                                  const uint8_t* __restrict__ clamped, const float* __restrict__ cov3D,
                                                  float* __restrict__ grad_rec, float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
                                                  float* __restrict__ dL_dcolor, float* __restrict__ dL_dmean3D,
                                                  float* __restrict__ dL_dcov3D, float* __restrict__ dL_dscale, float* __restrict__ dL_drot,
                                                  const float* __restrict__ opacities, float* __restrict__ dL_dopacity) {
    // the blend kernel's sums of this Gaussian: one 64-byte record (see k_zero_grads); the API's per-quantity tensors are written from it
    const float4 r0 = *reinterpret_cast<const float4*>(grad_rec + (size_t)i * GREC), r1 = *reinterpret_cast<const float4*>(grad_rec + (size_t)i * GREC + 4);
    const float r2x = grad_rec[(size_t)i * GREC + 8];
    // ... and the record is left cleared for the next backward (the caller keeps the buffer: no clearing launch in front of k_render_bw)
    *reinterpret_cast<float4*>(grad_rec + (size_t)i * GREC) = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(grad_rec + (size_t)i * GREC + 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    grad_rec[(size_t)i * GREC + 8] = 0.f;
    const float gcol[3] = {r0.x, r0.y, r0.z};
    {
        float g_op = r0.w;
        if (cam.raw && opacities) {  // dL/dlogit = dL/dopacity * o (1 - o)
            const float o_act = act_sigmoid(opacities[i]);
            g_op *= o_act * (1.f - o_act);
        }
        dL_dopacity[i] = g_op;
        dL_dmean2D[3 * i] = r1.x; dL_dmean2D[3 * i + 1] = r1.y; dL_dmean2D[3 * i + 2] = 0.f;
        if (dL_dconic) *reinterpret_cast<float4*>(dL_dconic + 4 * (size_t)i) = make_float4(r1.z, r1.w, 0.f, r2x);
        if (dL_dcolor) { dL_dcolor[3 * i] = gcol[0]; dL_dcolor[3 * i + 1] = gcol[1]; dL_dcolor[3 * i + 2] = gcol[2]; }
    }
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
    const float gcx = r1.z, gcy = r1.w, gcz = r2x;
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
    const float g2x = r1.x, g2y = r1.y;
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
        for (int ch = 0; ch < 3; ch++) dRGB[ch] = ((cl >> ch) & 1) ? 0.f : gcol[ch];
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
        float q[4], s[3];
        const float q_norm = act_rotation(rotations + 4 * i, cam.raw, q);
        act_scales(scales + 3 * i, cam.raw, s);
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
            const float ds = mod * (dA[k] * R[k] + dA[3 + k] * R[3 + k] + dA[6 + k] * R[6 + k]);
            dL_dscale[3 * i + k] = cam.raw ? ds * s[k] : ds;  // d exp(l) / dl = exp(l)
#pragma unroll
            for (int r_ = 0; r_ < 3; r_++) dR[3 * r_ + k] = dA[3 * r_ + k] * (mod * s[k]);
        }
        const float r = q[0], x = q[1], y = q[2], z = q[3];
        float g[4];
        g[0] = 2 * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
        g[1] = 2 * (y * dR[1] + z * dR[2] + y * dR[3] - 2 * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2 * x * dR[8]);
        g[2] = 2 * (-2 * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2 * y * dR[8]);
        g[3] = 2 * (-2 * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2 * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
        if (cam.raw) {  // through q / |q|: (g - q^ (q^ . g)) / |q|
            const float dot = q[0] * g[0] + q[1] * g[1] + q[2] * g[2] + q[3] * g[3];
#pragma unroll
            for (int k = 0; k < 4; k++) g[k] = (g[k] - q[k] * dot) / q_norm;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) dL_drot[4 * i + k] = g[k];
    }
}

// Workgroup = 128 Gaussians.  Their SH coefficients (and, on the way out, the SH gradients) are one contiguous 128 x 3M float
// block of shs / dL_dsh: it is staged through LDS with coalesced transfers (a lane reading or writing its own 192-byte row
// touches 48 different cache lines per wave instruction; measured 1.86 GB of HBM traffic for 0.5 GB of data).  Every output
// row is written by this kernel, zeros for culled Gaussians, so the caller does not clear 300 B per Gaussian beforehand.
#ifndef PBW_BLOCK
#define PBW_BLOCK 128
#endif
#ifndef PBW_SHB_U
#define PBW_SHB_U 12
#endif
// Adam on the `rest` SH coefficients (45 of a Gaussian's 59 parameters: 76 % of the optimizer's traffic) INSIDE the preprocessing backward: the gradient
// rows sit in LDS at the end of this kernel, so the step reads them there instead of the kernel writing 180 B per Gaussian and the optimizer reading them and
// the parameters back (nrc_gs_backward_rest_step).  p == nullptr: the gradient is written as usual.  Same per-element arithmetic as k_adam (adam_math.h).
struct RestAdam { float* p; float* m; float* v; float lr, beta1, beta2, eps, bc1, bc2; };
#ifndef PBW_MAXM
#define PBW_MAXM 16
#endif
template <int FORM>   // 0: one (P, M, 3) SH tensor; 1: dc + rest; 2: dc + rest with the Adam step of `rest` inside (RestAdam)
__global__ void __launch_bounds__(PBW_BLOCK) k_preprocess_bw(int P, GsCam cam_arg, const float* __restrict__ pose, const float* __restrict__ means3D, const float* __restrict__ shs,
                                                             const float* __restrict__ shs_rest, const float* __restrict__ opacities, int use_sh, const float* __restrict__ scales, const float* __restrict__ rotations,
                                                             int use_scale_rot, const int32_t* __restrict__ radii, const uint8_t* __restrict__ clamped,
                                                             const float* __restrict__ cov3D, float* __restrict__ grad_rec, float* __restrict__ dL_dmean2D,
                                                             float* __restrict__ dL_dconic, float* __restrict__ dL_dcolor,
                                                             float* __restrict__ dL_dmean3D, float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
                                                             float* __restrict__ dL_dsh_rest, float* __restrict__ dL_dscale, float* __restrict__ dL_drot,
                                                             float* __restrict__ dL_dopacity, RestAdam ra) {
    __shared__ float s_sh[PBW_BLOCK * (3 * PBW_MAXM + 1)];
    const GsCam cam = cam_with_pose(cam_arg, pose);
    const int first = blockIdx.x * PBW_BLOCK, i = first + threadIdx.x;
    const int row_len = 3 * cam.M, pitch = row_len + 1;  // +1: rows start in different LDS banks
    const int count = min(PBW_BLOCK, P - first);
    const bool visible = i < P && radii[i] > 0;
    if (use_sh) {
        // 12 elements per trip here (k_preprocess: 16): with 16 this kernel's 204 VGPRs left the split-SH form of the training step at 168-179 us against
        // 133-138 with 12 or 8 (the concatenated form 139 -> 128; tools/exp_gs_pre.py, round 6)
        sh_rows_copy<true, (FORM == 0 ? SHB_U : PBW_SHB_U), FORM != 0>(s_sh, pitch, row_len, count, (size_t)first, const_cast<float*>(shs), const_cast<float*>(shs_rest), PBW_BLOCK);
        __syncthreads();
    }
    float* sh_row = s_sh + threadIdx.x * pitch;
    if (visible) {
        preprocess_bw_one(i, cam, means3D, sh_row, use_sh, scales, rotations, use_scale_rot, clamped, cov3D, grad_rec, dL_dmean2D, dL_dconic, dL_dcolor,
                          dL_dmean3D, dL_dcov3D, dL_dscale, dL_drot, opacities, dL_dopacity);
    } else if (i < P) {
        if (use_sh) for (int k = 0; k < row_len; k++) sh_row[k] = 0.f;
        dL_dopacity[i] = 0.f;
        dL_dmean2D[3 * i] = 0.f; dL_dmean2D[3 * i + 1] = 0.f; dL_dmean2D[3 * i + 2] = 0.f;
        if (dL_dconic) *reinterpret_cast<float4*>(dL_dconic + 4 * (size_t)i) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (dL_dcolor) { dL_dcolor[3 * i] = 0.f; dL_dcolor[3 * i + 1] = 0.f; dL_dcolor[3 * i + 2] = 0.f; }
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
        if constexpr (FORM != 2) {
            sh_rows_copy<false, (FORM == 0 ? SHB_U : PBW_SHB_U), FORM != 0>(s_sh, pitch, row_len, count, (size_t)first, dL_dsh, dL_dsh_rest, PBW_BLOCK);
        } else {
            // dc gradient rows as usual (3 floats per Gaussian), then Adam over the block's `rest` rows: element k of the contiguous (count, row_len - 3) block, gradient
            // from LDS (invisible Gaussians: zeros -- their moments decay and their parameters move like in the optimizer's own launch)
            sh_block_copy<false, PBW_SHB_U>(s_sh, pitch, 0, 3, count, dL_dsh + (size_t)first * 3, PBW_BLOCK);
            const int len = row_len - 3, n = count * len;
            const size_t g0 = (size_t)first * len;
            const int dq = PBW_BLOCK / len, dr = PBW_BLOCK - dq * len;
            int r = (int)threadIdx.x / len, c = (int)threadIdx.x - r * len;
            const NrcAdamHyper h{ra.lr, ra.beta1, ra.beta2, ra.eps, 0.f, 0, ra.bc1, ra.bc2, 1.0f};
            enum { RU = 4 };
            for (int k = threadIdx.x; k < n; k += RU * PBW_BLOCK) {
                float pv[RU], mv[RU], vv[RU], gv[RU];
#pragma unroll
                for (int u = 0; u < RU; u++) {
                    const int kk = k + u * PBW_BLOCK;
                    const bool in = kk < n;
                    gv[u] = in ? s_sh[r * pitch + 3 + c] : 0.f;
                    pv[u] = in ? ra.p[g0 + kk] : 0.f; mv[u] = in ? ra.m[g0 + kk] : 0.f; vv[u] = in ? ra.v[g0 + kk] : 0.f;
                    r += dq; c += dr;
                    if (c >= len) { c -= len; r++; }
                }
#pragma unroll
                for (int u = 0; u < RU; u++) {
                    const int kk = k + u * PBW_BLOCK;
                    if (kk < n) {
                        nrc_adam_update(pv[u], gv[u], mv[u], vv[u], h, false, 0.f);
                        ra.p[g0 + kk] = pv[u]; ra.m[g0 + kk] = mv[u]; ra.v[g0 + kk] = vv[u];
                    }
                }
            }
        }
    }
}

// slices = waves; all of them should be resident at once: 160 KB of LDS per CU, (4 B x n_tiles) per wave, 256 CUs
// binning workspace (u32 words): [keyA P][valA P][rectA P][keyB P][valB P][rectB P][header RS_HDR_WORDS][status 4 x nblk x 256][group totals 4 x ngroups x 256] -- the depth pre-sort --
// [rowtot gy][roff gy][nitems gy][ioff gy][meta 4][tcount n_tiles][cnt2 item_cap x gx][spans 2 x cap]
struct BinWs {
    uint32_t *keyA, *valA, *rectA, *keyB, *valB, *rectB, *hdr, *status, *gstat, *rowtot, *roff, *nitems, *ioff, *meta, *tcount, *cnt2;
    uint2* spans;
    int nblk, ngroups, item_cap;
    int64_t cap, words, n_status;
};
// The library's one side stream (per device, created on first use) and the two events of its fork / join: the colour pass of the
// preprocessing runs there while the caller's stream works through the depth sort and the binning.  Never used inside a stream capture and
// never while the stage timer is armed (the pass then runs in line, where its time can be attributed).
struct GsSide { hipStream_t stream; hipEvent_t fork, join; bool ok; };
GsSide* gs_side() {
    static GsSide side[64] = {};
    static bool tried[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (!tried[dev]) {
        tried[dev] = true;
        GsSide& s = side[dev];
        s.ok = hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&s.join, hipEventDisableTiming) == hipSuccess;
    }
    return side[dev].ok ? &side[dev] : nullptr;
}
int64_t gs_default_span_cap(int P) { return 4 * (int64_t)(P > 0 ? P : 1) + 65536; }
BinWs gs_bin_ws(uint32_t* base, int P, int gx, int gy, int64_t cap) {
    BinWs w;
    const int64_t p1 = P > 0 ? P : 1;
    w.nblk = (int)nrc_cdiv(p1, RS_TILE);
    w.ngroups = (w.nblk + RS_GROUP - 1) / RS_GROUP;
    w.cap = cap;
    w.item_cap = (int)(cap / SPAN_CH_MIN + gy + 1);
    uint32_t* q = base;
    auto take = [&](int64_t n, int64_t align_words) { q = base + ((q - base) + align_words - 1) / align_words * align_words; uint32_t* r = q; q += n; return r; };
    w.keyA = take(p1, 4); w.valA = take(p1, 4); w.rectA = take(p1, 4); w.keyB = take(p1, 4); w.valB = take(p1, 4); w.rectB = take(p1, 4);
    w.hdr = take(RS_HDR_WORDS, 4);
    w.n_status = (int64_t)5 * 256 * (w.nblk + w.ngroups);   // status of sort pass p (4 = the span sweep) at + p * 256 * nblk, the group totals behind all five
    w.status = take(w.n_status, 4);
    w.gstat = w.status + (int64_t)5 * 256 * w.nblk;
    w.rowtot = take(gy, 4); w.roff = take(gy, 4); w.nitems = take(gy, 4); w.ioff = take(gy, 4);
    w.meta = take(4, 4);
    w.tcount = take((int64_t)gx * gy, 4);
    w.cnt2 = take((int64_t)w.item_cap * gx, 4);
    w.spans = reinterpret_cast<uint2*>(take(2 * cap, 4));
    w.words = (q - base) + 64;
    return w;
}

// NRC_GS_OVERLAP=1 (read once): the colour outputs of the preprocessing as a separate pass on the side stream (k_preprocess<1> + <2>).  OFF by
// default -- measured at 1 M / 6 M Gaussians: 0.48 / 1.50 ms per forward in one pass, 0.50-0.58 / 1.78-1.95 ms split, whatever the number of
// persistent workgroups: next to a kernel that streams 233 MB the latency-bound sort passes take 2-4 x as long (k_depth_keys 17 -> 33 us, the
// first two radix passes 16 -> 64 and 35 us), which costs more than the 45 us the geometry pass saves over the single pass.
const bool g_gs_no_overlap = [] { const char* e = getenv("NRC_GS_OVERLAP"); return !(e && e[0] == '1'); }();
// NRC_GS_COLOR_BLOCKS (read once): persistent workgroups of the colour pass per compute unit (default 2)
const int g_gs_color_blocks_per_cu = [] { const char* e = getenv("NRC_GS_COLOR_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 16 ? v : 2; }();
// The camera block of a frame (GS_POSE_FLOATS) from a camera-to-world pose that lives on the DEVICE: what GaussianSplatting/Renderer.py:60-74 computes on the
// host -- viewmatrix = w2c^T = [[R, 0], [-(R^T t)^T, 1]], projmatrix = viewmatrix @ P^T, campos = t -- as one launch of one wave (the tensor-op form was
// eleven launches of 4-5 us in front of every training step: transpose, matrix-vector product, negation, four concatenations, two fills, a 4x4 product).
__global__ void k_camera_block(const float* __restrict__ c2w, const float* __restrict__ proj_t, const float* __restrict__ bg3, float* __restrict__ out) {
    __shared__ float view[16];
    const int k = threadIdx.x;
    if (k < 16) {
        const int r = k >> 2, c = k & 3;
        float v;
        if (r < 3) v = c < 3 ? c2w[4 * r + c] : 0.f;
        else if (c == 3) v = 1.f;
        else v = -(c2w[c] * c2w[3] + c2w[4 + c] * c2w[7] + c2w[8 + c] * c2w[11]);   // -(R^T t)_c
        view[k] = v;
        out[k] = v;
    }
    __syncthreads();
    if (k < 16) {
        const int r = k >> 2, c = k & 3;
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) a = fmaf(view[4 * r + j], proj_t[4 * j + c], a);
        out[16 + k] = a;
    }
    if (k < 3) { out[32 + k] = c2w[4 * k + 3]; out[35 + k] = bg3 ? bg3[k] : 0.f; }
}

int make_cam(GsCam& cam, int W, int H, int D, int M, const float* view, const float* proj, const float* campos, const float* camera_dev, float tanx,
             float tany, float scale_modifier, int raw) {
    if (W < 1 || H < 1 || D < 0 || D > 3 || !(tanx > 0.f) || !(tany > 0.f)) return NRC_ERR_INVALID;
    if (!camera_dev && (!view || !proj || !campos)) return NRC_ERR_INVALID;  // the pose comes from the host arrays or from the device block
    for (int k = 0; k < 16; k++) { cam.view[k] = camera_dev ? 0.f : view[k]; cam.proj[k] = camera_dev ? 0.f : proj[k]; }
    for (int k = 0; k < 3; k++) cam.campos[k] = camera_dev ? 0.f : campos[k];
    cam.tan_fovx = tanx; cam.tan_fovy = tany;
    cam.focal_x = W / (2.0f * tanx); cam.focal_y = H / (2.0f * tany);
    cam.scale_modifier = scale_modifier;
    cam.W = W; cam.H = H; cam.gx = (W + TILE - 1) / TILE; cam.gy = (H + TILE - 1) / TILE; cam.D = D; cam.M = M;
    cam.raw = raw ? 1 : 0;
    return NRC_OK;
}

}  // namespace

extern "C" {

#if defined(NRC_SORT_PROBE)
int nrc_debug_sort_probe(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sort_probe), sizeof(g_sort_probe)); }
#endif

int nrc_host_mailbox_alloc(int64_t** mailbox) {
    NRC_ENTER();
    if (!mailbox) return NRC_ERR_INVALID;
    *mailbox = nullptr;
    void* host = nullptr;
    const hipError_t e = hipHostMalloc(&host, 8 * sizeof(int64_t), hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent);
    if (e != hipSuccess) { (void)hipGetLastError(); nrc_set_last_hip_error((int)e); return NRC_ERR_LAUNCH; }
    void* dev = nullptr;
    // one address on both sides (unified addressing): the kernels take the pointer the host reads through
    if (hipHostGetDevicePointer(&dev, host, 0) != hipSuccess || dev != host) { (void)hipGetLastError(); (void)hipHostFree(host); return NRC_ERR_UNSUPPORTED; }
    for (int k = 0; k < 8; k++) ((volatile int64_t*)host)[k] = 0;
    *mailbox = (int64_t*)host;
    return NRC_OK;
}
int nrc_host_mailbox_free(int64_t* mailbox) {
    if (mailbox) {
        const hipError_t e = hipHostFree(mailbox);
        if (e != hipSuccess) { (void)hipGetLastError(); nrc_set_last_hip_error((int)e); return NRC_ERR_LAUNCH; }
    }
    return NRC_OK;
}

int64_t nrc_gs_bin_hist_bytes(int32_t P, int32_t W, int32_t H, int64_t span_capacity) {
    if (P < 0 || W < 1 || H < 1 || span_capacity < 0) return NRC_ERR_INVALID;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    if (gx > SPAN_DIM_MAX || gy > SPAN_DIM_MAX) return 0;  // per-tile key sort fallback: no binning workspace
    return gs_bin_ws(nullptr, P, gx, gy, span_capacity > 0 ? span_capacity : gs_default_span_cap(P)).words * (int64_t)sizeof(uint32_t);
}

int nrc_gs_preprocess(int32_t P, int32_t D, int32_t M, int32_t W, int32_t H, const float* means3D, const float* shs, const float* shs_rest,
                      int32_t raw_parameters, const float* colors_precomp, const float* opacities, const float* scales, float scale_modifier,
                      const float* rotations, const float* cov3D_precomp, const float* viewmatrix_host, const float* projmatrix_host,
                      const float* campos_host, const float* camera_dev, float tan_fovx, float tan_fovy, int32_t* radii, float* depths,
                      float* points_xy, float* conic_opacity, float* rgb, uint8_t* clamped, float* cov3D, uint32_t* tiles_touched,
                      uint32_t* tile_counts, uint32_t* ranges, uint32_t* tile_fill, uint32_t* bin_hist, int64_t span_capacity,
                      int64_t instance_capacity, float* splat_records, int64_t* num_rendered, int64_t* count_mailbox, int64_t mailbox_ticket,
                      nrc_stream_t stream) {
    NRC_ENTER();
    GsCam cam;
    const int rc = make_cam(cam, W, H, D, M, viewmatrix_host, projmatrix_host, campos_host, camera_dev, tan_fovx, tan_fovy, scale_modifier, raw_parameters);
    if (rc != NRC_OK) return rc;
    if (P < 0 || !tile_counts || !ranges || !tile_fill || !num_rendered || instance_capacity < 0 || instance_capacity > 0xfffffffell) return NRC_ERR_INVALID;
    const uint32_t list_cap = instance_capacity > 0 ? (uint32_t)instance_capacity : 0xffffffffu;
    if (P > 0) {
        if ((shs == nullptr) == (colors_precomp == nullptr)) return NRC_ERR_INVALID;                      // exactly one colour source
        if (((scales != nullptr) && (rotations != nullptr)) == (cov3D_precomp != nullptr)) return NRC_ERR_INVALID;  // exactly one covariance source
        if (shs && M < (D + 1) * (D + 1)) return NRC_ERR_INVALID;
        if (shs_rest && (!shs || M < 2)) return NRC_ERR_INVALID;   // split SH: shs = (P,1,3) dc part, shs_rest = (P,M-1,3)
        if (raw_parameters && cov3D_precomp) return NRC_ERR_INVALID;  // raw log-scales / quaternions are what gets activated
    }
    hipStream_t s = (hipStream_t)stream;
    const int n_tiles = cam.gx * cam.gy;
    const bool lds_path = cam.gx <= SPAN_DIM_MAX && cam.gy <= SPAN_DIM_MAX && bin_hist != nullptr;
    if (instance_capacity > 0 && P > 0 && !lds_path) return NRC_ERR_UNSUPPORTED;  // the per-tile key sort fallback sizes its keys from the count
    if (count_mailbox && !(P > 0 && lds_path)) return NRC_ERR_UNSUPPORTED;        // the counts reach the mailbox from the last workgroup of k_item_scan only
    NRC_STAGE(s, nullptr);
    if (!(P > 0 && lds_path)) nrc_zero_async(tile_counts, sizeof(uint32_t) * n_tiles, s);  // only the global-atomic fallback counts into it
    if (P > 0) {
        if (!means3D || !opacities || !radii || !depths || !points_xy || !conic_opacity || !rgb || !clamped || !cov3D || !tiles_touched || !splat_records)
            return NRC_ERR_INVALID;
        if (shs && M > PRE_MAXM) return NRC_ERR_UNSUPPORTED;
        const BinWs w = lds_path ? gs_bin_ws(bin_hist, P, cam.gx, cam.gy, span_capacity > 0 ? span_capacity : gs_default_span_cap(P)) : BinWs{};
        // The colour outputs (192 of the 236 bytes read per Gaussian are SH coefficients) on the side stream, forked HERE -- in front of the geometry
        // pass, so that no event sits between the kernels of the caller's stream -- and joined at the end of this call.  In line (one pass) for
        // the fallback binning, given colours, inside a stream capture, and while the stage timer is armed.
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &capturing);
        GsSide* side = (lds_path && shs && !colors_precomp && capturing == hipStreamCaptureStatusNone && !g_nrc_stage_timer_armed && !g_gs_no_overlap) ? gs_side() : nullptr;
        bool forked = false;
        const int pre_blocks = (int)nrc_cdiv(P, PRE_BLOCK);
#define GS_PRE_ARGS P, cam, camera_dev, means3D, shs, shs_rest, colors_precomp, opacities, scales, rotations, cov3D_precomp, radii, depths, points_xy, conic_opacity, rgb, \
                    clamped, cov3D, tiles_touched, lds_path ? (uint32_t*)nullptr : tile_counts, (float4*)splat_records, lds_path ? w.hdr : (uint32_t*)nullptr
        if (side && hipEventRecord(side->fork, s) == hipSuccess && hipStreamWaitEvent(side->stream, side->fork, 0) == hipSuccess) {
            const int cus = 256;   // MI355X; only the experiment's grid size depends on it
            if (shs_rest) hipLaunchKernelGGL((k_preprocess<2, true>), dim3(pre_blocks < g_gs_color_blocks_per_cu * cus ? pre_blocks : g_gs_color_blocks_per_cu * cus), dim3(PRE_BLOCK), 0, side->stream, GS_PRE_ARGS);
            else hipLaunchKernelGGL((k_preprocess<2, false>), dim3(pre_blocks < g_gs_color_blocks_per_cu * cus ? pre_blocks : g_gs_color_blocks_per_cu * cus), dim3(PRE_BLOCK), 0, side->stream, GS_PRE_ARGS);
            forked = hipEventRecord(side->join, side->stream) == hipSuccess;
            if (!forked) (void)hipStreamSynchronize(side->stream);   // cannot happen short of a broken runtime: stay correct
            hipLaunchKernelGGL((k_preprocess<1, false>), dim3(pre_blocks), dim3(PRE_BLOCK), 0, s, GS_PRE_ARGS);
        } else {
            if (shs_rest) hipLaunchKernelGGL((k_preprocess<0, true>), dim3(pre_blocks), dim3(PRE_BLOCK), 0, s, GS_PRE_ARGS);
            else hipLaunchKernelGGL((k_preprocess<0, false>), dim3(pre_blocks), dim3(PRE_BLOCK), 0, s, GS_PRE_ARGS);
        }
        NRC_STAGE(s, "k_preprocess");
#undef GS_PRE_ARGS
        if (lds_path) {
            // depth pre-sort of the Gaussians: 4 stable 8-bit passes, one launch each, (keyA, rectA) -> B -> A -> B -> (valA, rectA) = depth order
            hipLaunchKernelGGL(k_depth_keys, dim3(nrc_cdiv(P, DK_BLOCK * DK_ITEMS)), dim3(DK_BLOCK), 0, s, P, cam.gx, cam.gy, radii, depths, points_xy, w.keyA, w.rectA,
                               w.hdr, w.status, w.n_status);
            NRC_STAGE(s, "k_depth_keys");
            for (int pass = 0; pass < 4; pass++) {
                const uint32_t *ki = (pass & 1) ? w.keyB : w.keyA, *vi = (pass & 1) ? w.valB : w.valA, *ri = (pass & 1) ? w.rectB : w.rectA;
                uint32_t *ko = (pass & 1) ? w.keyA : w.keyB, *vo = (pass & 1) ? w.valA : w.valB, *ro = (pass & 1) ? w.rectA : w.rectB;
                hipLaunchKernelGGL(k_radix_pass, dim3(w.nblk), dim3(RS_THREADS), 0, s, P, 8 * pass, pass, w.nblk, ki, pass == 0 ? (const uint32_t*)nullptr : vi, ri, w.hdr,
                                   w.status + (int64_t)pass * 256 * w.nblk, w.gstat + (int64_t)pass * 256 * w.ngroups, pass == 3 ? (uint32_t*)nullptr : ko, vo, ro);
                NRC_STAGE(s, "k_radix_pass");
            }
            // level 1: row spans in depth order (one launch); level 2 counting + scans: tile ranges and the per-(item, tile) cursors
            static_assert(SP_TILE == RS_TILE, "the span sweep shares the sort's tile count and status layout");
            hipLaunchKernelGGL(k_span_sweep, dim3(w.nblk), dim3(SP_THREADS), 0, s, P, cam.gy, w.nblk, w.valA, w.rectA, w.hdr, w.status + (int64_t)4 * 256 * w.nblk,
                               w.gstat + (int64_t)4 * 256 * w.ngroups, w.cap, w.spans, w.rowtot, w.roff, w.nitems, w.ioff, num_rendered + 1, w.meta);
            NRC_STAGE(s, "k_span_sweep");
            hipLaunchKernelGGL(k_item_count, dim3(SPAN_GRID_COUNT), dim3(64), 0, s, cam.gx, cam.gy, w.rowtot, w.roff, w.nitems, w.ioff, w.meta, w.cap, w.item_cap,
                               w.spans, w.cnt2);
            NRC_STAGE(s, "k_item_count");
            // per-(item, tile) cursors, tile totals; its last workgroup: ranges, instance count and the launch order of the tiles (into tile_fill)
            hipLaunchKernelGGL(k_item_scan, dim3((unsigned)nrc_cdiv(cam.gx, IS_XL), cam.gy), dim3(1024), 0, s, cam.gx, cam.gy, w.item_cap, w.nitems, w.ioff, w.cnt2, w.tcount,
                               w.hdr + RS_HDR_TICKET + 5, list_cap, ranges, tile_fill, num_rendered, count_mailbox, mailbox_ticket);
            NRC_STAGE(s, "k_item_scan");
            if (forked && hipStreamWaitEvent(s, side->join, 0) != hipSuccess) (void)hipStreamSynchronize(side->stream);   // the colours are in place for whatever the caller enqueues next
        }
    }
    if (!(P > 0 && lds_path)) {
        hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, tile_counts, n_tiles, list_cap, ranges, tile_fill, num_rendered);
        nrc_zero_async(num_rendered + 1, sizeof(int64_t), s);  // no span workspace in use
    }
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_gs_camera_block(const float* c2w_dev, const float* proj_t_dev, const float* bg3_dev, float* camera_dev_out, nrc_stream_t stream) {
    NRC_ENTER();
    if (!c2w_dev || !proj_t_dev || !camera_dev_out) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_camera_block, dim3(1), dim3(64), 0, (hipStream_t)stream, c2w_dev, proj_t_dev, bg3_dev, camera_dev_out);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_gs_bin_render(int32_t P, int32_t W, int32_t H, const float* bg_host, const float* camera_dev, const int32_t* radii, const float* depths,
                      const float* points_xy, const float* conic_opacity, const float* rgb, const uint32_t* ranges, uint32_t* tile_fill,
                      const uint32_t* bin_hist, int64_t span_capacity, int64_t instance_capacity, uint64_t* keys, int32_t* point_list,
                      const float* splat_records, float* out_color, uint32_t* n_contrib, float* final_T, nrc_stream_t stream) {
    NRC_ENTER();
    if (P < 0 || W < 1 || H < 1 || (!bg_host && !camera_dev) || !ranges || !tile_fill || !out_color || !n_contrib || !final_T) return NRC_ERR_INVALID;
    if (instance_capacity < 0 || instance_capacity > 0xfffffffell) return NRC_ERR_INVALID;
    const uint32_t list_cap = instance_capacity > 0 ? (uint32_t)instance_capacity : 0xffffffffu;
    const float bg[3] = {bg_host ? bg_host[0] : 0.f, bg_host ? bg_host[1] : 0.f, bg_host ? bg_host[2] : 0.f};
    GsCam cam = {};
    cam.W = W; cam.H = H; cam.gx = (W + TILE - 1) / TILE; cam.gy = (H + TILE - 1) / TILE;
    hipStream_t s = (hipStream_t)stream;
    NRC_STAGE(s, nullptr);
    if (P > 0) {
        if (!radii || !depths || !points_xy || !conic_opacity || !rgb || !point_list || !splat_records) return NRC_ERR_INVALID;
        const int n_tiles = cam.gx * cam.gy;
        const bool lds_path = cam.gx <= SPAN_DIM_MAX && cam.gy <= SPAN_DIM_MAX && bin_hist;
        if (!lds_path && (!keys || instance_capacity > 0)) return instance_capacity > 0 ? NRC_ERR_UNSUPPORTED : NRC_ERR_INVALID;
        if (lds_path) {
            const BinWs w = gs_bin_ws(const_cast<uint32_t*>(bin_hist), P, cam.gx, cam.gy, span_capacity > 0 ? span_capacity : gs_default_span_cap(P));
            if (instance_capacity > 0)
                hipLaunchKernelGGL(k_item_scatter<true>, dim3(SPAN_GRID), dim3(64), 0, s, cam.gx, cam.gy, w.rowtot, w.roff, w.nitems, w.ioff, w.meta, w.cap, w.item_cap, w.spans,
                                   w.cnt2, ranges, list_cap, point_list);
            else
                hipLaunchKernelGGL(k_item_scatter<false>, dim3(SPAN_GRID), dim3(64), 0, s, cam.gx, cam.gy, w.rowtot, w.roff, w.nitems, w.ioff, w.meta, w.cap, w.item_cap, w.spans,
                                   w.cnt2, ranges, list_cap, point_list);
            NRC_STAGE(s, "k_item_scatter");
        } else {
            hipLaunchKernelGGL(k_scatter, dim3(nrc_cdiv(P, 256)), dim3(256), 0, s, P, cam.gx, cam.gy, radii, depths, points_xy, ranges, tile_fill, keys);
            hipLaunchKernelGGL((k_sort_tiles<0, 1024>), dim3(n_tiles), dim3(256), 0, s, ranges, keys, point_list);
            hipLaunchKernelGGL((k_sort_tiles<1024, 4096>), dim3(n_tiles), dim3(256), 0, s, ranges, keys, point_list);
            hipLaunchKernelGGL((k_sort_tiles<4096, SORT_LDS_CAP>), dim3(n_tiles), dim3(256), 0, s, ranges, keys, point_list);
            NRC_STAGE(s, "k_scatter+k_sort_tiles");
        }
    }
    // tile_fill has served the fallback scatter (if any): it now carries the launch order of the tiles, longest list first, for both render kernels
    if (!(P > 0 && cam.gx <= SPAN_DIM_MAX && cam.gy <= SPAN_DIM_MAX && bin_hist)) {   // the span path left the order in tile_fill already (k_item_scan)
        hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, s, cam.gx * cam.gy, ranges, tile_fill);
        NRC_STAGE(s, "k_tile_order");
    }
    hipLaunchKernelGGL(k_render, dim3(cam.gx * cam.gy), dim3(256), 0, s, cam, ranges, point_list, (const float4*)splat_records, tile_fill, bg[0], bg[1], bg[2],
                       camera_dev, out_color, n_contrib, final_T);
    NRC_STAGE(s, "k_render");
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

static int gs_backward_impl(int32_t P, int32_t D, int32_t M, int32_t W, int32_t H, const float* bg_host, const float* means3D, const float* shs,
                    const float* shs_rest, int32_t raw_parameters, const float* opacities, const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                    const float* cov3D_precomp, const float* viewmatrix_host, const float* projmatrix_host, const float* campos_host,
                    const float* camera_dev, float tan_fovx, float tan_fovy, const int32_t* radii, const float* points_xy, const float* conic_opacity,
                    const float* rgb, const uint8_t* clamped, const float* cov3D, const int32_t* point_list, const uint32_t* ranges,
                    const float* splat_records, const uint32_t* tile_order, const uint32_t* n_contrib, const float* final_T, const float* dL_dpix,
                    float* dL_dmean2D, float* dL_dconic,
                    float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dsh_rest, float* dL_dscale,
                    float* dL_drot, float* grad_records, int32_t records_clear, nrc_stream_t stream, RestAdam ra) {
    GsCam cam;
    const int rc = make_cam(cam, W, H, D, M, viewmatrix_host, projmatrix_host, campos_host, camera_dev, tan_fovx, tan_fovy, scale_modifier, raw_parameters);
    if (rc != NRC_OK) return rc;
    if (P < 0 || (!bg_host && !camera_dev)) return NRC_ERR_INVALID;
    const float bg[3] = {bg_host ? bg_host[0] : 0.f, bg_host ? bg_host[1] : 0.f, bg_host ? bg_host[2] : 0.f};
    if (P == 0) return NRC_OK;
    if (!means3D || !radii || !points_xy || !conic_opacity || !rgb || !clamped || !cov3D || !point_list || !ranges || !n_contrib || !final_T ||
        !dL_dpix || !dL_dmean2D || !dL_dopacity || !dL_dmean3D || !dL_dcov3D || !grad_records || (reinterpret_cast<uintptr_t>(grad_records) & 63u))
        return NRC_ERR_INVALID;
    const int use_sh = colors_precomp == nullptr, use_sr = cov3D_precomp == nullptr;
    if ((use_sh && (!shs || !dL_dsh)) || (use_sr && (!scales || !rotations || !dL_dscale || !dL_drot))) return NRC_ERR_INVALID;
    if ((shs_rest != nullptr) != (dL_dsh_rest != nullptr || ra.p != nullptr) || (raw_parameters && (!opacities || !use_sr))) return NRC_ERR_INVALID;
    if (ra.p && (ra.p != shs_rest || !ra.m || !ra.v || !use_sh || dL_dsh_rest || !(ra.bc1 > 0.f) || !(ra.bc2 > 0.f))) return NRC_ERR_INVALID;   // the step updates the tensor the kernels read
    hipStream_t s = (hipStream_t)stream;
    NRC_STAGE(s, nullptr);
    if (use_sh && M > PBW_MAXM) return NRC_ERR_UNSUPPORTED;
    if (!records_clear) {
        hipLaunchKernelGGL(k_zero_grads, dim3((unsigned)nrc_cdiv((int64_t)P * (GREC / 4), 256)), dim3(256), 0, s, P, (float4*)grad_records);
        NRC_STAGE(s, "k_zero_grads");
    }
    hipLaunchKernelGGL(k_render_bw, dim3(cam.gx * cam.gy), dim3(256), 0, s, cam, ranges, point_list, (const float4*)splat_records, tile_order, bg[0], bg[1], bg[2],
                       camera_dev, n_contrib, final_T, dL_dpix, grad_records);
    NRC_STAGE(s, "k_render_bw");
#define GS_PBW_ARGS P, cam, camera_dev, means3D, shs, shs_rest, opacities, use_sh, scales, rotations, use_sr, radii, clamped, cov3D, grad_records, dL_dmean2D, dL_dconic, dL_dcolor, \
                    dL_dmean3D, dL_dcov3D, dL_dsh, dL_dsh_rest, dL_dscale, dL_drot, dL_dopacity, ra
    if (ra.p) hipLaunchKernelGGL(k_preprocess_bw<2>, dim3(nrc_cdiv(P, PBW_BLOCK)), dim3(PBW_BLOCK), 0, s, GS_PBW_ARGS);
    else if (shs_rest) hipLaunchKernelGGL(k_preprocess_bw<1>, dim3(nrc_cdiv(P, PBW_BLOCK)), dim3(PBW_BLOCK), 0, s, GS_PBW_ARGS);
    else hipLaunchKernelGGL(k_preprocess_bw<0>, dim3(nrc_cdiv(P, PBW_BLOCK)), dim3(PBW_BLOCK), 0, s, GS_PBW_ARGS);
#undef GS_PBW_ARGS
    NRC_STAGE(s, "k_preprocess_bw");
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_gs_backward(int32_t P, int32_t D, int32_t M, int32_t W, int32_t H, const float* bg_host, const float* means3D, const float* shs,
                    const float* shs_rest, int32_t raw_parameters, const float* opacities, const float* colors_precomp, const float* scales, float scale_modifier, const float* rotations,
                    const float* cov3D_precomp, const float* viewmatrix_host, const float* projmatrix_host, const float* campos_host,
                    const float* camera_dev, float tan_fovx, float tan_fovy, const int32_t* radii, const float* points_xy, const float* conic_opacity,
                    const float* rgb, const uint8_t* clamped, const float* cov3D, const int32_t* point_list, const uint32_t* ranges,
                    const float* splat_records, const uint32_t* tile_order, const uint32_t* n_contrib, const float* final_T, const float* dL_dpix,
                    float* dL_dmean2D, float* dL_dconic,
                    float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dsh_rest, float* dL_dscale,
                    float* dL_drot, float* grad_records, int32_t records_clear, nrc_stream_t stream) {
    NRC_ENTER();
    return gs_backward_impl(P, D, M, W, H, bg_host, means3D, shs, shs_rest, raw_parameters, opacities, colors_precomp, scales, scale_modifier, rotations, cov3D_precomp,
                            viewmatrix_host, projmatrix_host, campos_host, camera_dev, tan_fovx, tan_fovy, radii, points_xy, conic_opacity, rgb, clamped, cov3D, point_list,
                            ranges, splat_records, tile_order, n_contrib, final_T, dL_dpix, dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh,
                            dL_dsh_rest, dL_dscale, dL_drot, grad_records, records_clear, stream, RestAdam{nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f, 1.f, 1.f});
}

/* nrc_gs_backward whose preprocessing backward applies the optimizer's Adam step to the `rest` SH tensor itself (shs_rest_param = the tensor passed as shs_rest in the
 * forward; exp_avg / exp_avg_sq its moments): dL/d shs_rest is not materialised.  Everything else as nrc_gs_backward. */
int nrc_gs_backward_rest_step(int32_t P, int32_t D, int32_t M, int32_t W, int32_t H, const float* bg_host, const float* means3D, const float* shs,
                    float* shs_rest_param, int32_t raw_parameters, const float* opacities, const float* scales, float scale_modifier, const float* rotations,
                    const float* camera_dev, float tan_fovx, float tan_fovy, const int32_t* radii, const float* points_xy, const float* conic_opacity,
                    const float* rgb, const uint8_t* clamped, const float* cov3D, const int32_t* point_list, const uint32_t* ranges,
                    const float* splat_records, const uint32_t* tile_order, const uint32_t* n_contrib, const float* final_T, const float* dL_dpix,
                    float* dL_dmean2D, float* dL_dopacity, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale,
                    float* dL_drot, float* grad_records, int32_t records_clear, float* rest_exp_avg, float* rest_exp_avg_sq, float lr, float beta1, float beta2,
                    float eps, float bias_correction1, float bias_correction2, nrc_stream_t stream) {
    NRC_ENTER();
    if (!shs_rest_param || !rest_exp_avg || !rest_exp_avg_sq || !camera_dev) return NRC_ERR_INVALID;
    return gs_backward_impl(P, D, M, W, H, nullptr, means3D, shs, shs_rest_param, raw_parameters, opacities, nullptr, scales, scale_modifier, rotations, nullptr,
                            nullptr, nullptr, nullptr, camera_dev, tan_fovx, tan_fovy, radii, points_xy, conic_opacity, rgb, clamped, cov3D, point_list,
                            ranges, splat_records, tile_order, n_contrib, final_T, dL_dpix, dL_dmean2D, nullptr, dL_dopacity, nullptr, dL_dmean3D, dL_dcov3D, dL_dsh,
                            nullptr, dL_dscale, dL_drot, grad_records, records_clear, stream,
                            RestAdam{shs_rest_param, rest_exp_avg, rest_exp_avg_sq, lr, beta1, beta2, eps, bias_correction1, bias_correction2});
}

}  // extern "C"
