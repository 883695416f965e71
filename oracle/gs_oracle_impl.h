/*
 * oracle/gs_oracle_impl.h -- body of the 3D Gaussian Splatting rasterizer oracle, instantiated twice by gs_oracle.c:
 *   REAL = float  -> gsf_*  : the parity oracle for the HIP rasterizer (same f32 arithmetic, source order, no FMA -- except in the
 *                             per-pixel blend, see GS_BLEND_POWER)
 *   REAL = double -> gsd_*  : used only by tests to validate the analytic backward against finite differences
 *
 * TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in diff-gaussian-rasterization @
 * 59f5f77e3ddbac3ed9db93ec2cfe99ed6c5d121d (src/Thirdparty/DiffGaussianRasterization.py:9), a pip-from-git dependency
 * that is not under /root/reference; the reference holds no test or golden vector at this boundary.  This file restates
 * the PUBLISHED algorithm (Kerbl et al. 2023, "3D Gaussian Splatting for Real-Time Radiance Field Rendering", sec. 4-6
 * and appendix A) with the conventions the reference's call sites depend on (SURVEY.md Appendix C.2;
 * src/Methods/GaussianSplatting/Renderer.py:60-81):
 *   matrices arrive transposed (w2c.T, (P @ w2c).T) i.e. column-vector matrices in column-major memory; quaternion (w,x,y,z);
 *   SH layout (P,16,3); view-space cull z <= 0.2; EWA projection with the 1.3 x tan(fov) guard band and the +0.3 px
 *   low-pass; radius = ceil(3 sqrt(lambda_max)); 16x16 tiles; keys (tile, depth) with ties broken by Gaussian index
 *   (stable radix sort); alpha = min(0.99, o exp(power)), skip alpha < 1/255, stop before T would drop below 1e-4.
 * In-tree pins: the SH -> RGB step equals src/Methods/GaussianSplatting/utils.py:21-59 (convert_sh_features) and the
 * covariance step equals utils.py:10-18 (build_covariances); both are checked against golden vectors generated from the
 * reference (tests/test_oracle_gs.py), as is the projection-matrix marshalling (Cameras/Perspective.py:96-119).
 */

#define TILE 16

/* The per-pixel blend arithmetic (round 5).  The upstream rasterizer is an nvcc build (-fmad=true): its forward / backward blend loops contain fused
 * multiply-adds, which pairs exactly is not recoverable from source.  The HIP blend kernels are bound by VALU issue, so they state the exponent, the
 * transmittance update and the colour sums with EXPLICIT fused multiply-adds (csrc/gs_raster.hip: blend_power); the oracle states the same
 * fusions here, so the integer outputs that depend on them (n_contrib) and final_T stay bit-exact.  FMA is fmaf / fma: correctly rounded. */
#define GS_BLEND_POWER(A, B, C, dx, dy) FMA((REAL)-0.5, FMA((C) * (dy), (dy), ((A) * (dx)) * (dx)), -(((B) * (dx)) * (dy)))

/* The SAME blend in the order the published source states it (forward.cu renderCUDA of diff-gaussian-rasterization, as described in Kerbl et al. 2023, appendix A):
 *   power = -0.5 (A dx^2 + C dy^2) - B dx dy;  alpha = min(0.99, o exp(power));  test_T = T (1 - alpha);  C_ch += c_ch alpha T
 * with NO fused multiply-add anywhere (this file is built with -ffp-contract=off).  FN(render_source_order) below uses it: the kernels and FN(bin_and_render)
 * state explicit fusions (GS_BLEND_POWER), an algebraic restatement of these expressions -- tests/test_oracle_gs.py and tests/test_gpu_gs_parity.py bound how far
 * the fused statement drifts from this one (last contributor on a tiny fraction of pixels, colour and transmittance by ulps), so that the bit-exact comparison of
 * kernel and fused oracle is not a comparison of the kernel with itself (advisor finding, round 5). */
static inline void FN(xform43)(const REAL* p, const REAL* m, REAL* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void FN(xform44)(const REAL* p, const REAL* m, REAL* o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* standard rotation matrix of an (unnormalised) quaternion (w,x,y,z), row-major; == Cameras/utils.py:180-208 */
static inline void FN(quat_R)(const REAL* q, REAL* R) {
    const REAL r = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - r * z);     R[2] = 2 * (x * z + r * y);
    R[3] = 2 * (x * y + r * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - r * x);
    R[6] = 2 * (x * z - r * y);     R[7] = 2 * (y * z + r * x);     R[8] = 1 - 2 * (x * x + y * y);
}
/* Sigma = (R S)(R S)^T, upper triangle (xx, xy, xz, yy, yz, zz) */
static inline void FN(cov3d)(const REAL* scale, REAL mod, const REAL* q, REAL* c) {
    REAL R[9], A[9];
    FN(quat_R)(q, R);
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) A[3 * i + k] = R[3 * i + k] * (mod * scale[k]);
    c[0] = A[0] * A[0] + A[1] * A[1] + A[2] * A[2];
    c[1] = A[0] * A[3] + A[1] * A[4] + A[2] * A[5];
    c[2] = A[0] * A[6] + A[1] * A[7] + A[2] * A[8];
    c[3] = A[3] * A[3] + A[4] * A[4] + A[5] * A[5];
    c[4] = A[3] * A[6] + A[4] * A[7] + A[5] * A[8];
    c[5] = A[6] * A[6] + A[7] * A[7] + A[8] * A[8];
}
/* M = J R_wc (2x3) for the clamped view-space point t */
static inline void FN(proj_jac)(const REAL* t_in, REAL fx, REAL fy, REAL tanx, REAL tany, const REAL* vm, REAL* Mx, REAL* My,
                                REAL* t_cl, int* gx, int* gy) {
    const REAL limx = (REAL)1.3 * tanx, limy = (REAL)1.3 * tany;
    const REAL txtz = t_in[0] / t_in[2], tytz = t_in[1] / t_in[2];
    t_cl[0] = FMIN(limx, FMAX(-limx, txtz)) * t_in[2];
    t_cl[1] = FMIN(limy, FMAX(-limy, tytz)) * t_in[2];
    t_cl[2] = t_in[2];
    *gx = (txtz < -limx || txtz > limx) ? 0 : 1;
    *gy = (tytz < -limy || tytz > limy) ? 0 : 1;
    const REAL j00 = fx / t_cl[2], j02 = -(fx * t_cl[0]) / (t_cl[2] * t_cl[2]);
    const REAL j11 = fy / t_cl[2], j12 = -(fy * t_cl[1]) / (t_cl[2] * t_cl[2]);
    /* R_wc[i][k] = vm[i + 4k] */
    for (int k = 0; k < 3; k++) {
        Mx[k] = j00 * vm[0 + 4 * k] + j02 * vm[2 + 4 * k];
        My[k] = j11 * vm[1 + 4 * k] + j12 * vm[2 + 4 * k];
    }
}
static inline void FN(sym_mul)(const REAL* c, const REAL* v, REAL* o) { /* o = Sigma v */
    o[0] = c[0] * v[0] + c[1] * v[1] + c[2] * v[2];
    o[1] = c[1] * v[0] + c[3] * v[1] + c[4] * v[2];
    o[2] = c[2] * v[0] + c[4] * v[1] + c[5] * v[2];
}
/* EWA 2-D covariance (a, b, c) with the 0.3 low-pass */
static inline void FN(cov2d)(const REAL* mean, REAL fx, REAL fy, REAL tanx, REAL tany, const REAL* cov3, const REAL* vm, REAL* out) {
    REAL t[3], tc[3], Mx[3], My[3], sx[3], sy[3]; int gx, gy;
    FN(xform43)(mean, vm, t);
    FN(proj_jac)(t, fx, fy, tanx, tany, vm, Mx, My, tc, &gx, &gy);
    FN(sym_mul)(cov3, Mx, sx); FN(sym_mul)(cov3, My, sy);
    out[0] = Mx[0] * sx[0] + Mx[1] * sx[1] + Mx[2] * sx[2] + (REAL)0.3;
    out[1] = Mx[0] * sy[0] + Mx[1] * sy[1] + Mx[2] * sy[2];
    out[2] = My[0] * sy[0] + My[1] * sy[1] + My[2] * sy[2] + (REAL)0.3;
}

#define SH_C0 ((REAL)0.28209479177387814)
#define SH_C1 ((REAL)0.4886025119029199)
static const double FN(SH_C2)[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396};
static const double FN(SH_C3)[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
                                    1.445305721320277, -0.5900435899266435};

/* SH -> RGB for one Gaussian: sh (M,3); returns rgb (+0.5, clamped at 0) and the clamp mask */
static inline void FN(sh_color)(int deg, const REAL* pos, const REAL* campos, const REAL* sh, REAL* rgb, uint8_t* clamped) {
    REAL d[3] = {pos[0] - campos[0], pos[1] - campos[1], pos[2] - campos[2]};
    const REAL len = SQRT(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    d[0] /= len; d[1] /= len; d[2] /= len;
    const REAL x = d[0], y = d[1], z = d[2];
    for (int c = 0; c < 3; c++) {
        REAL r = SH_C0 * sh[c];
        if (deg > 0) {
            r = r - SH_C1 * y * sh[3 + c] + SH_C1 * z * sh[6 + c] - SH_C1 * x * sh[9 + c];
            if (deg > 1) {
                const REAL xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + (REAL)FN(SH_C2)[0] * xy * sh[12 + c] + (REAL)FN(SH_C2)[1] * yz * sh[15 + c] +
                    (REAL)FN(SH_C2)[2] * ((REAL)2 * zz - xx - yy) * sh[18 + c] + (REAL)FN(SH_C2)[3] * xz * sh[21 + c] +
                    (REAL)FN(SH_C2)[4] * (xx - yy) * sh[24 + c];
                if (deg > 2) {
                    r = r + (REAL)FN(SH_C3)[0] * y * ((REAL)3 * xx - yy) * sh[27 + c] + (REAL)FN(SH_C3)[1] * xy * z * sh[30 + c] +
                        (REAL)FN(SH_C3)[2] * y * ((REAL)4 * zz - xx - yy) * sh[33 + c] +
                        (REAL)FN(SH_C3)[3] * z * ((REAL)2 * zz - (REAL)3 * xx - (REAL)3 * yy) * sh[36 + c] +
                        (REAL)FN(SH_C3)[4] * x * ((REAL)4 * zz - xx - yy) * sh[39 + c] + (REAL)FN(SH_C3)[5] * z * (xx - yy) * sh[42 + c] +
                        (REAL)FN(SH_C3)[6] * x * (xx - (REAL)3 * yy) * sh[45 + c];
                }
            }
        }
        r += (REAL)0.5;
        clamped[c] = r < 0;
        rgb[c] = r < 0 ? 0 : r;
    }
}

static inline void FN(tile_rect)(const REAL* p, int radius, int gx, int gy, int* rmin, int* rmax) {
    rmin[0] = IMIN(gx, IMAX(0, (int)((p[0] - radius) / TILE)));
    rmin[1] = IMIN(gy, IMAX(0, (int)((p[1] - radius) / TILE)));
    rmax[0] = IMIN(gx, IMAX(0, (int)((p[0] + radius + TILE - 1) / TILE)));
    rmax[1] = IMIN(gy, IMAX(0, (int)((p[1] + radius + TILE - 1) / TILE)));
}

/* ---- stage A: per-Gaussian preprocess.  Returns the number of (tile, Gaussian) instances. ---- */
int64_t FN(preprocess)(int P, int D, int M, int W, int H, const REAL* means3D, const REAL* shs, const REAL* colors_precomp,
                       const REAL* opacities, const REAL* scales, REAL scale_modifier, const REAL* rotations, const REAL* cov3D_precomp,
                       const REAL* viewmatrix, const REAL* projmatrix, const REAL* campos, REAL tan_fovx, REAL tan_fovy,
                       int32_t* radii, REAL* depths, REAL* points_xy, REAL* conic_opacity, REAL* rgb, uint8_t* clamped, REAL* cov3D,
                       uint32_t* tiles_touched) {
    const REAL fx = W / ((REAL)2 * tan_fovx), fy = H / ((REAL)2 * tan_fovy);
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    int64_t total = 0;
    /* OpenMP: every Gaussian is independent, the instance total is an integer reduction -- results do not depend on the thread count */
#pragma omp parallel for schedule(static) reduction(+ : total)
    for (int i = 0; i < P; i++) {
        radii[i] = 0; tiles_touched[i] = 0;
        depths[i] = 0; points_xy[2 * i] = points_xy[2 * i + 1] = 0;
        for (int k = 0; k < 4; k++) conic_opacity[4 * i + k] = 0;
        for (int k = 0; k < 3; k++) { rgb[3 * i + k] = 0; clamped[3 * i + k] = 0; }
        const REAL* p = means3D + 3 * i;
        REAL pv[3]; FN(xform43)(p, viewmatrix, pv);
        if (cov3D_precomp) for (int k = 0; k < 6; k++) cov3D[6 * i + k] = cov3D_precomp[6 * i + k];
        else FN(cov3d)(scales + 3 * i, scale_modifier, rotations + 4 * i, cov3D + 6 * i);
        if (pv[2] <= (REAL)0.2) continue;
        REAL ph[4]; FN(xform44)(p, projmatrix, ph);
        const REAL pw = (REAL)1 / (ph[3] + (REAL)0.0000001);
        const REAL ndc[2] = {ph[0] * pw, ph[1] * pw};
        REAL cov[3]; FN(cov2d)(p, fx, fy, tan_fovx, tan_fovy, cov3D + 6 * i, viewmatrix, cov);
        const REAL det = cov[0] * cov[2] - cov[1] * cov[1];
        if (det == 0) continue;
        const REAL det_inv = (REAL)1 / det;
        const REAL conic[3] = {cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv};
        const REAL mid = (REAL)0.5 * (cov[0] + cov[2]);
        const REAL lambda1 = mid + SQRT(FMAX((REAL)0.1, mid * mid - det));
        const REAL lambda2 = mid - SQRT(FMAX((REAL)0.1, mid * mid - det));
        const int my_radius = (int)CEIL((REAL)3 * SQRT(FMAX(lambda1, lambda2)));
        const REAL pix[2] = {((ndc[0] + (REAL)1) * W - (REAL)1) * (REAL)0.5, ((ndc[1] + (REAL)1) * H - (REAL)1) * (REAL)0.5};
        int rmin[2], rmax[2];
        FN(tile_rect)(pix, my_radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (colors_precomp) for (int k = 0; k < 3; k++) rgb[3 * i + k] = colors_precomp[3 * i + k];
        else FN(sh_color)(D, p, campos, shs + (size_t)i * M * 3, rgb + 3 * i, clamped + 3 * i);
        depths[i] = pv[2]; radii[i] = my_radius;
        points_xy[2 * i] = pix[0]; points_xy[2 * i + 1] = pix[1];
        conic_opacity[4 * i] = conic[0]; conic_opacity[4 * i + 1] = conic[1]; conic_opacity[4 * i + 2] = conic[2];
        conic_opacity[4 * i + 3] = opacities[i];
        tiles_touched[i] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
        total += tiles_touched[i];
    }
    return total;
}

/* ---- stage B: binning (instances sorted by (tile, depth) with index tie-break) + tile ranges + blending ---- */
typedef struct { uint64_t key; int32_t id; } FN(inst_t);
static int FN(inst_cmp)(const void* a, const void* b) {
    const FN(inst_t)* x = (const FN(inst_t)*)a; const FN(inst_t)* y = (const FN(inst_t)*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);
}
void FN(bin_and_render)(int P, int W, int H, const REAL* bg, const int32_t* radii, const REAL* depths, const REAL* points_xy,
                        const REAL* conic_opacity, const REAL* rgb, int64_t num_rendered, int32_t* point_list, uint32_t* ranges,
                        REAL* out_color, uint32_t* n_contrib, REAL* final_T) {
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    FN(inst_t)* inst = (FN(inst_t)*)malloc(sizeof(FN(inst_t)) * (size_t)(num_rendered > 0 ? num_rendered : 1));
    int64_t n = 0;
    for (int i = 0; i < P; i++) {
        if (radii[i] <= 0) continue;
        int rmin[2], rmax[2];
        FN(tile_rect)(points_xy + 2 * i, radii[i], gx, gy, rmin, rmax);
        const float df = (float)depths[i];
        uint32_t dbits; memcpy(&dbits, &df, 4);
        for (int y = rmin[1]; y < rmax[1]; y++)
            for (int x = rmin[0]; x < rmax[0]; x++) {
                inst[n].key = ((uint64_t)(y * gx + x) << 32) | dbits; inst[n].id = i; n++;
            }
    }
    qsort(inst, (size_t)n, sizeof(FN(inst_t)), FN(inst_cmp));
    for (int t = 0; t < gx * gy; t++) { ranges[2 * t] = 0; ranges[2 * t + 1] = 0; }
    for (int64_t k = 0; k < n; k++) {
        point_list[k] = inst[k].id;
        const uint32_t tile = (uint32_t)(inst[k].key >> 32);
        if (k == 0 || (uint32_t)(inst[k - 1].key >> 32) != tile) ranges[2 * tile] = (uint32_t)k;
        if (k == n - 1 || (uint32_t)(inst[k + 1].key >> 32) != tile) ranges[2 * tile + 1] = (uint32_t)(k + 1);
    }
    free(inst);
    /* OpenMP: pixels are independent */
#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tile = (py / TILE) * gx + px / TILE;
            REAL T = 1, C[3] = {0, 0, 0};
            uint32_t contributor = 0, last = 0;
            for (uint32_t k = ranges[2 * tile]; k < ranges[2 * tile + 1]; k++) {
                contributor++;
                const int id = point_list[k];
                const REAL dx = points_xy[2 * id] - (REAL)px, dy = points_xy[2 * id + 1] - (REAL)py;
                const REAL* co = conic_opacity + 4 * id;
                const REAL power = GS_BLEND_POWER(co[0], co[1], co[2], dx, dy);
                if (power > 0) continue;
                const REAL alpha = FMIN((REAL)0.99, co[3] * EXP(power));
                if (alpha < (REAL)1 / (REAL)255) continue;
                const REAL test_T = FMA(-T, alpha, T);
                if (test_T < (REAL)0.0001) break;
                const REAL w = alpha * T;
                for (int c = 0; c < 3; c++) C[c] = FMA(rgb[3 * id + c], w, C[c]);
                T = test_T;
                last = contributor;
            }
            final_T[py * W + px] = T;
            n_contrib[py * W + px] = last;
            for (int c = 0; c < 3; c++) out_color[(size_t)c * H * W + py * W + px] = C[c] + T * bg[c];
        }
}

/* the image of given tile lists (ranges / point_list of FN(bin_and_render)) with the source-order blend */
void FN(render_source_order)(int W, int H, const REAL* bg, const REAL* points_xy, const REAL* conic_opacity, const REAL* rgb, const int32_t* point_list,
                             const uint32_t* ranges, REAL* out_color, uint32_t* n_contrib, REAL* final_T) {
    const int gx = (W + TILE - 1) / TILE;
#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int tile = (py / TILE) * gx + px / TILE;
            REAL T = 1, C[3] = {0, 0, 0};
            uint32_t contributor = 0, last = 0;
            for (uint32_t k = ranges[2 * tile]; k < ranges[2 * tile + 1]; k++) {
                contributor++;
                const int id = point_list[k];
                const REAL dx = points_xy[2 * id] - (REAL)px, dy = points_xy[2 * id + 1] - (REAL)py;
                const REAL* co = conic_opacity + 4 * id;
                const REAL power = (REAL)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                if (power > 0) continue;
                const REAL alpha = FMIN((REAL)0.99, co[3] * EXP(power));
                if (alpha < (REAL)1 / (REAL)255) continue;
                const REAL test_T = T * (1 - alpha);
                if (test_T < (REAL)0.0001) break;
                for (int c = 0; c < 3; c++) C[c] += rgb[3 * id + c] * alpha * T;
                T = test_T;
                last = contributor;
            }
            final_T[py * W + px] = T;
            n_contrib[py * W + px] = last;
            for (int c = 0; c < 3; c++) out_color[(size_t)c * H * W + py * W + px] = C[c] + T * bg[c];
        }
}

/* ---- backward ---- */
void FN(backward)(int P, int D, int M, int W, int H, const REAL* bg, const REAL* means3D, const REAL* shs, const REAL* colors_precomp,
                  const REAL* scales, REAL scale_modifier, const REAL* rotations, const REAL* cov3D_precomp, const REAL* viewmatrix,
                  const REAL* projmatrix, const REAL* campos, REAL tan_fovx, REAL tan_fovy, const int32_t* radii, const REAL* points_xy,
                  const REAL* conic_opacity, const REAL* rgb, const uint8_t* clamped, const REAL* cov3D, const int32_t* point_list,
                  const uint32_t* ranges, const uint32_t* n_contrib, const REAL* final_T, const REAL* dL_dpix,
                  REAL* dL_dmean2D, REAL* dL_dconic, REAL* dL_dopacity, REAL* dL_dcolor, REAL* dL_dmean3D, REAL* dL_dcov3D, REAL* dL_dsh,
                  REAL* dL_dscale, REAL* dL_drot) {
    const int gx = (W + TILE - 1) / TILE;
    const REAL fx = W / ((REAL)2 * tan_fovx), fy = H / ((REAL)2 * tan_fovy);
    memset(dL_dmean2D, 0, sizeof(REAL) * 3 * P); memset(dL_dconic, 0, sizeof(REAL) * 4 * P); memset(dL_dopacity, 0, sizeof(REAL) * P);
    memset(dL_dcolor, 0, sizeof(REAL) * 3 * P); memset(dL_dmean3D, 0, sizeof(REAL) * 3 * P); memset(dL_dcov3D, 0, sizeof(REAL) * 6 * P);
    memset(dL_dsh, 0, sizeof(REAL) * 3 * (size_t)M * P); memset(dL_dscale, 0, sizeof(REAL) * 3 * P); memset(dL_drot, 0, sizeof(REAL) * 4 * P);
    /* (1) blending backward: per pixel, back to front */
    const REAL ddelx_dx = (REAL)0.5 * W, ddely_dy = (REAL)0.5 * H;
    /* OpenMP over pixel rows with atomic adds into the per-Gaussian sums (what the reference's CUDA kernel does with atomicAdd).  With ONE
     * thread (oracle.gs_backward's default, used by every parity test) the order of the additions is the serial one and the result is
     * reproducible bit for bit; more threads (bench.py's CPU baseline) change only the order of the float additions. */
#pragma omp parallel for schedule(dynamic, 4)
    for (int py = 0; py < H; py++)
        for (int px = 0; px < W; px++) {
            const int pix = py * W + px, tile = (py / TILE) * gx + px / TILE;
            const REAL T_final = final_T[pix];
            REAL T = T_final;
            const uint32_t last = n_contrib[pix];
            REAL accum[3] = {0, 0, 0}, last_color[3] = {0, 0, 0}, last_alpha = 0;
            const REAL g[3] = {dL_dpix[pix], dL_dpix[(size_t)H * W + pix], dL_dpix[(size_t)2 * H * W + pix]};
            const uint32_t r0 = ranges[2 * tile];
            for (int64_t k = (int64_t)r0 + last - 1; k >= (int64_t)r0; k--) {
                const int id = point_list[k];
                const REAL dx = points_xy[2 * id] - (REAL)px, dy = points_xy[2 * id + 1] - (REAL)py;
                const REAL* co = conic_opacity + 4 * id;
                const REAL power = GS_BLEND_POWER(co[0], co[1], co[2], dx, dy);
                if (power > 0) continue;
                const REAL G = EXP(power);
                const REAL alpha = FMIN((REAL)0.99, co[3] * G);
                if (alpha < (REAL)1 / (REAL)255) continue;
                T = T / (1 - alpha);
                const REAL dchannel_dcolor = alpha * T;
                REAL dL_dalpha = 0;
                for (int c = 0; c < 3; c++) {
                    const REAL col = rgb[3 * id + c];
                    accum[c] = last_alpha * last_color[c] + (1 - last_alpha) * accum[c];
                    last_color[c] = col;
                    dL_dalpha += (col - accum[c]) * g[c];
                    { const REAL add_ = dchannel_dcolor * g[c];
_Pragma("omp atomic")
                    dL_dcolor[3 * id + c] += add_; }
                }
                dL_dalpha *= T;
                last_alpha = alpha;
                REAL bg_dot = 0;
                for (int c = 0; c < 3; c++) bg_dot += bg[c] * g[c];
                dL_dalpha += (-T_final / (1 - alpha)) * bg_dot;
                const REAL dL_dG = co[3] * dL_dalpha;
                const REAL gdx = G * dx, gdy = G * dy;
                const REAL dG_ddelx = -gdx * co[0] - gdy * co[1];
                const REAL dG_ddely = -gdy * co[2] - gdx * co[1];
                { const REAL add_ = dL_dG * dG_ddelx * ddelx_dx;
_Pragma("omp atomic")
                dL_dmean2D[3 * id] += add_; }
                { const REAL add_ = dL_dG * dG_ddely * ddely_dy;
_Pragma("omp atomic")
                dL_dmean2D[3 * id + 1] += add_; }
                { const REAL add_ = (REAL)-0.5 * gdx * dx * dL_dG;
_Pragma("omp atomic")
                dL_dconic[4 * id] += add_; }
                { const REAL add_ = (REAL)-0.5 * gdx * dy * dL_dG;
_Pragma("omp atomic")
                dL_dconic[4 * id + 1] += add_; }
                { const REAL add_ = (REAL)-0.5 * gdy * dy * dL_dG;
_Pragma("omp atomic")
                dL_dconic[4 * id + 3] += add_; }
                { const REAL add_ = G * dL_dalpha;
_Pragma("omp atomic")
                dL_dopacity[id] += add_; }
            }
        }
    /* (2) per-Gaussian backward: independent Gaussians */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        if (!(radii[i] > 0)) continue;
        const REAL* mean = means3D + 3 * i;
        const REAL* c3 = cov3D + 6 * i;
        /* 2a: conic -> cov2D -> (cov3D, mean via the projection Jacobian) */
        REAL t[3], tc[3], Mx[3], My[3], sx[3], sy[3]; int gmx, gmy;
        FN(xform43)(mean, viewmatrix, t);
        FN(proj_jac)(t, fx, fy, tan_fovx, tan_fovy, viewmatrix, Mx, My, tc, &gmx, &gmy);
        FN(sym_mul)(c3, Mx, sx); FN(sym_mul)(c3, My, sy);
        const REAL a = Mx[0] * sx[0] + Mx[1] * sx[1] + Mx[2] * sx[2] + (REAL)0.3;
        const REAL b = Mx[0] * sy[0] + Mx[1] * sy[1] + Mx[2] * sy[2];
        const REAL c = My[0] * sy[0] + My[1] * sy[1] + My[2] * sy[2] + (REAL)0.3;
        const REAL gcx = dL_dconic[4 * i], gcy = dL_dconic[4 * i + 1], gcz = dL_dconic[4 * i + 3];
        const REAL denom = a * c - b * b;
        const REAL denom2inv = (REAL)1 / (denom * denom + (REAL)0.0000001);
        REAL dL_da = 0, dL_db = 0, dL_dc = 0;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * gcx + 2 * b * c * gcy + (denom - a * c) * gcz);
            dL_dc = denom2inv * (-a * a * gcz + 2 * a * b * gcy + (denom - a * c) * gcx);
            dL_db = denom2inv * 2 * (b * c * gcx - (denom + 2 * b * b) * gcy + a * b * gcz);
            REAL* o = dL_dcov3D + 6 * i;
            o[0] = Mx[0] * Mx[0] * dL_da + Mx[0] * My[0] * dL_db + My[0] * My[0] * dL_dc;
            o[3] = Mx[1] * Mx[1] * dL_da + Mx[1] * My[1] * dL_db + My[1] * My[1] * dL_dc;
            o[5] = Mx[2] * Mx[2] * dL_da + Mx[2] * My[2] * dL_db + My[2] * My[2] * dL_dc;
            o[1] = 2 * Mx[0] * Mx[1] * dL_da + (Mx[0] * My[1] + Mx[1] * My[0]) * dL_db + 2 * My[0] * My[1] * dL_dc;
            o[2] = 2 * Mx[0] * Mx[2] * dL_da + (Mx[0] * My[2] + Mx[2] * My[0]) * dL_db + 2 * My[0] * My[2] * dL_dc;
            o[4] = 2 * Mx[2] * Mx[1] * dL_da + (Mx[1] * My[2] + Mx[2] * My[1]) * dL_db + 2 * My[1] * My[2] * dL_dc;
        }
        /* dL/dM rows */
        REAL dMx[3], dMy[3];
        for (int k = 0; k < 3; k++) { dMx[k] = 2 * sx[k] * dL_da + sy[k] * dL_db; dMy[k] = 2 * sy[k] * dL_dc + sx[k] * dL_db; }
        /* dL/dJ = dL/dM R_wc^T ; only the four non-constant entries */
        REAL dJ00 = 0, dJ02 = 0, dJ11 = 0, dJ12 = 0;
        for (int k = 0; k < 3; k++) {
            dJ00 += viewmatrix[0 + 4 * k] * dMx[k]; dJ02 += viewmatrix[2 + 4 * k] * dMx[k];
            dJ11 += viewmatrix[1 + 4 * k] * dMy[k]; dJ12 += viewmatrix[2 + 4 * k] * dMy[k];
        }
        const REAL tz = (REAL)1 / tc[2], tz2 = tz * tz, tz3 = tz2 * tz;
        const REAL dtx = (REAL)gmx * -fx * tz2 * dJ02;
        const REAL dty = (REAL)gmy * -fy * tz2 * dJ12;
        const REAL dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2 * fx * tc[0]) * tz3 * dJ02 + (2 * fy * tc[1]) * tz3 * dJ12;
        REAL dmean[3];
        for (int k = 0; k < 3; k++) dmean[k] = viewmatrix[0 + 4 * k] * dtx + viewmatrix[1 + 4 * k] * dty + viewmatrix[2 + 4 * k] * dtz;
        /* 2b: screen-space mean -> 3-D mean through the projection */
        REAL mh[4]; FN(xform44)(mean, projmatrix, mh);
        const REAL mw = (REAL)1 / (mh[3] + (REAL)0.0000001);
        const REAL mul1 = mh[0] * mw * mw, mul2 = mh[1] * mw * mw;
        const REAL g2x = dL_dmean2D[3 * i], g2y = dL_dmean2D[3 * i + 1];
        dmean[0] += (projmatrix[0] * mw - projmatrix[3] * mul1) * g2x + (projmatrix[1] * mw - projmatrix[3] * mul2) * g2y;
        dmean[1] += (projmatrix[4] * mw - projmatrix[7] * mul1) * g2x + (projmatrix[5] * mw - projmatrix[7] * mul2) * g2y;
        dmean[2] += (projmatrix[8] * mw - projmatrix[11] * mul1) * g2x + (projmatrix[9] * mw - projmatrix[11] * mul2) * g2y;
        /* 2c: colour -> SH coefficients and (through the view direction) the mean */
        if (!colors_precomp) {
            REAL dir0[3] = {mean[0] - campos[0], mean[1] - campos[1], mean[2] - campos[2]};
            const REAL sum2 = dir0[0] * dir0[0] + dir0[1] * dir0[1] + dir0[2] * dir0[2];
            const REAL len = SQRT(sum2);
            const REAL x = dir0[0] / len, y = dir0[1] / len, z = dir0[2] / len;
            const REAL* sh = shs + (size_t)i * M * 3;
            REAL* gsh = dL_dsh + (size_t)i * M * 3;
            REAL dRGB[3], ddir[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ch++) dRGB[ch] = clamped[3 * i + ch] ? 0 : dL_dcolor[3 * i + ch];
            /* basis values and their gradients w.r.t. (x,y,z) */
            REAL Bv[16], Bx[16], By[16], Bz[16];
            for (int k = 0; k < 16; k++) { Bv[k] = Bx[k] = By[k] = Bz[k] = 0; }
            Bv[0] = SH_C0;
            if (D > 0) {
                Bv[1] = -SH_C1 * y; By[1] = -SH_C1;
                Bv[2] = SH_C1 * z;  Bz[2] = SH_C1;
                Bv[3] = -SH_C1 * x; Bx[3] = -SH_C1;
                if (D > 1) {
                    const REAL xx = x * x, yy = y * y, zz = z * z;
                    const REAL c0 = (REAL)FN(SH_C2)[0], c1 = (REAL)FN(SH_C2)[1], c2 = (REAL)FN(SH_C2)[2], c3_ = (REAL)FN(SH_C2)[3], c4 = (REAL)FN(SH_C2)[4];
                    Bv[4] = c0 * x * y; Bx[4] = c0 * y; By[4] = c0 * x;
                    Bv[5] = c1 * y * z; By[5] = c1 * z; Bz[5] = c1 * y;
                    Bv[6] = c2 * (2 * zz - xx - yy); Bx[6] = c2 * -2 * x; By[6] = c2 * -2 * y; Bz[6] = c2 * 4 * z;
                    Bv[7] = c3_ * x * z; Bx[7] = c3_ * z; Bz[7] = c3_ * x;
                    Bv[8] = c4 * (xx - yy); Bx[8] = c4 * 2 * x; By[8] = c4 * -2 * y;
                    if (D > 2) {
                        const REAL e0 = (REAL)FN(SH_C3)[0], e1 = (REAL)FN(SH_C3)[1], e2 = (REAL)FN(SH_C3)[2], e3 = (REAL)FN(SH_C3)[3],
                                   e4 = (REAL)FN(SH_C3)[4], e5 = (REAL)FN(SH_C3)[5], e6 = (REAL)FN(SH_C3)[6];
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
                for (int ch = 0; ch < 3; ch++) {
                    gsh[3 * k + ch] = Bv[k] * dRGB[ch];
                    ddir[0] += Bx[k] * sh[3 * k + ch] * dRGB[ch];
                    ddir[1] += By[k] * sh[3 * k + ch] * dRGB[ch];
                    ddir[2] += Bz[k] * sh[3 * k + ch] * dRGB[ch];
                }
            /* normalisation backward: d(v/|v|) */
            const REAL inv32 = (REAL)1 / SQRT(sum2 * sum2 * sum2);
            dmean[0] += ((sum2 - dir0[0] * dir0[0]) * ddir[0] - dir0[1] * dir0[0] * ddir[1] - dir0[2] * dir0[0] * ddir[2]) * inv32;
            dmean[1] += (-dir0[0] * dir0[1] * ddir[0] + (sum2 - dir0[1] * dir0[1]) * ddir[1] - dir0[2] * dir0[1] * ddir[2]) * inv32;
            dmean[2] += (-dir0[0] * dir0[2] * ddir[0] - dir0[1] * dir0[2] * ddir[1] + (sum2 - dir0[2] * dir0[2]) * ddir[2]) * inv32;
        }
        for (int k = 0; k < 3; k++) dL_dmean3D[3 * i + k] = dmean[k];
        /* 2d: cov3D -> scale, rotation */
        if (!cov3D_precomp) {
            const REAL* q = rotations + 4 * i; const REAL* s = scales + 3 * i;
            REAL R[9], A[9], Gs[9], dA[9];
            FN(quat_R)(q, R);
            for (int r_ = 0; r_ < 3; r_++) for (int k = 0; k < 3; k++) A[3 * r_ + k] = R[3 * r_ + k] * (scale_modifier * s[k]);
            const REAL* o = dL_dcov3D + 6 * i;
            Gs[0] = o[0]; Gs[1] = (REAL)0.5 * o[1]; Gs[2] = (REAL)0.5 * o[2];
            Gs[3] = (REAL)0.5 * o[1]; Gs[4] = o[3]; Gs[5] = (REAL)0.5 * o[4];
            Gs[6] = (REAL)0.5 * o[2]; Gs[7] = (REAL)0.5 * o[4]; Gs[8] = o[5];
            for (int r_ = 0; r_ < 3; r_++) for (int k = 0; k < 3; k++)
                dA[3 * r_ + k] = 2 * (Gs[3 * r_] * A[k] + Gs[3 * r_ + 1] * A[3 + k] + Gs[3 * r_ + 2] * A[6 + k]);
            REAL dR[9];
            for (int k = 0; k < 3; k++) {
                dL_dscale[3 * i + k] = scale_modifier * (dA[k] * R[k] + dA[3 + k] * R[3 + k] + dA[6 + k] * R[6 + k]);
                for (int r_ = 0; r_ < 3; r_++) dR[3 * r_ + k] = dA[3 * r_ + k] * (scale_modifier * s[k]);
            }
            const REAL r = q[0], x = q[1], y = q[2], z = q[3];
            dL_drot[4 * i] = 2 * (-z * dR[1] + y * dR[2] + z * dR[3] - x * dR[5] - y * dR[6] + x * dR[7]);
            dL_drot[4 * i + 1] = 2 * (y * dR[1] + z * dR[2] + y * dR[3] - 2 * x * dR[4] - r * dR[5] + z * dR[6] + r * dR[7] - 2 * x * dR[8]);
            dL_drot[4 * i + 2] = 2 * (-2 * y * dR[0] + x * dR[1] + r * dR[2] + x * dR[3] + z * dR[5] - r * dR[6] + z * dR[7] - 2 * y * dR[8]);
            dL_drot[4 * i + 3] = 2 * (-2 * z * dR[0] - r * dR[1] + x * dR[2] + r * dR[3] - 2 * z * dR[4] + y * dR[5] + x * dR[6] + y * dR[7]);
        }
    }
}
#undef TILE
#undef SH_C0
#undef SH_C1
