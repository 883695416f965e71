/*
 * oracle/ngp_oracle.c -- CPU restatement (plain C, scalar, single thread per call) of the reference's in-tree
 * CUDA extension `VolumeRenderingV2` and of `MortonEncoding`.
 *
 * THIS IS TEST INFRASTRUCTURE. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product path (nerficg_amd/) never links, imports or calls anything in oracle/.
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference/):
 *   R  = src/Methods/InstantNGP/VolumeRenderingV2/csrc/raymarching.cu
 *   V  = src/Methods/InstantNGP/VolumeRenderingV2/csrc/volumerendering.cu
 *   I  = src/Methods/InstantNGP/VolumeRenderingV2/csrc/intersection.cu
 *   L  = src/Methods/InstantNGP/VolumeRenderingV2/csrc/losses.cu
 *   M  = src/CudaUtils/MortonEncoding/MortonEncoding/morton_encoding.cu
 *
 * Parity pin status: the reference ships no tests/golden vectors for these kernels and they cannot be compiled here
 * (nvcc + CUDA torch). Pins used instead (tests/test_oracle_*.py): integrate_samples golden vectors generated from
 * the reference's own PyTorch code (compositing math), numpy.packbits, Morton round trips, finite differences.
 *
 * Arithmetic policy: IEEE-754 binary32, source-order evaluation, NO fused multiply-add contraction (built with
 * -ffp-contract=off); the HIP kernels that must be index-exact are built the same way. (nvcc's default -fmad=true
 * may contract a*b+c in the reference binary; that sub-ulp behaviour is not recoverable from source and is
 * documented as unpinned in DESIGN.md.)  `__expf` (V:30,127,233) is restated with expf(): tolerance in tests.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SQRT3 1.73205080757f

static inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); } /* helper_math.h:280 */
static inline float signf_(float x) { return copysignf(1.0f, x); }                       /* R:7 */
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* R:11-13 */
static inline float calc_dt(float t, float exp_step_factor, int max_samples, int grid_size, float scale) {
    return clampf(t * exp_step_factor, SQRT3 / max_samples, SQRT3 * 2 * scale / grid_size);
}
/* R:19-23 */
static inline int mip_from_pos(float x, float y, float z, int cascades) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int e; frexpf(mx, &e);
    return imin(cascades - 1, imax(0, e + 1));
}
/* R:29-32 */
static inline int mip_from_dt(float dt, int grid_size, int cascades) {
    int e; frexpf(dt * grid_size, &e);
    return imin(cascades - 1, imax(0, e));
}
/* R:35-42 */
static inline uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
/* R:44-50 */
static inline uint32_t morton3D_(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
/* R:52-60 */
static inline uint32_t morton3D_invert_(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

/* R:62-70 (kernel), R:72-88 (host) */
void oracle_morton3D(const int32_t* coords, int64_t n, int32_t* indices) {
    for (int64_t i = 0; i < n; i++)
        indices[i] = (int32_t)morton3D_((uint32_t)coords[3 * i], (uint32_t)coords[3 * i + 1], (uint32_t)coords[3 * i + 2]);
}
/* R:90-101 : note the arithmetic (sign-propagating) shift of the *int* index before the unsigned mask */
void oracle_morton3D_invert(const int32_t* indices, int64_t n, int32_t* coords) {
    for (int64_t i = 0; i < n; i++) {
        const int32_t ind = indices[i];
        coords[3 * i + 0] = (int32_t)morton3D_invert_((uint32_t)(ind >> 0));
        coords[3 * i + 1] = (int32_t)morton3D_invert_((uint32_t)(ind >> 1));
        coords[3 * i + 2] = (int32_t)morton3D_invert_((uint32_t)(ind >> 2));
    }
}
/* R:123-141 : one output byte = 8 consecutive cells, bit i <- cell 8n+i (little-endian within the byte) */
void oracle_packbits(const float* grid, int64_t n_bytes, float thr, uint8_t* bitfield) {
    for (int64_t n = 0; n < n_bytes; n++) {
        uint8_t bits = 0;
        for (int i = 0; i < 8; i++) bits |= (grid[8 * n + i] > thr) ? (uint8_t)(1u << i) : 0;
        bitfield[n] = bits;
    }
}

/* ---- shared DDA step (R:205-233 == R:246-278 == R:367-401).  Returns 1 if the cell is occupied. ---- */
typedef struct { float ox, oy, oz, dx, dy, dz, dxi, dyi, dzi; } ray_t;

static inline int march_probe(const ray_t* r, float t, float dt, const uint8_t* bitfield, int cascades, int grid_size,
                              float scale, float* x, float* y, float* z, int* nx, int* ny, int* nz, float* mip_bound) {
    const uint32_t grid_size3 = (uint32_t)grid_size * grid_size * grid_size;
    *x = r->ox + t * r->dx; *y = r->oy + t * r->dy; *z = r->oz + t * r->dz;
    const int mip = imax(mip_from_pos(*x, *y, *z, cascades), mip_from_dt(dt, grid_size, cascades));
    *mip_bound = fminf(scalbnf(1.0f, mip - 1), scale);
    const float inv = 1 / *mip_bound;
    *nx = (int)clampf(0.5f * (*x * inv + 1) * grid_size, 0.0f, grid_size - 1.0f);
    *ny = (int)clampf(0.5f * (*y * inv + 1) * grid_size, 0.0f, grid_size - 1.0f);
    *nz = (int)clampf(0.5f * (*z * inv + 1) * grid_size, 0.0f, grid_size - 1.0f);
    const uint32_t idx = (uint32_t)mip * grid_size3 + morton3D_((uint32_t)*nx, (uint32_t)*ny, (uint32_t)*nz);
    return (bitfield[idx / 8] & (1 << (idx % 8))) != 0;
}
/* the voxel-skip branch R:224-233; `scale_for_dt` is `scale` in the train kernel and `cascades` in the test kernel (R:370,399) */
static inline float march_skip(const ray_t* r, float t, float x, float y, float z, int nx, int ny, int nz, float mip_bound,
                               float exp_step_factor, int max_samples, int grid_size, float scale_for_dt) {
    const float gi = 1.0f / grid_size;
    const float tx = (((nx + 0.5f + 0.5f * signf_(r->dx)) * gi * 2 - 1) * mip_bound - x) * r->dxi;
    const float ty = (((ny + 0.5f + 0.5f * signf_(r->dy)) * gi * 2 - 1) * mip_bound - y) * r->dyi;
    const float tz = (((nz + 0.5f + 0.5f * signf_(r->dz)) * gi * 2 - 1) * mip_bound - z) * r->dzi;
    const float t_target = t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
    do { t += calc_dt(t, exp_step_factor, max_samples, grid_size, scale_for_dt); } while (t < t_target);
    return t;
}
static inline ray_t load_ray(const float* o, const float* d, int64_t r) {
    ray_t q;
    q.ox = o[3 * r]; q.oy = o[3 * r + 1]; q.oz = o[3 * r + 2];
    q.dx = d[3 * r]; q.dy = d[3 * r + 1]; q.dz = d[3 * r + 2];
    q.dxi = 1.0f / q.dx; q.dyi = 1.0f / q.dy; q.dzi = 1.0f / q.dz;
    return q;
}

/*
 * R:166-280 (kernel) + R:283-332 (host).  The reference reserves output slots with two atomicAdds (R:237-238), so the
 * order of rays in rays_a / of segments in the sample arrays is arrival order (non-deterministic).  This restatement
 * serialises rays in index order: rays_a[n] = (n, start, count) with start = exclusive prefix sum of counts -- one of
 * the orders the reference can produce; per-ray contents (count, xyz, dir, dt, t) are order-independent.
 * pass 0: only counts (outputs may be NULL) ; returns total sample count.
 */
int64_t oracle_raymarching_train(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* bitfield,
                                 int cascades, float scale, float exp_step_factor, const float* noise, int grid_size,
                                 int max_samples, int64_t n_rays, int64_t* rays_a, float* xyzs, float* dirs, float* deltas,
                                 float* ts, int32_t* counter) {
    /* Rays are independent: both passes run over the rays in parallel (OpenMP; bench.py's cpu_baseline uses every host core), the exclusive
     * prefix sum of the counts between them is serial.  Per-ray arithmetic and the ray-index order of the outputs are unchanged. */
    int32_t* cnt = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_rays > 0 ? n_rays : 1));
    float* first_t = (float*)malloc(sizeof(float) * (size_t)(n_rays > 0 ? n_rays : 1));
    int64_t* starts = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n_rays > 0 ? n_rays : 1));
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t r = 0; r < n_rays; r++) {
        const ray_t q = load_ray(rays_o, rays_d, r);
        float t1 = hits_t[2 * r];
        const float t2 = hits_t[2 * r + 1];
        if (t1 >= 0) { /* R:195-198 */
            const float dt = calc_dt(t1, exp_step_factor, max_samples, grid_size, scale);
            t1 += dt * noise[r];
        }
        /* first pass R:200-234 */
        float t = t1; int n_samples = 0;
        while (0 <= t && t < t2 && n_samples < max_samples) {
            float x, y, z, mb; int nx, ny, nz;
            const float dt = calc_dt(t, exp_step_factor, max_samples, grid_size, scale);
            if (march_probe(&q, t, dt, bitfield, cascades, grid_size, scale, &x, &y, &z, &nx, &ny, &nz, &mb)) { t += dt; n_samples++; }
            else t = march_skip(&q, t, x, y, z, nx, ny, nz, mb, exp_step_factor, max_samples, grid_size, scale);
        }
        cnt[r] = n_samples; first_t[r] = t1;
    }
    int64_t total = 0;
    for (int64_t r = 0; r < n_rays; r++) { starts[r] = total; total += cnt[r]; }
    if (rays_a) {
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < n_rays; r++) { rays_a[3 * r] = r; rays_a[3 * r + 1] = starts[r]; rays_a[3 * r + 2] = cnt[r]; }
    }
    if (xyzs) {
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t r = 0; r < n_rays; r++) {
            /* second pass R:243-279 */
            const ray_t q = load_ray(rays_o, rays_d, r);
            const float t2 = hits_t[2 * r + 1];
            const int64_t start = starts[r];
            const int n_samples = cnt[r];
            float t = first_t[r]; int s = 0;
            while (t < t2 && s < n_samples) {
                float x, y, z, mb; int nx, ny, nz;
                const float dt = calc_dt(t, exp_step_factor, max_samples, grid_size, scale);
                if (march_probe(&q, t, dt, bitfield, cascades, grid_size, scale, &x, &y, &z, &nx, &ny, &nz, &mb)) {
                    const int64_t k = start + s;
                    xyzs[3 * k] = x; xyzs[3 * k + 1] = y; xyzs[3 * k + 2] = z;
                    dirs[3 * k] = q.dx; dirs[3 * k + 1] = q.dy; dirs[3 * k + 2] = q.dz;
                    ts[k] = t; deltas[k] = dt;
                    t += dt; s++;
                } else t = march_skip(&q, t, x, y, z, nx, ny, nz, mb, exp_step_factor, max_samples, grid_size, scale);
            }
        }
    }
    free(cnt); free(first_t); free(starts);
    if (counter) { counter[0] = (int32_t)total; counter[1] = (int32_t)n_rays; }
    return total;
}

/* R:335-404 (kernel) + R:407-454 (host). hits_t is (n_total_rays,2) and its [r][0] is advanced in place (R:390).
 * Outputs are (n_alive, N_samples[,3]) and must be zero-initialised by the caller (R:421-426). */
void oracle_raymarching_test(const float* rays_o, const float* rays_d, float* hits_t, const int64_t* alive, int64_t n_alive,
                             const uint8_t* bitfield, int cascades, float scale, float exp_step_factor, int grid_size,
                             int max_samples, int N_samples, float* xyzs, float* dirs, float* deltas, float* ts,
                             int32_t* n_eff) {
    for (int64_t n = 0; n < n_alive; n++) {
        const int64_t r = alive[n];
        const ray_t q = load_ray(rays_o, rays_d, r);
        float t = hits_t[2 * r]; const float t2 = hits_t[2 * r + 1];
        int s = 0;
        while (t < t2 && s < N_samples) {
            float x, y, z, mb; int nx, ny, nz;
            const float dt = calc_dt(t, exp_step_factor, max_samples, grid_size, (float)cascades); /* quirk R:370 */
            if (march_probe(&q, t, dt, bitfield, cascades, grid_size, scale, &x, &y, &z, &nx, &ny, &nz, &mb)) {
                const int64_t k = n * N_samples + s;
                xyzs[3 * k] = x; xyzs[3 * k + 1] = y; xyzs[3 * k + 2] = z;
                dirs[3 * k] = q.dx; dirs[3 * k + 1] = q.dy; dirs[3 * k + 2] = q.dz;
                ts[k] = t; deltas[k] = dt;
                t += dt; hits_t[2 * r] = t; s++;
            } else t = march_skip(&q, t, x, y, z, nx, ny, nz, mb, exp_step_factor, max_samples, grid_size, (float)cascades);
        }
        n_eff[n] = s;
    }
}

/* ---------------------------------------------------------------- intersection.cu */
static inline void ray_aabb(const float* o, const float* inv_d, const float* c, const float* h, float* t1o, float* t2o) { /* I:5-22 */
    float t1 = -INFINITY, t2 = INFINITY;
    for (int k = 0; k < 3; k++) {
        const float tmin = (c[k] - h[k] - o[k]) * inv_d[k];
        const float tmax = (c[k] + h[k] - o[k]) * inv_d[k];
        t1 = fmaxf(t1, fminf(tmin, tmax));
        t2 = fminf(t2, fmaxf(tmin, tmax));
    }
    if (t1 > t2) { *t1o = -1.0f; *t2o = -1.0f; } else { *t1o = t1; *t2o = t2; }
}
/* sort one ray's hit list ascending by t1, exactly as the host code I:94-97 (torch::sort on hits_t[...,0]; unused
 * slots hold -1 and therefore sort to the front). Insertion sort = stable; ties only occur between identical (-1,-1). */
static void sort_hits(float* ht, int64_t* hv, int max_hits) {
    for (int i = 1; i < max_hits; i++) {
        const float a = ht[2 * i], b = ht[2 * i + 1]; const int64_t v = hv[i];
        int j = i - 1;
        while (j >= 0 && ht[2 * j] > a) { ht[2 * j + 2] = ht[2 * j]; ht[2 * j + 3] = ht[2 * j + 1]; hv[j + 1] = hv[j]; j--; }
        ht[2 * j + 2] = a; ht[2 * j + 3] = b; hv[j + 1] = v;
    }
}
/* I:25-56 + I:59-100. The reference keeps the first max_hits hits in atomic-arrival order; here: voxel-index order. */
void oracle_ray_aabb_intersect(const float* rays_o, const float* rays_d, const float* centers, const float* half_sizes,
                               int64_t n_rays, int64_t n_voxels, int max_hits, int32_t* hit_cnt, float* hits_t,
                               int64_t* hits_voxel_idx) {
    for (int64_t r = 0; r < n_rays; r++) {
        float* ht = hits_t + r * max_hits * 2; int64_t* hv = hits_voxel_idx + r * max_hits;
        for (int i = 0; i < max_hits; i++) { ht[2 * i] = ht[2 * i + 1] = -1.0f; hv[i] = -1; }
        const float inv_d[3] = {1.0f / rays_d[3 * r], 1.0f / rays_d[3 * r + 1], 1.0f / rays_d[3 * r + 2]};
        int cnt = 0;
        for (int64_t v = 0; v < n_voxels; v++) {
            float t1, t2;
            ray_aabb(rays_o + 3 * r, inv_d, centers + 3 * v, half_sizes + 3 * v, &t1, &t2);
            if (t2 > 0) {
                if (cnt < max_hits) { ht[2 * cnt] = fmaxf(t1, 0.0f); ht[2 * cnt + 1] = t2; hv[cnt] = v; }
                cnt++;
            }
        }
        hit_cnt[r] = cnt;
        sort_hits(ht, hv, max_hits);
    }
}
/* I:103-121 + I:124-153 + I:156-196 */
void oracle_ray_sphere_intersect(const float* rays_o, const float* rays_d, const float* centers, const float* radii,
                                 int64_t n_rays, int64_t n_spheres, int max_hits, int32_t* hit_cnt, float* hits_t,
                                 int64_t* hits_sphere_idx) {
    for (int64_t r = 0; r < n_rays; r++) {
        float* ht = hits_t + r * max_hits * 2; int64_t* hv = hits_sphere_idx + r * max_hits;
        for (int i = 0; i < max_hits; i++) { ht[2 * i] = ht[2 * i + 1] = -1.0f; hv[i] = -1; }
        const float* o = rays_o + 3 * r; const float* d = rays_d + 3 * r;
        int cnt = 0;
        for (int64_t s = 0; s < n_spheres; s++) {
            const float* c = centers + 3 * s;
            const float cx = o[0] - c[0], cy = o[1] - c[1], cz = o[2] - c[2];
            const float a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];          /* helper_math dot(): x*x + y*y + z*z */
            const float half_b = d[0] * cx + d[1] * cy + d[2] * cz;
            const float cc = (cx * cx + cy * cy + cz * cz) - radii[s] * radii[s];
            const float disc = half_b * half_b - a * cc;
            float t1 = -1.0f, t2 = -1.0f;
            if (!(disc < 0)) { const float sq = sqrtf(disc); t1 = (-half_b - sq) / a; t2 = (-half_b + sq) / a; }
            if (t2 > 0) {
                if (cnt < max_hits) { ht[2 * cnt] = fmaxf(t1, 0.0f); ht[2 * cnt + 1] = t2; hv[cnt] = s; }
                cnt++;
            }
        }
        hit_cnt[r] = cnt;
        sort_hits(ht, hv, max_hits);
    }
}

/* ---------------------------------------------------------------- volumerendering.cu */
/* V:6-45 + V:48-84. Outputs zero-initialised here like the host code V:58-62. */
void oracle_composite_train_fw(const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                               const int64_t* rays_a, int64_t n_rays, int64_t n_total, float T_threshold,
                               int64_t* total_samples, float* opacity, float* depth, float* rgb, float* ws) {
    memset(total_samples, 0, sizeof(int64_t) * n_rays); memset(opacity, 0, sizeof(float) * n_rays);
    memset(depth, 0, sizeof(float) * n_rays); memset(rgb, 0, sizeof(float) * 3 * n_rays); memset(ws, 0, sizeof(float) * n_total);
    /* one ray = one independent serial chain (V:6-45 is a thread per ray): parallel over rays, every ray writes its own outputs only */
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t n = 0; n < n_rays; n++) {
        const int ray_idx = (int)rays_a[3 * n], start = (int)rays_a[3 * n + 1], N = (int)rays_a[3 * n + 2];
        int samples = 0; float T = 1.0f;
        while (samples < N) {
            const int s = start + samples;
            const float a = 1.0f - expf(-sigmas[s] * deltas[s]);
            const float w = a * T;
            rgb[3 * ray_idx] += w * rgbs[3 * s]; rgb[3 * ray_idx + 1] += w * rgbs[3 * s + 1]; rgb[3 * ray_idx + 2] += w * rgbs[3 * s + 2];
            depth[ray_idx] += w * ts[s];
            opacity[ray_idx] += w;
            ws[s] = w;
            T *= 1.0f - a;
            if (T <= T_threshold) break; /* V:41 : the saturating sample is not counted */
            samples++;
        }
        total_samples[ray_idx] = samples;
    }
}
/* V:87-151 + V:154-202 */
void oracle_composite_train_bw(const float* dL_dopacity, const float* dL_ddepth, const float* dL_drgb, const float* dL_dws,
                               const float* sigmas, const float* rgbs, const float* ws, const float* deltas, const float* ts,
                               const int64_t* rays_a, const float* opacity, const float* depth, const float* rgb,
                               int64_t n_rays, int64_t n_total, float T_threshold, float* dL_dsigmas, float* dL_drgbs) {
    memset(dL_dsigmas, 0, sizeof(float) * n_total); memset(dL_drgbs, 0, sizeof(float) * 3 * n_total);
    float* scan = (float*)malloc(sizeof(float) * (n_total > 0 ? n_total : 1));
    for (int64_t i = 0; i < n_total; i++) scan[i] = dL_dws[i] * ws[i]; /* V:178 */
    for (int64_t n = 0; n < n_rays; n++) {
        const int ray_idx = (int)rays_a[3 * n], start = (int)rays_a[3 * n + 1], N = (int)rays_a[3 * n + 2];
        if (N <= 0) continue; /* the reference reads scan[start-1] here (V:123); nothing is written for an empty ray */
        const float R = rgb[3 * ray_idx], G = rgb[3 * ray_idx + 1], B = rgb[3 * ray_idx + 2];
        const float O = opacity[ray_idx], D = depth[ray_idx];
        float T = 1.0f, r = 0.0f, g = 0.0f, b = 0.0f, d = 0.0f;
        for (int i = 1; i < N; i++) scan[start + i] += scan[start + i - 1]; /* V:119-122 sequential inclusive scan */
        const float sum = scan[start + N - 1];
        int samples = 0;
        while (samples < N) {
            const int s = start + samples;
            const float a = 1.0f - expf(-sigmas[s] * deltas[s]);
            const float w = a * T;
            r += w * rgbs[3 * s]; g += w * rgbs[3 * s + 1]; b += w * rgbs[3 * s + 2];
            d += w * ts[s];
            T *= 1.0f - a;
            dL_drgbs[3 * s] = dL_drgb[3 * ray_idx] * w;
            dL_drgbs[3 * s + 1] = dL_drgb[3 * ray_idx + 1] * w;
            dL_drgbs[3 * s + 2] = dL_drgb[3 * ray_idx + 2] * w;
            dL_dsigmas[s] = deltas[s] * (
                dL_drgb[3 * ray_idx] * (rgbs[3 * s] * T - (R - r)) +
                dL_drgb[3 * ray_idx + 1] * (rgbs[3 * s + 1] * T - (G - g)) +
                dL_drgb[3 * ray_idx + 2] * (rgbs[3 * s + 2] * T - (B - b)) +
                dL_dopacity[ray_idx] * (1 - O) +
                dL_ddepth[ray_idx] * (ts[s] * T - (D - d)) +
                T * dL_dws[s] - (sum - scan[s]));
            if (T <= T_threshold) break;
            samples++;
        }
    }
    free(scan);
}
/* V:205-249 + V:252-285. opacity/depth/rgb are accumulated in place, alive entries set to -1 when a ray dies. */
void oracle_composite_test_fw(const float* sigmas, const float* rgbs, const float* deltas, const float* ts,
                              int64_t* alive, int64_t n_alive, int N_samples, float T_threshold, const int32_t* n_eff,
                              float* opacity, float* depth, float* rgb) {
    for (int64_t n = 0; n < n_alive; n++) {
        if (n_eff[n] == 0) { alive[n] = -1; continue; }
        const int64_t r = alive[n];
        int s = 0; float T = 1 - opacity[r];
        while (s < n_eff[n]) {
            const int64_t k = n * N_samples + s;
            const float a = 1.0f - expf(-sigmas[k] * deltas[k]);
            const float w = a * T;
            rgb[3 * r] += w * rgbs[3 * k]; rgb[3 * r + 1] += w * rgbs[3 * k + 1]; rgb[3 * r + 2] += w * rgbs[3 * k + 2];
            depth[r] += w * ts[k];
            opacity[r] += w;
            T *= 1.0f - a;
            if (T <= T_threshold) { alive[n] = -1; break; }
            s++;
        }
    }
}

/* ---------------------------------------------------------------- losses.cu */
/* L:9-43 + L:46-61 + L:64-109 */
void oracle_distortion_loss_fw(const float* ws, const float* deltas, const float* ts, const int64_t* rays_a, int64_t n_rays,
                               int64_t n_total, float* loss, float* ws_incl, float* wts_incl) {
    memset(loss, 0, sizeof(float) * n_rays); memset(ws_incl, 0, sizeof(float) * n_total); memset(wts_incl, 0, sizeof(float) * n_total);
    for (int64_t n = 0; n < n_rays; n++) {
        const int ray_idx = (int)rays_a[3 * n], start = (int)rays_a[3 * n + 1], N = (int)rays_a[3 * n + 2];
        float wi = 0.0f, wti = 0.0f, acc = 0.0f;
        for (int i = 0; i < N; i++) {
            const int s = start + i;
            const float we = wi, wte = wti;          /* exclusive scans */
            const float wt = ws[s] * ts[s];          /* L:75 */
            wi = wi + ws[s]; wti = wti + wt;         /* inclusive scans, sequential order like in-thread thrust */
            ws_incl[s] = wi; wts_incl[s] = wti;
            const float l = 2 * (wti * we - wi * wte) + 1.0f / 3 * ws[s] * ws[s] * deltas[s]; /* L:94-95 */
            acc = acc + l;                            /* L:57 thrust::reduce, init 0 */
        }
        loss[ray_idx] = acc;
    }
}
/* L:112-142 + L:145-174 */
void oracle_distortion_loss_bw(const float* dL_dloss, const float* ws_incl, const float* wts_incl, const float* ws,
                               const float* deltas, const float* ts, const int64_t* rays_a, int64_t n_rays, int64_t n_total,
                               float* dL_dws) {
    memset(dL_dws, 0, sizeof(float) * n_total);
    for (int64_t n = 0; n < n_rays; n++) {
        const int ray_idx = (int)rays_a[3 * n], start = (int)rays_a[3 * n + 1], N = (int)rays_a[3 * n + 2];
        if (N <= 0) continue;
        const int end = start + N - 1;
        const float ws_sum = ws_incl[end], wts_sum = wts_incl[end];
        for (int s = start; s <= end; s++) {
            float v = dL_dloss[ray_idx] * 2 * (
                (s == start ? 0.0f : (ts[s] * ws_incl[s - 1] - wts_incl[s - 1])) +
                (wts_sum - wts_incl[s] - ts[s] * (ws_sum - ws_incl[s])));
            v += dL_dloss[ray_idx] * 2.0f / 3 * ws[s] * deltas[s];
            dL_dws[s] = v;
        }
    }
}

/* ---------------------------------------------------------------- morton_encoding.cu */
static inline uint64_t split_by_3(uint32_t a) { /* M:15-23 */
    uint64_t x = a & 0x1fffff;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
static inline float saturatef(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); } /* __saturatef: NaN -> 0 */
/* M:25-46 (kernel) + M:48-74 (host: aminmax over dim 0, cube = max extent) */
void oracle_morton_encode(const float* positions, int64_t n, int64_t* out) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], positions[3 * i + k]); mx[k] = fmaxf(mx[k], positions[3 * i + k]); }
    const float cube = fmaxf(mx[0] - mn[0], fmaxf(mx[1] - mn[1], mx[2] - mn[2]));
    const float rcp = 1.0f / cube;
    const float factor = 2097151.0f;
    for (int64_t i = 0; i < n; i++) {
        const uint32_t x = (uint32_t)(saturatef((positions[3 * i] - mn[0]) * rcp) * factor);
        const uint32_t y = (uint32_t)(saturatef((positions[3 * i + 1] - mn[1]) * rcp) * factor);
        const uint32_t z = (uint32_t)(saturatef((positions[3 * i + 2] - mn[2]) * rcp) * factor);
        out[i] = (int64_t)(split_by_3(x) | split_by_3(y) << 1 | split_by_3(z) << 2);
    }
}
