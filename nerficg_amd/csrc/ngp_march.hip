// ngp_march.hip -- occupancy-grid utilities, ray/box and ray/sphere intersection, DDA ray marching (train + test),
// Morton codes and ray generation for gfx950.
//
// BUILD NOTE: this translation unit is compiled with -ffp-contract=off.  Everything here decides *indices* (which
// cell, how many samples, where a ray starts) from f32 arithmetic, and the parity contract for indices is bit-exact
// against oracle/ngp_oracle.c, which is built without FMA contraction as well.
//
// Reference semantics (never copied, restated for wave64 / deterministic compaction):
//   raymarching.cu:11-60 (helpers) :62-161 (morton, packbits) :166-332 (train) :335-454 (test)
//   intersection.cu:5-196, morton_encoding.cu:15-74, Cameras/Perspective.py:64-94, Datasets/utils.py:1033-1074
#include "common.h"
#include <hip/hip_fp16.h>

#define SQRT3 1.73205080757f
#define NRC_WAVE_MARCH_MAX_RAYS 32768  // up to here one wave per ray (latency-bound batches), above one thread per ray

namespace {

__device__ __forceinline__ float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }
__device__ __forceinline__ float signf_(float x) { return copysignf(1.0f, x); }

__device__ __forceinline__ float calc_dt(float t, float esf, int max_samples, int grid_size, float scale) {
    return clampf(t * esf, SQRT3 / max_samples, SQRT3 * 2 * scale / grid_size);
}
__device__ __forceinline__ int mip_from_pos(float x, float y, float z, int cascades) {
    const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
    int e; frexpf(mx, &e);
    return min(cascades - 1, max(0, e + 1));
}
__device__ __forceinline__ int mip_from_dt(float dt, int grid_size, int cascades) {
    int e; frexpf(dt * grid_size, &e);
    return min(cascades - 1, max(0, e));
}
__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3D(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}
__device__ __forceinline__ uint32_t morton3D_invert(uint32_t x) {
    x = x & 0x49249249u;
    x = (x | (x >> 2)) & 0xc30c30c3u;
    x = (x | (x >> 4)) & 0x0f00f00fu;
    x = (x | (x >> 8)) & 0xff0000ffu;
    x = (x | (x >> 16)) & 0x0000ffffu;
    return x;
}

struct Ray {
    float ox, oy, oz, dx, dy, dz, dxi, dyi, dzi;
};
__device__ __forceinline__ Ray load_ray(const float* __restrict__ o, const float* __restrict__ d, int64_t r) {
    Ray q;
    q.ox = o[3 * r]; q.oy = o[3 * r + 1]; q.oz = o[3 * r + 2];
    q.dx = d[3 * r]; q.dy = d[3 * r + 1]; q.dz = d[3 * r + 2];
    q.dxi = 1.0f / q.dx; q.dyi = 1.0f / q.dy; q.dzi = 1.0f / q.dz;
    return q;
}

struct MarchCfg {
    const uint8_t* __restrict__ bitfield;
    int cascades, grid_size, max_samples;
    float scale, esf, dt_scale;  // dt_scale: `scale` (train) or `(float)cascades` (test kernel quirk, raymarching.cu:370)
    uint32_t grid_size3;
    float grid_size_inv;
    float mip0_bound, mip0_bound_inv;  // cascades == 1: the mip level is always 0, its bound and reciprocal are constants (same f32 values)
    int use_lut;                       // the kernel staged expand_bits(0 .. grid_size-1) in LDS (march_lut) and passes the array to cell_probe
};

// The cell containing o + t d: sample position, step, occupancy bit and (for an empty cell) the t beyond which the march resumes.
// Same f32 operation sequence as raymarching.cu:200-234 / 243-279 (this file is compiled with -ffp-contract=off).
// `lut`: the kernel's __shared__ table, passed as a plain argument (not through the struct) so that after inlining the compiler knows it is LDS
// and reads it with ds_read_b32 -- behind a generic pointer in MarchCfg the lookups were flat_load_dword through the vector memory path.
struct CellAt { int nx, ny, nz; float mip_bound; };
// sample position, step and occupancy bit of the cell containing o + t d (no exit distance: a sample does not need it)
__device__ __forceinline__ bool cell_locate(const Ray& q, const MarchCfg& c, const uint32_t* lut, float t, float& x, float& y, float& z, float& dt, CellAt& at) {
    x = q.ox + t * q.dx; y = q.oy + t * q.dy; z = q.oz + t * q.dz;
    dt = calc_dt(t, c.esf, c.max_samples, c.grid_size, c.dt_scale);
    int mip = 0;
    float mip_bound = c.mip0_bound, mip_bound_inv = c.mip0_bound_inv;
    if (c.cascades > 1) {  // wave-uniform; with one cascade min(cascades - 1, .) pins the level to 0 and the frexp / scalbn / divide drop out
        mip = max(mip_from_pos(x, y, z, c.cascades), mip_from_dt(dt, c.grid_size, c.cascades));
        mip_bound = fminf(scalbnf(1.0f, mip - 1), c.scale);
        mip_bound_inv = 1 / mip_bound;
    }
    const int nx = (int)clampf(0.5f * (x * mip_bound_inv + 1) * c.grid_size, 0.0f, c.grid_size - 1.0f);
    const int ny = (int)clampf(0.5f * (y * mip_bound_inv + 1) * c.grid_size, 0.0f, c.grid_size - 1.0f);
    const int nz = (int)clampf(0.5f * (z * mip_bound_inv + 1) * c.grid_size, 0.0f, c.grid_size - 1.0f);
    // Morton index: three LDS lookups instead of 27 VALU instructions (a third of the step) when the kernel staged the table
    const uint32_t mort = c.use_lut ? (lut[nx] | (lut[ny] << 1) | (lut[nz] << 2)) : morton3D(nx, ny, nz);
    const uint32_t idx = (uint32_t)mip * c.grid_size3 + mort;
    at.nx = nx; at.ny = ny; at.nz = nz; at.mip_bound = mip_bound;
    return c.bitfield[idx >> 3] & (1 << (idx & 7));
}
// the t beyond which the march resumes after an EMPTY cell (raymarching.cu:221-230)
__device__ __forceinline__ float cell_exit(const Ray& q, const MarchCfg& c, const CellAt& at, float t, float x, float y, float z) {
    const float tx = (((at.nx + 0.5f + 0.5f * signf_(q.dx)) * c.grid_size_inv * 2 - 1) * at.mip_bound - x) * q.dxi;
    const float ty = (((at.ny + 0.5f + 0.5f * signf_(q.dy)) * c.grid_size_inv * 2 - 1) * at.mip_bound - y) * q.dyi;
    const float tz = (((at.nz + 0.5f + 0.5f * signf_(q.dz)) * c.grid_size_inv * 2 - 1) * at.mip_bound - z) * q.dzi;
    return t + fmaxf(0.0f, fminf(tx, fminf(ty, tz)));
}
__device__ __forceinline__ bool cell_probe(const Ray& q, const MarchCfg& c, const uint32_t* lut, float t, float& x, float& y, float& z, float& dt, float& t_target) {
    CellAt at;
    const bool occ = cell_locate(q, c, lut, t, x, y, z, dt, at);
    t_target = cell_exit(q, c, at, t, x, y, z);
    return occ;
}
// stages expand_bits(i), i < grid_size <= MARCH_LUT_MAX, in LDS and points the configuration at it (all threads of the block call this)
#define MARCH_LUT_MAX 1024
__device__ __forceinline__ void march_lut(MarchCfg& c, uint32_t* s_lut) {
    if (c.grid_size <= MARCH_LUT_MAX) {
        for (int i = threadIdx.x; i < c.grid_size; i += blockDim.x) s_lut[i] = expand_bits((uint32_t)i);
        c.use_lut = 1;
    }
    __syncthreads();
}
// One DDA step.  Returns true when the cell containing o + t d is occupied (sample taken at t with step dt);
// otherwise advances t to beyond the cell's exit face.  x,y,z,dt are outputs for the occupied case.
__device__ __forceinline__ bool march_step(const Ray& q, const MarchCfg& c, const uint32_t* lut, float& t, float& x, float& y, float& z, float& dt) {
    // the exit distance of the cell (a dozen instructions) is worked out only when the cell is empty: the per-thread marches are VALU-bound
    CellAt at;
    if (cell_locate(q, c, lut, t, x, y, z, dt, at)) return true;
    const float t_target = cell_exit(q, c, at, t, x, y, z);
    do { t += calc_dt(t, c.esf, c.max_samples, c.grid_size, c.dt_scale); } while (t < t_target);
    return false;
}

// ------------------------------------------------------------------------------------------------ small utilities
__global__ void k_morton3D(const int32_t* __restrict__ coords, int64_t n, int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = (int32_t)morton3D(coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]);
}
__global__ void k_morton3D_invert(const int32_t* __restrict__ idx, int64_t n, int32_t* __restrict__ coords) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t ind = idx[i];  // arithmetic shifts on the signed value, like the reference (raymarching.cu:97-100)
    coords[3 * i + 0] = (int32_t)morton3D_invert((uint32_t)(ind >> 0));
    coords[3 * i + 1] = (int32_t)morton3D_invert((uint32_t)(ind >> 1));
    coords[3 * i + 2] = (int32_t)morton3D_invert((uint32_t)(ind >> 2));
}

// packbits: one lane owns 8 consecutive cells = two 16-byte loads (f32) / one 16-byte load (f16), one byte out.
// Four lanes are merged with DPP so that a lane quartet issues a single 4-byte store.
template <typename T>
__global__ void k_packbits_unaligned(const T* __restrict__ grid, int64_t n_bytes, float thr, uint8_t* __restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_bytes) return;
    uint32_t bits = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) bits |= (uint32_t)((float)grid[8 * n + i] > thr) << i;
    out[n] = (uint8_t)bits;
}
template <typename T>
__global__ void k_packbits(const T* __restrict__ grid, int64_t n_bytes, float thr, uint8_t* __restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t bits = 0;
    if (n < n_bytes) {
        if constexpr (sizeof(T) == 4) {
            const float4 a = reinterpret_cast<const float4*>(grid)[2 * n];
            const float4 b = reinterpret_cast<const float4*>(grid)[2 * n + 1];
            bits = (a.x > thr) | ((a.y > thr) << 1) | ((a.z > thr) << 2) | ((a.w > thr) << 3) | ((b.x > thr) << 4) |
                   ((b.y > thr) << 5) | ((b.z > thr) << 6) | ((b.w > thr) << 7);
        } else {
            const uint4 raw = reinterpret_cast<const uint4*>(grid)[n];
            const __half* h = reinterpret_cast<const __half*>(&raw);
#pragma unroll
            for (int i = 0; i < 8; i++) bits |= (uint32_t)(__half2float(h[i]) > thr) << i;
        }
    }
    // gather the 4 bytes of a lane quartet into its first lane
    const uint32_t b1 = __shfl_down(bits, 1, 4), b2 = __shfl_down(bits, 2, 4), b3 = __shfl_down(bits, 3, 4);
    if ((threadIdx.x & 3) == 0 && n < n_bytes) {
        if (n + 3 < n_bytes) reinterpret_cast<uint32_t*>(out)[n >> 2] = bits | (b1 << 8) | (b2 << 16) | (b3 << 24);
        else {
            out[n] = (uint8_t)bits;
            if (n + 1 < n_bytes) out[n + 1] = (uint8_t)b1;
            if (n + 2 < n_bytes) out[n + 2] = (uint8_t)b2;
        }
    }
}

// ------------------------------------------------------------------------------------------------ intersections
__device__ __forceinline__ void sort_hits(float* ht, int64_t* hv, int max_hits) {
    // ascending by t1 (unused -1 slots first), the order torch::sort gives the reference (intersection.cu:94-97)
    for (int i = 1; i < max_hits; i++) {
        const float a = ht[2 * i], b = ht[2 * i + 1];
        const int64_t v = hv[i];
        int j = i - 1;
        while (j >= 0 && ht[2 * j] > a) {
            ht[2 * j + 2] = ht[2 * j]; ht[2 * j + 3] = ht[2 * j + 1]; hv[j + 1] = hv[j];
            j--;
        }
        ht[2 * j + 2] = a; ht[2 * j + 3] = b; hv[j + 1] = v;
    }
}

// one lane per ray, loop over the (few) primitives: deterministic slot order (voxel order), no atomics.
template <bool SPHERE>
__global__ void k_ray_prim_intersect(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                     const float* __restrict__ centers, const float* __restrict__ ext, int64_t n_rays,
                                     int64_t n_prims, int max_hits, int32_t* __restrict__ hit_cnt,
                                     float* __restrict__ hits_t, int64_t* __restrict__ hits_idx) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const float ox = rays_o[3 * r], oy = rays_o[3 * r + 1], oz = rays_o[3 * r + 2];
    const float dx = rays_d[3 * r], dy = rays_d[3 * r + 1], dz = rays_d[3 * r + 2];
    float* ht = hits_t + r * max_hits * 2;
    int64_t* hv = hits_idx + r * max_hits;
    for (int i = 0; i < max_hits; i++) { ht[2 * i] = -1.0f; ht[2 * i + 1] = -1.0f; hv[i] = -1; }
    int cnt = 0;
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    for (int64_t v = 0; v < n_prims; v++) {
        const float cx = centers[3 * v], cy = centers[3 * v + 1], cz = centers[3 * v + 2];
        float t1, t2;
        if constexpr (!SPHERE) {
            const float hx = ext[3 * v], hy = ext[3 * v + 1], hz = ext[3 * v + 2];
            const float ax = (cx - hx - ox) * ix, bx = (cx + hx - ox) * ix;
            const float ay = (cy - hy - oy) * iy, by = (cy + hy - oy) * iy;
            const float az = (cz - hz - oz) * iz, bz = (cz + hz - oz) * iz;
            t1 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
            t2 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
            if (t1 > t2) { t1 = -1.0f; t2 = -1.0f; }
        } else {
            const float px = ox - cx, py = oy - cy, pz = oz - cz;
            const float a = dx * dx + dy * dy + dz * dz;
            const float half_b = dx * px + dy * py + dz * pz;
            const float c = (px * px + py * py + pz * pz) - ext[v] * ext[v];
            const float disc = half_b * half_b - a * c;
            t1 = -1.0f; t2 = -1.0f;
            if (!(disc < 0)) { const float sq = sqrtf(disc); t1 = (-half_b - sq) / a; t2 = (-half_b + sq) / a; }
        }
        if (t2 > 0) {
            if (cnt < max_hits) { ht[2 * cnt] = fmaxf(t1, 0.0f); ht[2 * cnt + 1] = t2; hv[cnt] = v; }
            cnt++;
        }
    }
    hit_cnt[r] = cnt;
    if (max_hits > 1) sort_hits(ht, hv, max_hits);
}

// Box-centred origins and the [t_in, t_out] span of every ray in ONE launch: what InstantNGPRenderer.render_rays does with a subtraction, the
// slab test against the scene box (the expressions of k_ray_prim_intersect with the box at the origin) and two clamps (Renderer.py:55-77).
__global__ void __launch_bounds__(256) k_clip_rays(int64_t n, const float* __restrict__ origin, const float* __restrict__ dirs, float cx, float cy, float cz,
                                                   float hx, float hy, float hz, float near_plane, float far_plane, float* __restrict__ o_out,
                                                   float* __restrict__ span) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const float ox = origin[3 * r] - cx, oy = origin[3 * r + 1] - cy, oz = origin[3 * r + 2] - cz;
    o_out[3 * r] = ox; o_out[3 * r + 1] = oy; o_out[3 * r + 2] = oz;
    const float ix = 1.0f / dirs[3 * r], iy = 1.0f / dirs[3 * r + 1], iz = 1.0f / dirs[3 * r + 2];
    const float ax = (0.0f - hx - ox) * ix, bx = (0.0f + hx - ox) * ix;
    const float ay = (0.0f - hy - oy) * iy, by = (0.0f + hy - oy) * iy;
    const float az = (0.0f - hz - oz) * iz, bz = (0.0f + hz - oz) * iz;
    float t1 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    float t2 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    if (t1 > t2) { t1 = -1.0f; t2 = -1.0f; }
    float s0 = -1.0f, s1 = -1.0f;
    if (t2 > 0) { s0 = fmaxf(t1, 0.0f); s1 = t2; }
    span[2 * r] = fmaxf(s0, near_plane);      // a miss stays (near, -1): an empty interval
    span[2 * r + 1] = fminf(s1, far_plane);
}

// ------------------------------------------------------------------------------------------------ ray marching (train)
// Stage A: per-ray sample count (pass 1 of raymarching.cu:200-234) + per-block sums.
__global__ void __launch_bounds__(256) k_march_count(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                     const float* __restrict__ hits_t, const float* __restrict__ noise,
                                                     MarchCfg c, int64_t n_rays, int32_t* __restrict__ counts,
                                                     int32_t* __restrict__ block_sums) {
    __shared__ int smem[8];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int n = 0;
    if (r < n_rays) {
        const Ray q = load_ray(rays_o, rays_d, r);
        float t1 = hits_t[2 * r];
        const float t2 = hits_t[2 * r + 1];
        if (t1 >= 0) t1 += calc_dt(t1, c.esf, c.max_samples, c.grid_size, c.scale) * noise[r];
        float t = t1, x, y, z, dt;
        while (0 <= t && t < t2 && n < c.max_samples) {
            if (march_step(q, c, (const uint32_t*)nullptr, t, x, y, z, dt)) { t += dt; n++; }
        }
        counts[r] = n;
    }
    int total;
    (void)nrc_block256_excl_scan_i(n, smem, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}
// Stage B: exclusive scan of the block sums by ONE 1024-thread block (n_blocks <= a few thousand) -> block offsets, total.
__global__ void __launch_bounds__(1024) k_scan_block_sums(int32_t* __restrict__ block_sums, int64_t n_blocks,
                                                          int64_t n_rays, int32_t* __restrict__ counter) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_blocks; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int v = i < n_blocks ? block_sums[i] : 0;
        const int incl = nrc_wave_incl_sum_i(v, lane);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int off = carry_s;
        for (int w = 0; w < wave; w++) off += wave_tot[w];
        if (i < n_blocks) block_sums[i] = off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) { counter[0] = carry_s; counter[1] = (int32_t)n_rays; }
}
// Stage C: rays_a[r] = (r, block_offset + in-block exclusive scan, count)
__global__ void __launch_bounds__(256) k_assign_rays_a(const int32_t* __restrict__ counts, const int32_t* __restrict__ block_offs,
                                                       int64_t n_rays, int64_t* __restrict__ rays_a) {
    __shared__ int smem[8];
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int n = r < n_rays ? counts[r] : 0;
    const int excl = nrc_block256_excl_scan_i(n, smem, nullptr);
    if (r < n_rays) {
        rays_a[3 * r] = r;
        rays_a[3 * r + 1] = (int64_t)block_offs[blockIdx.x] + excl;
        rays_a[3 * r + 2] = n;
    }
}
// Stages B + C + the capacity cut for small batches (one workgroup over all per-ray counts): rays_a[r] = (r, start, n) with segments cut at `cap`
// rows, counter = (uncut total, n_rays), *overflow = max(total - cap, 0).  Replaces the block-sum clear, k_scan_block_sums, k_assign_rays_a and
// k_march_cap of a recorded training iteration (a few thousand rays) by one launch.
__global__ void __launch_bounds__(1024) k_scan_assign_cap(const int32_t* __restrict__ counts, int64_t n_rays, int64_t cap, int64_t* __restrict__ rays_a,
                                                          int32_t* __restrict__ counter, int64_t* __restrict__ overflow, int64_t* __restrict__ mailbox = nullptr,
                                                          int64_t mailbox_ticket = 0) {
    __shared__ int wave_tot[16];
    __shared__ int64_t carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_rays; base += 1024) {
        const int64_t r = base + threadIdx.x;
        const int v = r < n_rays ? counts[r] : 0;
        const int incl = nrc_wave_incl_sum_i(v, lane);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int64_t off = carry_s;
        for (int w = 0; w < wave; w++) off += wave_tot[w];
        if (r < n_rays) {
            const int64_t start = off + incl - v;
            rays_a[3 * r] = r;
            rays_a[3 * r + 1] = start + v > cap ? min(start, cap) : start;
            rays_a[3 * r + 2] = start + v > cap ? max(cap - start, (int64_t)0) : (int64_t)v;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counter[0] = (int32_t)carry_s; counter[1] = (int32_t)n_rays;
        if (overflow) overflow[0] = max(carry_s - cap, (int64_t)0);
        if (mailbox) {   // the total straight into mapped host memory, the ticket LAST (nrc_host_mailbox_alloc): the host sizes the sample buffers without a copy or a stream wait
            __hip_atomic_store(mailbox, carry_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mailbox + 1, n_rays, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mailbox + 2, mailbox_ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// Stage D: pass 2 (raymarching.cu:243-279): emit the samples of each ray into its reserved, contiguous segment.
__global__ void __launch_bounds__(256) k_march_write(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                     const float* __restrict__ hits_t, const float* __restrict__ noise,
                                                     MarchCfg c, int64_t n_rays, const int64_t* __restrict__ rays_a,
                                                     float* __restrict__ xyzs, float* __restrict__ dirs,
                                                     float* __restrict__ deltas, float* __restrict__ ts) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= n_rays) return;
    const int64_t r = rays_a[3 * n], start = rays_a[3 * n + 1];
    const int N = (int)rays_a[3 * n + 2];
    if (N == 0) return;
    const Ray q = load_ray(rays_o, rays_d, r);
    float t1 = hits_t[2 * r];
    const float t2 = hits_t[2 * r + 1];
    if (t1 >= 0) t1 += calc_dt(t1, c.esf, c.max_samples, c.grid_size, c.scale) * noise[r];
    float t = t1, x, y, z, dt;
    int s = 0;
    while (t < t2 && s < N) {
        if (march_step(q, c, (const uint32_t*)nullptr, t, x, y, z, dt)) {
            const int64_t k = start + s;
            xyzs[3 * k] = x; xyzs[3 * k + 1] = y; xyzs[3 * k + 2] = z;
            dirs[3 * k] = q.dx; dirs[3 * k + 1] = q.dy; dirs[3 * k + 2] = q.dz;
            ts[k] = t; deltas[k] = dt;
            t += dt; s++;
        }
    }
}

// ---- wave-per-ray variants (small batches) ---------------------------------------------------------------------------
// A training batch is a few thousand rays: with one thread per ray the march is a chain of ~400 dependent occupancy-byte loads
// per ray on a machine that is 97 % idle (0.41 + 0.25 ms per iteration measured).  The sequence of candidate positions of a ray,
// t <- t + calc_dt(t), does not depend on the occupancy (a sample step and a skip step advance by the same rule), so a WAVE
// takes the next 64 candidates of one ray at once: the chain of 64 additions runs on the VALU, the 64 occupancy probes are
// independent loads, and the sequential semantics -- which candidates the reference loop actually visits -- are replayed on
// the ballots with a scalar loop (one turn per visited empty cell or run of samples).  Bit-identical to the per-thread loop.
// the march of ONE ray by one wave (all 64 lanes call this with the same ray): returns the number of samples, parks their t (count pass) or writes
// the sample rows (write pass)
template <bool WRITE>
__device__ __forceinline__ int wave_march_ray(const Ray& q, float t1, const float t2, const MarchCfg& c, const uint32_t* s_lut, const int N, const int64_t start,
                                              float* __restrict__ park_row, float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas,
                                              float* __restrict__ ts) {
    const int lane = threadIdx.x & 63;
    float t = t1;                                     // wave-uniform: first candidate of the next chunk
    float skip_until = -__builtin_inff();             // wave-uniform: candidates below it are jumped over (empty-cell skip)
    int s = 0;
    bool done = !(0 <= t);
    while (!done && t < t2 && s < N) {
        float my_t = t;
        if (c.esf == 0.f) {  // wave-uniform.  Fixed step: clamp(t * 0, lo, hi) = lo for every finite t, so the chain is one add per candidate instead of four
            const float dt0 = calc_dt(0.f, 0.f, c.max_samples, c.grid_size, c.dt_scale);
#pragma unroll 8
            for (int k = 0; k < 64; k++) {
                if (lane == k) my_t = t;
                t += dt0;
            }
        } else {
#pragma unroll 8
            for (int k = 0; k < 64; k++) {
                if (lane == k) my_t = t;
                t += calc_dt(t, c.esf, c.max_samples, c.grid_size, c.dt_scale);
            }
        }
        const bool valid = my_t < t2;
        float x, y, z, dt, t_target;
        const bool occ = cell_probe(q, c, s_lut, valid ? my_t : t1, x, y, z, dt, t_target) && valid;
        const unsigned long long validm = __ballot(valid), occm = __ballot(occ);
        unsigned long long samples = 0ull;
        const int s_before = s;
        int pos = 0;
        while (pos < 64 && s < N) {
            const unsigned long long reach = __ballot(my_t >= skip_until) & (~0ull << pos);
            if (reach == 0ull) break;                              // the pending skip passes the whole chunk
            const int j = __builtin_ctzll(reach);
            if (!((validm >> j) & 1ull)) { done = true; break; }   // t >= t2
            if ((occm >> j) & 1ull) {                              // a run of samples
                const unsigned long long stop = ~occm & (~0ull << j);
                const int e = stop ? __builtin_ctzll(stop) : 64;
                const int cnt = min(e - j, N - s);
                samples |= (cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull)) << j;
                s += cnt;
                pos = j + cnt;
            } else {                                               // empty cell: resume at the first candidate >= its exit
                skip_until = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t_target), j));
                pos = j + 1;                                       // (the reference's do-while advances at least once)
            }
        }
        if (!WRITE && park_row && ((samples >> lane) & 1ull))  // the count pass parks the sample positions: k_march_expand instead of a second march
            park_row[s_before + __popcll(samples & ((1ull << lane) - 1ull))] = my_t;
        if (WRITE && ((samples >> lane) & 1ull)) {
            const int64_t k = start + s_before + __popcll(samples & ((1ull << lane) - 1ull));
            xyzs[3 * k] = x; xyzs[3 * k + 1] = y; xyzs[3 * k + 2] = z;
            dirs[3 * k] = q.dx; dirs[3 * k + 1] = q.dy; dirs[3 * k + 2] = q.dz;
            ts[k] = my_t; deltas[k] = dt;
        }
    }
    return s;
}

template <bool WRITE>
__global__ void __launch_bounds__(256) k_march_wave(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                    const float* __restrict__ hits_t, const float* __restrict__ noise, MarchCfg c,
                                                    int64_t n_rays, int32_t* __restrict__ counts, int32_t* __restrict__ block_sums,
                                                    const int64_t* __restrict__ rays_a, float* __restrict__ xyzs,
                                                    float* __restrict__ dirs, float* __restrict__ deltas, float* __restrict__ ts,
                                                    float* __restrict__ park) {
    __shared__ uint32_t s_lut[MARCH_LUT_MAX];
    march_lut(c, s_lut);
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_rays) return;
    int64_t r = n, start = 0;
    int N = c.max_samples;
    if (WRITE) {
        r = rays_a[3 * n]; start = rays_a[3 * n + 1]; N = (int)rays_a[3 * n + 2];
        if (N == 0) return;
    }
    const Ray q = load_ray(rays_o, rays_d, r);
    float t1 = hits_t[2 * r];
    const float t2 = hits_t[2 * r + 1];
    if (t1 >= 0) t1 += calc_dt(t1, c.esf, c.max_samples, c.grid_size, c.scale) * noise[r];
    const int s = wave_march_ray<WRITE>(q, t1, t2, c, s_lut, N, start, park ? park + r * c.max_samples : nullptr, xyzs, dirs, deltas, ts);
    if (!WRITE && lane == 0) {
        counts[r] = s;
        if (s && block_sums) atomicAdd(&block_sums[r >> 8], s);
    }
}

// Second pass when the count pass parked the sample positions: a wave per ray expands t into (xyz, dir, dt, t) -- the same f32
// expressions as cell_probe / calc_dt, no march, no occupancy probes.
__global__ void __launch_bounds__(256) k_march_expand(const float* __restrict__ rays_o, const float* __restrict__ rays_d, MarchCfg c, int64_t n_rays,
                                                      const int64_t* __restrict__ rays_a, const float* __restrict__ park, float* __restrict__ xyzs,
                                                      float* __restrict__ dirs, float* __restrict__ deltas, float* __restrict__ ts,
                                                      const int32_t* __restrict__ counter = nullptr, int64_t cap = 0) {
    // (fixed capacity, nrc_raymarching_train_capped: the workgroups behind the rays' make the rows [counter[0], cap) inert samples)
    const int64_t ray_blocks = (n_rays + 3) / 4;
    if ((int64_t)blockIdx.x >= ray_blocks) {
        const int64_t i = ((int64_t)blockIdx.x - ray_blocks) * 256 + threadIdx.x;
        if (i < cap && i >= (int64_t)counter[0]) {
            xyzs[3 * i] = 0.f; xyzs[3 * i + 1] = 0.f; xyzs[3 * i + 2] = 0.f;
            dirs[3 * i] = 0.f; dirs[3 * i + 1] = 0.f; dirs[3 * i + 2] = 1.f;
            deltas[i] = 0.f; ts[i] = 0.f;
        }
        return;
    }
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= n_rays) return;
    const int64_t r = rays_a[3 * n], start = rays_a[3 * n + 1];
    const int N = (int)rays_a[3 * n + 2];
    if (N == 0) return;
    const Ray q = load_ray(rays_o, rays_d, r);
    const float* src = park + r * c.max_samples;
    for (int k = lane; k < N; k += 64) {
        const float t = src[k];
        const int64_t o = start + k;
        xyzs[3 * o] = q.ox + t * q.dx; xyzs[3 * o + 1] = q.oy + t * q.dy; xyzs[3 * o + 2] = q.oz + t * q.dz;
        dirs[3 * o] = q.dx; dirs[3 * o + 1] = q.dy; dirs[3 * o + 2] = q.dz;
        ts[o] = t;
        deltas[o] = calc_dt(t, c.esf, c.max_samples, c.grid_size, c.dt_scale);
    }
}

// ---- the batch of a fused training iteration (include/nerficg_hip.h group 13) -----------------------------------------------------------------
// What Trainer.py:84-87 + Renderer.py:55-77 + the count pass of the march do before the first network call, as ONE launch: wave n takes ray
// order[cursor + n] (or ids[n]) out of the resident pool, centres its origin, clips it against the scene box and the depth range, draws its march
// jitter (Philox4x32-10 keyed by the iteration, counter = the ray's index in the batch: the draw does not depend on how the batch is cut over ranks)
// and marches it (wave_march_ray: parked positions); the LAST workgroup to finish scans the per-ray counts into rays_a with the capacity cut,
// writes counter / overflow, and advances the cursor and the iteration counter of the generator -- k_gather_ray_batch, two uniform_ launches,
// k_clip_rays, k_march_wave and k_scan_assign_cap of the recorded iteration of round 4 (33 us of 4-5 us launches + 63 us).
struct TrainBatch {
    const int64_t* ids;          // (ray_capacity) or NULL: then order / cursor
    const int64_t* order;        // resident sampling order (RandomSequentialSampler's permutation on the device)
    int64_t* cursor;             // DEVICE i64[1]: first position of this batch; advanced by the batch size
    const int32_t* n_rays_dev;   // DEVICE i32[1] or NULL: live rays of the batch (<= ray_capacity; the rows behind are inert)
    int64_t n_pool, ray_offset;  // rays in the pool; index of row 0 in the GLOBAL batch (data parallel: the jitter is a function of the global index)
    const float *pool_o, *pool_d, *pool_rgb, *pool_alpha;
    float cx, cy, cz, hx, hy, hz, near_plane, far_plane;
    uint64_t* rng;               // DEVICE u64[2]: seed, iterations drawn so far (advanced by one)
    const float *bg_in, *noise_in;   // explicit background colour (3) / jitter (ray_capacity) instead of draws (tests, replays of a recorded batch)
    float *rays_o, *rays_d, *hits_t, *target, *bg_out;
    int64_t cap;                 // sample capacity
    int64_t* rays_a; int32_t* counter; int64_t* overflow; uint32_t* ticket;
};

// Philox4x32-10 (Salmon et al. 2011): counter (c0..c3), key (k0, k1) -> four 32-bit words
__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += 0x9E3779B9u; key.y += 0xBB67AE85u;
    }
    return ctr;
}
__device__ __forceinline__ float u01(uint32_t w) { return (float)(w >> 8) * 0x1p-24f; }   // [0, 1): 24 random bits, like torch's uniform_ for f32
// stream 0: the jitter of global ray `index`; stream 1: the background colour of the iteration
__device__ __forceinline__ uint4 train_draw(uint64_t seed, uint64_t iteration, uint32_t stream, uint64_t index) {
    return philox4x32_10(make_uint4((uint32_t)index, (uint32_t)(index >> 32), (uint32_t)iteration, (uint32_t)(iteration >> 32) ^ (stream << 31)),
                         make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
}

__global__ void __launch_bounds__(256) k_train_march(TrainBatch b, MarchCfg c, int64_t ray_cap, int32_t* __restrict__ counts, float* __restrict__ park) {
    __shared__ uint32_t s_lut[MARCH_LUT_MAX];
    __shared__ int wave_tot[4];
    __shared__ int64_t carry_s;
    march_lut(c, s_lut);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 4 + wave;
    int64_t n_live = ray_cap;
    if (b.n_rays_dev) n_live = max((int64_t)0, min((int64_t)*b.n_rays_dev, ray_cap));
    const uint64_t seed = b.rng ? b.rng[0] : 0ull, iteration = b.rng ? b.rng[1] : 0ull;
    float bg[3];
    {
        const uint4 w = train_draw(seed, iteration, 1u, 0ull);
        bg[0] = b.bg_in ? b.bg_in[0] : u01(w.x); bg[1] = b.bg_in ? b.bg_in[1] : u01(w.y); bg[2] = b.bg_in ? b.bg_in[2] : u01(w.z);
    }
    if (blockIdx.x == 0 && threadIdx.x < 3) b.bg_out[threadIdx.x] = bg[threadIdx.x];
    if (n < ray_cap) {
        int s = 0;
        if (n < n_live) {
            int64_t id = b.ids ? b.ids[n] : b.order[*b.cursor + n];
            if (id < 0) id += b.n_pool;                        // torch's indexing: -1 is the last row
            const bool ok = id >= 0 && id < b.n_pool;          // out of range: a NaN row, which no loss survives silently (k_gather_ray_batch)
            const float bad = __int_as_float(0x7fc00000);
            Ray q;
            const float gx = ok ? b.pool_o[3 * id] : bad, gy = ok ? b.pool_o[3 * id + 1] : bad, gz = ok ? b.pool_o[3 * id + 2] : bad;
            q.dx = ok ? b.pool_d[3 * id] : bad; q.dy = ok ? b.pool_d[3 * id + 1] : bad; q.dz = ok ? b.pool_d[3 * id + 2] : bad;
            // k_clip_rays, same expressions
            q.ox = gx - b.cx; q.oy = gy - b.cy; q.oz = gz - b.cz;
            q.dxi = 1.0f / q.dx; q.dyi = 1.0f / q.dy; q.dzi = 1.0f / q.dz;
            const float ax = (0.0f - b.hx - q.ox) * q.dxi, bx = (0.0f + b.hx - q.ox) * q.dxi;
            const float ay = (0.0f - b.hy - q.oy) * q.dyi, by = (0.0f + b.hy - q.oy) * q.dyi;
            const float az = (0.0f - b.hz - q.oz) * q.dzi, bz = (0.0f + b.hz - q.oz) * q.dzi;
            float ta = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
            float tb = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
            if (ta > tb) { ta = -1.0f; tb = -1.0f; }
            float s0 = -1.0f, s1 = -1.0f;
            if (tb > 0) { s0 = fmaxf(ta, 0.0f); s1 = tb; }
            const float h0 = fmaxf(s0, b.near_plane), t2 = fminf(s1, b.far_plane);
            const float jitter = b.noise_in ? b.noise_in[n] : u01(train_draw(seed, iteration, 0u, (uint64_t)(b.ray_offset + n)).x);
            if (lane == 0) {
                b.rays_o[3 * n] = q.ox; b.rays_o[3 * n + 1] = q.oy; b.rays_o[3 * n + 2] = q.oz;
                b.rays_d[3 * n] = q.dx; b.rays_d[3 * n + 1] = q.dy; b.rays_d[3 * n + 2] = q.dz;
                b.hits_t[2 * n] = h0; b.hits_t[2 * n + 1] = t2;
                if (b.target) {
                    const float a = b.pool_alpha ? (ok ? b.pool_alpha[id] : bad) : 1.f;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        float v = b.pool_rgb ? (ok ? b.pool_rgb[3 * id + k] : bad) : 0.f;
                        if (b.pool_alpha) {   // apply_background_color (Datasets/utils.py:185-189): lerp(bg, rgb, alpha).clamp(0, 1), torch's two-branch lerp
                            const float diff = v - bg[k];
                            v = a < 0.5f ? fmaf(a, diff, bg[k]) : fmaf(-diff, 1.f - a, v);   // (torch's kernel is built with contraction: one rounding)
                            v = fminf(fmaxf(v, 0.f), 1.f);
                        }
                        b.target[3 * n + k] = v;
                    }
                }
            }
            float t1 = h0;
            if (t1 >= 0) t1 += calc_dt(t1, c.esf, c.max_samples, c.grid_size, c.scale) * jitter;
            s = wave_march_ray<false>(q, t1, t2, c, s_lut, c.max_samples, 0, park + n * c.max_samples, nullptr, nullptr, nullptr, nullptr);
        } else if (lane == 0) {   // a row behind the live rays: a ray that misses everything
            b.rays_o[3 * n] = 0.f; b.rays_o[3 * n + 1] = 0.f; b.rays_o[3 * n + 2] = 0.f;
            b.rays_d[3 * n] = 0.f; b.rays_d[3 * n + 1] = 0.f; b.rays_d[3 * n + 2] = 1.f;
            b.hits_t[2 * n] = b.near_plane; b.hits_t[2 * n + 1] = -1.f;
            if (b.target) { b.target[3 * n] = 0.f; b.target[3 * n + 1] = 0.f; b.target[3 * n + 2] = 0.f; }
        }
        if (lane == 0) __hip_atomic_store(&counts[n], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the last workgroup to get here: k_scan_assign_cap over all counts, then the cursor and the generator move on
    if (!nrc_last_workgroup(b.ticket, blockIdx.x, gridDim.x)) return;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < ray_cap; base += 256) {
        const int64_t r = base + threadIdx.x;
        const int v = r < ray_cap ? __hip_atomic_load(&counts[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        const int incl = nrc_wave_incl_sum_i(v, lane);
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int64_t off = carry_s;
        for (int w = 0; w < wave; w++) off += wave_tot[w];
        if (r < ray_cap) {
            const int64_t start = off + incl - v;
            b.rays_a[3 * r] = r;
            b.rays_a[3 * r + 1] = start + v > b.cap ? min(start, b.cap) : start;
            b.rays_a[3 * r + 2] = start + v > b.cap ? max(b.cap - start, (int64_t)0) : (int64_t)v;
        }
        __syncthreads();
        if (threadIdx.x == 255) carry_s = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        b.counter[0] = (int32_t)carry_s; b.counter[1] = (int32_t)n_live;
        if (b.overflow) b.overflow[0] = max(carry_s - b.cap, (int64_t)0);
        if (b.cursor && !b.ids) *b.cursor += n_live;
        if (b.rng) b.rng[1] = iteration + 1ull;
    }
}

// the batch of a training iteration out of the resident ray pool: one launch for all fields (see the header)
__global__ void __launch_bounds__(256) k_gather_ray_batch(const int64_t* __restrict__ ids, int64_t n, int64_t n_pool, const float* __restrict__ a,
                                                          const float* __restrict__ b, const float* __restrict__ c, const float* __restrict__ d,
                                                          float* __restrict__ oa, float* __restrict__ ob, float* __restrict__ oc, float* __restrict__ od) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int64_t r = ids[i];
    if (r < 0) r += n_pool;                    // torch's indexing: -1 is the last row
    const bool ok = r >= 0 && r < n_pool;      // out of range (torch raises a device assert): the row is NaN, which no loss survives silently
    const float bad = __int_as_float(0x7fc00000);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (a) oa[3 * i + k] = ok ? a[3 * r + k] : bad;
        if (b) ob[3 * i + k] = ok ? b[3 * r + k] : bad;
        if (c) oc[3 * i + k] = ok ? c[3 * r + k] : bad;
    }
    if (d) od[i] = ok ? d[r] : bad;
}

// Fixed sample capacity (graph capture): cut the ray segments at `cap` rows and make the unused tail inert, all from the device-side total.
__global__ void __launch_bounds__(256) k_march_cap(int64_t n_rays, int64_t cap, const int32_t* __restrict__ counter, int64_t* __restrict__ rays_a,
                                                   float* __restrict__ xyzs, float* __restrict__ dirs, float* __restrict__ deltas, float* __restrict__ ts,
                                                   int64_t* __restrict__ overflow) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && overflow) overflow[0] = max((int64_t)counter[0] - cap, (int64_t)0);
    if (i < n_rays) {
        const int64_t start = rays_a[3 * i + 1], n = rays_a[3 * i + 2];
        if (start + n > cap) {
            rays_a[3 * i + 1] = min(start, cap);
            rays_a[3 * i + 2] = max(cap - start, (int64_t)0);
        }
    }
    if (i < cap && i >= (int64_t)counter[0]) {
        xyzs[3 * i] = 0.f; xyzs[3 * i + 1] = 0.f; xyzs[3 * i + 2] = 0.f;
        dirs[3 * i] = 0.f; dirs[3 * i + 1] = 0.f; dirs[3 * i + 2] = 1.f;
        deltas[i] = 0.f; ts[i] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ ray marching (test)
__global__ void __launch_bounds__(256) k_march_test(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                                    float* __restrict__ hits_t, const int64_t* __restrict__ alive,
                                                    int64_t n_alive, MarchCfg c, int N_samples, float* __restrict__ xyzs,
                                                    float* __restrict__ dirs, float* __restrict__ deltas,
                                                    float* __restrict__ ts, int32_t* __restrict__ n_eff) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= n_alive) return;
    const int64_t r = alive[n];
    const Ray q = load_ray(rays_o, rays_d, r);
    float t = hits_t[2 * r];
    const float t2 = hits_t[2 * r + 1];
    float x, y, z, dt, t_next = t;
    int s = 0;
    const int64_t base = n * N_samples;
    while (t < t2 && s < N_samples) {
        if (march_step(q, c, (const uint32_t*)nullptr, t, x, y, z, dt)) {
            const int64_t k = base + s;
            xyzs[3 * k] = x; xyzs[3 * k + 1] = y; xyzs[3 * k + 2] = z;
            dirs[3 * k] = q.dx; dirs[3 * k + 1] = q.dy; dirs[3 * k + 2] = q.dz;
            ts[k] = t; deltas[k] = dt;
            t += dt; s++;
            t_next = t;  // the reference stores t after every ACCEPTED sample only (raymarching.cu:390)
        }
    }
    if (s > 0) hits_t[2 * r] = t_next;
    n_eff[n] = s;
    // the reference returns torch::zeros outputs: clear the unused tail of this ray's rows
    for (int k = s; k < N_samples; k++) {
        const int64_t j = base + k;
        xyzs[3 * j] = 0.f; xyzs[3 * j + 1] = 0.f; xyzs[3 * j + 2] = 0.f;
        dirs[3 * j] = 0.f; dirs[3 * j + 1] = 0.f; dirs[3 * j + 2] = 0.f;
        ts[j] = 0.f; deltas[j] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ 63-bit Morton codes
__device__ __forceinline__ uint64_t split_by_3(uint32_t a) {
    uint64_t x = a & 0x1fffff;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, 64));
    return v;
}
// grid-stride min/max; partial[b*6 + {0..2}] = min xyz, {3..5} = max xyz
__global__ void __launch_bounds__(256) k_bounds_partial(const float* __restrict__ pos, int64_t n, float* __restrict__ partial) {
    __shared__ float sm[4][6];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int k = 0; k < 3; k++) { const float v = pos[3 * i + k]; mn[k] = fminf(mn[k], v); mx[k] = fmaxf(mx[k], v); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; k++) { mn[k] = wave_min(mn[k]); mx[k] = wave_max(mx[k]); }
    if (lane == 0) { for (int k = 0; k < 3; k++) { sm[wave][k] = mn[k]; sm[wave][3 + k] = mx[k]; } }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = sm[0][threadIdx.x];
        for (int w = 1; w < 4; w++) v = threadIdx.x < 3 ? fminf(v, sm[w][threadIdx.x]) : fmaxf(v, sm[w][threadIdx.x]);
        partial[blockIdx.x * 6 + threadIdx.x] = v;
    }
}
// final[0..2] = min, final[3] = cube size (max extent)
__global__ void __launch_bounds__(64) k_bounds_final(const float* __restrict__ partial, int n_partial, float* __restrict__ fin) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = threadIdx.x; i < n_partial; i += 64) {
#pragma unroll
        for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], partial[i * 6 + k]); mx[k] = fmaxf(mx[k], partial[i * 6 + 3 + k]); }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { mn[k] = wave_min(mn[k]); mx[k] = wave_max(mx[k]); }
    if (threadIdx.x == 0) {
        fin[0] = mn[0]; fin[1] = mn[1]; fin[2] = mn[2];
        fin[3] = fmaxf(mx[0] - mn[0], fmaxf(mx[1] - mn[1], mx[2] - mn[2]));
    }
}
__global__ void k_morton_encode(const float* __restrict__ pos, int64_t n, const float* __restrict__ fin, int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float rcp = 1.0f / fin[3];
    const float factor = 2097151.0f;
    // __saturatef semantics: clamp to [0,1], NaN -> 0
    const float nx = fminf(fmaxf((pos[3 * i] - fin[0]) * rcp, 0.0f), 1.0f);
    const float ny = fminf(fmaxf((pos[3 * i + 1] - fin[1]) * rcp, 0.0f), 1.0f);
    const float nz = fminf(fmaxf((pos[3 * i + 2] - fin[2]) * rcp, 0.0f), 1.0f);
    const uint32_t x = (uint32_t)(nx * factor), y = (uint32_t)(ny * factor), z = (uint32_t)(nz * factor);
    out[i] = (int64_t)(split_by_3(x) | split_by_3(y) << 1 | split_by_3(z) << 2);
}

// ------------------------------------------------------------------------------------------------ ray generation
struct RayGenCfg {
    int width, height;
    float min_x, max_x, min_y, max_y, step_x, step_y;  // torch.linspace(start, end, steps) in f32
    float R[9], pos[3];                                // c2w rotation (row-major) and position as f32
};
// torch.linspace f32 device formula: idx < steps/2 ? start + step*idx : end - step*(steps-idx-1)
__device__ __forceinline__ float linspace_at(float start, float end, float step, int steps, int idx) {
    return idx < steps / 2 ? start + step * idx : end - step * (steps - idx - 1);
}
__global__ void __launch_bounds__(256) k_generate_rays(RayGenCfg g, float* __restrict__ origin, float* __restrict__ direction,
                                                       float* __restrict__ view_dir) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n = (int64_t)g.width * g.height;
    if (i >= n) return;
    const int px = (int)(i % g.width), py = (int)(i / g.width);
    const float lx = linspace_at(g.min_x, g.max_x, g.step_x, g.width, px);
    const float ly = linspace_at(g.min_y, g.max_y, g.step_y, g.height, py);
    // dir = local @ R^T  (Datasets/utils.py:1035): dir[k] = sum_j local[j] * R[k][j], local = (lx, ly, 1)
    const float dx = lx * g.R[0] + ly * g.R[1] + g.R[2];
    const float dy = lx * g.R[3] + ly * g.R[4] + g.R[5];
    const float dz = lx * g.R[6] + ly * g.R[7] + g.R[8];
    if (direction) { direction[3 * i] = dx; direction[3 * i + 1] = dy; direction[3 * i + 2] = dz; }
    if (origin) { origin[3 * i] = g.pos[0]; origin[3 * i + 1] = g.pos[1]; origin[3 * i + 2] = g.pos[2]; }
    if (view_dir) {
        // torch.nn.functional.normalize: v / max(||v||_2, 1e-12)
        const float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
        view_dir[3 * i] = dx / nrm; view_dir[3 * i + 1] = dy / nrm; view_dir[3 * i + 2] = dz / nrm;
    }
}


// ------------------------------------------------------------------------------------------------ fused image pipeline, stage 1
// Camera rays are generated on the fly (no ray tensors read from HBM): pixel -> ray (View.get_rays semantics) -> shift by the
// model centre (InstantNGP/Renderer.py:39) -> slab test against the model box (intersection.cu:5-22, :51) -> near/far clamp
// (Renderer.py:42-43) -> DDA march with the TEST kernel's step rule (raymarching.cu:367-401).  Each ray's (o, d, t1, t2) is
// kept (32 B/ray) for the later stages.
struct RenderCam {
    RayGenCfg g;
    float center[3], half[3];
    float near_plane, far_plane;
    int64_t ray_begin;  // first pixel (row-major) of this shard
};
__device__ __forceinline__ void camera_ray(const RenderCam& cam, int64_t pix, Ray& q, float& t1, float& t2) {
    const int px = (int)(pix % cam.g.width), py = (int)(pix / cam.g.width);
    const float lx = linspace_at(cam.g.min_x, cam.g.max_x, cam.g.step_x, cam.g.width, px);
    const float ly = linspace_at(cam.g.min_y, cam.g.max_y, cam.g.step_y, cam.g.height, py);
    const float dx = lx * cam.g.R[0] + ly * cam.g.R[1] + cam.g.R[2];
    const float dy = lx * cam.g.R[3] + ly * cam.g.R[4] + cam.g.R[5];
    const float dz = lx * cam.g.R[6] + ly * cam.g.R[7] + cam.g.R[8];
    const float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-12f);
    q.dx = dx / nrm; q.dy = dy / nrm; q.dz = dz / nrm;
    q.ox = cam.g.pos[0] - cam.center[0]; q.oy = cam.g.pos[1] - cam.center[1]; q.oz = cam.g.pos[2] - cam.center[2];
    q.dxi = 1.0f / q.dx; q.dyi = 1.0f / q.dy; q.dzi = 1.0f / q.dz;
    const float ax = (0.f - cam.half[0] - q.ox) * q.dxi, bx = (0.f + cam.half[0] - q.ox) * q.dxi;
    const float ay = (0.f - cam.half[1] - q.oy) * q.dyi, by = (0.f + cam.half[1] - q.oy) * q.dyi;
    const float az = (0.f - cam.half[2] - q.oz) * q.dzi, bz = (0.f + cam.half[2] - q.oz) * q.dzi;
    float a = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    float b = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    if (a > b) { a = -1.0f; b = -1.0f; }
    if (b > 0) a = fmaxf(a, 0.0f); else { a = -1.0f; b = -1.0f; }  // hits_t keeps -1 unless t2 > 0 (intersection.cu:48-53)
    t1 = fmaxf(a, cam.near_plane);
    t2 = fminf(b, cam.far_plane);
}
// Tile-interleaved sample layout (the MI355X-native core of the image pipeline).
// A tile = 8x8 pixels = one wave, lane = pixel (lx = lane & 7, ly = lane >> 3).  Sample k of the ray on lane l of tile T lives
// in slot (tile_off[T] + k) * 64 + l: a "row" holds the k-th samples of the 64 rays of a tile.  Consequences:
//   * every kernel of the pipeline reads and writes rows = 256-byte coalesced segments (the ray-major layout of the
//     reference makes each lane stride through its own ray: one cache line per lane per access);
//   * the 64 samples of a row are neighbours in space (adjacent pixels, same step), so the hash-grid gathers of a wave
//     fall into few cache lines -- the texture-address path (one distinct line per clock per CU) is THE bottleneck of
//     the encoding, see profiles/;
//   * a sample record shrinks to its t (4 B): the ray is implied by the slot, dt is a function of t;
//   * compositing walks a ray serially per lane in exactly the reference's order (volumerendering.cu:205-249).
// Rays of a tile with fewer samples leave holes in the tail rows (ts = -1).
__device__ __forceinline__ int64_t tile_pixel(const RayGenCfg& g, int64_t tile, int lane, int tiles_x) {
    const int tx = (int)(tile % tiles_x), ty = (int)(tile / tiles_x);
    const int px = tx * NRC_TILE_W + (lane & (NRC_TILE_W - 1)), py = ty * NRC_TILE_H + (lane >> NRC_TILE_W_LOG2);
    return (px < g.width && py < g.height) ? (int64_t)py * g.width + px : -1;
}
__global__ void __launch_bounds__(256) k_render_count(RenderCam cam, MarchCfg c, int tiles_x, int64_t tile_begin, int64_t n_tiles,
                                                      float* __restrict__ ray_od, float* __restrict__ ray_t,
                                                      int32_t* __restrict__ ray_cnt, int32_t* __restrict__ tile_rows, float* __restrict__ ts_prov) {
    __shared__ uint32_t s_lut[MARCH_LUT_MAX];
    march_lut(c, s_lut);
    const int lane = threadIdx.x & 63;
    const int64_t lt = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // local tile
    if (lt >= n_tiles) return;
    const int64_t pix = tile_pixel(cam.g, tile_begin + lt, lane, tiles_x);
    const int64_t q = lt * 64 + lane;
    int n = 0;
    Ray ry = {0.f, 0.f, 0.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    float t1 = 0.f, t2 = -1.f;
    if (pix >= 0) {
        camera_ray(cam, pix, ry, t1, t2);
        float t = t1, x, y, z, dt;
        if (ts_prov) {
            // the samples are parked in a per-tile arena of max_samples rows (sample k of lane l at (lt * max_samples + k) * 64 + l):
            // the second pass copies them to their final rows instead of marching again
            float* park = ts_prov + (size_t)lt * c.max_samples * 64 + lane;
            while (t < t2 && n < c.max_samples) {
                if (march_step(ry, c, s_lut, t, x, y, z, dt)) { park[(size_t)n * 64] = t; t += dt; n++; }
            }
        } else {
            while (t < t2 && n < c.max_samples) {
                if (march_step(ry, c, s_lut, t, x, y, z, dt)) { t += dt; n++; }
            }
        }
    }
    float* od = ray_od + lt * 384 + lane;  // per-tile SoA [6][64]: every component is one 256-byte row
    od[0] = ry.ox; od[64] = ry.oy; od[128] = ry.oz; od[192] = ry.dx; od[256] = ry.dy; od[320] = ry.dz;
    ray_t[2 * q] = t1; ray_t[2 * q + 1] = t2;
    ray_cnt[q] = n;
    int m = n;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    if (ts_prov) {   // holes of the tile's rows (rays with fewer samples than its longest): the arena can then be queried in place (nrc_ngp_query_samples)
        float* park = ts_prov + (size_t)lt * c.max_samples * 64 + lane;
        for (int k = n; k < m; k++) park[(size_t)k * 64] = -1.0f;
    }
    const int tile_samples = nrc_group_sum_i<64>(n);
    if (lane == 0) { tile_rows[lt] = m; tile_rows[n_tiles + lt] = tile_samples; }   // second half of the array: samples per tile (for the total)
}
// exclusive scan of tile_rows[0..n_tiles) -> tile_off[0..n_tiles] (tile_off[n_tiles] = total rows); counter = (total rows, total samples: the
// sum of tile_rows[n_tiles..2 n_tiles)), so that the caller's ONE host read sizes the sample buffers and reports the marched samples.
// One workgroup; thread t owns the contiguous chunk [t c, (t + 1) c) (c = ceil(n / 1024): 10 tiles of an 800x800 image) -- one block scan over the
// chunk sums instead of one per 1024 tiles (round 4: 14 -> 6 us).  mailbox (optional, nrc_host_mailbox_alloc): the two totals and the caller's
// ticket also go straight to host memory, the ticket last -- the host polls it instead of copying `counter` back.
__global__ void __launch_bounds__(1024) k_scan_tiles(const int32_t* __restrict__ tile_rows, int64_t n_tiles, int32_t* __restrict__ tile_off,
                                                     int32_t* __restrict__ counter, int64_t* __restrict__ mailbox, int64_t mailbox_ticket) {
    __shared__ int wave_tot[16];
    __shared__ int samp_tot[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t chunk = (n_tiles + 1023) / 1024;
    const int64_t i0 = (int64_t)threadIdx.x * chunk, i1 = i0 + chunk < n_tiles ? i0 + chunk : n_tiles;
    int rows = 0, samples = 0;
    for (int64_t i = i0; i < i1; i++) { rows += tile_rows[i]; samples += tile_rows[n_tiles + i]; }
    const int incl = nrc_wave_incl_sum_i(rows, lane);
    samples = nrc_group_sum_i<64>(samples);
    if (lane == 63) wave_tot[wave] = incl;
    if (lane == 0) samp_tot[wave] = samples;
    __syncthreads();
    int off = incl - rows, total = 0;
#pragma unroll
    for (int w = 0; w < 16; w++) { off += w < wave ? wave_tot[w] : 0; total += wave_tot[w]; }
    for (int64_t i = i0; i < i1; i++) { tile_off[i] = off; off += tile_rows[i]; }
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int w = 0; w < 16; w++) tot += samp_tot[w];
        tile_off[n_tiles] = total; counter[0] = total; counter[1] = tot;
        if (mailbox) {
            __hip_atomic_store(mailbox, (int64_t)total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mailbox + 1, (int64_t)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(mailbox + 2, mailbox_ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// ---- slab-major row order (front-to-back processing with early termination, see nrc_ngp_render_layers) -------------------------
// Row k of tile T normally sits at tile_off[T] + k.  In slab order the rows of all tiles with k in [0, G) come first (tile after
// tile, k ascending inside a tile), then k in [G, 2G), ...:
//   row_of[tile_off[T] + k] = slab_off[k / G] + sum over T' < T of rows_in_slab(T', k / G) + k % G
// so that a chunk of consecutive rows is a slab of the image in DEPTH and the chunks of an image can be composited front to back.
// G = 16 consecutive samples of a tile stay adjacent: with G = 1 (pure layer order) the encoder lost 24 % to worse cache reuse
// between neighbouring waves.
#ifndef NRC_SLAB_G
#define NRC_SLAB_G 16   // rows of a tile per slab (measured: 8 and 32 in profiles/ -- see DESIGN 3.1)
#endif
__device__ __forceinline__ int rows_in_slab(int tile_rows, int slab) { return min(NRC_SLAB_G, max(0, tile_rows - slab * NRC_SLAB_G)); }
__global__ void __launch_bounds__(256) k_slab_totals(const int32_t* __restrict__ tile_rows, int64_t n_tiles, int32_t* __restrict__ slab_tot) {
    __shared__ int smem[8];
    const int slab = blockIdx.x;
    int acc = 0;
    for (int64_t t = threadIdx.x; t < n_tiles; t += 256) acc += rows_in_slab(tile_rows[t], slab);
    int total;
    (void)nrc_block256_excl_scan_i(acc, smem, &total);
    if (threadIdx.x == 0) slab_tot[slab] = total;
}
__global__ void k_slab_offsets(int n_slabs, int32_t* __restrict__ slab_off) {  // in: totals, out: exclusive prefix (+ grand total at [n_slabs])
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int run = 0;
    for (int s = 0; s < n_slabs; s++) { const int c = slab_off[s]; slab_off[s] = run; run += c; }
    slab_off[n_slabs] = run;
}
__global__ void __launch_bounds__(256) k_slab_rows(const int32_t* __restrict__ tile_rows, const int32_t* __restrict__ tile_off, int64_t n_tiles,
                                                   const int32_t* __restrict__ slab_off, int32_t* __restrict__ row_of) {
    __shared__ int smem[8];
    const int slab = blockIdx.x;
    const int first = slab_off[slab];
    if (slab_off[slab + 1] == first) return;
    // thread x owns the contiguous tiles [x c, (x + 1) c): one block scan over the chunk sums (a scan per 256 tiles was 40 of them at 800x800: 44 us)
    const int64_t per = (n_tiles + 255) / 256, t0 = (int64_t)threadIdx.x * per, t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
    int mine = 0;
    for (int64_t t = t0; t < t1; t++) mine += rows_in_slab(tile_rows[t], slab);
    int total;
    int run = first + nrc_block256_excl_scan_i(mine, smem, &total);
    for (int64_t t = t0; t < t1; t++) {
        const int c = rows_in_slab(tile_rows[t], slab);
        for (int j = 0; j < c; j++) row_of[tile_off[t] + slab * NRC_SLAB_G + j] = run + j;
        run += c;
    }
}
// second march: ts[row(k) * 64 + lane] = t of the k-th sample; holes = -1; row_tile[row] = local tile of the row;
// row(k) = tile_off + k (tile-major) or row_of[tile_off + k] (layer-major)
__global__ void __launch_bounds__(256) k_render_write(MarchCfg c, int64_t n_tiles, const float* __restrict__ ray_od,
                                                      const float* __restrict__ ray_t, const int32_t* __restrict__ ray_cnt,
                                                      const int32_t* __restrict__ tile_off, float* __restrict__ ts,
                                                      int32_t* __restrict__ row_tile, const int32_t* __restrict__ row_of,
                                                      const float* __restrict__ ts_prov, int64_t row_cap, int32_t* __restrict__ row_k = nullptr) {
    __shared__ uint32_t s_lut[MARCH_LUT_MAX];
    march_lut(c, s_lut);
    const int lane = threadIdx.x & 63;
    const int64_t lt = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (lt >= n_tiles) return;
    const int64_t q = lt * 64 + lane;
    const int64_t row0 = tile_off[lt];
    // row_cap: rows the sample buffers hold (fixed-capacity frames; INT64_MAX otherwise).  A tile whose rows would cross it writes the rows that fit:
    // nothing goes out of bounds, the frame is reported as overflowed (counter[0] > capacity) and is rendered again by the caller.
    const int rows = (int)min((int64_t)(tile_off[lt + 1] - (int)row0), max(row_cap - row0, (int64_t)0));
    const int N = ray_cnt[q];
    auto row_at = [&](int k) -> int64_t { return row_of ? (int64_t)row_of[row0 + k] : row0 + k; };
    for (int k = lane; k < rows; k += 64) {
        const int64_t r = row_at(k);
        row_tile[r] = (int32_t)lt;
        if (row_k) row_k[r] = k;   // slab-major rows queried from the arena: which sample of the tile a row is
    }
    if (!ts) return;   // the arena is queried in place: only the rows' tiles are wanted
    if (ts_prov) {  // the count pass parked the samples: a coalesced copy (256-byte rows) instead of the second march
        const float* park = ts_prov + (size_t)lt * c.max_samples * 64 + lane;
        for (int k0 = 0; k0 < rows; k0 += 8) {  // eight rows in flight: the copy is latency-bound otherwise (327 us against 300 for the march)
            float v[8];
            int64_t dst[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int k = k0 + u;
                v[u] = k < N ? park[(size_t)k * 64] : -1.0f;
                dst[u] = k < rows ? row_at(k) : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (k0 + u < rows) ts[dst[u] * 64 + lane] = v[u];
        }
        return;
    }
    const float* od = ray_od + lt * 384 + lane;
    Ray ry;
    ry.ox = od[0]; ry.oy = od[64]; ry.oz = od[128]; ry.dx = od[192]; ry.dy = od[256]; ry.dz = od[320];
    ry.dxi = 1.0f / ry.dx; ry.dyi = 1.0f / ry.dy; ry.dzi = 1.0f / ry.dz;
    float t = ray_t[2 * q], x, y, z, dt;
    const float t2 = ray_t[2 * q + 1];
    int s = 0;
    while (t < t2 && s < N) {
        if (march_step(ry, c, s_lut, t, x, y, z, dt)) {
            if (s < rows) ts[row_at(s) * 64 + lane] = t;
            t += dt; s++;
        }
    }
    for (int k = N; k < rows; k++) ts[row_at(k) * 64 + lane] = -1.0f;
}

MarchCfg make_cfg(const uint8_t* bitfield, int cascades, float scale, float esf, int grid_size, int max_samples, float dt_scale) {
    MarchCfg c;
    c.bitfield = bitfield; c.cascades = cascades; c.grid_size = grid_size; c.max_samples = max_samples;
    c.scale = scale; c.esf = esf; c.dt_scale = dt_scale;
    c.grid_size3 = (uint32_t)grid_size * grid_size * grid_size;
    c.grid_size_inv = 1.0f / grid_size;
    c.mip0_bound = fminf(0.5f, scale);  // fminf(scalbnf(1, -1), scale)
    c.mip0_bound_inv = 1.0f / c.mip0_bound;
    c.use_lut = 0;
    return c;
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

int nrc_morton3D(const int32_t* coords, int64_t n, int32_t* indices, nrc_stream_t stream) {
    NRC_ENTER();
    if (n < 0 || (n > 0 && (!coords || !indices))) return NRC_ERR_INVALID;
    if (n == 0) return NRC_OK;
    hipLaunchKernelGGL(k_morton3D, dim3(nrc_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, coords, n, indices);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_morton3D_invert(const int32_t* indices, int64_t n, int32_t* coords, nrc_stream_t stream) {
    NRC_ENTER();
    if (n < 0 || (n > 0 && (!coords || !indices))) return NRC_ERR_INVALID;
    if (n == 0) return NRC_OK;
    hipLaunchKernelGGL(k_morton3D_invert, dim3(nrc_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, indices, n, coords);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_packbits(const void* grid, int32_t grid_dtype, int64_t n_bytes, float thr, uint8_t* bitfield, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_bytes < 0 || (n_bytes > 0 && (!grid || !bitfield)) || (grid_dtype != 0 && grid_dtype != 1)) return NRC_ERR_INVALID;
    if (n_bytes == 0) return NRC_OK;
    const bool aligned = (((uintptr_t)grid & 15) == 0) && (((uintptr_t)bitfield & 3) == 0);
    if (!aligned) {
        if (grid_dtype == 0)
            hipLaunchKernelGGL(k_packbits_unaligned<float>, dim3(nrc_cdiv(n_bytes, 256)), dim3(256), 0, (hipStream_t)stream,
                               (const float*)grid, n_bytes, thr, bitfield);
        else
            hipLaunchKernelGGL(k_packbits_unaligned<__half>, dim3(nrc_cdiv(n_bytes, 256)), dim3(256), 0, (hipStream_t)stream,
                               (const __half*)grid, n_bytes, thr, bitfield);
    } else if (grid_dtype == 0)
        hipLaunchKernelGGL(k_packbits<float>, dim3(nrc_cdiv(n_bytes, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)grid, n_bytes, thr, bitfield);
    else
        hipLaunchKernelGGL(k_packbits<__half>, dim3(nrc_cdiv(n_bytes, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const __half*)grid, n_bytes, thr, bitfield);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_ray_aabb_intersect(const float* rays_o, const float* rays_d, const float* centers, const float* half_sizes,
                           int64_t n_rays, int64_t n_voxels, int32_t max_hits, int32_t* hit_cnt, float* hits_t,
                           int64_t* hits_voxel_idx, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_rays < 0 || n_voxels < 0 || max_hits < 1) return NRC_ERR_INVALID;
    if (n_rays == 0) return NRC_OK;
    if (!rays_o || !rays_d || !hit_cnt || !hits_t || !hits_voxel_idx || (n_voxels > 0 && (!centers || !half_sizes))) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_ray_prim_intersect<false>, dim3(nrc_cdiv(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, rays_o,
                       rays_d, centers, half_sizes, n_rays, n_voxels, max_hits, hit_cnt, hits_t, hits_voxel_idx);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_ray_sphere_intersect(const float* rays_o, const float* rays_d, const float* centers, const float* radii,
                             int64_t n_rays, int64_t n_spheres, int32_t max_hits, int32_t* hit_cnt, float* hits_t,
                             int64_t* hits_sphere_idx, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_rays < 0 || n_spheres < 0 || max_hits < 1) return NRC_ERR_INVALID;
    if (n_rays == 0) return NRC_OK;
    if (!rays_o || !rays_d || !hit_cnt || !hits_t || !hits_sphere_idx || (n_spheres > 0 && (!centers || !radii))) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_ray_prim_intersect<true>, dim3(nrc_cdiv(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, rays_o,
                       rays_d, centers, radii, n_rays, n_spheres, max_hits, hit_cnt, hits_t, hits_sphere_idx);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

static int64_t train_ws_head_bytes(int64_t n_rays) {
    // counts[n_rays] + block_sums[ceil(n_rays/256)], both i32, 256-byte aligned sections
    const int64_t a = (n_rays * 4 + 255) / 256 * 256;
    const int64_t b = (nrc_cdiv(n_rays, 256) * 4 + 255) / 256 * 256;
    return a + b + 256;
}
int64_t nrc_raymarching_train_ws_bytes(int64_t n_rays, int32_t max_samples) {
    if (n_rays < 0 || max_samples < 1) return NRC_ERR_INVALID;
    // + the parked sample positions (max_samples f32 per ray) for batches that take the wave-per-ray path
    const int64_t park = n_rays <= NRC_WAVE_MARCH_MAX_RAYS ? n_rays * (int64_t)max_samples * 4 : 0;
    return train_ws_head_bytes(n_rays) + park;
}
int nrc_raymarching_train_count(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* bitfield,
                                int32_t cascades, float scale, float esf, const float* noise, int32_t grid_size,
                                int32_t max_samples, int64_t n_rays, int64_t* rays_a, int32_t* counter, void* workspace,
                                nrc_stream_t stream) {
    NRC_ENTER();
    if (n_rays < 0 || !counter || cascades < 1 || grid_size < 1 || max_samples < 1) return NRC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (n_rays == 0) { nrc_zero_async(counter, 8, s); return NRC_OK; }
    if (!rays_o || !rays_d || !hits_t || !bitfield || !noise || !rays_a || !workspace) return NRC_ERR_INVALID;
    const int64_t nb = nrc_cdiv(n_rays, 256);
    int32_t* counts = (int32_t*)workspace;
    int32_t* block_sums = (int32_t*)((char*)workspace + (n_rays * 4 + 255) / 256 * 256);
    const MarchCfg c = make_cfg(bitfield, cascades, scale, esf, grid_size, max_samples, scale);
    if (n_rays <= NRC_WAVE_MARCH_MAX_RAYS) {
        nrc_zero_async(block_sums, sizeof(int32_t) * nb, s);
        float* park = (float*)((char*)workspace + train_ws_head_bytes(n_rays));
        hipLaunchKernelGGL(k_march_wave<false>, dim3((unsigned)nrc_cdiv(n_rays, 4)), dim3(256), 0, s, rays_o, rays_d, hits_t, noise, c, n_rays, counts,
                           block_sums, (const int64_t*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, park);
    } else {
        hipLaunchKernelGGL(k_march_count, dim3(nb), dim3(256), 0, s, rays_o, rays_d, hits_t, noise, c, n_rays, counts, block_sums);
    }
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, s, block_sums, nb, n_rays, counter);
    hipLaunchKernelGGL(k_assign_rays_a, dim3(nb), dim3(256), 0, s, counts, block_sums, n_rays, rays_a);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_raymarching_train_count_posted(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* bitfield, int32_t cascades, float scale,
                                       float esf, const float* noise, int32_t grid_size, int32_t max_samples, int64_t n_rays, int64_t* rays_a, int32_t* counter,
                                       void* workspace, int64_t* count_mailbox, int64_t mailbox_ticket, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_rays < 1 || n_rays > NRC_WAVE_MARCH_MAX_RAYS || !counter || cascades < 1 || grid_size < 1 || max_samples < 1) return NRC_ERR_INVALID;
    if (!rays_o || !rays_d || !hits_t || !bitfield || !noise || !rays_a || !workspace || !count_mailbox) return NRC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    int32_t* counts = (int32_t*)workspace;
    float* park = (float*)((char*)workspace + train_ws_head_bytes(n_rays));
    const MarchCfg c = make_cfg(bitfield, cascades, scale, esf, grid_size, max_samples, scale);
    hipLaunchKernelGGL(k_march_wave<false>, dim3((unsigned)nrc_cdiv(n_rays, 4)), dim3(256), 0, s, rays_o, rays_d, hits_t, noise, c, n_rays, counts,
                       (int32_t*)nullptr, (const int64_t*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, park);
    hipLaunchKernelGGL(k_scan_assign_cap, dim3(1), dim3(1024), 0, s, (const int32_t*)counts, n_rays, (int64_t)1 << 62, rays_a, counter, (int64_t*)nullptr,
                       count_mailbox, mailbox_ticket);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_raymarching_train_write(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* bitfield,
                                int32_t cascades, float scale, float esf, const float* noise, int32_t grid_size,
                                int32_t max_samples, int64_t n_rays, const int64_t* rays_a, float* xyzs, float* dirs,
                                float* deltas, float* ts, const void* workspace, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_rays < 0 || cascades < 1 || grid_size < 1 || max_samples < 1) return NRC_ERR_INVALID;
    if (n_rays == 0) return NRC_OK;
    if (!rays_o || !rays_d || !hits_t || !bitfield || !noise || !rays_a) return NRC_ERR_INVALID;
    const MarchCfg c = make_cfg(bitfield, cascades, scale, esf, grid_size, max_samples, scale);
    if (n_rays <= NRC_WAVE_MARCH_MAX_RAYS && workspace)  // the count pass of the same workspace parked the sample positions
        hipLaunchKernelGGL(k_march_expand, dim3((unsigned)nrc_cdiv(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, c, n_rays, rays_a,
                           (const float*)((const char*)workspace + train_ws_head_bytes(n_rays)), xyzs, dirs, deltas, ts);
    else if (n_rays <= NRC_WAVE_MARCH_MAX_RAYS)
        hipLaunchKernelGGL(k_march_wave<true>, dim3((unsigned)nrc_cdiv(n_rays, 4)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, hits_t, noise, c,
                           n_rays, (int32_t*)nullptr, (int32_t*)nullptr, rays_a, xyzs, dirs, deltas, ts, (float*)nullptr);
    else
        hipLaunchKernelGGL(k_march_write, dim3(nrc_cdiv(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, hits_t,
                           noise, c, n_rays, rays_a, xyzs, dirs, deltas, ts);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_ngp_clip_rays(int64_t n_rays, const float* origin, const float* dirs, const float* center3, const float* half3, float near_plane,
                      float far_plane, float* origin_centred, float* spans, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_rays < 0 || !center3 || !half3) return NRC_ERR_INVALID;
    if (n_rays == 0) return NRC_OK;
    if (!origin || !dirs || !origin_centred || !spans) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_clip_rays, dim3(nrc_cdiv(n_rays, 256)), dim3(256), 0, (hipStream_t)stream, n_rays, origin, dirs, center3[0], center3[1], center3[2],
                       half3[0], half3[1], half3[2], near_plane, far_plane, origin_centred, spans);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_raymarching_train_cap_overflow(int64_t n_rays, int64_t sample_capacity, const int32_t* counter, int64_t* rays_a, float* xyzs, float* dirs,
                              float* deltas, float* ts, int64_t* overflow, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_rays < 0 || sample_capacity < 0) return NRC_ERR_INVALID;
    const int64_t n = n_rays > sample_capacity ? n_rays : sample_capacity;
    if (n == 0) return NRC_OK;
    if (!counter || (n_rays && !rays_a) || (sample_capacity && (!xyzs || !dirs || !deltas || !ts))) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_march_cap, dim3((unsigned)nrc_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, n_rays, sample_capacity, counter, rays_a,
                       xyzs, dirs, deltas, ts, overflow);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_raymarching_train_cap(int64_t n_rays, int64_t sample_capacity, const int32_t* counter, int64_t* rays_a, float* xyzs, float* dirs,
                              float* deltas, float* ts, nrc_stream_t stream) {
    return nrc_raymarching_train_cap_overflow(n_rays, sample_capacity, counter, rays_a, xyzs, dirs, deltas, ts, nullptr, stream);
}
int nrc_raymarching_train_capped(const float* rays_o, const float* rays_d, const float* hits_t, const uint8_t* bitfield, int32_t cascades, float scale,
                                 float esf, const float* noise, int32_t grid_size, int32_t max_samples, int64_t n_rays, int64_t sample_capacity,
                                 int64_t* rays_a, int32_t* counter, float* xyzs, float* dirs, float* deltas, float* ts, int64_t* overflow, void* workspace,
                                 nrc_stream_t stream) {
    NRC_ENTER();
    if (n_rays < 1 || n_rays > NRC_WAVE_MARCH_MAX_RAYS || sample_capacity < 1 || cascades < 1 || grid_size < 1 || max_samples < 1) return NRC_ERR_INVALID;
    if (!rays_o || !rays_d || !hits_t || !bitfield || !noise || !rays_a || !counter || !xyzs || !dirs || !deltas || !ts || !workspace) return NRC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    int32_t* counts = (int32_t*)workspace;
    float* park = (float*)((char*)workspace + train_ws_head_bytes(n_rays));
    const MarchCfg c = make_cfg(bitfield, cascades, scale, esf, grid_size, max_samples, scale);
    hipLaunchKernelGGL(k_march_wave<false>, dim3((unsigned)nrc_cdiv(n_rays, 4)), dim3(256), 0, s, rays_o, rays_d, hits_t, noise, c, n_rays, counts,
                       (int32_t*)nullptr, (const int64_t*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, park);
    hipLaunchKernelGGL(k_scan_assign_cap, dim3(1), dim3(1024), 0, s, (const int32_t*)counts, n_rays, sample_capacity, rays_a, counter, overflow);
    hipLaunchKernelGGL(k_march_expand, dim3((unsigned)(nrc_cdiv(n_rays, 4) + nrc_cdiv(sample_capacity, 256))), dim3(256), 0, s, rays_o, rays_d, c, n_rays,
                       (const int64_t*)rays_a, (const float*)park, xyzs, dirs, deltas, ts, (const int32_t*)counter, sample_capacity);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
static int64_t train_march_head_bytes(int64_t ray_capacity) { return NRC_TICKET_WORDS * 4 + (ray_capacity * 4 + 255) / 256 * 256; }   // [tickets][counts]
int64_t nrc_ngp_train_march_ws_bytes(int64_t ray_capacity, int32_t max_samples) {
    if (ray_capacity < 1 || ray_capacity > NRC_WAVE_MARCH_MAX_RAYS || max_samples < 1) return NRC_ERR_INVALID;
    return train_march_head_bytes(ray_capacity) + ray_capacity * (int64_t)max_samples * 4;
}
int nrc_ngp_train_march(const int64_t* ids, const int64_t* order, int64_t* cursor, const int32_t* n_rays_dev, int64_t ray_capacity, int64_t n_pool,
                        int64_t ray_offset, const float* pool_origin, const float* pool_dir, const float* pool_rgb, const float* pool_alpha,
                        const float* center3, const float* half3, float near_plane, float far_plane, const uint8_t* density_bitfield, int32_t cascades,
                        float scale, float exp_step_factor, int32_t grid_size, int32_t max_samples, uint64_t* rng_state, const float* bg_in,
                        const float* noise_in, int64_t sample_capacity, float* rays_o, float* rays_d, float* hits_t, float* target_rgb, float* bg,
                        int64_t* rays_a, int32_t* counter, float* xyzs, float* dirs, float* deltas, float* ts, int64_t* overflow, void* workspace,
                        nrc_stream_t stream) {
    NRC_ENTER();
    if (ray_capacity < 1 || ray_capacity > NRC_WAVE_MARCH_MAX_RAYS || sample_capacity < 1 || n_pool < 1 || cascades < 1 || grid_size < 1 || max_samples < 1)
        return NRC_ERR_INVALID;
    if ((!ids && (!order || !cursor)) || !pool_origin || !pool_dir || !center3 || !half3 || !density_bitfield || (!rng_state && (!bg_in || !noise_in)) ||
        (target_rgb && !pool_rgb) || !rays_o || !rays_d || !hits_t || !bg || !rays_a || !counter || !xyzs || !dirs || !deltas || !ts || !workspace)
        return NRC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    TrainBatch b;
    b.ids = ids; b.order = order; b.cursor = cursor; b.n_rays_dev = n_rays_dev; b.n_pool = n_pool; b.ray_offset = ray_offset;
    b.pool_o = pool_origin; b.pool_d = pool_dir; b.pool_rgb = pool_rgb; b.pool_alpha = pool_alpha;
    b.cx = center3[0]; b.cy = center3[1]; b.cz = center3[2]; b.hx = half3[0]; b.hy = half3[1]; b.hz = half3[2];
    b.near_plane = near_plane; b.far_plane = far_plane; b.rng = rng_state; b.bg_in = bg_in; b.noise_in = noise_in;
    b.rays_o = rays_o; b.rays_d = rays_d; b.hits_t = hits_t; b.target = target_rgb; b.bg_out = bg; b.cap = sample_capacity;
    b.rays_a = rays_a; b.counter = counter; b.overflow = overflow; b.ticket = (uint32_t*)workspace;
    int32_t* counts = (int32_t*)((char*)workspace + NRC_TICKET_WORDS * 4);
    float* park = (float*)((char*)workspace + train_march_head_bytes(ray_capacity));
    const MarchCfg c = make_cfg(density_bitfield, cascades, scale, exp_step_factor, grid_size, max_samples, scale);
    NRC_STAGE(s, nullptr);
    hipLaunchKernelGGL(k_train_march, dim3((unsigned)nrc_cdiv(ray_capacity, 4)), dim3(256), 0, s, b, c, ray_capacity, counts, park);
    NRC_STAGE(s, "k_train_march");
    hipLaunchKernelGGL(k_march_expand, dim3((unsigned)(nrc_cdiv(ray_capacity, 4) + nrc_cdiv(sample_capacity, 256))), dim3(256), 0, s, (const float*)rays_o,
                       (const float*)rays_d, c, ray_capacity, (const int64_t*)rays_a, (const float*)park, xyzs, dirs, deltas, ts, (const int32_t*)counter,
                       sample_capacity);
    NRC_STAGE(s, "k_march_expand");
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_gather_ray_batch(const int64_t* ids, int64_t n, int64_t n_pool, const float* pool_a3, const float* pool_b3, const float* pool_c3,
                         const float* pool_d1, float* out_a3, float* out_b3, float* out_c3, float* out_d1, nrc_stream_t stream) {
    NRC_ENTER();
    if (n < 0 || n_pool < 0) return NRC_ERR_INVALID;
    if (n == 0) return NRC_OK;
    if (!ids || (pool_a3 && !out_a3) || (pool_b3 && !out_b3) || (pool_c3 && !out_c3) || (pool_d1 && !out_d1)) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_gather_ray_batch, dim3((unsigned)nrc_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, ids, n, n_pool, pool_a3, pool_b3, pool_c3,
                       pool_d1, out_a3, out_b3, out_c3, out_d1);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_raymarching_test(const float* rays_o, const float* rays_d, float* hits_t, const int64_t* alive, int64_t n_alive,
                         const uint8_t* bitfield, int32_t cascades, float scale, float esf, int32_t grid_size,
                         int32_t max_samples, int32_t N_samples, float* xyzs, float* dirs, float* deltas, float* ts,
                         int32_t* n_eff, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_alive < 0 || cascades < 1 || grid_size < 1 || max_samples < 1 || N_samples < 1) return NRC_ERR_INVALID;
    if (n_alive == 0) return NRC_OK;
    if (!rays_o || !rays_d || !hits_t || !alive || !bitfield || !xyzs || !dirs || !deltas || !ts || !n_eff) return NRC_ERR_INVALID;
    const MarchCfg c = make_cfg(bitfield, cascades, scale, esf, grid_size, max_samples, (float)cascades);
    hipLaunchKernelGGL(k_march_test, dim3(nrc_cdiv(n_alive, 256)), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, hits_t,
                       alive, n_alive, c, N_samples, xyzs, dirs, deltas, ts, n_eff);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int64_t nrc_morton_encode_ws_bytes(int64_t n) {
    (void)n;
    return (1024 * 6 + 4) * (int64_t)sizeof(float);
}
int nrc_morton_encode(const float* positions, int64_t n, int64_t* codes, void* workspace, nrc_stream_t stream) {
    NRC_ENTER();
    if (n < 0) return NRC_ERR_INVALID;
    if (n == 0) return NRC_OK;
    if (!positions || !codes || !workspace) return NRC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* fin = partial + 1024 * 6;
    const int nb = (int)(nrc_cdiv(n, 256) < 1024 ? nrc_cdiv(n, 256) : 1024);
    hipLaunchKernelGGL(k_bounds_partial, dim3(nb), dim3(256), 0, s, positions, n, partial);
    hipLaunchKernelGGL(k_bounds_final, dim3(1), dim3(64), 0, s, partial, nb, fin);
    hipLaunchKernelGGL(k_morton_encode, dim3(nrc_cdiv(n, 256)), dim3(256), 0, s, positions, n, fin, codes);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}


static void fill_raygen(RayGenCfg& g, int32_t width, int32_t height, const double* intr, const double* c2w) {
    g.width = width; g.height = height;
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    g.min_x = (float)((0.5 - cx) / fx); g.max_x = (float)((width - 1 + 0.5 - cx) / fx);
    g.min_y = (float)((0.5 - cy) / fy); g.max_y = (float)((height - 1 + 0.5 - cy) / fy);
    g.step_x = width > 1 ? (g.max_x - g.min_x) / (float)(width - 1) : 0.0f;
    g.step_y = height > 1 ? (g.max_y - g.min_y) / (float)(height - 1) : 0.0f;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) g.R[3 * r + c] = (float)c2w[4 * r + c];
        g.pos[r] = (float)c2w[4 * r + 3];
    }
}

int64_t nrc_ngp_render_provisional_bytes(int64_t n_tiles, int32_t max_samples) {
    return (n_tiles < 0 || max_samples < 1) ? -1 : n_tiles * (int64_t)max_samples * 256 + 256;
}
int nrc_ngp_render_count(int32_t width, int32_t height, const double* intr, const double* c2w, const float* center3, const float* half3,
                         float near_plane, float far_plane, int64_t tile_begin, int64_t n_tiles, const uint8_t* bitfield,
                         int32_t cascades, float scale, float esf, int32_t grid_size, int32_t max_samples, float* ray_od,
                         float* ray_t, int32_t* ray_cnt, int32_t* tile_rows, int32_t* tile_off, int32_t* counter, float* ts_provisional,
                         int64_t* count_mailbox, int64_t mailbox_ticket, nrc_stream_t stream) {
    NRC_ENTER();
    if (width < 1 || height < 1 || !intr || !c2w || !center3 || !half3 || n_tiles < 0 || tile_begin < 0 || cascades < 1 ||
        grid_size < 1 || max_samples < 1 || !counter || !tile_off)
        return NRC_ERR_INVALID;
    const int tiles_x = (width + NRC_TILE_W - 1) / NRC_TILE_W, tiles_y = (height + NRC_TILE_H - 1) / NRC_TILE_H;
    if (tile_begin + n_tiles > (int64_t)tiles_x * tiles_y) return NRC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    if (n_tiles > 0 && (!bitfield || !ray_od || !ray_t || !ray_cnt || !tile_rows)) return NRC_ERR_INVALID;
    RenderCam cam;
    fill_raygen(cam.g, width, height, intr, c2w);
    for (int k = 0; k < 3; k++) { cam.center[k] = center3[k]; cam.half[k] = half3[k]; }
    cam.near_plane = near_plane; cam.far_plane = far_plane; cam.ray_begin = 0;
    const MarchCfg c = make_cfg(bitfield, cascades, scale, esf, grid_size, max_samples, (float)cascades);
    if (n_tiles > 0)
        hipLaunchKernelGGL(k_render_count, dim3(nrc_cdiv(n_tiles, 4)), dim3(256), 0, s, cam, c, tiles_x, tile_begin, n_tiles, ray_od, ray_t,
                           ray_cnt, tile_rows, ts_provisional);
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, s, tile_rows, n_tiles, tile_off, counter, count_mailbox, mailbox_ticket);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_ngp_render_write(int64_t n_tiles, const uint8_t* bitfield, int32_t cascades, float scale, float esf, int32_t grid_size,
                         int32_t max_samples, const float* ray_od, const float* ray_t, const int32_t* ray_cnt, const int32_t* tile_off,
                         float* ts, int32_t* row_tile, const float* ts_provisional, int64_t row_capacity, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_tiles < 0 || cascades < 1 || grid_size < 1 || max_samples < 1 || row_capacity < 0) return NRC_ERR_INVALID;
    if (n_tiles == 0) return NRC_OK;
    if (!bitfield || !ray_od || !ray_t || !ray_cnt || !tile_off || !row_tile) return NRC_ERR_INVALID;
    if (!ts && !ts_provisional) return NRC_ERR_INVALID;   // ts == NULL: the parked samples stay where they are (arena queried in place), row_tile only
    const MarchCfg c = make_cfg(bitfield, cascades, scale, esf, grid_size, max_samples, (float)cascades);
    hipLaunchKernelGGL(k_render_write, dim3(nrc_cdiv(n_tiles, 4)), dim3(256), 0, (hipStream_t)stream, c, n_tiles, ray_od, ray_t, ray_cnt,
                       tile_off, ts, row_tile, (const int32_t*)nullptr, ts_provisional, row_capacity > 0 ? row_capacity : INT64_MAX);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_ngp_render_write_layers(int64_t n_tiles, const uint8_t* bitfield, int32_t cascades, float scale, float esf, int32_t grid_size,
                                int32_t max_samples, const float* ray_od, const float* ray_t, const int32_t* ray_cnt, const int32_t* tile_rows,
                                const int32_t* tile_off, float* ts, int32_t* row_tile, int32_t* layer_off, int32_t* row_of,
                                const float* ts_provisional, int32_t* row_k, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_tiles < 0 || cascades < 1 || grid_size < 1 || max_samples < 1 || max_samples > 1024) return NRC_ERR_INVALID;
    if (n_tiles == 0) return NRC_OK;
    if (!bitfield || !ray_od || !ray_t || !ray_cnt || !tile_rows || !tile_off || !row_tile || !layer_off || !row_of) return NRC_ERR_INVALID;
    if (!ts && !(ts_provisional && row_k)) return NRC_ERR_INVALID;   // ts == NULL: arena queried in place, the rows' tiles and sample numbers only
    hipStream_t s = (hipStream_t)stream;
    const MarchCfg c = make_cfg(bitfield, cascades, scale, esf, grid_size, max_samples, (float)cascades);
    const int n_slabs = (max_samples + NRC_SLAB_G - 1) / NRC_SLAB_G;
    hipLaunchKernelGGL(k_slab_totals, dim3(n_slabs), dim3(256), 0, s, tile_rows, n_tiles, layer_off);
    hipLaunchKernelGGL(k_slab_offsets, dim3(1), dim3(64), 0, s, n_slabs, layer_off);
    hipLaunchKernelGGL(k_slab_rows, dim3(n_slabs), dim3(256), 0, s, tile_rows, tile_off, n_tiles, (const int32_t*)layer_off, row_of);
    hipLaunchKernelGGL(k_render_write, dim3(nrc_cdiv(n_tiles, 4)), dim3(256), 0, s, c, n_tiles, ray_od, ray_t, ray_cnt, tile_off, ts, row_tile,
                       (const int32_t*)row_of, ts_provisional, INT64_MAX, row_k);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_generate_rays(int32_t width, int32_t height, const double* intr, const double* c2w, float* origin, float* direction,
                      float* view_direction, nrc_stream_t stream) {
    NRC_ENTER();
    if (width < 1 || height < 1 || !intr || !c2w) return NRC_ERR_INVALID;
    RayGenCfg g;
    g.width = width; g.height = height;
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3];
    // Perspective.py:73-79 (python doubles, rounded to f32 when torch.linspace materialises its scalars)
    g.min_x = (float)((0.5 - cx) / fx); g.max_x = (float)((width - 1 + 0.5 - cx) / fx);
    g.min_y = (float)((0.5 - cy) / fy); g.max_y = (float)((height - 1 + 0.5 - cy) / fy);
    g.step_x = width > 1 ? (g.max_x - g.min_x) / (float)(width - 1) : 0.0f;
    g.step_y = height > 1 ? (g.max_y - g.min_y) / (float)(height - 1) : 0.0f;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) g.R[3 * r + c] = (float)c2w[4 * r + c];
        g.pos[r] = (float)c2w[4 * r + 3];
    }
    const int64_t n = (int64_t)width * height;
    hipLaunchKernelGGL(k_generate_rays, dim3(nrc_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, g, origin, direction, view_direction);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

}  // extern "C"
