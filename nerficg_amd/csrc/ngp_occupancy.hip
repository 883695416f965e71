// ngp_occupancy.hip -- occupancy-grid maintenance of InstantNGP fused on the device.
//
// Reference: InstantNGPRenderer.update_occupancy_grid (src/Methods/InstantNGP/Renderer.py:247-272): a zero-filled scratch grid receives
// the freshly queried densities by index_put, `where(grid < 0, grid, max(grid * decay, scratch))` rebuilds the grid, a masked mean is
// pulled to the host (`.item()`), and packbits (raymarching.cu:138-161) thresholds at min(mean, density_threshold).
// Here: scatter (max over duplicate cells: index_put with duplicates keeps an arbitrary one, the maximum is one of them), one EMA pass
// that also produces the positive-cell sum / count as per-block partials, a fixed-order final reduction that leaves the threshold ON THE
// DEVICE, and the bit packing reading it from there -- no host round trip, 5 small launches.  HBM: 4 + 4 + 4 + 4 + 1/8 B per cell.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

constexpr int OCC_THREADS = 256;
constexpr int OCC_CELLS = 8;  // per lane: one bitfield byte

template <typename T>
__global__ void __launch_bounds__(256) k_occ_scatter(const int64_t* __restrict__ cell, const T* __restrict__ density, int64_t per_cascade, int64_t total,
                                                     int64_t cells_per_cascade, int* __restrict__ scratch) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const float d = (float)density[e];
    const int64_t idx = cell[e];
    if (!(d > 0.f) || idx < 0 || idx >= cells_per_cascade) return;  // max(grid * decay, scratch) ignores scratch <= 0 for every cell it updates
    atomicMax(&scratch[(e / per_cascade) * cells_per_cascade + idx], __float_as_int(d));  // positive floats order like their bit patterns
}

__global__ void __launch_bounds__(OCC_THREADS) k_occ_ema(float* __restrict__ grid, const float* __restrict__ scratch, int64_t n_cells, float decay,
                                                         float* __restrict__ partial_sum, float* __restrict__ partial_cnt) {
    const int64_t i0 = ((int64_t)blockIdx.x * OCC_THREADS + threadIdx.x) * OCC_CELLS;
    float sum = 0.f, cnt = 0.f;
    if (i0 < n_cells) {  // n_cells is a multiple of 8 (checked by the launcher)
        float g[8], s[8];
        *reinterpret_cast<float4*>(g) = *reinterpret_cast<const float4*>(grid + i0);
        *reinterpret_cast<float4*>(g + 4) = *reinterpret_cast<const float4*>(grid + i0 + 4);
        *reinterpret_cast<float4*>(s) = *reinterpret_cast<const float4*>(scratch + i0);
        *reinterpret_cast<float4*>(s + 4) = *reinterpret_cast<const float4*>(scratch + i0 + 4);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (!(g[k] < 0.f)) g[k] = fmaxf(g[k] * decay, s[k]);  // carved cells (< 0) stay as they are
            if (g[k] > 0.f) { sum += g[k]; cnt += 1.f; }
        }
        *reinterpret_cast<float4*>(grid + i0) = *reinterpret_cast<float4*>(g);
        *reinterpret_cast<float4*>(grid + i0 + 4) = *reinterpret_cast<float4*>(g + 4);
    }
    __shared__ float ssum[OCC_THREADS / 64], scnt[OCC_THREADS / 64];
    sum = nrc_group_sum<64>(sum);
    cnt = nrc_group_sum<64>(cnt);
    if ((threadIdx.x & 63) == 0) { ssum[threadIdx.x >> 6] = sum; scnt[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial_sum[blockIdx.x] = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
        partial_cnt[blockIdx.x] = (scnt[0] + scnt[1]) + (scnt[2] + scnt[3]);
    }
}

// one workgroup, fixed summation order, f64: threshold[0] = min(mean, density_threshold) the way Python's min() treats a NaN mean (no
// positive cell -> NaN -> every bit 0), threshold[1] = mean
__global__ void __launch_bounds__(256) k_occ_threshold(const float* __restrict__ partial_sum, const float* __restrict__ partial_cnt, int nblk,
                                                       float density_threshold, float* __restrict__ threshold) {
    __shared__ double ssum[256], scnt[256];
    double s = 0.0, c = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 256) { s += (double)partial_sum[b]; c += (double)partial_cnt[b]; }
    ssum[threadIdx.x] = s; scnt[threadIdx.x] = c;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) { ssum[threadIdx.x] += ssum[threadIdx.x + d]; scnt[threadIdx.x] += scnt[threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mean = scnt[0] > 0.0 ? (float)(ssum[0] / scnt[0]) : __int_as_float(0x7fc00000);
        threshold[0] = density_threshold < mean ? density_threshold : mean;
        threshold[1] = mean;
    }
}

__global__ void __launch_bounds__(256) k_occ_pack(const float* __restrict__ grid, int64_t n_bytes, const float* __restrict__ threshold,
                                                  uint8_t* __restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= n_bytes) return;
    const float thr = threshold[0];
    const float4 a = reinterpret_cast<const float4*>(grid)[2 * n];
    const float4 b = reinterpret_cast<const float4*>(grid)[2 * n + 1];
    out[n] = (uint8_t)((a.x > thr) | ((a.y > thr) << 1) | ((a.z > thr) << 2) | ((a.w > thr) << 3) | ((b.x > thr) << 4) | ((b.y > thr) << 5) |
                       ((b.z > thr) << 6) | ((b.w > thr) << 7));
}

// ------------------------------------------------------------------------------------------------ cell sampling for the update
// Which cells get a fresh density this round (Renderer.py:183-206): every cell (warm-up), or n uniformly drawn cells + n cells drawn
// uniformly from the currently occupied ones (grid > threshold) per cascade; and where inside its cell each one is queried
// (Renderer.py:251-258: cell centre lattice (c / (G-1) * 2 - 1) * (s - s/G), jitter U(-1,1) * s/G).  One lane per drawn cell, counter-based
// generator (splitmix64 of seed and draw number) -- the draws are a pure function of *seed, whatever the launch geometry.
__device__ __forceinline__ uint32_t occ_expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu; v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t occ_compact_bits(uint32_t x) {
    x &= 0x49249249u; x = (x | (x >> 2)) & 0xC30C30C3u; x = (x | (x >> 4)) & 0x0F00F00Fu; x = (x | (x >> 8)) & 0xFF0000FFu; x = (x | (x >> 16)) & 0x0000FFFFu;
    return x;
}
__device__ __forceinline__ uint64_t occ_mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float occ_unit(uint32_t bits24) { return (float)(bits24 & 0xFFFFFFu) * (1.0f / 16777216.0f); }  // [0, 1)

__global__ void __launch_bounds__(256) k_occ_draw(int cascades, int G, float scale, int mode, int64_t n_half, int64_t per_cascade,
                                                  const int64_t* __restrict__ seed, const int32_t* __restrict__ occ_idx, const int32_t* __restrict__ occ_cnt,
                                                  int64_t occ_stride, int64_t* __restrict__ indices, float* __restrict__ points) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)cascades * per_cascade) return;
    const int c = (int)(e / per_cascade);
    const int64_t k = e - (int64_t)c * per_cascade;
    const uint64_t base = occ_mix((uint64_t)seed[0] ^ ((uint64_t)e * 0xD1342543DE82EF95ull));
    const uint64_t r0 = occ_mix(base), r1 = occ_mix(base + 1), r2 = occ_mix(base + 2);
    uint32_t x, y, z;
    int64_t cell;
    if (mode == 0) {            // all cells, in Morton order
        cell = k;
        x = occ_compact_bits((uint32_t)k); y = occ_compact_bits((uint32_t)k >> 1); z = occ_compact_bits((uint32_t)k >> 2);
    } else if (k < n_half) {    // uniform over the grid
        x = (uint32_t)(((r0 & 0xFFFFFFFFull) * (uint64_t)G) >> 32); y = (uint32_t)(((r0 >> 32) * (uint64_t)G) >> 32);
        z = (uint32_t)(((r1 & 0xFFFFFFFFull) * (uint64_t)G) >> 32);
        cell = (int64_t)(occ_expand_bits(x) | (occ_expand_bits(y) << 1) | (occ_expand_bits(z) << 2));
    } else {                    // uniform over the occupied cells of this cascade (none: an ignored entry)
        const int cnt = occ_cnt[c];
        if (cnt <= 0) { indices[e] = -1; points[3 * e] = points[3 * e + 1] = points[3 * e + 2] = 0.f; return; }
        const uint32_t pick = (uint32_t)(((r1 >> 32) * (uint64_t)cnt) >> 32);
        cell = occ_idx[(int64_t)c * occ_stride + pick];
        x = occ_compact_bits((uint32_t)cell); y = occ_compact_bits((uint32_t)cell >> 1); z = occ_compact_bits((uint32_t)cell >> 2);
    }
    const float s = fminf(scalbnf(1.0f, c - 1), scale);
    const float half_cell = s / (float)G;
    const float inv = 1.0f / (float)(G - 1);
    const float jx = occ_unit((uint32_t)r2) * 2.f - 1.f, jy = occ_unit((uint32_t)(r2 >> 24)) * 2.f - 1.f, jz = occ_unit((uint32_t)(r2 >> 40)) * 2.f - 1.f;
    indices[e] = cell;
    points[3 * e] = ((float)x * inv * 2.f - 1.f) * (s - half_cell) + jx * half_cell;
    points[3 * e + 1] = ((float)y * inv * 2.f - 1.f) * (s - half_cell) + jy * half_cell;
    points[3 * e + 2] = ((float)z * inv * 2.f - 1.f) * (s - half_cell) + jz * half_cell;
}

// ------------------------------------------------------------------------------------------------ carving
// Renderer.py:208-245: a cell survives when a camera sees its centre (optionally: through a pixel of the 3x3-dilated alpha mask); union over
// the views (or intersection, "subtractive"); the survivors are dilated by one cell and everything else is frozen at -1.
struct CarveView {
    float R[9];       // camera-to-world rotation, row-major
    float pos[3];     // camera position
    float fx, fy, cx, cy, width, height, near_plane, far_plane;
};

__global__ void __launch_bounds__(256) k_carve_view(uint8_t* __restrict__ remaining, int cascades, int G, float scale, float cen_x, float cen_y, float cen_z,
                                                    CarveView v, const float* __restrict__ alpha, int alpha_w, int alpha_h, int subtractive) {
    const int64_t cells = (int64_t)G * G * G;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= cascades * cells) return;
    const int c = (int)(e / cells);
    const int64_t k = e - c * cells;
    const int x = (int)(k % G), y = (int)((k / G) % G), z = (int)(k / ((int64_t)G * G));
    const float s = fminf(scalbnf(1.0f, c - 1), scale);
    const float half_cell = s / (float)G, inv = 1.0f / (float)(G - 1);
    const float wx = ((float)x * inv * 2.f - 1.f) * (s - half_cell) + cen_x - v.pos[0];
    const float wy = ((float)y * inv * 2.f - 1.f) * (s - half_cell) + cen_y - v.pos[1];
    const float wz = ((float)z * inv * 2.f - 1.f) * (s - half_cell) + cen_z - v.pos[2];
    // world -> camera: (p - position) @ R  (Datasets/utils.py:1027-1031)
    const float px = wx * v.R[0] + wy * v.R[3] + wz * v.R[6];
    const float py = wx * v.R[1] + wy * v.R[4] + wz * v.R[7];
    const float depth = wx * v.R[2] + wy * v.R[5] + wz * v.R[8];
    const float zc = fmaxf(depth, 1.0e-8f);
    const float sx = px / zc * v.fx + v.cx, sy = py / zc * v.fy + v.cy;
    bool seen = sx >= 0.f && sy >= 0.f && sx < v.width && sy < v.height && depth > v.near_plane && depth < v.far_plane;
    if (seen && alpha) {
        const int ix = (int)floorf(sx), iy = (int)floorf(sy);
        bool lit = false;
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int qx = ix + dx, qy = iy + dy;
                if (qx >= 0 && qy >= 0 && qx < alpha_w && qy < alpha_h) lit |= alpha[(int64_t)qy * alpha_w + qx] > 0.f;
            }
        seen = lit;
    }
    remaining[e] = subtractive ? (uint8_t)(remaining[e] && seen) : (uint8_t)(remaining[e] || seen);
}

__global__ void __launch_bounds__(256) k_carve_finish(const uint8_t* __restrict__ remaining, int cascades, int G, float* __restrict__ grid) {
    const int64_t cells = (int64_t)G * G * G;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= cascades * cells) return;
    const int c = (int)(e / cells);
    const int64_t k = e - c * cells;
    const int x = (int)(k % G), y = (int)((k / G) % G), z = (int)(k / ((int64_t)G * G));
    const uint8_t* r = remaining + c * cells;
    bool keep = false;
    for (int dz = -1; dz <= 1; dz++)
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int qx = x + dx, qy = y + dy, qz = z + dz;
                if (qx >= 0 && qy >= 0 && qz >= 0 && qx < G && qy < G && qz < G) keep |= r[qx + (int64_t)G * (qy + (int64_t)G * qz)] != 0;
            }
    grid[c * cells + (occ_expand_bits((uint32_t)x) | (occ_expand_bits((uint32_t)y) << 1) | (occ_expand_bits((uint32_t)z) << 2))] = keep ? 0.f : -1.f;
}

}  // namespace

extern "C" {

int64_t nrc_occupancy_update_ws_bytes(int64_t n_cells_total) {
    if (n_cells_total < 0) return -1;
    const int64_t nblk = nrc_cdiv(n_cells_total, OCC_THREADS * OCC_CELLS);
    return n_cells_total * 4 + 2 * nblk * 4 + 64;
}

int nrc_occupancy_update(float* grid, const int64_t* cell_indices, const void* densities, int32_t densities_dtype, int32_t cascades,
                         int64_t cells_per_cascade, int64_t samples_per_cascade, float decay, float density_threshold, uint8_t* bitfield,
                         float* threshold_out, void* workspace, nrc_stream_t stream) {
    NRC_ENTER();
    if (cascades < 1 || cells_per_cascade < 8 || (cells_per_cascade & 7) || samples_per_cascade < 0) return NRC_ERR_INVALID;
    if (!grid || !bitfield || !threshold_out || !workspace || (densities_dtype != 0 && densities_dtype != 1)) return NRC_ERR_INVALID;
    if (samples_per_cascade > 0 && (!cell_indices || !densities)) return NRC_ERR_INVALID;
    if ((((uintptr_t)grid | (uintptr_t)workspace) & 15u) != 0) return NRC_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_cells = (int64_t)cascades * cells_per_cascade;
    const int nblk = (int)nrc_cdiv(n_cells, OCC_THREADS * OCC_CELLS);
    float* scratch = reinterpret_cast<float*>(workspace);
    float* partial_sum = scratch + n_cells;
    float* partial_cnt = partial_sum + nblk;
    if (nrc_zero_async(scratch, (size_t)n_cells * 4, st) != hipSuccess) return NRC_ERR_LAUNCH;
    const int64_t total = (int64_t)cascades * samples_per_cascade;
    if (total > 0) {
        if (densities_dtype == 0)
            hipLaunchKernelGGL(k_occ_scatter<float>, dim3((unsigned)nrc_cdiv(total, 256)), dim3(256), 0, st, cell_indices, (const float*)densities,
                               samples_per_cascade, total, cells_per_cascade, reinterpret_cast<int*>(scratch));
        else
            hipLaunchKernelGGL(k_occ_scatter<__half>, dim3((unsigned)nrc_cdiv(total, 256)), dim3(256), 0, st, cell_indices, (const __half*)densities,
                               samples_per_cascade, total, cells_per_cascade, reinterpret_cast<int*>(scratch));
    }
    hipLaunchKernelGGL(k_occ_ema, dim3(nblk), dim3(OCC_THREADS), 0, st, grid, scratch, n_cells, decay, partial_sum, partial_cnt);
    hipLaunchKernelGGL(k_occ_threshold, dim3(1), dim3(256), 0, st, partial_sum, partial_cnt, nblk, density_threshold, threshold_out);
    hipLaunchKernelGGL(k_occ_pack, dim3((unsigned)nrc_cdiv(n_cells / 8, 256)), dim3(256), 0, st, grid, n_cells / 8, threshold_out, bitfield);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_occupancy_draw_cells(int32_t cascades, int32_t grid_size, float scale, int32_t mode, int64_t n_per_half, const int64_t* seed,
                             const int32_t* occupied_indices, const int32_t* occupied_counts, int64_t occupied_stride, int64_t* cell_indices,
                             float* points, nrc_stream_t stream) {
    NRC_ENTER();
    if (cascades < 1 || grid_size < 2 || grid_size > 1024 || (mode != 0 && mode != 1) || !seed || !cell_indices || !points) return NRC_ERR_INVALID;
    if (mode == 1 && (n_per_half < 0 || !occupied_indices || !occupied_counts)) return NRC_ERR_INVALID;
    const int64_t per = mode == 0 ? (int64_t)grid_size * grid_size * grid_size : 2 * n_per_half;
    if (per == 0) return NRC_OK;
    hipLaunchKernelGGL(k_occ_draw, dim3((unsigned)nrc_cdiv(cascades * per, 256)), dim3(256), 0, (hipStream_t)stream, (int)cascades, (int)grid_size, scale,
                       (int)mode, n_per_half, per, seed, occupied_indices, occupied_counts, occupied_stride, cell_indices, points);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_occupancy_carve_view(uint8_t* remaining, int32_t cascades, int32_t grid_size, float scale, const float* center3, const float* c2w_rotation9,
                             const float* position3, float focal_x, float focal_y, float center_x, float center_y, int32_t width, int32_t height,
                             float near_plane, float far_plane, const float* alpha_mask, int32_t subtractive, nrc_stream_t stream) {
    NRC_ENTER();
    if (!remaining || cascades < 1 || grid_size < 2 || grid_size > 1024 || !center3 || !c2w_rotation9 || !position3 || width < 1 || height < 1)
        return NRC_ERR_INVALID;
    CarveView v;
    for (int i = 0; i < 9; i++) v.R[i] = c2w_rotation9[i];
    for (int i = 0; i < 3; i++) v.pos[i] = position3[i];
    v.fx = focal_x; v.fy = focal_y; v.cx = center_x; v.cy = center_y; v.width = (float)width; v.height = (float)height;
    v.near_plane = near_plane; v.far_plane = far_plane;
    const int64_t total = (int64_t)cascades * grid_size * grid_size * grid_size;
    hipLaunchKernelGGL(k_carve_view, dim3((unsigned)nrc_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, remaining, (int)cascades, (int)grid_size, scale,
                       center3[0], center3[1], center3[2], v, alpha_mask, (int)width, (int)height, (int)subtractive);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_occupancy_carve_finish(const uint8_t* remaining, int32_t cascades, int32_t grid_size, float* grid, nrc_stream_t stream) {
    NRC_ENTER();
    if (!remaining || !grid || cascades < 1 || grid_size < 2 || grid_size > 1024) return NRC_ERR_INVALID;
    const int64_t total = (int64_t)cascades * grid_size * grid_size * grid_size;
    hipLaunchKernelGGL(k_carve_finish, dim3((unsigned)nrc_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, remaining, (int)cascades, (int)grid_size, grid);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

}  // extern "C"
