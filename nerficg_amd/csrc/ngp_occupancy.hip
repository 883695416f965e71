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
    if (hipMemsetAsync(scratch, 0, (size_t)n_cells * 4, st) != hipSuccess) return NRC_ERR_LAUNCH;
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

}  // extern "C"
