// knn.hip -- mean squared distance to the 3 nearest neighbours of every point (3DGS scale initialisation).
// Replaces simple_knn._C.distCUDA2 as the reference binds it: src/Thirdparty/SimpleKNN.py:17-18
// (compute_mean_squared_knn_distances = _C.distCUDA2), called at src/Optim/knn_utils.py:34-38 -- the reference has an
// scikit-learn fallback, this removes the need for it.  Runs once per training.
//
// Exact 3-NN: the caller passes the points in Morton order (nerficg_amd.MortonEncoding + a sort); consecutive runs of KNN_BOX points
// form boxes with an axis-aligned bound; a point scans the boxes whose bound is closer than its current third-nearest distance,
// starting from its own neighbourhood in the sorted order.  Distances are (dx*dx + dy*dy) + dz*dz in f32 (file compiled with
// -ffp-contract=off): identical to oracle/knn_oracle.c for every point.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

#define KNN_BOX 1024

__global__ void __launch_bounds__(256) k_knn_boxes(const float* __restrict__ p, int64_t n, float* __restrict__ boxes) {
    __shared__ float s[6][256];
    const int64_t b = blockIdx.x;
    float mn[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, mx[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int k = threadIdx.x; k < KNN_BOX; k += 256) {
        const int64_t i = b * KNN_BOX + k;
        if (i < n)
            for (int c = 0; c < 3; c++) { const float v = p[3 * i + c]; mn[c] = fminf(mn[c], v); mx[c] = fmaxf(mx[c], v); }
    }
    for (int c = 0; c < 3; c++) { s[c][threadIdx.x] = mn[c]; s[3 + c][threadIdx.x] = mx[c]; }
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if ((int)threadIdx.x < d)
            for (int c = 0; c < 3; c++) {
                s[c][threadIdx.x] = fminf(s[c][threadIdx.x], s[c][threadIdx.x + d]);
                s[3 + c][threadIdx.x] = fmaxf(s[3 + c][threadIdx.x], s[3 + c][threadIdx.x + d]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) boxes[6 * b + threadIdx.x] = s[threadIdx.x][0];
}

__device__ __forceinline__ float dist2(const float* a, float bx, float by, float bz) {
    const float dx = a[0] - bx, dy = a[1] - by, dz = a[2] - bz;
    return (dx * dx + dy * dy) + dz * dz;
}
__device__ __forceinline__ void push3(float (&best)[3], float d) {
    if (d < best[2]) {
        if (d < best[1]) {
            best[2] = best[1];
            if (d < best[0]) { best[1] = best[0]; best[0] = d; } else best[1] = d;
        } else best[2] = d;
    }
}
__global__ void __launch_bounds__(256) k_knn3(const float* __restrict__ p, int64_t n, const float* __restrict__ boxes, int64_t n_boxes,
                                             float* __restrict__ out_sorted) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float q[3] = {p[3 * i], p[3 * i + 1], p[3 * i + 2]};
    float best[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    for (int64_t j = max((int64_t)0, i - 3); j <= min(n - 1, i + 3); j++)
        if (j != i) push3(best, dist2(q, p[3 * j], p[3 * j + 1], p[3 * j + 2]));
    for (int64_t b = 0; b < n_boxes; b++) {
        const float* bb = boxes + 6 * b;
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float lo = bb[c] - q[c], hi = q[c] - bb[3 + c];
            const float e = fmaxf(0.f, fmaxf(lo, hi));
            d += e * e;
        }
        if (d > best[2]) continue;
        const int64_t j0 = b * KNN_BOX, j1 = min(n, j0 + KNN_BOX);
        for (int64_t j = j0; j < j1; j++) {
            if (j >= i - 3 && j <= i + 3) continue;  // already seen (and never the point itself)
            push3(best, dist2(q, p[3 * j], p[3 * j + 1], p[3 * j + 2]));
        }
    }
    out_sorted[i] = (best[0] + best[1] + best[2]) / 3.0f;
}

}  // namespace

extern "C" {

int64_t nrc_knn3_ws_bytes(int64_t n) {
    if (n < 0) return NRC_ERR_INVALID;
    return nrc_cdiv(n > 0 ? n : 1, KNN_BOX) * 6 * (int64_t)sizeof(float) + 256;
}

int nrc_knn3_mean_sq_dist(const float* points_morton_sorted, int64_t n, float* out_sorted, void* workspace, nrc_stream_t stream) {
    NRC_ENTER();
    if (n < 0 || (n > 0 && n < 4)) return NRC_ERR_INVALID;  // three neighbours besides the point itself
    if (n == 0) return NRC_OK;
    if (!points_morton_sorted || !out_sorted || !workspace) return NRC_ERR_INVALID;
    hipStream_t s = (hipStream_t)stream;
    const int64_t nb = nrc_cdiv(n, KNN_BOX);
    float* boxes = (float*)workspace;
    hipLaunchKernelGGL(k_knn_boxes, dim3((unsigned)nb), dim3(256), 0, s, points_morton_sorted, n, boxes);
    hipLaunchKernelGGL(k_knn3, dim3((unsigned)nrc_cdiv(n, 256)), dim3(256), 0, s, points_morton_sorted, n, (const float*)boxes, nb, out_sorted);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

}  // extern "C"
