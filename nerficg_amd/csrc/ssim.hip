// ssim.hip -- SSIM map + gradient w.r.t. the first image (the 3DGS photometric loss term).
// Replaces the external package fused-ssim as the reference uses it: src/Thirdparty/FusedSSIM.py:10-15,
// src/Optim/Losses/DSSIM.py:11-18, src/Methods/GaussianSplatting/Loss.py:14-15.
// 11x11 Gaussian window (sigma 1.5), zero "same" padding, per channel plane; see oracle/ssim_oracle.c for the formulas.
//
// HBM-bound stencil: a 32x32 output tile per workgroup of 256 threads (round 6; 16x16 before); the 42x42 input halo tile of both images is staged in LDS once, the
// horizontal pass of the five moments (x1, x2, x1^2, x2^2, x1 x2) goes LDS -> LDS, the vertical pass LDS -> registers -- and every thread computes FOUR
// neighbouring outputs per pass from one sliding window (14 LDS reads serve 4 x 11 taps), so a pixel costs ~27 LDS reads instead of 91 (the kernels were bound by
// LDS instruction issue, not by their 26-40 MB of traffic: 60 + 44 us per 1297x840x3 frame).  Per output the eleven products are added in the same order as before.  Training
// mode also stores the three partial-derivative maps, so the backward pass is one more separable blur of three maps
// (the window is symmetric: the adjoint of the blur is the blur).  Algorithmic bytes per pixel and channel: forward 8 B read +
// 4 (+12 training) written; backward 24 B read + 4 written.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

#define ST 32          // output tile edge
#define SR 5           // window radius
#define SI (ST + 2 * SR)  // 42: input tile edge
#define SP (SI + 1)    // LDS pitch of the input tiles
#define XP 33          // LDS pitch of the horizontally blurred rows: odd, so that the 4 rows x 8 column groups a half-wave writes in the horizontal pass land in 32 different banks
#define SQ 4           // outputs per thread and pass
#define SLD ((SI * SI + 255) / 256)  // halo elements per thread

__constant__ float SSIM_G[11] = {0.001028380123898387f, 0.0075987582094967365f, 0.036000773310661316f, 0.10936068743467331f,
                                 0.21300552785396576f,  0.26601171493530273f,   0.21300552785396576f,  0.10936068743467331f,
                                 0.036000773310661316f, 0.0075987582094967365f, 0.001028380123898387f};

// LOSS (the fused photometric loss, nrc_photometric_loss_forward): no SSIM map is written; the workgroup's sums of the SSIM values and of |img1 - img2|
// over its pixels go to partial[2 * block], partial[2 * block + 1] (a fixed tree inside the workgroup: the sums do not depend on scheduling)
template <bool TRAIN, bool LOSS = false>
__global__ void __launch_bounds__(256) k_ssim_fwd(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W, float C1, float C2,
                                                  float* __restrict__ ssim_map, float* __restrict__ dm_dmu1, float* __restrict__ dm_dsigma1_sq,
                                                  float* __restrict__ dm_dsigma12, float* __restrict__ partial = nullptr) {
    __shared__ float s1[SI][SP], s2[SI][SP];
    __shared__ float xb[5][SI][XP];
    __shared__ float red[2][4];
    // the plane's base is wave-uniform (scalar registers); inside the plane 32-bit offsets, stepped from load to load: element k + 256 of the (SI x SI) halo
    // is 6 rows and 4 columns further (256 = 6 x 42 + 4), one row more when the column wraps -- no multiply, no 64-bit vector arithmetic per load
    const size_t plane = (size_t)blockIdx.z * H * W;
    img1 += plane; img2 += plane;
    const int x0 = blockIdx.x * ST - SR, y0 = blockIdx.y * ST - SR;
    {   // all of a thread's halo loads are issued before the first LDS store: one HBM latency per tile, not one per loop trip
        float r1[SLD], r2[SLD];
        int r = (int)threadIdx.x / SI, c = (int)threadIdx.x - r * SI;
        int off = (y0 + r) * W + x0 + c;
#pragma unroll
        for (int it = 0; it < SLD; it++) {
            const int y = y0 + r, x = x0 + c;
            const bool in = r < SI && y >= 0 && y < H && x >= 0 && x < W;
            r1[it] = in ? img1[off] : 0.f;
            r2[it] = in ? img2[off] : 0.f;
            r += 256 / SI; c += 256 % SI; off += (256 / SI) * W + 256 % SI;
            if (c >= SI) { c -= SI; r++; off += W - SI; }
        }
        r = (int)threadIdx.x / SI; c = (int)threadIdx.x - r * SI;
#pragma unroll
        for (int it = 0; it < SLD; it++) {
            if (r < SI) { s1[r][c] = r1[it]; s2[r][c] = r2[it]; }
            r += 256 / SI; c += 256 % SI;
            if (c >= SI) { c -= SI; r++; }
        }
    }
    __syncthreads();
    // horizontal pass: item = (input row r, group of SQ output columns); the 10 + SQ inputs of both images are read once
    for (int k = threadIdx.x; k < SI * (ST / SQ); k += 256) {
        const int r = k / (ST / SQ), c0 = (k - r * (ST / SQ)) * SQ;
        float u[10 + SQ], v[10 + SQ];
#pragma unroll
        for (int j = 0; j < 10 + SQ; j++) { u[j] = s1[r][c0 + j]; v[j] = s2[r][c0 + j]; }
#pragma unroll
        for (int i = 0; i < SQ; i++) {
            float a = 0.f, b = 0.f, aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
            for (int t = 0; t < 11; t++) {
                const float g = SSIM_G[t], uu = u[i + t], vv = v[i + t];
                a += g * uu; b += g * vv; aa += g * uu * uu; bb += g * vv * vv; ab += g * uu * vv;
            }
            xb[0][r][c0 + i] = a; xb[1][r][c0 + i] = b; xb[2][r][c0 + i] = aa; xb[3][r][c0 + i] = bb; xb[4][r][c0 + i] = ab;
        }
    }
    __syncthreads();
    // vertical pass: thread = (column tx, group of SQ output rows)
    const int tx = threadIdx.x & 31, ty0 = (threadIdx.x >> 5) * SQ;
    const int x = blockIdx.x * ST + tx;
    float m[SQ][5];
#pragma unroll
    for (int i = 0; i < SQ; i++)
#pragma unroll
        for (int q = 0; q < 5; q++) m[i][q] = 0.f;
#pragma unroll
    for (int q = 0; q < 5; q++) {
        float col[10 + SQ];
#pragma unroll
        for (int j = 0; j < 10 + SQ; j++) col[j] = xb[q][ty0 + j][tx];
#pragma unroll
        for (int i = 0; i < SQ; i++)
#pragma unroll
            for (int t = 0; t < 11; t++) m[i][q] += SSIM_G[t] * col[i + t];
    }
    float sum_ssim = 0.f, sum_l1 = 0.f;
#pragma unroll
    for (int i = 0; i < SQ; i++) {
        const int y = blockIdx.y * ST + ty0 + i;
        const bool inside = x < W && y < H;
        const float mu1 = m[i][0], mu2 = m[i][1];
        const float sg1 = m[i][2] - mu1 * mu1, sg2 = m[i][3] - mu2 * mu2, sg12 = m[i][4] - mu1 * mu2;
        const float A = mu1 * mu1 + mu2 * mu2 + C1, B = sg1 + sg2 + C2, C = 2.f * mu1 * mu2 + C1, D = 2.f * sg12 + C2;
        const float iAB = 1.f / (A * B);
        const float ssim = C * D * iAB;
        if (inside) {
            const size_t o = plane + (size_t)(y * W + x);
            if (!LOSS) ssim_map[o] = ssim;
            if (TRAIN) {
                dm_dmu1[o] = (mu2 * 2.f * D) * iAB - (mu2 * 2.f * C) * iAB - (mu1 * 2.f * C * D) * iAB / A + (mu1 * 2.f * C * D) * iAB / B;
                dm_dsigma1_sq[o] = (-C * D) * iAB / B;
                dm_dsigma12[o] = (2.f * C) * iAB;
            }
            if (LOSS) { sum_ssim += ssim; sum_l1 += fabsf(s1[ty0 + i + SR][tx + SR] - s2[ty0 + i + SR][tx + SR]); }
        }
    }
    if (LOSS) {
        float a = sum_ssim, b = sum_l1;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            partial[2 * blk] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
            partial[2 * blk + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        }
    }
}
// the loss value from the workgroups' partial sums: one workgroup, double accumulators, a fixed order.  out[0] = lambda_l1 * mean|a - b| +
// lambda_dssim * (1 - mean SSIM), out[1] = mean|a - b|, out[2] = mean SSIM
__global__ void __launch_bounds__(1024) k_photo_reduce(const float* __restrict__ partial, int64_t n_blocks, double inv_n, float lambda_l1, float lambda_dssim,
                                                       float* __restrict__ out) {
    __shared__ double red[2][16];
    double a = 0.0, b = 0.0;
    for (int64_t k0 = threadIdx.x; k0 < n_blocks; k0 += 4 * 1024) {   // four independent loads per turn: the launch is one latency chain otherwise
        float2 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int64_t k = k0 + (int64_t)u * 1024;
            v[u] = k < n_blocks ? reinterpret_cast<const float2*>(partial)[k] : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) { a += (double)v[u].x; b += (double)v[u].y; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sa = 0.0, sb = 0.0;
        for (int w = 0; w < 16; w++) { sa += red[0][w]; sb += red[1][w]; }
        const double ssim = sa * inv_n, l1 = sb * inv_n;
        out[0] = (float)((double)lambda_l1 * l1 + (double)lambda_dssim * (1.0 - ssim));
        out[1] = (float)l1; out[2] = (float)ssim;
    }
}
// LOSS (nrc_photometric_loss_backward): dL/dmap is the same number for every pixel, -lambda_dssim * g / n with the upstream gradient g of the loss
// value read from the device (no map is read), and lambda_l1 * g / n * sign(img1 - img2) is added to the result
template <bool LOSS = false>
__global__ void __launch_bounds__(256) k_ssim_bwd(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W,
                                                  const float* __restrict__ dL_dmap, const float* __restrict__ dm_dmu1,
                                                  const float* __restrict__ dm_dsigma1_sq, const float* __restrict__ dm_dsigma12,
                                                  float* __restrict__ dL_dimg1, const float* __restrict__ upstream = nullptr, float c_ssim = 0.f,
                                                  float c_l1 = 0.f) {
    __shared__ float p[3][SI][SP];
    const float g_up = LOSS ? (upstream ? upstream[0] : 1.f) : 0.f;
    __shared__ float xb[3][SI][XP];
    const size_t plane = (size_t)blockIdx.z * H * W;   // wave-uniform base, 32-bit offsets inside the plane stepped from load to load (see k_ssim_fwd)
    img1 += plane; img2 += plane; dm_dmu1 += plane; dm_dsigma1_sq += plane; dm_dsigma12 += plane; dL_dimg1 += plane;
    if (!LOSS) dL_dmap += plane;
    const int x0 = blockIdx.x * ST - SR, y0 = blockIdx.y * ST - SR;
    {
        float r0[SLD], r1[SLD], r2[SLD], dl[SLD];
        int r = (int)threadIdx.x / SI, c = (int)threadIdx.x - r * SI;
        int off = (y0 + r) * W + x0 + c;
#pragma unroll
        for (int it = 0; it < SLD; it++) {
            const int y = y0 + r, x = x0 + c;
            const bool in = r < SI && y >= 0 && y < H && x >= 0 && x < W;
            dl[it] = in ? (LOSS ? c_ssim * g_up : dL_dmap[off]) : 0.f;
            r0[it] = in ? dm_dmu1[off] : 0.f;
            r1[it] = in ? dm_dsigma1_sq[off] : 0.f;
            r2[it] = in ? dm_dsigma12[off] : 0.f;
            r += 256 / SI; c += 256 % SI; off += (256 / SI) * W + 256 % SI;
            if (c >= SI) { c -= SI; r++; off += W - SI; }
        }
        r = (int)threadIdx.x / SI; c = (int)threadIdx.x - r * SI;
#pragma unroll
        for (int it = 0; it < SLD; it++) {
            if (r < SI) { p[0][r][c] = dl[it] * r0[it]; p[1][r][c] = dl[it] * r1[it]; p[2][r][c] = dl[it] * r2[it]; }
            r += 256 / SI; c += 256 % SI;
            if (c >= SI) { c -= SI; r++; }
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < SI * (ST / SQ); k += 256) {
        const int r = k / (ST / SQ), c0 = (k - r * (ST / SQ)) * SQ;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            float u[10 + SQ];
#pragma unroll
            for (int j = 0; j < 10 + SQ; j++) u[j] = p[q][r][c0 + j];
#pragma unroll
            for (int i = 0; i < SQ; i++) {
                float a = 0.f;
#pragma unroll
                for (int t = 0; t < 11; t++) a += SSIM_G[t] * u[i + t];
                xb[q][r][c0 + i] = a;
            }
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 31, ty0 = (threadIdx.x >> 5) * SQ;
    const int x = blockIdx.x * ST + tx;
    float b[SQ][3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        float col[10 + SQ];
#pragma unroll
        for (int j = 0; j < 10 + SQ; j++) col[j] = xb[q][ty0 + j][tx];
#pragma unroll
        for (int i = 0; i < SQ; i++) {
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < 11; t++) a += SSIM_G[t] * col[i + t];
            b[i][q] = a;
        }
    }
    if (x >= W) return;
#pragma unroll
    for (int i = 0; i < SQ; i++) {
        const int y = blockIdx.y * ST + ty0 + i;
        if (y >= H) break;
        const int o = y * W + x;
        const float u = img1[o], v = img2[o];
        float r = b[i][0] + 2.f * u * b[i][1] + v * b[i][2];
        if (LOSS) { const float d = u - v; r += c_l1 * g_up * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)); }
        dL_dimg1[o] = r;
    }
}

}  // namespace

extern "C" {

int nrc_ssim_forward(const float* img1, const float* img2, int64_t planes, int32_t H, int32_t W, float C1, float C2, float* ssim_map,
                     float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, nrc_stream_t stream) {
    NRC_ENTER();
    if (planes < 0 || H < 1 || W < 1 || planes > 65535 || (int64_t)H * W > 0x7fffffff) return NRC_ERR_INVALID;
    if (planes == 0) return NRC_OK;
    if (!img1 || !img2 || !ssim_map) return NRC_ERR_INVALID;
    const bool train = dm_dmu1 || dm_dsigma1_sq || dm_dsigma12;
    if (train && !(dm_dmu1 && dm_dsigma1_sq && dm_dsigma12)) return NRC_ERR_INVALID;
    const dim3 grid((W + ST - 1) / ST, (H + ST - 1) / ST, (unsigned)planes);
    if (train)
        hipLaunchKernelGGL(k_ssim_fwd<true>, grid, dim3(256), 0, (hipStream_t)stream, img1, img2, (int)H, (int)W, C1, C2, ssim_map, dm_dmu1, dm_dsigma1_sq,
                           dm_dsigma12);
    else
        hipLaunchKernelGGL(k_ssim_fwd<false>, grid, dim3(256), 0, (hipStream_t)stream, img1, img2, (int)H, (int)W, C1, C2, ssim_map, dm_dmu1, dm_dsigma1_sq,
                           dm_dsigma12);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_ssim_backward(const float* img1, const float* img2, int64_t planes, int32_t H, int32_t W, const float* dL_dmap, const float* dm_dmu1,
                      const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1, nrc_stream_t stream) {
    NRC_ENTER();
    if (planes < 0 || H < 1 || W < 1 || planes > 65535 || (int64_t)H * W > 0x7fffffff) return NRC_ERR_INVALID;
    if (planes == 0) return NRC_OK;
    if (!img1 || !img2 || !dL_dmap || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimg1) return NRC_ERR_INVALID;
    const dim3 grid((W + ST - 1) / ST, (H + ST - 1) / ST, (unsigned)planes);
    hipLaunchKernelGGL(k_ssim_bwd<false>, grid, dim3(256), 0, (hipStream_t)stream, img1, img2, (int)H, (int)W, dL_dmap, dm_dmu1, dm_dsigma1_sq, dm_dsigma12,
                       dL_dimg1, (const float*)nullptr, 0.f, 0.f);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int64_t nrc_photometric_loss_ws_floats(int64_t planes, int32_t H, int32_t W) {
    if (planes < 0 || H < 1 || W < 1) return NRC_ERR_INVALID;
    return 2 * planes * (int64_t)((W + ST - 1) / ST) * ((H + ST - 1) / ST) + 4;
}
int nrc_photometric_loss_forward(const float* image, const float* target, int64_t planes, int32_t H, int32_t W, float C1, float C2, float lambda_l1,
                                 float lambda_dssim, float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12, float* workspace, float* loss3,
                                 nrc_stream_t stream) {
    NRC_ENTER();
    if (planes < 1 || H < 1 || W < 1 || planes > 65535 || (int64_t)H * W > 0x7fffffff || !image || !target || !workspace || !loss3) return NRC_ERR_INVALID;
    const bool train = dm_dmu1 || dm_dsigma1_sq || dm_dsigma12;
    if (train && !(dm_dmu1 && dm_dsigma1_sq && dm_dsigma12)) return NRC_ERR_INVALID;
    const dim3 grid((W + ST - 1) / ST, (H + ST - 1) / ST, (unsigned)planes);
    hipStream_t s = (hipStream_t)stream;
    if (train)
        hipLaunchKernelGGL((k_ssim_fwd<true, true>), grid, dim3(256), 0, s, image, target, (int)H, (int)W, C1, C2, (float*)nullptr, dm_dmu1, dm_dsigma1_sq,
                           dm_dsigma12, workspace);
    else
        hipLaunchKernelGGL((k_ssim_fwd<false, true>), grid, dim3(256), 0, s, image, target, (int)H, (int)W, C1, C2, (float*)nullptr, dm_dmu1, dm_dsigma1_sq,
                           dm_dsigma12, workspace);
    const int64_t n_blocks = (int64_t)grid.x * grid.y * grid.z;
    hipLaunchKernelGGL(k_photo_reduce, dim3(1), dim3(1024), 0, s, (const float*)workspace, n_blocks, 1.0 / ((double)planes * H * W), lambda_l1, lambda_dssim, loss3);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}
int nrc_photometric_loss_backward(const float* image, const float* target, int64_t planes, int32_t H, int32_t W, float lambda_l1, float lambda_dssim,
                                  const float* upstream_dev, const float* dm_dmu1, const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimage,
                                  nrc_stream_t stream) {
    NRC_ENTER();
    if (planes < 1 || H < 1 || W < 1 || planes > 65535 || (int64_t)H * W > 0x7fffffff) return NRC_ERR_INVALID;
    if (!image || !target || !dm_dmu1 || !dm_dsigma1_sq || !dm_dsigma12 || !dL_dimage) return NRC_ERR_INVALID;
    const dim3 grid((W + ST - 1) / ST, (H + ST - 1) / ST, (unsigned)planes);
    const double inv_n = 1.0 / ((double)planes * H * W);
    hipLaunchKernelGGL(k_ssim_bwd<true>, grid, dim3(256), 0, (hipStream_t)stream, image, target, (int)H, (int)W, (const float*)nullptr, dm_dmu1, dm_dsigma1_sq,
                       dm_dsigma12, dL_dimage, upstream_dev, (float)(-(double)lambda_dssim * inv_n), (float)((double)lambda_l1 * inv_n));
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

}  // extern "C"
