// gs_densify.hip -- 3DGS densification bookkeeping on the device.
//
// What the reference does with a dozen boolean-mask copies, torch.cat calls and host syncs per parameter group
// (src/Methods/GaussianSplatting/Model.py:157-246 driving src/Optim/adam_utils.py:21-61) is here ONE plan and ONE gather:
//   nrc_gs_densify_stats          Model.py:243-246  accum[visible] += |grad_xy|, n_observations[visible] += 1
//   nrc_gs_densify_plan           Model.py:189-241  classify every Gaussian (duplicate / split / prune), prefix-scan, emit the row list of
//                                                   the array AFTER densify_and_prune in the reference's order:
//                                                   [kept originals][kept clones][kept split children, copy 0][copy 1]
//   nrc_gather_rows               adam_utils.py:21-61,81-98  every parameter tensor and both Adam moments through that row list in one launch
//                                                   (new rows get zero moments, like extend_param_groups' torch.zeros_like)
//   nrc_gs_densify_split_children Model.py:199-205  positions / log-scales of the split children from the host's standard-normal draws
//   nrc_compact_mask              adam_utils.py:30  mask -> ascending index list (prune_param_groups, bake_activations' 1/255 prune)
// The host reads the five counters once (it needs the sizes to allocate); nothing else leaves the device.
// HBM streaming + a 1024-wide block scan; no LDS tricks needed: this runs every 100 iterations.
#include <hip/hip_runtime.h>

#include "common.h"

namespace {

constexpr int DP_THREADS = 256;
constexpr int DP_ITEMS = 4;
constexpr int DP_BLOCK = DP_THREADS * DP_ITEMS;  // Gaussians per workgroup
constexpr int GATHER_MAX = 24;

enum : unsigned { F_KEEP = 1u, F_DUP = 2u, F_SPLIT = 4u, F_CHILD = 8u };

struct DensifyIn {
    const float* grad_accum;
    const int* n_obs;
    const float* log_scales;
    const float* opacity_logits;
    float grad_threshold, dense_extent, min_opacity, max_scale;
};

struct DensifyFlags {
    DensifyIn in;
    __device__ unsigned operator()(int64_t i) const {
        const float l0 = in.log_scales[3 * i], l1 = in.log_scales[3 * i + 1], l2 = in.log_scales[3 * i + 2];
        const float s0 = expf(l0), s1 = expf(l1), s2 = expf(l2);
        const float smax = fmaxf(fmaxf(s0, s1), s2);
        const float opacity = 1.0f / (1.0f + expf(-in.opacity_logits[i]));
        const bool large_rule = in.max_scale > 0.f;
        const bool faint = opacity < in.min_opacity;
        bool dup = false, split = false;
        if (in.grad_accum) {
            const int n = in.n_obs[i];
            const float g = in.grad_accum[i] / (float)(n < 1 ? 1 : n);
            const bool hot = g >= in.grad_threshold;
            dup = hot && smax <= in.dense_extent;
            split = hot && smax > in.dense_extent;
        }
        const bool prune_orig = faint || (large_rule && smax > in.max_scale);
        unsigned f = 0;
        if (!split && !prune_orig) f |= F_KEEP;
        if (dup && !prune_orig) f |= F_DUP;
        if (split) {
            f |= F_SPLIT;
            // the children carry log(s / 1.6); the reference's final prune looks at exp() of that (Model.py:202, 236-238)
            const float c0 = expf(logf(s0 / 1.6f)), c1 = expf(logf(s1 / 1.6f)), c2 = expf(logf(s2 / 1.6f));
            const float cmax = fmaxf(fmaxf(c0, c1), c2);
            if (!(faint || (large_rule && cmax > in.max_scale))) f |= F_CHILD;
        }
        return f;
    }
};

struct MaskFlags {
    const uint8_t* mask;
    __device__ unsigned operator()(int64_t i) const { return mask[i] ? F_KEEP : 0u; }
};

// four 16-bit counters in two words: lo = keep | dup << 16, hi = split | child << 16 (a block holds 1024 Gaussians: no carry between fields)
struct Cnt {
    unsigned lo, hi;
};
__device__ __forceinline__ Cnt cnt_of(unsigned f) { return {(f & 1u) | ((f >> 1 & 1u) << 16), (f >> 2 & 1u) | ((f >> 3 & 1u) << 16)}; }
__device__ __forceinline__ Cnt operator+(Cnt a, Cnt b) { return {a.lo + b.lo, a.hi + b.hi}; }

// exclusive scan of one Cnt per thread over the 256-thread block; returns the exclusive prefix, *total = block sum
__device__ __forceinline__ Cnt block_excl_scan(Cnt v, Cnt* total) {
    __shared__ Cnt wave_tot[DP_THREADS / NRC_WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    Cnt inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned ol = __shfl_up(inc.lo, d, 64), oh = __shfl_up(inc.hi, d, 64);
        if (lane >= d) { inc.lo += ol; inc.hi += oh; }
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    Cnt base = {0, 0}, all = {0, 0};
#pragma unroll
    for (int w = 0; w < DP_THREADS / NRC_WAVE; w++) {
        if (w < wave) base = base + wave_tot[w];
        all = all + wave_tot[w];
    }
    *total = all;
    __syncthreads();
    return {base.lo + inc.lo - v.lo, base.hi + inc.hi - v.hi};
}

template <class F>
__global__ void __launch_bounds__(DP_THREADS) k_plan_count(F flags, int64_t P, int* __restrict__ blk) {
    const int64_t i0 = (int64_t)blockIdx.x * DP_BLOCK + threadIdx.x * DP_ITEMS;
    Cnt c = {0, 0};
#pragma unroll
    for (int k = 0; k < DP_ITEMS; k++)
        if (i0 + k < P) c = c + cnt_of(flags(i0 + k));
    Cnt tot;
    block_excl_scan(c, &tot);
    if (threadIdx.x == 0) {
        blk[4 * blockIdx.x + 0] = tot.lo & 0xffff;
        blk[4 * blockIdx.x + 1] = tot.lo >> 16;
        blk[4 * blockIdx.x + 2] = tot.hi & 0xffff;
        blk[4 * blockIdx.x + 3] = tot.hi >> 16;
    }
}

// one workgroup: per-block totals -> exclusive bases (in place) and the five counters
__global__ void __launch_bounds__(1024) k_plan_scan(int* __restrict__ blk, int nblk, int* __restrict__ counts) {
    __shared__ int s[4][1024];
    __shared__ int carry[4];
    if (threadIdx.x < 4) carry[threadIdx.x] = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblk; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        int v[4];
#pragma unroll
        for (int f = 0; f < 4; f++) { v[f] = b < nblk ? blk[4 * b + f] : 0; s[f][threadIdx.x] = v[f]; }
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            int o[4];
#pragma unroll
            for (int f = 0; f < 4; f++) o[f] = threadIdx.x >= d ? s[f][threadIdx.x - d] : 0;
            __syncthreads();
#pragma unroll
            for (int f = 0; f < 4; f++) s[f][threadIdx.x] += o[f];
            __syncthreads();
        }
        if (b < nblk) {
#pragma unroll
            for (int f = 0; f < 4; f++) blk[4 * b + f] = carry[f] + s[f][threadIdx.x] - v[f];
        }
        __syncthreads();
        if (threadIdx.x < 4) carry[threadIdx.x] += s[threadIdx.x][1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int keep = carry[0], dup = carry[1], split = carry[2], child = carry[3];
        counts[0] = keep + dup + 2 * child;  // rows after densify_and_prune
        counts[1] = keep;
        counts[2] = dup;
        counts[3] = child;  // per copy
        counts[4] = split;  // the host draws 2 * split standard-normal rows
    }
}

template <class F>
__global__ void __launch_bounds__(DP_THREADS) k_plan_emit(F flags, int64_t P, const int* __restrict__ blk, const int* __restrict__ counts,
                                                          int* __restrict__ src, int* __restrict__ kind, int* __restrict__ aux) {
    const int64_t i0 = (int64_t)blockIdx.x * DP_BLOCK + threadIdx.x * DP_ITEMS;
    unsigned f[DP_ITEMS];
    Cnt c = {0, 0};
#pragma unroll
    for (int k = 0; k < DP_ITEMS; k++) {
        f[k] = i0 + k < P ? flags(i0 + k) : 0u;
        c = c + cnt_of(f[k]);
    }
    Cnt tot;
    Cnt ex = block_excl_scan(c, &tot);
    int keep = blk[4 * blockIdx.x + 0] + (int)(ex.lo & 0xffff);
    int dup = blk[4 * blockIdx.x + 1] + (int)(ex.lo >> 16);
    int split = blk[4 * blockIdx.x + 2] + (int)(ex.hi & 0xffff);
    int child = blk[4 * blockIdx.x + 3] + (int)(ex.hi >> 16);
    const int n_keep = counts[1], n_dup = counts[2], n_child = counts[3], n_split = counts[4];
#pragma unroll
    for (int k = 0; k < DP_ITEMS; k++) {
        const int i = (int)(i0 + k);
        if (f[k] & F_KEEP) {
            src[keep] = i;
            if (kind) { kind[keep] = 0; aux[keep] = -1; }
            keep++;
        }
        if (f[k] & F_DUP) {
            const int o = n_keep + dup++;
            src[o] = i; kind[o] = 1; aux[o] = -1;
        }
        if (f[k] & F_CHILD) {
#pragma unroll
            for (int copy = 0; copy < 2; copy++) {
                const int o = n_keep + n_dup + copy * n_child + child;
                src[o] = i; kind[o] = 2; aux[o] = copy * n_split + split;  // row of the (2 * n_split, 3) noise tensor (Model.py:196-198)
            }
            child++;
        }
        if (f[k] & F_SPLIT) split++;
    }
}

__global__ void __launch_bounds__(256) k_densify_stats(const float* __restrict__ grad, int ld, const int* __restrict__ radii, int64_t P,
                                                       float* __restrict__ accum, int* __restrict__ n_obs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P || radii[i] <= 0) return;
    const float gx = grad[i * ld], gy = grad[i * ld + 1];
    accum[i] += sqrtf(gx * gx + gy * gy);
    n_obs[i] += 1;
}

struct GatherSet {
    const float* in[GATHER_MAX];
    float* out[GATHER_MAX];
    int row[GATHER_MAX];
    int zero_new[GATHER_MAX];
};

__global__ void __launch_bounds__(256) k_gather_rows(GatherSet s, const int* __restrict__ src, const int* __restrict__ kind, int64_t n_out) {
    const int t = blockIdx.y;
    const int row = s.row[t];
    const bool zero_new = s.zero_new[t] != 0 && kind != nullptr;
    const float* __restrict__ in = s.in[t];
    float* __restrict__ out = s.out[t];
    const int64_t total = n_out * row;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / row;
        const int c = (int)(e - r * row);
        float v = 0.f;
        if (!(zero_new && kind[r] != 0)) v = in[(int64_t)src[r] * row + c];
        out[e] = v;
    }
}

// the inverse of k_gather_rows for a list of DISTINCT rows: out[t][dst[r], :] = in[t][r, :] (unpacking the reduced union rows of a view-parallel step)
__global__ void __launch_bounds__(256) k_scatter_rows(GatherSet s, const int* __restrict__ dst, int64_t n_in) {
    const int t = blockIdx.y;
    const int row = s.row[t];
    const float* __restrict__ in = s.in[t];
    float* __restrict__ out = s.out[t];
    const int64_t total = n_in * row;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t r = e / row;
        const int c = (int)(e - r * row);
        out[(int64_t)dst[r] * row + c] = in[e];
    }
}

__global__ void __launch_bounds__(256) k_split_children(const int* __restrict__ src, const int* __restrict__ kind, const int* __restrict__ aux, int64_t n_out,
                                                        const float* __restrict__ positions, const float* __restrict__ log_scales,
                                                        const float* __restrict__ rotations, const float* __restrict__ noise,
                                                        float* __restrict__ positions_out, float* __restrict__ log_scales_out) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= n_out || kind[o] != 2) return;
    const int64_t i = src[o], a = aux[o];
    const float s0 = expf(log_scales[3 * i]), s1 = expf(log_scales[3 * i + 1]), s2 = expf(log_scales[3 * i + 2]);
    const float v0 = noise[3 * a] * s0, v1 = noise[3 * a + 1] * s1, v2 = noise[3 * a + 2] * s2;  // torch.normal(0, std) = z * std
    // quaternion_to_rotation_matrix with normalisation (src/Cameras/utils.py:180-208), real part first
    float qr = rotations[4 * i], qi = rotations[4 * i + 1], qj = rotations[4 * i + 2], qk = rotations[4 * i + 3];
    const float norm = fmaxf(sqrtf(qr * qr + qi * qi + qj * qj + qk * qk), 1e-12f);  // torch.nn.functional.normalize eps
    qr /= norm; qi /= norm; qj /= norm; qk /= norm;
    const float ii2 = qi * qi * 2.f, jj2 = qj * qj * 2.f, kk2 = qk * qk * 2.f;
    const float ij2 = qi * qj * 2.f, ik2 = qi * qk * 2.f, jk2 = qj * qk * 2.f;
    const float ri2 = qr * qi * 2.f, rj2 = qr * qj * 2.f, rk2 = qr * qk * 2.f;
    const float x = (1.f - (jj2 + kk2)) * v0 + (ij2 - rk2) * v1 + (ik2 + rj2) * v2;
    const float y = (ij2 + rk2) * v0 + (1.f - (ii2 + kk2)) * v1 + (jk2 - ri2) * v2;
    const float z = (ik2 - rj2) * v0 + (jk2 + ri2) * v1 + (1.f - (ii2 + jj2)) * v2;
    positions_out[3 * o] = x + positions[3 * i];
    positions_out[3 * o + 1] = y + positions[3 * i + 1];
    positions_out[3 * o + 2] = z + positions[3 * i + 2];
    log_scales_out[3 * o] = logf(s0 / 1.6f);
    log_scales_out[3 * o + 1] = logf(s1 / 1.6f);
    log_scales_out[3 * o + 2] = logf(s2 / 1.6f);
}

template <class F>
int run_plan(F flags, int64_t P, int* src, int* kind, int* aux, int* counts, void* ws, hipStream_t st) {
    int* blk = reinterpret_cast<int*>(ws);
    const int nblk = (int)nrc_cdiv(P, DP_BLOCK);
    if (nblk > 0) hipLaunchKernelGGL(k_plan_count<F>, dim3(nblk), dim3(DP_THREADS), 0, st, flags, P, blk);
    hipLaunchKernelGGL(k_plan_scan, dim3(1), dim3(1024), 0, st, blk, nblk, counts);
    if (nblk > 0) hipLaunchKernelGGL(k_plan_emit<F>, dim3(nblk), dim3(DP_THREADS), 0, st, flags, P, blk, counts, src, kind, aux);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

}  // namespace

extern "C" {

int nrc_gs_densify_stats(const float* viewspace_grad, int32_t ld, const int32_t* radii, int64_t P, float* grad_accum, int32_t* n_observations,
                         nrc_stream_t stream) {
    NRC_ENTER();
    if (P < 0 || ld < 2) return NRC_ERR_INVALID;
    if (P == 0) return NRC_OK;
    if (!viewspace_grad || !radii || !grad_accum || !n_observations) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_densify_stats, dim3((unsigned)nrc_cdiv(P, 256)), dim3(256), 0, (hipStream_t)stream, viewspace_grad, (int)ld, radii, P, grad_accum,
                       n_observations);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

// per-block bases (4 ints each) + room for the five counters of nrc_compact_mask
int64_t nrc_gs_densify_plan_ws_bytes(int64_t P) { return P < 0 ? -1 : (nrc_cdiv(P, DP_BLOCK) + 2) * 4 * (int64_t)sizeof(int); }

int nrc_gs_densify_plan(const float* grad_accum, const int32_t* n_observations, const float* log_scales, const float* opacity_logits, int64_t P,
                        float grad_threshold, float dense_extent, float min_opacity, float max_scale, int32_t* src, int32_t* kind, int32_t* aux,
                        int32_t* counts, void* workspace, nrc_stream_t stream) {
    NRC_ENTER();
    if (P < 0 || P > (int64_t)1 << 30 || !counts || !workspace) return NRC_ERR_INVALID;
    if ((grad_accum == nullptr) != (n_observations == nullptr)) return NRC_ERR_INVALID;
    // a threshold <= 0 would also select the fresh clones for splitting (their padded gradient is 0, Model.py:191-193): not planned here
    if (grad_accum && !(grad_threshold > 0.f)) return NRC_ERR_UNSUPPORTED;
    if (P > 0 && (!log_scales || !opacity_logits || !src || !kind || !aux)) return NRC_ERR_INVALID;
    DensifyFlags f{{grad_accum, n_observations, log_scales, opacity_logits, grad_threshold, dense_extent, min_opacity, max_scale}};
    return run_plan(f, P, src, kind, aux, counts, workspace, (hipStream_t)stream);
}

int64_t nrc_compact_mask_ws_bytes(int64_t n) { return nrc_gs_densify_plan_ws_bytes(n); }

int nrc_compact_mask(const uint8_t* mask, int64_t n, int32_t* indices, int32_t* count, void* workspace, nrc_stream_t stream) {
    NRC_ENTER();
    if (n < 0 || n > (int64_t)1 << 30 || !count || !workspace) return NRC_ERR_INVALID;
    if (n > 0 && (!mask || !indices)) return NRC_ERR_INVALID;
    int* counts = reinterpret_cast<int*>(workspace) + 4 * nrc_cdiv(n, DP_BLOCK);  // five counters behind the per-block bases
    const int rc = run_plan(MaskFlags{mask}, n, indices, nullptr, nullptr, counts, workspace, (hipStream_t)stream);
    if (rc != NRC_OK) return rc;
    if (hipMemcpyAsync(count, counts, sizeof(int), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return NRC_ERR_LAUNCH;
    return NRC_OK;
}

int nrc_gather_rows(const float* const* in, float* const* out, const int32_t* row_floats, const int32_t* zero_new, int32_t n_tensors,
                    const int32_t* src, const int32_t* kind, int64_t n_out, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_tensors < 0 || n_tensors > GATHER_MAX || n_out < 0) return NRC_ERR_INVALID;
    if (n_tensors == 0 || n_out == 0) return NRC_OK;
    if (!in || !out || !row_floats || !src) return NRC_ERR_INVALID;
    GatherSet s{};
    int max_row = 1;
    for (int t = 0; t < n_tensors; t++) {
        if (!in[t] || !out[t] || row_floats[t] < 1) return NRC_ERR_INVALID;
        s.in[t] = in[t]; s.out[t] = out[t]; s.row[t] = row_floats[t]; s.zero_new[t] = zero_new ? zero_new[t] : 0;
        if (row_floats[t] > max_row) max_row = row_floats[t];
    }
    int64_t bx = nrc_cdiv(n_out * max_row, 256 * 4);
    if (bx > 65535 * 16) bx = 65535 * 16;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)bx, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream, s, src, kind, n_out);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_scatter_rows(const float* const* in, float* const* out, const int32_t* row_floats, int32_t n_tensors, const int32_t* dst, int64_t n_in,
                     nrc_stream_t stream) {
    NRC_ENTER();
    if (n_tensors < 0 || n_tensors > GATHER_MAX || n_in < 0) return NRC_ERR_INVALID;
    if (n_tensors == 0 || n_in == 0) return NRC_OK;
    if (!in || !out || !row_floats || !dst) return NRC_ERR_INVALID;
    GatherSet s{};
    int max_row = 1;
    for (int t = 0; t < n_tensors; t++) {
        if (!in[t] || !out[t] || row_floats[t] < 1) return NRC_ERR_INVALID;
        s.in[t] = in[t]; s.out[t] = out[t]; s.row[t] = row_floats[t]; s.zero_new[t] = 0;
        if (row_floats[t] > max_row) max_row = row_floats[t];
    }
    int64_t bx = nrc_cdiv(n_in * max_row, 256 * 4);
    if (bx > 65535 * 16) bx = 65535 * 16;
    if (bx < 1) bx = 1;
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)bx, (unsigned)n_tensors), dim3(256), 0, (hipStream_t)stream, s, dst, n_in);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

int nrc_gs_densify_split_children(const int32_t* src, const int32_t* kind, const int32_t* aux, int64_t n_out, const float* positions,
                                  const float* log_scales, const float* rotations, const float* noise, float* positions_out,
                                  float* log_scales_out, nrc_stream_t stream) {
    NRC_ENTER();
    if (n_out < 0) return NRC_ERR_INVALID;
    if (n_out == 0) return NRC_OK;
    if (!src || !kind || !aux || !positions || !log_scales || !rotations || !noise || !positions_out || !log_scales_out) return NRC_ERR_INVALID;
    hipLaunchKernelGGL(k_split_children, dim3((unsigned)nrc_cdiv(n_out, 256)), dim3(256), 0, (hipStream_t)stream, src, kind, aux, n_out, positions,
                       log_scales, rotations, noise, positions_out, log_scales_out);
    NRC_LAUNCH_CHECK();
    return NRC_OK;
}

}  // extern "C"
