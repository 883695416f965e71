// common.h -- shared device helpers for the gfx950 (MI355X, CDNA4) kernels of libnerficg_hip.so.
// wave = 64 lanes everywhere; no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nerficg_hip.h"

#define NRC_WAVE 64

void nrc_set_last_hip_error(int e);  // lib_info.hip

// hipGetLastError() is per-thread and sticky across ALL HIP users of the thread (torch included): clear it on entry so that
// NRC_LAUNCH_CHECK() reports only errors raised by this library's own launches.
#define NRC_ENTER() (void)hipGetLastError()

#define NRC_LAUNCH_CHECK()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) { nrc_set_last_hip_error((int)e__); return NRC_ERR_LAUNCH; } \
    } while (0)

static inline int64_t nrc_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Stage timer (include/nerficg_hip.h group 12): when armed, NRC_STAGE(s, "name") records a HIP event on the launch stream `s` behind the kernel(s)
// just launched; the time between two consecutive marks is the stage `name`.  NRC_STAGE(s, nullptr) opens an entry point (what ran before it
// belongs to nobody).  Disarmed (the default, and always inside a stream capture) it is one load and a predictable branch.
extern int g_nrc_stage_timer_armed;            // lib_info.hip
void nrc_stage_mark(hipStream_t s, const char* name);
#define NRC_STAGE(s, name) do { if (g_nrc_stage_timer_armed) nrc_stage_mark((s), (name)); } while (0)

// Clearing device memory with a kernel of our own instead of hipMemsetAsync.  Outside a capture the two cost the same (ROCm runs a fill
// kernel for the memset); inside a stream capture the memset becomes a graph memset node, and replays of graphs holding such nodes were
// measured to finish AFTER the launch stream considered them done (an event recorded behind the replay fires early; only a device-wide
// synchronize waits for them), so back-to-back replays raced with their own predecessor.  Kernel nodes keep the stream order.
// `p` and `bytes` must be multiples of 4 (every caller clears 32-bit or wider elements).
static __global__ void __launch_bounds__(256) nrc_k_zero_words(uint32_t* __restrict__ p, size_t n_words) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n_vec = ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) ? n_words / 4 : 0;
    if (i < n_vec) reinterpret_cast<uint4*>(p)[i] = make_uint4(0u, 0u, 0u, 0u);
    const size_t tail = n_words - 4 * n_vec;   // at most 3 words when the pointer is 16-byte aligned, everything otherwise
    for (size_t k = i; k < tail; k += (size_t)gridDim.x * 256) p[4 * n_vec + k] = 0u;
}
static inline hipError_t nrc_zero_async(void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    if (!p || (bytes & 3u) || (reinterpret_cast<uintptr_t>(p) & 3u)) return hipErrorInvalidValue;
    const size_t n_words = bytes / 4;
    const bool aligned = (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
    const size_t threads = aligned ? (n_words / 4 > 0 ? n_words / 4 : 1) : (n_words < ((size_t)1 << 22) ? n_words : ((size_t)1 << 22));
    hipLaunchKernelGGL(nrc_k_zero_words, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, (uint32_t*)p, n_words);
    return hipGetLastError();
}

// ---- wave-level primitives (64-wide) ---------------------------------------------------------
__device__ __forceinline__ int nrc_lane() { return __lane_id(); }

// inclusive scans over the lanes of a G-lane group (G power of two <= 64), Hillis-Steele with DPP-lowered shuffles
template <int G>
__device__ __forceinline__ float nrc_group_incl_sum(float v, int lane_in_group) {
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
        const float o = __shfl_up(v, d, G);
        if (lane_in_group >= d) v += o;
    }
    return v;
}
template <int G>
__device__ __forceinline__ float nrc_group_incl_prod(float v, int lane_in_group) {
#pragma unroll
    for (int d = 1; d < G; d <<= 1) {
        const float o = __shfl_up(v, d, G);
        if (lane_in_group >= d) v *= o;
    }
    return v;
}
template <int G>
__device__ __forceinline__ float nrc_group_sum(float v) {
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, G);
    return v;
}
template <int G>
__device__ __forceinline__ int nrc_group_sum_i(int v) {
#pragma unroll
    for (int d = G / 2; d >= 1; d >>= 1) v += __shfl_xor(v, d, G);
    return v;
}
__device__ __forceinline__ int nrc_wave_incl_sum_i(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// block-wide exclusive scan of one int per thread (blockDim.x == 256 -> 4 waves). `smem` needs 8 ints.
__device__ __forceinline__ int nrc_block256_excl_scan_i(int v, int* smem, int* block_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = nrc_wave_incl_sum_i(v, lane);
    if (lane == 63) smem[wave] = incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        const int s = smem[w];
        if (w < wave) base += s;
    }
    if (block_total) *block_total = smem[0] + smem[1] + smem[2] + smem[3];
    __syncthreads();
    return base + incl - v;
}

// "Am I the last workgroup of this launch to get here?" -- for kernels whose closing step (a scan over per-workgroup results, a final sum)
// runs in whichever workgroup finishes last instead of in a launch of its own.  A single ticket word serialises: returning atomics on ONE
// address retire at ~11 ns each on this chip (4 096 workgroups = 47 us, measured on k_amp_check_prepare's first version), so the ticket has
// two levels -- 16 first-level words on 16 different cache lines, the last arrival of each moves the second-level word.  `tickets`: 17 x 16
// u32 words (NRC_TICKET_WORDS), zero before the first launch; the last workgroup puts them back to zero.  All threads of the workgroup call
// this (it contains barriers); results written before the call must have been stored with agent-scope atomics (or be followed by
// s_waitcnt vmcnt(0), which the call issues) and are to be read back with agent-scope loads.
#define NRC_TICKET_WORDS (17 * 16)
__device__ __forceinline__ bool nrc_last_workgroup(uint32_t* tickets, uint32_t block_id, uint32_t n_blocks) {
    __shared__ bool nrc_last_s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0) {
        const uint32_t g = block_id & 15u;
        const uint32_t in_group = (n_blocks - g + 15u) / 16u;          // workgroups whose id is congruent to g
        const uint32_t groups = n_blocks < 16u ? n_blocks : 16u;
        bool last = false;
        if (__hip_atomic_fetch_add(&tickets[16 * g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == in_group - 1u)
            last = __hip_atomic_fetch_add(&tickets[16 * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == groups - 1u;
        if (last)
            for (int k = 0; k <= 16; k++) __hip_atomic_store(&tickets[16 * k], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nrc_last_s = last;
    }
    __syncthreads();
    return nrc_last_s;
}

// Pixel footprint of a sample-layout tile (= one wave) of the fused InstantNGP image pipeline: 2^NRC_TILE_W_LOG2 x 64/2^NRC_TILE_W_LOG2.
#ifndef NRC_TILE_W_LOG2
#define NRC_TILE_W_LOG2 3
#endif
#define NRC_TILE_W (1 << NRC_TILE_W_LOG2)
#define NRC_TILE_H (64 >> NRC_TILE_W_LOG2)

// internal (C++ linkage) launchers shared between translation units: layer-major image compositing (ngp_composite.hip)
void nrc_launch_layers_init(int64_t n_tiles, const int32_t* ray_cnt, float* state, uint8_t* ray_alive, int32_t* next_k, uint8_t* tile_alive,
                            int32_t* skipped_rows, hipStream_t s);
void nrc_launch_composite_layers(const void* packed, const float* ts, const int32_t* ray_cnt, const int32_t* tile_rows, const int32_t* tile_off,
                                 const int32_t* row_of, int32_t* row_tile, int64_t row_end, int width, int height, int64_t tile_begin, int64_t n_tiles, int cascades, float esf,
                                 int grid_size, int max_samples, float thr, const float* bg3, float* state, uint8_t* ray_alive, int32_t* next_k,
                                 uint8_t* tile_alive, float* rgb, float* alpha, float* depth, int32_t* skipped_rows, int arena_rows, hipStream_t s);
// internal launchers of the optimizer step (adam.hip), used by nrc_ngp_train_backward_step (ngp_net.hip)
void nrc_launch_amp_prepare(int32_t* device_step, float* bias_corrections, float* scale, int32_t* growth_tracker, float growth_factor, float backoff_factor,
                            int32_t growth_interval, float beta1, float beta2, float* state4, hipStream_t s);
void nrc_launch_amp_adam(float* pa, const float* ga, float* ma, float* va, void* ha, int64_t na, float l2c_a, int64_t l2n_a, float* pb, const float* gb, float* mb,
                         float* vb, void* hb, int64_t nb, float l2c_b, int64_t l2n_b, const float* state4, const float* bias_corrections, const float* lr_dev, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int adam_w_mode, hipStream_t s);
