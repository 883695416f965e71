// tools/micro/gather_fetch_calib.hip -- what rocprofv3's FETCH_SIZE reports for the hash-grid encoder's access shape on gfx950.
//
// MI355X_MICROARCH.md calibrates FETCH_SIZE only for wide coalesced streaming reads (it reports HALF of the bytes: 128-byte requests tallied at
// 64 B) and calls every other access width uncalibrated.  k_grid_encode reads 16-byte pairs at hashed addresses of a 24.4 MB fp16 table.
// Four kernels with a KNOWN number of requested bytes / touched lines (printed), to be run under
//     rocprofv3 --pmc FETCH_SIZE                 --kernel-trace ...   (and, separately)
//     rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace ...
//   k_stream16   every lane reads consecutive 16-byte pieces of a 1 GiB buffer once            (the guide's calibrated case: expect 0.5 x bytes)
//   k_line16     ONE 16-byte load per 128-byte line of the 1 GiB buffer, every line exactly once  (a cold gather miss: what does one miss tally?)
//   k_half16     ONE 16-byte load per 64-byte half line, every half line exactly once, the two halves of a line far apart in time
//   k_table16    the encoder's shape: 16-byte loads at pseudo-random 16-byte-aligned offsets of a 24.4 MB table (L2 4 MB per XCD: mostly misses
//                served by the Infinity Cache), 64 loads per lane
// hipcc --offload-arch=gfx950 -O3 -o gather_fetch_calib gather_fetch_calib.hip && ./gather_fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__global__ void __launch_bounds__(256) k_stream16(const uint4* __restrict__ buf, size_t n_vec, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * 256) { const uint4 v = buf[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) out[0] = acc;
}
// one 16-byte load per `stride`-byte unit; unit u is read at chunk (u % chunks_per_unit) so that all positions inside a line occur
__global__ void __launch_bounds__(256) k_unit16(const uint4* __restrict__ buf, size_t n_units, int vec_per_unit, size_t unit_offset_vec, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < n_units; u += (size_t)gridDim.x * 256) {
        // a bijective scramble of the unit index inside blocks of 2^20 units: neighbouring lanes touch lines that are megabytes apart (a gather, not a stream)
        const size_t blk = u >> 20, in = u & 0xfffff;
        const size_t v = (blk << 20) | ((in * 0x9E3779B1ull) & 0xfffff);   // odd multiplier: a permutation of 20-bit values
        const uint4 q = buf[unit_offset_vec + v * vec_per_unit + (v % (size_t)vec_per_unit)];
        acc ^= q.x ^ q.y ^ q.z ^ q.w;
    }
    if (acc == 0x12345u) out[0] = acc;
}
// --- round 4: the 3DGS preprocessing kernels' access shapes (scalar 4-byte loads), for which the streaming correction had been applied unchecked ---
//   k_stream4    every lane reads consecutive 4-byte words (a wave instruction = 256 contiguous bytes)
//   k_stream8    consecutive 8-byte pairs (512 contiguous bytes per wave instruction)
//   k_soa12      a (P, 3) f32 array read the way k_preprocess reads means3D / scales: lane i loads words 3 i, 3 i + 1, 3 i + 2 (three instructions,
//                12-byte lane stride: each touches the same six lines)
//   k_rows192    a (P, 48) f32 array (the SH coefficients) read the way sh_rows_copy stages it: 128 rows = 24 576 contiguous bytes per workgroup,
//                thread t loads words t, t + 128, t + 256, ... (coalesced 4-byte loads)
//   k_preproc    all of k_preprocess's reads of one Gaussian together: (P,3) + (P,48) rows + (P,1) + (P,3) + (P,4) = 236 B per Gaussian
__global__ void __launch_bounds__(256) k_stream4(const uint32_t* __restrict__ buf, size_t n, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= buf[i];
    if (acc == 0x12345u) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_stream8(const uint2* __restrict__ buf, size_t n, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint2 v = buf[i]; acc ^= v.x ^ v.y; }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ void __launch_bounds__(128) k_soa12(const float* __restrict__ a, int P, uint32_t* __restrict__ out) {
    const int i = blockIdx.x * 128 + threadIdx.x;
    if (i >= P) return;
    const float s = a[3 * i] + a[3 * i + 1] + a[3 * i + 2];
    if (s == 12345.f) out[0] = 1u;
}
__global__ void __launch_bounds__(128) k_rows192(const float* __restrict__ rows, int P, uint32_t* __restrict__ out) {
    const size_t first = (size_t)blockIdx.x * 128;
    const int count = (int)((size_t)P - first < 128 ? (size_t)P - first : 128);
    const float* g = rows + first * 48;
    float s = 0.f;
    for (int k = threadIdx.x; k < count * 48; k += 128) s += g[k];
    if (s == 12345.f) out[0] = 1u;
}
__global__ void __launch_bounds__(128) k_preproc(const float* __restrict__ means, const float* __restrict__ sh, const float* __restrict__ op,
                                                 const float* __restrict__ sc, const float* __restrict__ rot, int P, uint32_t* __restrict__ out) {
    const size_t first = (size_t)blockIdx.x * 128;
    const int count = (int)((size_t)P - first < 128 ? (size_t)P - first : 128);
    const float* g = sh + first * 48;
    float s = 0.f;
    for (int k = threadIdx.x; k < count * 48; k += 128) s += g[k];
    const int i = (int)first + threadIdx.x;
    if (i < P) {
        s += means[3 * i] + means[3 * i + 1] + means[3 * i + 2] + op[i] + sc[3 * i] + sc[3 * i + 1] + sc[3 * i + 2];
        s += rot[4 * i] + rot[4 * i + 1] + rot[4 * i + 2] + rot[4 * i + 3];
    }
    if (s == 12345.f) out[0] = 1u;
}
__global__ void __launch_bounds__(256) k_table16(const uint4* __restrict__ table, uint32_t n_vec, int loads_per_lane, uint32_t* __restrict__ out) {
    uint32_t acc = 0, s = mix((uint32_t)(blockIdx.x * 256 + threadIdx.x) + 1u);
    for (int k = 0; k < loads_per_lane; k++) {
        s = mix(s + 0x9E3779B9u);
        const uint4 q = table[s % n_vec];
        acc ^= q.x ^ q.y ^ q.z ^ q.w;
    }
    if (acc == 0x12345u) out[0] = acc;
}

int main() {
    const size_t big = (size_t)1 << 30;                       // 1 GiB: every line is used once, nothing is re-read
    const size_t table_bytes = (size_t)6098120 * 4;           // the fp16 hash table of the shipped configuration: 6 098 120 entries x 2 halves
    uint4 *buf, *table; uint32_t* out;
    hipMalloc(&buf, big); hipMalloc(&table, table_bytes); hipMalloc(&out, 256);
    hipMemset(buf, 1, big); hipMemset(table, 2, table_bytes);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto timed = [&](const char* name, auto launch, double req_bytes, double lines) {
        launch();                                             // once untimed
        hipEventRecord(a);
        for (int r = 0; r < 3; r++) launch();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b); ms /= 3;
        printf("%-12s %8.3f ms  requested %10.1f MB  distinct 128-B lines %10.0f (= %9.1f MB)  -> %7.1f GB/s requested, %7.1f GB/s of lines\n", name, ms,
               req_bytes / 1e6, lines, lines * 128 / 1e6, req_bytes / ms / 1e6, lines * 128 / ms / 1e6);
    };
    const int grid = 256 * 16;
    timed("k_stream16", [&] { hipLaunchKernelGGL(k_stream16, dim3(grid), dim3(256), 0, 0, buf, big / 16, out); }, (double)big, (double)big / 128);
    const size_t n_lines = big / 128;
    timed("k_line16", [&] { hipLaunchKernelGGL(k_unit16, dim3(grid), dim3(256), 0, 0, buf, n_lines, 8, (size_t)0, out); }, (double)n_lines * 16, (double)n_lines);
    // half lines: first all lower halves, then (second launch inside the same "kernel name") all upper halves would need two kernels; instead units of 64 B,
    // scrambled over 2^20-unit blocks (64 MB): the partner half of a line is touched ~unrelatedly far away in the iteration order
    const size_t n_half = big / 64;
    timed("k_half16", [&] { hipLaunchKernelGGL(k_unit16, dim3(grid), dim3(256), 0, 0, buf, n_half, 4, (size_t)0, out); }, (double)n_half * 16, (double)n_lines);
    const int lpl = 64;
    const double n_loads = (double)grid * 256 * lpl;
    timed("k_table16", [&] { hipLaunchKernelGGL(k_table16, dim3(grid), dim3(256), 0, 0, table, (uint32_t)(table_bytes / 16), lpl, out); }, n_loads * 16,
          (double)table_bytes / 128);
    {   // round 4: the scalar-load shapes of the 3DGS preprocessing (buffers of the 1 M-Gaussian bench scene's sizes x 4, so that nothing is re-read from a cache)
        const int P = 4000000;
        float* f = reinterpret_cast<float*>(buf);
        const float *means = f, *sh = f + (size_t)3 * P, *op = sh + (size_t)48 * P, *sc = op + P, *rot = sc + (size_t)3 * P;   // 59 floats x P = 944 MB of the 1 GiB
        timed("k_stream4", [&] { hipLaunchKernelGGL(k_stream4, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, big / 4, out); }, (double)big, (double)big / 128);
        timed("k_stream8", [&] { hipLaunchKernelGGL(k_stream8, dim3(grid), dim3(256), 0, 0, (const uint2*)buf, big / 8, out); }, (double)big, (double)big / 128);
        timed("k_soa12", [&] { hipLaunchKernelGGL(k_soa12, dim3((P + 127) / 128), dim3(128), 0, 0, means, P, out); }, 12.0 * P, 12.0 * P / 128);
        timed("k_rows192", [&] { hipLaunchKernelGGL(k_rows192, dim3((P + 127) / 128), dim3(128), 0, 0, sh, P, out); }, 192.0 * P, 192.0 * P / 128);
        timed("k_preproc", [&] { hipLaunchKernelGGL(k_preproc, dim3((P + 127) / 128), dim3(128), 0, 0, means, sh, op, sc, rot, P, out); }, 236.0 * P, 236.0 * P / 128);
    }
    printf("k_table16: %.0f loads of 16 B over a %.1f MB table (%.0f lines): each line is read %.1f times per launch\n", n_loads, table_bytes / 1e6,
           (double)table_bytes / 128, n_loads / ((double)table_bytes / 128));
    return 0;
}
