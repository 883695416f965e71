// tools/micro/lds_atomics.hip -- LDS atomic throughput on gfx950: f32 add vs u32 add vs u64 add vs plain store, random addresses in a 128 KB slice.
// hipcc --offload-arch=gfx950 -O3 -o lds_atomics lds_atomics.hip && ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void __launch_bounds__(1024) k(const uint32_t* __restrict__ idx, int n, float* out) {
    extern __shared__ float acc[];
    for (int j = threadIdx.x; j < 32768; j += 1024) acc[j] = 0.f;
    __syncthreads();
    const uint32_t* p = idx + (size_t)blockIdx.x * n;
    for (int k0 = threadIdx.x; k0 < n; k0 += 1024) {
        const uint32_t e = p[k0];
        if (MODE == 0) atomicAdd(&acc[e & 32767], 1.0f);
        else if (MODE == 1) atomicAdd(reinterpret_cast<uint32_t*>(acc) + (e & 32767), 1u);
        else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long*>(acc) + (e & 16383), 1ull);
        else if (MODE == 3) acc[e & 32767] = 1.0f;
        else if (MODE == 4) __hip_atomic_fetch_add(&acc[e & 32767], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 5) { float old = atomicAdd(&acc[e & 32767], 1.0f); if (old == 12345.f) out[0] = old; }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[0] + acc[77];
}
int main() {
    const int n = 131072, blocks = 256;
    uint32_t* h = (uint32_t*)malloc((size_t)n * blocks * 4);
    uint32_t s = 12345;
    for (size_t i = 0; i < (size_t)n * blocks; i++) { s = s * 1664525u + 1013904223u; h[i] = s >> 8; }
    uint32_t* d; float* o;
    hipMalloc(&d, (size_t)n * blocks * 4); hipMalloc(&o, blocks * 4);
    hipMemcpy(d, h, (size_t)n * blocks * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto kern, const char* name) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 131072, 0, d, n, o);
        hipEventRecord(a);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(1024), 131072, 0, d, n, o);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        printf("%-28s %8.1f us  -> %.2f lane-ops per clock per CU (2.4 GHz)\n", name, ms * 1e3, (double)n / (ms * 1e-3 * 2.4e9));
    };
    run(k<3>, "plain ds_write_b32");
    run(k<0>, "atomicAdd float");
    run(k<4>, "hip_atomic relaxed wg float");
    run(k<5>, "atomicAdd float (returning)");
    run(k<1>, "atomicAdd u32");
    run(k<2>, "atomicAdd u64");
    return 0;
}
