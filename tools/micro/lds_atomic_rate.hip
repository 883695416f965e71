// tools/micro/lds_atomic_rate.hip -- rate of LDS atomics on gfx950 in the access pattern of k_render_bw: 4 waves per workgroup, in every wave the
// first lane of each 16-lane row (4 active lanes) adds NQ values of one list entry to s_acc[entry][q]; entries differ between rows.
// Variants: ds_add_f32 (what the kernel does), ds_add_u32, ds_add_u64, and ONE instruction with NQ active lanes per row (quantity q in lane q).
// hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define NQ 9
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    __shared__ float s_f[256][NQ + 1];
    __shared__ unsigned long long s_u64[256][NQ + 1];
    unsigned* s_u = reinterpret_cast<unsigned*>(&s_f[0][0]);
    for (int i = threadIdx.x; i < 256 * (NQ + 1); i += 256) { (&s_f[0][0])[i] = 0.f; (&s_u64[0][0])[i] = 0ull; }
    __syncthreads();
    const int lane = threadIdx.x & 63, row = threadIdx.x >> 4, q = lane & 15;
    unsigned e = row * 37u + 11u;
    const float v = 1.0f + lane;
    for (int i = 0; i < iters; i++) {
        e = (e * 1664525u + 1013904223u);
        const int ent = (e >> 8) & 255;
        if (MODE == 0) { if (q == 0) { for (int k = 0; k < NQ; k++) atomicAdd(&s_f[ent][k], v); } }
        if (MODE == 1) { if (q == 0) { for (int k = 0; k < NQ; k++) atomicAdd(&s_u[ent * (NQ + 1) + k], (unsigned)lane + 1u); } }
        if (MODE == 2) { if (q == 0) { for (int k = 0; k < NQ; k++) atomicAdd(&s_u64[ent][k], (unsigned long long)lane + 1ull); } }
        if (MODE == 3) { if (q < NQ) atomicAdd(&s_f[ent][q], v); }
        if (MODE == 4) { if (q < NQ) atomicAdd(&s_u64[ent][q], (unsigned long long)lane + 1ull); }
        if (MODE == 5) { if (q < NQ) atomicAdd(&s_u[ent * (NQ + 1) + q], (unsigned)lane + 1u); }
    }
    __syncthreads();
    float s = 0.f;
    for (int i = threadIdx.x; i < 256 * (NQ + 1); i += 256) s += (&s_f[0][0])[i] + (float)(&s_u64[0][0])[i];
    if (s == 123.456f) out[0] = s;
}
template <int MODE>
void run(const char* name, int blocks_per_cu) {
    float* o; hipMalloc(&o, 4);
    const int iters = 20000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per CU: blocks_per_cu workgroups x 16 rows x iters (row, entry) updates of NQ values
    const double updates = (double)blocks_per_cu * 16 * iters;
    printf("%-44s %d WG/CU : %8.1f us -> %6.1f cycles per (row, entry) update per CU, %.2f lane-ops/clk/CU\n", name, blocks_per_cu, ms * 1e3,
           ms * 1e-3 * 2.4e9 / updates, updates * NQ / (ms * 1e-3 * 2.4e9));
    hipFree(o);
}
int main() {
    for (int b : {1, 2, 4}) {
        run<0>("ds_add_f32, 9 instr x 4 lanes", b);
        run<1>("ds_add_u32, 9 instr x 4 lanes", b);
        run<2>("ds_add_u64, 9 instr x 4 lanes", b);
        run<3>("ds_add_f32, 1 instr x 36 lanes", b);
        run<4>("ds_add_u64, 1 instr x 36 lanes", b);
        run<5>("ds_add_u32, 1 instr x 36 lanes", b);
    }
    return 0;
}
