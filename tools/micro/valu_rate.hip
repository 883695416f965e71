// tools/micro/valu_rate.hip -- VALU issue rate per SIMD on gfx950 as a function of resident waves per SIMD and of instruction-level parallelism
// inside a wave: independent v_fma_f32 chains (CHAINS per lane), W waves per SIMD (blocks of 256 threads = 1 wave per SIMD each, W blocks per CU).
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    float v[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) v[c] = (float)(threadIdx.x + c);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) v[c] = __builtin_fmaf(v[c], a, b);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += v[c];
    if (s == 123.456f) out[0] = s;
}
template <int CHAINS>
void run(int waves_per_simd) {
    float* o; hipMalloc(&o, 4);
    const int iters = 4096, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, o, iters, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, o, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 16 * CHAINS * waves_per_simd;  // wave-instructions issued on one SIMD
    printf("chains %d  waves/SIMD %d : %8.1f us  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", CHAINS, waves_per_simd, ms * 1e3,
           ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(o);
}
int main() {
    for (int w : {1, 2, 4, 8}) { run<1>(w); run<2>(w); run<4>(w); run<8>(w); }
    return 0;
}
