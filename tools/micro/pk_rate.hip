// tools/micro/pk_rate.hip -- does packed f32 VALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two f32 per lane per instruction) buy throughput on
// gfx950?  Same harness as valu_rate.hip: 4 independent chains per lane, W waves per SIMD; variants: v_fma_f32 (1 result per lane and instruction),
// v_pk_fma_f32, v_pk_mul_f32 + v_pk_add_f32 (the unfused pair -ffp-contract=off code needs), plus v_exp_f32 and v_cndmask for reference.
// The round-2 review proposed two pixels per lane with packed math for the 3DGS blend kernels; this measures what that can give.
// hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    f2 v[4];
#pragma unroll
    for (int c = 0; c < 4; c++) { v[c].x = (float)(threadIdx.x + c); v[c].y = (float)(threadIdx.x + 7 * c); }
    const f2 A = {a, a * 1.0001f}, B = {b, b * 0.999f};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[c].x) : "v"(a), "v"(b));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(A), "v"(B));
                if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[c]) : "v"(A));
                if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[c]) : "v"(B));
                if (MODE == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[c].x) : "v"(a));
                if (MODE == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(v[c].x));
                if (MODE == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[c].x) : "v"(a));
            }
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; c++) s += v[c].x + v[c].y;
    if (s == 123.456f) out[0] = s;
}
template <int MODE>
void run(const char* name, int results_per_lane, int waves_per_simd) {
    float* o; hipMalloc(&o, 4);
    const int iters = 4096, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, iters, 1.0001f, 0.5f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, o, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 16 * 4 * waves_per_simd;
    const double cyc = ms * 1e-3 * 2.4e9 / instr_per_simd;
    printf("%-14s waves/SIMD %d : %6.2f cycles per wave-instruction per SIMD (2.4 GHz) = %5.2f cycles per f32 result per lane\n", name, waves_per_simd, cyc,
           cyc / results_per_lane);
    hipFree(o);
}
int main() {
    for (int w : {1, 4, 8}) {
        run<0>("v_fma_f32", 1, w); run<4>("v_mul_f32", 1, w); run<1>("v_pk_fma_f32", 2, w); run<2>("v_pk_mul_f32", 2, w); run<3>("v_pk_add_f32", 2, w);
        run<5>("v_exp_f32", 1, w); run<6>("v_cndmask_b32", 1, w);
    }
    return 0;
}
