// tools/micro/tr_read_map.hip -- the lane map of gfx950's ds_read_b64_tr_b16, checked with exact integers.
// hipcc --offload-arch=gfx950 -O2 -o tr_read_map tr_read_map.hip && ./tr_read_map
// Expectation (cdna_hip_programming.md T10): per group of 16 lanes, lane 4q+p supplies the address of row q, columns 4p..4p+3 of a 4 x 16 block of
// 16-bit elements; lane i of the group receives column i, row q in element q.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __fp16 f4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define PITCH 72
__global__ void k(float* out) {
    __shared__ __fp16 img[64 * PITCH];
    for (int i = threadIdx.x; i < 64 * PITCH; i += 64) img[i] = (__fp16)(float)((i / PITCH) * 32 + (i % PITCH) % 32);   // row * 32 + col (col < 32 used)
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, q = i >> 2, p = i & 3, g = l >> 4;
    // group g reads the block rows 4g..4g+3, columns 16 (g & 1) .. +15
    const __fp16* a = img + (4 * g + q) * PITCH + 16 * (g & 1) + 4 * p;
    f4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) f4*)a);
    for (int j = 0; j < 4; j++) out[l * 4 + j] = (float)v[j];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        const int g = l >> 4, i = l & 15;
        for (int j = 0; j < 4; j++) {
            const float want = (float)((4 * g + j) * 32 + 16 * (g & 1) + i);
            if (h[l * 4 + j] != want) { if (bad < 8) printf("lane %d elem %d: got %g want %g\n", l, j, h[l * 4 + j], want); bad++; }
        }
    }
    printf("%s (%d mismatches)\n", bad ? "MAP DIFFERS" : "map as documented", bad);
    return bad != 0;
}
