/* oracle/ssim_oracle.c -- CPU restatement of the SSIM map and its gradient w.r.t. the first image, as the reference's 3DGS loss
 * uses it: src/Optim/Losses/DSSIM.py:11-18 (fused_dssim = 1 - fused_ssim(input, target)), src/Methods/GaussianSplatting/Loss.py:14-15.
 *
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
 *
 * The arithmetic lives in the third-party package fused-ssim (github.com/rahul-goel/fused-ssim, unpinned in
 * src/Thirdparty/FusedSSIM.py:10), which is NOT under /root/reference: PARITY UNPINNED against that binary.  Restated from its
 * published algorithm = the SSIM of Wang et al. 2004 with an 11x11 Gaussian window (sigma 1.5), zero "same" padding,
 * C1 = 0.01^2, C2 = 0.03^2, per channel:
 *   mu_k = G * x_k,  s_k = G * x_k^2 - mu_k^2,  s_12 = G * x_1 x_2 - mu_1 mu_2
 *   ssim = (2 mu_1 mu_2 + C1)(2 s_12 + C2) / ((mu_1^2 + mu_2^2 + C1)(s_1 + s_2 + C2))
 * and d ssim / d x_1 through the three maps d/dmu_1, d/ds_1, d/ds_12 (G is symmetric, so the adjoint of G* is G* again).
 * Pinned in tests/test_oracle_ssim.py against a plain PyTorch conv2d implementation + autograd.                              */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static void gauss11(double* g) {
    double s = 0;
    for (int i = 0; i < 11; i++) { g[i] = exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); s += g[i]; }
    for (int i = 0; i < 11; i++) g[i] /= s;
}
/* separable blur with zero padding, double accumulation; src/dst: H x W */
static void blur(const double* src, double* dst, double* tmp, int H, int W, const double* g) {
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            double a = 0;
            for (int k = -5; k <= 5; k++) { const int xx = x + k; if (xx >= 0 && xx < W) a += g[k + 5] * src[(size_t)y * W + xx]; }
            tmp[(size_t)y * W + x] = a;
        }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            double a = 0;
            for (int k = -5; k <= 5; k++) { const int yy = y + k; if (yy >= 0 && yy < H) a += g[k + 5] * tmp[(size_t)yy * W + x]; }
            dst[(size_t)y * W + x] = a;
        }
}
/* img1, img2: (planes, H, W) f32.  Outputs (all (planes, H, W) f32): ssim_map and, if non-null, the three partial-derivative maps. */
void oracle_ssim_forward(const float* img1, const float* img2, int64_t planes, int64_t H, int64_t W, float C1, float C2, float* ssim_map,
                         float* dm_dmu1, float* dm_dsigma1_sq, float* dm_dsigma12) {
    double g[11];
    gauss11(g);
    const size_t n = (size_t)H * W;
#pragma omp parallel for
    for (int64_t p = 0; p < planes; p++) {
        double* buf = (double*)malloc(sizeof(double) * n * 7);
        double *a = buf, *mu1 = buf + n, *mu2 = buf + 2 * n, *e11 = buf + 3 * n, *e22 = buf + 4 * n, *e12 = buf + 5 * n, *tmp = buf + 6 * n;
        const float *x1 = img1 + p * n, *x2 = img2 + p * n;
        for (size_t i = 0; i < n; i++) a[i] = x1[i];
        blur(a, mu1, tmp, (int)H, (int)W, g);
        for (size_t i = 0; i < n; i++) a[i] = x2[i];
        blur(a, mu2, tmp, (int)H, (int)W, g);
        for (size_t i = 0; i < n; i++) a[i] = (double)x1[i] * x1[i];
        blur(a, e11, tmp, (int)H, (int)W, g);
        for (size_t i = 0; i < n; i++) a[i] = (double)x2[i] * x2[i];
        blur(a, e22, tmp, (int)H, (int)W, g);
        for (size_t i = 0; i < n; i++) a[i] = (double)x1[i] * x2[i];
        blur(a, e12, tmp, (int)H, (int)W, g);
        for (size_t i = 0; i < n; i++) {
            const double m1 = mu1[i], m2 = mu2[i];
            const double s1 = e11[i] - m1 * m1, s2 = e22[i] - m2 * m2, s12 = e12[i] - m1 * m2;
            const double A = m1 * m1 + m2 * m2 + C1, B = s1 + s2 + C2, C = 2 * m1 * m2 + C1, D = 2 * s12 + C2;
            const double m = (C * D) / (A * B);
            ssim_map[p * n + i] = (float)m;
            if (dm_dmu1) {
                dm_dmu1[p * n + i] = (float)((m2 * 2.0 * D) / (A * B) - (m2 * 2.0 * C) / (A * B) - (m1 * 2.0 * C * D) / (A * A * B) + (m1 * 2.0 * C * D) / (A * B * B));
                dm_dsigma1_sq[p * n + i] = (float)((-C * D) / (A * B * B));
                dm_dsigma12[p * n + i] = (float)((2.0 * C) / (A * B));
            }
        }
        free(buf);
    }
}
/* dL/dimg1 = G*(dL_dmap dm_dmu1) + 2 img1 G*(dL_dmap dm_dsigma1_sq) + img2 G*(dL_dmap dm_dsigma12) */
void oracle_ssim_backward(const float* img1, const float* img2, int64_t planes, int64_t H, int64_t W, const float* dL_dmap, const float* dm_dmu1,
                          const float* dm_dsigma1_sq, const float* dm_dsigma12, float* dL_dimg1) {
    double g[11];
    gauss11(g);
    const size_t n = (size_t)H * W;
#pragma omp parallel for
    for (int64_t p = 0; p < planes; p++) {
        double* buf = (double*)malloc(sizeof(double) * n * 5);
        double *a = buf, *b1 = buf + n, *b2 = buf + 2 * n, *b3 = buf + 3 * n, *tmp = buf + 4 * n;
        for (size_t i = 0; i < n; i++) a[i] = (double)dL_dmap[p * n + i] * dm_dmu1[p * n + i];
        blur(a, b1, tmp, (int)H, (int)W, g);
        for (size_t i = 0; i < n; i++) a[i] = (double)dL_dmap[p * n + i] * dm_dsigma1_sq[p * n + i];
        blur(a, b2, tmp, (int)H, (int)W, g);
        for (size_t i = 0; i < n; i++) a[i] = (double)dL_dmap[p * n + i] * dm_dsigma12[p * n + i];
        blur(a, b3, tmp, (int)H, (int)W, g);
        for (size_t i = 0; i < n; i++) dL_dimg1[p * n + i] = (float)(b1[i] + 2.0 * img1[p * n + i] * b2[i] + (double)img2[p * n + i] * b3[i]);
        free(buf);
    }
}
