/* oracle/convref.c -- plain C restatement of the convolution primitive.  TEST INFRASTRUCTURE.
 *
 * What the reference's nn.Conv2d layers compute on this path (python/models.py:29-39: Conv3x3 =
 * Conv2d(k=3, stride=1, padding=1); python/MoeNet_lite2.py:5-6, models.py:186-196: 1x1 convs):
 *   out[b][o][y][x] = bias[o] + sum_{c,ky,kx} in[b][c][y+ky-p][x+kx-p] * w[o][c][ky][kx]
 * with zero padding p, stride 1, fp32 everywhere, accumulation in input-channel-major order.
 * NCHW contiguous in/out, OIHW weights.  OpenMP over (b, o) planes only; the inner loops are
 * scalar so this file stays an independent check of torch/oneDNN (not a fast kernel).
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC convref.c -o _build/libconvref.so  (oracle/Makefile)
 */
#include <stdint.h>
#include <string.h>

void moe_oracle_conv2d_f32(const float* in, const float* w, const float* bias, float* out,
                           int B, int Cin, int H, int W, int Cout, int k, int pad)
{
    const int Ho = H + 2 * pad - k + 1, Wo = W + 2 * pad - k + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < Cout; ++o) {
            float* op = out + ((int64_t)b * Cout + o) * Ho * Wo;
            const float b0 = bias ? bias[o] : 0.0f;
            for (int i = 0; i < Ho * Wo; ++i) op[i] = b0;
            for (int c = 0; c < Cin; ++c) {
                const float* ip = in + ((int64_t)b * Cin + c) * H * W;
                const float* wp = w + ((int64_t)o * Cin + c) * k * k;
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        const float wv = wp[ky * k + kx];
                        const int y0 = pad - ky > 0 ? pad - ky : 0;
                        const int y1 = H + pad - ky < Ho ? H + pad - ky : Ho;
                        const int x0 = pad - kx > 0 ? pad - kx : 0;
                        const int x1 = W + pad - kx < Wo ? W + pad - kx : Wo;
                        for (int y = y0; y < y1; ++y) {
                            const float* irow = ip + (int64_t)(y + ky - pad) * W + (kx - pad);
                            float* orow = op + (int64_t)y * Wo;
                            for (int x = x0; x < x1; ++x) orow[x] += irow[x] * wv;
                        }
                    }
            }
        }
}

/* PReLU / LeakyReLU with one scalar slope (models.py:77, 181: nn.PReLU() has a single parameter). */
void moe_oracle_prelu_f32(float* x, int64_t n, float slope)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) x[i] = x[i] >= 0.0f ? x[i] : x[i] * slope;
}

int moe_oracle_abi_version(void) { return 1; }
