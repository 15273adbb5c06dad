// Micro-test (GPU box): operand layout and scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3 -> f32) on gfx950, and of
// v_cvt_pk_fp8_f32 -- what a split-operand conv with fp8 correction products would build on.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_scale_probe.hip -o gpurun_out/mfma_scale_probe && gpurun_out/mfma_scale_probe
// Hypotheses under test:
//   H1  A (32 x 64) / B (64 x 32): lane l holds row / column l & 31, K block l >> 5 (32 consecutive k = 8 VGPRs, byte b of register r is k = 4r + b)
//   H2  C/D: lane l, register r: column l & 31, row (r & 3) + 8 (r >> 2) + 4 (l >> 5)               (the map of the 32x32 f16 shapes)
//   H3  scale_a / scale_b: byte 0 of the lane's scale register is an E8M0 exponent e, the lane's 32 values count as value * 2^(e - 127)
//   H4  __builtin_amdgcn_cvt_pk_fp8_f32 on gfx950 produces OCP e4m3 (448 max, no fnuz), rounds to nearest even; saturation behaviour is printed
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void mfma_k(const unsigned char* A, const unsigned char* B, const int* sa, const int* sb, float* D)
{
    const int l = threadIdx.x;
    i8v a, b;
    for (int r = 0; r < 8; ++r) {
        unsigned va = 0, vb = 0;
        for (int q = 0; q < 4; ++q) {
            const int k = (l >> 5) * 32 + 4 * r + q;
            va |= (unsigned)A[(l & 31) * 64 + k] << (8 * q);       // A[i][k]
            vb |= (unsigned)B[k * 32 + (l & 31)] << (8 * q);       // B[k][j]
        }
        a[r] = (int)va; b[r] = (int)vb;
    }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 16; ++r) D[l * 16 + r] = c[r];
}

// the kernel's use: A parked in AGPRs, C non-zero (C[r] = 1000 + r), uniform scales 108 / 129 held in VGPRs behind an opaque asm
__global__ void mfma_k2(const unsigned char* A, const unsigned char* B, float* D)
{
    const int l = threadIdx.x;
    i8v a, b;
    for (int r = 0; r < 8; ++r) {
        unsigned va = 0, vb = 0;
        for (int q = 0; q < 4; ++q) {
            const int k = (l >> 5) * 32 + 4 * r + q;
            va |= (unsigned)A[(l & 31) * 64 + k] << (8 * q);
            vb |= (unsigned)B[k * 32 + (l & 31)] << (8 * q);
        }
        a[r] = (int)va; b[r] = (int)vb;
    }
    asm volatile("" : "+a"(a));
    int s_a = 127 - 19, s_b = 127 + 2;
    asm volatile("" : "+v"(s_a), "+v"(s_b));
    f16v c;
    for (int r = 0; r < 16; ++r) c[r] = 1000.f + r;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, s_a, 0, s_b);
    for (int r = 0; r < 16; ++r) D[l * 16 + r] = c[r];
}

__global__ void cvt_k(const float* x, unsigned* y, float* back, int n)
{
    const int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i >= n) return;
    const unsigned p = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x[i], -x[i], 0, false);
    y[i] = p;
    back[i] = __builtin_amdgcn_cvt_f32_fp8((int)p, 0);
}

static float e4m3_to_float(unsigned char v)      // OCP e4m3fn
{
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 15 && m == 7) f = NAN;
    else if (e == 0) f = std::ldexp((float)m, -9);
    else f = std::ldexp(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}

int main()
{
    std::vector<unsigned char> A(32 * 64), B(64 * 32);
    srand(1);
    for (auto& v : A) v = (unsigned char)(rand() & 0xFF);
    for (auto& v : B) v = (unsigned char)(rand() & 0xFF);
    for (auto& v : A) if ((v & 0x7F) == 0x7F) v ^= 1;       // no NaN codes
    for (auto& v : B) if ((v & 0x7F) == 0x7F) v ^= 1;
    for (auto& v : A) v = (v & 0x87) | (((v >> 3) & 7) + 4) << 3;      // exponents 4..11: moderate magnitudes
    for (auto& v : B) v = (v & 0x87) | (((v >> 3) & 7) + 4) << 3;
    unsigned char *dA, *dB; int *dsa, *dsb; float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 64 * 4); hipMalloc(&dsb, 64 * 4); hipMalloc(&dD, 64 * 16 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    for (int test = 0; test < 3; ++test) {
        int sa[64], sb[64];
        for (int l = 0; l < 64; ++l) {
            sa[l] = test == 0 ? 127 : test == 1 ? 127 - (l & 31) % 5 - 3 * (l >> 5) : 127 | 0x55000000;       // test 2: garbage in the other bytes
            sb[l] = test == 0 ? 127 : test == 1 ? 127 + (l & 31) % 3 - 11 : 120 | 0x00AA3300;
        }
        hipMemcpy(dsa, sa, sizeof sa, hipMemcpyHostToDevice); hipMemcpy(dsb, sb, sizeof sb, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_k, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
        float D[64 * 16];
        hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
        // which lane's scale byte applies to A[i][k] / B[k][j]?  candidates: 0 = the lane that holds the value (i + 32 (k >> 5)), 1 = lane i (K block ignored),
        // 2 = lane i + 32 (both halves must agree: take the upper), 3 = lane 2 (i & 15) + (k >> 5) + 32 (i >> 4)  (pairs of lanes)
        for (int hyp = 0; hyp < 4; ++hyp) {
            double worst = 0, scale = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 16; ++r) {
                    const int j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                    double ref = 0;
                    for (int k = 0; k < 64; ++k) {
                        const int kb = k >> 5;
                        const int la = hyp == 0 ? i + 32 * kb : hyp == 1 ? i : hyp == 2 ? i + 32 : 2 * (i & 15) + kb + 32 * (i >> 4);
                        const int lb = hyp == 0 ? j + 32 * kb : hyp == 1 ? j : hyp == 2 ? j + 32 : 2 * (j & 15) + kb + 32 * (j >> 4);
                        ref += (double)e4m3_to_float(A[i * 64 + k]) * std::ldexp(1.0, (sa[la] & 255) - 127) * (double)e4m3_to_float(B[k * 32 + j]) * std::ldexp(1.0, (sb[lb] & 255) - 127);
                    }
                    worst = std::fmax(worst, std::fabs(ref - D[l * 16 + r]));
                    scale = std::fmax(scale, std::fabs(ref));
                }
            printf("mfma_scale_f32_32x32x64 fp8 x fp8, test %d (%s), scale-lane hypothesis %d: max |device - hypothesis| = %.3e of max |C| = %.3e   %s\n", test,
                   test == 0 ? "scales 1" : test == 1 ? "per-lane scales" : "byte 0 only", hyp, worst, scale, worst <= 1e-4 * scale ? "HOLDS" : "mismatch");
        }
    }
    {
        hipLaunchKernelGGL(mfma_k2, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        float D[64 * 16];
        hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
        double worst = 0, scale = 0;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 16; ++r) {
                const int j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
                double ref = 0;
                for (int k = 0; k < 64; ++k) ref += (double)e4m3_to_float(A[i * 64 + k]) * (double)e4m3_to_float(B[k * 32 + j]);
                ref = ref * std::ldexp(1.0, -17) + 1000.0 + r;
                worst = std::fmax(worst, std::fabs(ref - D[l * 16 + r]));
                scale = std::fmax(scale, std::fabs(ref - 1000.0 - r));
            }
        printf("A in AGPRs, C = 1000 + r, scales 108 / 129: max |device - hypothesis| = %.3e, max |product term| = %.3e   %s\n", worst, scale, worst <= 2e-4 ? "HOLDS" : "MISMATCH");
    }
    // conversions
    const float xs[] = {0.f, 1.f, 1.0625f, 1.1875f, 0.3f, 447.f, 448.f, 449.f, 464.f, 480.f, 1000.f, 1e6f, 0.001953125f, 0.0009765625f, 0.0015f, 1e-4f, 17.f, 18.f, 19.f};
    const int n = sizeof xs / sizeof xs[0];
    float *dx, *dback; unsigned* dy;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dback, n * 4);
    hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cvt_k, dim3(1), dim3(64), 0, 0, dx, dy, dback, n);
    unsigned y[64]; float back[64];
    hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost); hipMemcpy(back, dback, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i)
        printf("cvt_pk_fp8_f32(%g, %g): bytes %02x %02x  -> OCP e4m3 reading %g %g ; cvt_f32_fp8(byte 0) = %g\n", xs[i], -xs[i], y[i] & 255, (y[i] >> 8) & 255,
               e4m3_to_float(y[i] & 255), e4m3_to_float((y[i] >> 8) & 255), back[i]);
    return 0;
}
