// Micro-test (GPU box): does MODE.FP16_OVFL (bit 23 of the MODE register) make v_cvt_scalef32_pk_fp8_f16 saturate on gfx950 (without it: NaN beyond 464)?
// and: back-to-back half-register conversions with / without a wait state (the forwarding hazard conv64_q8.hip tripped over)
//   hipcc --offload-arch=gfx950 -O2 tools/micro/cvt_ovfl_probe.hip -o tools/micro/cvt_ovfl_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const unsigned* x, unsigned* y, float sc, int n, int ovfl)
{
    if ((int)threadIdx.x >= n) return;
    const unsigned a0 = x[threadIdx.x], a1 = x[(threadIdx.x + 1) % n];
    unsigned p;
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");
    asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, %3\n\ts_nop 0\n\tv_cvt_scalef32_pk_fp8_f16 %0, %2, %3 op_sel:[0,0,1]\n\ts_nop 0" : "=&v"(p) : "v"(a0), "v"(a1), "v"(sc));
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 0");
    y[threadIdx.x] = p;
    // an fp16 overflow behind the switch-off: must be inf again
    const float big = 1e6f * (float)(threadIdx.x + 1);
    y[64 + threadIdx.x] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)big);
}
// the decode conv64_q8.hip uses for fp8 residual words: __builtin_amdgcn_cvt_pk_f32_fp8(word, sel) -> bytes 2 sel, 2 sel + 1 as OCP e4m3?
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void kd(const unsigned* x, float* y, int n)
{
    if ((int)threadIdx.x >= n) return;
    const f2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)x[threadIdx.x], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)x[threadIdx.x], true);
    y[4 * threadIdx.x] = lo[0]; y[4 * threadIdx.x + 1] = lo[1]; y[4 * threadIdx.x + 2] = hi[0]; y[4 * threadIdx.x + 3] = hi[1];
}
static float e4m3(unsigned char v)
{
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = (e == 15 && m == 7) ? NAN : e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}
int main()
{
    const float xs[] = {0.f, 1.f, 100.f, 1700.f, 1800.f, 1856.f, 1900.f, 4000.f, 60000.f, INFINITY, NAN, 3e-6f};
    const int n = sizeof xs / sizeof xs[0];
    h2 hx[64]; for (int i = 0; i < n; ++i) { hx[i][0] = (_Float16)xs[i]; hx[i][1] = (_Float16)(-xs[i]); }
    unsigned* dx; unsigned* dy; hipMalloc(&dx, sizeof hx); hipMalloc(&dy, 128 * 4);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    for (int ovfl = 0; ovfl < 2; ++ovfl) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy, 4.0f, n, ovfl);
        unsigned y[128]; hipMemcpy(y, dy, 128 * 4, hipMemcpyDeviceToHost);
        printf("scale operand 4, FP16_OVFL %d:\n", ovfl);
        for (int i = 0; i < n; ++i) printf("  (%g, %g) -> bytes %02x %02x = %g %g   (word %08x; fp16(1e6 (i+1)) behind it = %04x)\n", (float)hx[i][0], (float)hx[i][1], y[i] & 255, (y[i] >> 8) & 255, e4m3(y[i] & 255), e4m3((y[i] >> 8) & 255), y[i], y[64 + i]);
    }
    {
        unsigned w[64]; srand(3);
        for (int i = 0; i < 64; ++i) { w[i] = (unsigned)rand() ^ ((unsigned)rand() << 16); for (int b = 0; b < 4; ++b) if (((w[i] >> (8 * b)) & 0x7F) == 0x7F) w[i] ^= 1u << (8 * b); }
        float* dz; hipMalloc(&dz, 256 * 4);
        hipMemcpy(dx, w, sizeof w, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(kd, dim3(1), dim3(64), 0, 0, dx, dz, 64);
        float z[256]; hipMemcpy(z, dz, sizeof z, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 64; ++i) for (int b = 0; b < 4; ++b) bad += z[4 * i + b] != e4m3((w[i] >> (8 * b)) & 255);
        printf("cvt_pk_f32_fp8(word, sel): byte 2 sel + {0, 1} as OCP e4m3 -- %d of 256 values differ\n", bad);
    }
    return 0;
}
