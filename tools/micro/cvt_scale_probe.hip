// Micro-test (GPU box): v_cvt_scalef32_pk_fp8_f16 on gfx950 -- is the source multiplied or divided by the scale, does it saturate?
//   hipcc --offload-arch=gfx950 -O2 tools/micro/cvt_scale_probe.hip -o tools/micro/cvt_scale_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));
__global__ void k(const h2* x, unsigned* y, float sc, int n)
{
    if ((int)threadIdx.x >= n) return;
    s2 r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, x[threadIdx.x], sc, false);
    y[threadIdx.x] = __builtin_bit_cast(unsigned, r);
    // both halves: word_sel = false writes bits 15:0, word_sel = true bits 31:16 -- is the other half kept?
    s2 q = {0, 0};
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, x[threadIdx.x], sc, false);
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, x[(threadIdx.x + 1) % n], sc, true);
    y[64 + threadIdx.x] = __builtin_bit_cast(unsigned, q);
}
static float e4m3(unsigned char v)
{
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = (e == 15 && m == 7) ? NAN : e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}
int main()
{
    const float xs[] = {0.f, 1.f, 1.1875f, 0.3f, 3.f, 20.f, 28.f, 29.f, 30.f, 100.f, 1000.f, 60000.f, 1e-3f, 1e-4f, 6e-5f, 3e-6f};
    const int n = sizeof xs / sizeof xs[0];
    h2 hx[64]; for (int i = 0; i < n; ++i) { hx[i][0] = (_Float16)xs[i]; hx[i][1] = (_Float16)(-xs[i]); }
    h2* dx; unsigned* dy; hipMalloc(&dx, sizeof hx); hipMalloc(&dy, 128 * 4);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    for (float sc : {1.0f, 16.0f, 0.0625f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dy, sc, n);
        unsigned y[128]; hipMemcpy(y, dy, 128 * 4, hipMemcpyDeviceToHost);
        printf("scale operand %g:\n", sc);
        for (int i = 0; i < n; ++i) printf("  (%g, %g) -> bytes %02x %02x = %g %g\n", (float)hx[i][0], (float)hx[i][1], y[i] & 255, (y[i] >> 8) & 255, e4m3(y[i] & 255), e4m3((y[i] >> 8) & 255)), printf("      both words (this, next): %08x\n", y[64 + i]);
    }
    return 0;
}
