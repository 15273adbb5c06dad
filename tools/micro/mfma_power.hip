// mfma_power.hip -- what the package power cap lets v_mfma_f32_32x32x16_f16 deliver, on operands whose bits toggle and on zeros: the PRACTICAL matrix roofline
// the conv kernels are measured against in DESIGN.md section 7 (MI355X_MICROARCH.md: "the chip clocks to its power budget", zero-filled inputs ran +19 %).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
// One 4-wave workgroup per CU (512 registers per wave as the conv kernels), every wave issues back-to-back MFMAs on eight accumulators (A / B fragments in
// registers: no LDS, no memory traffic in the loop) for ~0.3 s per variant; effective clock = MFMAs x 32 cycles / time (the loop is nothing but MFMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));

template <int IDLE, bool SHAREB = false>      // SHAREB: the eight MFMAs of an iteration read the SAME B fragment (the conv kernels: six MFMAs per fragment); IDLE: s_nop states inserted behind every MFMA (a lower duty cycle of the matrix pipe: what the clock does when the pipe is not full)
__global__ __launch_bounds__(256) void mfma_loop(const half8_t* a, const half8_t* b, float* out, int iters)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    half8_t A[8], B[8];
    for (int i = 0; i < 8; ++i) { A[i] = a[(i * 4 + w) * 64 + lane]; B[i] = b[((i * 4 + w) * 64 + lane + blockIdx.x) % (32 * 64)]; }
    float16_t acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; it += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[i], SHAREB ? B[j] : B[(i + j) & 7], acc[i], 0, 0, 0);
                if (IDLE == 1) asm volatile("s_nop 7");
                if (IDLE == 2) asm volatile("s_nop 7\ns_nop 7\ns_nop 7");
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(A[i]), "+v"(B[i]));      // (keeps the loop from being folded)
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv)
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    std::vector<_Float16> ha(32 * 64 * 8), hb(32 * 64 * 8);
    half8_t *da, *db; float* dout;
    hipMalloc(&da, ha.size() * 2); hipMalloc(&db, hb.size() * 2); hipMalloc(&dout, cus * 256 * 4);
    const double peak = cus * 4.0 * 1024 * p.clockRate * 1e3 / 1e12;
    printf("%d CUs, max clock %.2f GHz, nominal fp16 MFMA peak %.0f TFLOP/s\n", cus, p.clockRate / 1e6, peak);
    for (int kind = 0; kind < 4; ++kind) {
        srand(1);
        for (size_t i = 0; i < ha.size(); ++i) {
            const float u = rand() / (float)RAND_MAX * 2.f - 1.f, v = rand() / (float)RAND_MAX * 2.f - 1.f;
            ha[i] = (_Float16)(kind == 0 ? 0.f : kind != 2 ? u * 0.05f : (u > 0 ? 0.03125f : -0.03125f));       // zeros | uniform random | +-2^-5 (only the sign toggles) | uniform random, B shared
            hb[i] = (_Float16)(kind == 0 ? 0.f : v);
        }
        hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
        for (int idle = 0; idle < 3; ++idle) {
            const int iters = argc > 1 ? atoi(argv[1]) : 150000;      // ~60 ms per launch on random data: long enough for the power management to settle
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&](int n) {
                if (kind == 3) { if (idle == 0) mfma_loop<0, true><<<cus, 256>>>(da, db, dout, n); else if (idle == 1) mfma_loop<1, true><<<cus, 256>>>(da, db, dout, n); else mfma_loop<2, true><<<cus, 256>>>(da, db, dout, n); return; }
                if (idle == 0) mfma_loop<0><<<cus, 256>>>(da, db, dout, n);
                else if (idle == 1) mfma_loop<1><<<cus, 256>>>(da, db, dout, n);
                else mfma_loop<2><<<cus, 256>>>(da, db, dout, n);
            };
            launch(iters / 4);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch(iters);
            launch(iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mfmas = 2.0 * iters * 8;               // per wave
            const double tf = mfmas * 32768.0 * cus * 4 / (ms / 1e3) / 1e12;
            printf("%-22s idle states %d: %7.1f TFLOP/s = %.3f of nominal, %.2f ms, MFMA issue rate = %.3f GHz x 32-cycle MFMAs\n",
                   kind == 0 ? "zeros" : kind == 1 ? "uniform random" : kind == 2 ? "random signs" : "random, B per 8 MFMAs", idle == 0 ? 0 : idle == 1 ? 8 : 24, tf, tf / peak, ms, mfmas * 32 / (ms / 1e3) / 1e9);
        }
    }
    return 0;
}
