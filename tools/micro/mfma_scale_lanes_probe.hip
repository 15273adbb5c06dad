// Micro-test (GPU box), groundwork for block-scaled fp6 correction products (tests/emu_precision.py corr6):
//   (1) v_mfma_scale_f32_32x32x64_f8f6f4: WHICH lane's scale byte applies to which (row, K block) of A and (column, K block) of B?   All operands 1.0 (fp8), all
//       scales 127 but one lane's (128 = x2): the outputs that grow tell the lane's reach.
//   (2) fp6 e2m3 as the A operand (cbsz = 2): 32 values in 24 bytes (6 VGPRs) -- in which order?   A[i][slot] carries a slot-specific code, B selects one k.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_scale_lanes_probe.hip -o tools/micro/mfma_scale_lanes_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int i8v __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int CBSZ>
__global__ void k(const unsigned* A, const unsigned* B, const int* sa, const int* sb, float* D)
{
    const int l = threadIdx.x;
    i8v a, b;
    for (int r = 0; r < 8; ++r) { a[r] = (int)A[l * 8 + r]; b[r] = (int)B[l * 8 + r]; }
    f16v c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, CBSZ, 0, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 16; ++r) D[l * 16 + r] = c[r];
}

// v_cvt_scalef32_pk_fp4_f16: two fp16 -> two e2m1 nibbles (one byte), source divided by the scale operand?  op_sel picks the destination byte?
__global__ void kc(const unsigned* x, unsigned* y, float sc, int n)
{
    if ((int)threadIdx.x >= n) return;
    const unsigned a = x[threadIdx.x], b = x[(threadIdx.x + 1) % n];
    unsigned p = 0xAAAAAAAAu, q = 0xAAAAAAAAu;
    asm volatile("v_cvt_scalef32_pk_fp4_f16 %0, %1, %2\n\ts_nop 1" : "+v"(p) : "v"(a), "v"(sc));
    asm volatile("v_cvt_scalef32_pk_fp4_f16 %0, %1, %2 op_sel:[0,0,1,0]\n\ts_nop 1" : "+v"(q) : "v"(b), "v"(sc));
    y[2 * threadIdx.x] = p; y[2 * threadIdx.x + 1] = q;
}

static float e2m1(unsigned v)      // OCP fp4: s ee m, bias 1: 0, 0.5, 1, 1.5, 2, 3, 4, 6
{
    const int s = (v >> 3) & 1, e = (v >> 1) & 3, m = v & 1;
    const float f = e == 0 ? m * 0.5f : std::ldexp(1.0f + m * 0.5f, e - 1);
    return s ? -f : f;
}

static float e2m3(unsigned v)      // OCP fp6 e2m3: s eeMMM, bias 1
{
    const int s = (v >> 5) & 1, e = (v >> 3) & 3, m = v & 7;
    const float f = e == 0 ? m / 8.0f : std::ldexp(1.0f + m / 8.0f, e - 1);
    return s ? -f : f;
}

int main()
{
    unsigned *dA, *dB; int *dsa, *dsb; float* dD;
    (void)hipMalloc(&dA, 64 * 8 * 4); (void)hipMalloc(&dB, 64 * 8 * 4); (void)hipMalloc(&dsa, 256); (void)hipMalloc(&dsb, 256); (void)hipMalloc(&dD, 64 * 16 * 4);
    std::vector<unsigned> ones(64 * 8, 0x38383838u);      // fp8 e4m3 1.0 = 0x38
    float D[64 * 16];
    auto rowof = [](int l, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); };      // C/D map of the 32x32 shapes: lane l, register r -> row; column = l & 31
    // ---- (1) scale lanes ----
    for (int side = 0; side < 2; ++side) {
        printf("%s scales: lane -> what grows (rows for A, columns for B) and by how much (64 = all ones; 96 = one K block of 32 doubled; 128 = both)\n", side ? "B" : "A");
        for (int l0 = 0; l0 < 64; ++l0) {
            int sa[64], sb[64];
            for (int l = 0; l < 64; ++l) { sa[l] = 127; sb[l] = 127; }
            (side ? sb : sa)[l0] = 128;
            (void)hipMemcpy(dA, ones.data(), 64 * 8 * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, ones.data(), 64 * 8 * 4, hipMemcpyHostToDevice);
            (void)hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
            (void)hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
            bool hit[32] = {}; float val = 64.f; int nonuniform = 0;
            for (int l = 0; l < 64; ++l)
                for (int r = 0; r < 16; ++r) {
                    const float v = D[l * 16 + r];
                    if (v != 64.f) { hit[side ? (l & 31) : rowof(l, r)] = true; if (val != 64.f && v != val) ++nonuniform; val = v; }
                }
            printf("  lane %2d:", l0);
            int nh = 0;
            for (int i = 0; i < 32; ++i) if (hit[i]) { if (nh < 6) printf(" %d", i); ++nh; }
            printf("%s  (%d %s, value %g%s)\n", nh > 6 ? " ..." : "", nh, side ? "columns" : "rows", val, nonuniform ? ", NOT uniform" : "");
        }
    }
    // ---- (2) fp6 e2m3 A operand: slot order ----
    {
        // A: slot t (0..31) of a lane's 24-byte stream carries code 1 + (t % 28): value e2m3(code); stream = little-endian bit string, slot t at bits [6t, 6t+6)
        std::vector<unsigned> A(64 * 8, 0);
        for (int l = 0; l < 64; ++l) {
            unsigned char bytes[32] = {};
            for (int t = 0; t < 32; ++t) {
                const unsigned code = 1 + (t % 28);
                for (int bit = 0; bit < 6; ++bit) if (code >> bit & 1) bytes[(6 * t + bit) >> 3] |= (unsigned char)(1u << ((6 * t + bit) & 7));
            }
            memcpy(&A[l * 8], bytes, 32);
        }
        int s1[64]; for (int l = 0; l < 64; ++l) s1[l] = 127;
        (void)hipMemcpy(dsa, s1, 256, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, s1, 256, hipMemcpyHostToDevice);
        (void)hipMemcpy(dA, A.data(), 64 * 8 * 4, hipMemcpyHostToDevice);
        printf("fp6 e2m3 A operand (cbsz 2), B = fp8 unit vector e_k: D[0][0] per k = the A value that sits at k; hypothesis: slot t of lane (row, K block kb) is k = 32 kb + t:\n");
        int good = 0;
        for (int kq = 0; kq < 64; ++kq) {
            std::vector<unsigned> B(64 * 8, 0);      // B[k][j] = 1 for k == kq: lane (j, kb = kq >> 5), byte kq & 31
            for (int j = 0; j < 32; ++j) { const int l = j + 32 * (kq >> 5); ((unsigned char*)&B[l * 8])[kq & 31] = 0x38; }
            (void)hipMemcpy(dB, B.data(), 64 * 8 * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
            (void)hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
            const float want = e2m3(1 + ((kq & 31) % 28));
            good += D[0] == want;
            if (kq < 8 || D[0] != want) printf("  k %2d: D[0][0] = %g (hypothesis %g)%s\n", kq, D[0], want, D[0] == want ? "" : "   <-- differs");
        }
        printf("  %d of 64 k agree with the hypothesis\n", good);
    }
    // ---- (3) fp4 e2m1 A operand (cbsz 4): 32 values in 16 bytes (4 VGPRs), nibble t at bits [4t, 4t+4)? ----
    {
        std::vector<unsigned> A(64 * 8, 0);
        for (int l = 0; l < 64; ++l) {
            unsigned char bytes[32] = {};
            for (int t = 0; t < 32; ++t) bytes[t >> 1] |= (unsigned char)((1 + (t % 7)) << (4 * (t & 1)));
            memcpy(&A[l * 8], bytes, 32);
        }
        (void)hipMemcpy(dA, A.data(), 64 * 8 * 4, hipMemcpyHostToDevice);
        printf("fp4 e2m1 A operand (cbsz 4), B = fp8 unit vector e_k: D[0][0] per k; code of slot t = 1 + t %% 7 -> value e2m1(code):\n  ");
        for (int kq = 0; kq < 64; ++kq) {
            std::vector<unsigned> B(64 * 8, 0);
            for (int j = 0; j < 32; ++j) { const int l = j + 32 * (kq >> 5); ((unsigned char*)&B[l * 8])[kq & 31] = 0x38; }
            (void)hipMemcpy(dB, B.data(), 64 * 8 * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
            (void)hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
            printf("%g%s", D[0], (kq & 15) == 15 ? "\n  " : " ");
        }
        printf("(slot values in stream order: ");
        for (int t = 0; t < 16; ++t) printf("%g ", e2m1(1 + (t % 7)));
        printf("...)\n");
    }
    // ---- (4) the fp4 conversion ----
    {
        const float xs[] = {0.f, 0.2f, 0.25f, 0.3f, 0.5f, 0.75f, 1.f, 1.25f, 1.5f, 1.75f, 2.f, 2.5f, 3.f, 3.5f, 4.f, 5.f, 6.f, 7.f, 100.f, -1.25f};
        const int n = sizeof xs / sizeof xs[0];
        unsigned hx[32];
        for (int i = 0; i < n; ++i) { const _Float16 h0 = (_Float16)xs[i], h1 = (_Float16)(-xs[i]); unsigned short u0, u1; memcpy(&u0, &h0, 2); memcpy(&u1, &h1, 2); hx[i] = u0 | ((unsigned)u1 << 16); }
        unsigned* dy; (void)hipMalloc(&dy, 64 * 4);
        (void)hipMemcpy(dA, hx, n * 4, hipMemcpyHostToDevice);
        for (float sc : {1.0f, 4.0f}) {
            hipLaunchKernelGGL(kc, dim3(1), dim3(64), 0, 0, dA, dy, sc, n);
            unsigned y[64]; (void)hipMemcpy(y, dy, 2 * n * 4, hipMemcpyDeviceToHost);
            printf("v_cvt_scalef32_pk_fp4_f16, scale operand %g (destination preset 0xAAAAAAAA; second column: op_sel:[0,0,1,0] on the NEXT input):\n", sc);
            for (int i = 0; i < n; ++i) printf("  (%g, %g) -> %08x = (%g, %g)      | %08x\n", xs[i], -xs[i], y[2 * i], e2m1(y[2 * i] & 15), e2m1((y[2 * i] >> 4) & 15), y[2 * i + 1]);
        }
    }
    return 0;
}
