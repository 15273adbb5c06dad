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
    return 0;
}
