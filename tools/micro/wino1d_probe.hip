// wino1d_probe.hip -- round 6, behind tools/micro/wino_probe.hip: the 2-D Winograd probe showed F(2x2, 3x3) with register-resident weights to be ISSUE-bound (12.8 instructions
// per MFMA on one wave per SIMD: MFMA busy 0.35; 2.25x fewer MFMAs bought 1.23x).  This probe measures the ONE-dimensional form: F(2, 3) along x, the three kernel rows as
// direct taps -- 12 position products per 2 output pixels instead of 18 tap products (1.5x fewer MFMAs), one 1-D transform each side:
//   input    V[y][p] = (d B)[p], p = 0..3: four positions per tile of 2 output pixels from 4 input pixels of ONE row: 16 packed-fp16 ops per (tile, 8 channels), once per input row
//   product  T[p] = sum over dy, cin of U[dy][p] V[o - 1 + dy][p]:  3 x 4 k-slices = 12 MFMAs per position and output row, 48 per wave (32 output channels) and row
//   output   Y0 = T0 + T1 + T2,  Y1 = T1 - T2 - T3: four fp32 adds per output pair (the 2-D form: 24 per four outputs)
// ~3.3 VALU + 1.2 LDS instructions per MFMA: inside what one wave per SIMD hides behind a 32-cycle MFMA.  Weights: U[dy 3][pos 4][cin 64] of the wave's 32 output channels =
// 48 A fragments = 192 registers, resident.  A workgroup (4 waves = 128 of the 256 output channels, half `ch` = blockIdx & 1) walks DOWN a 64-pixel strip one output row a step;
// the V rows live in an LDS ring of four (a step multiplies rows o - 1, o, o + 1 while the waves transform row o + 2), raw input rows arrive two at a time by LDS-DMA.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 tools/micro/wino1d_probe.hip -o tools/micro/bin/wino1d_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(2))) _Float16 half2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
typedef __attribute__((ext_vector_type(16))) float float16_t;
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;

// a - b on packed halves as ONE v_pk_fma_f16 with a -1 the compiler cannot see through (see wino_probe.hip)
#define hsub8(a_, b_) __builtin_elementwise_fma((b_), neg1_8, (a_))

#ifndef WP_PIN
#define WP_PIN 1
#endif
#ifndef WP_FILL
#define WP_FILL 5          // VALU / SALU instructions pinned behind each MFMA
#endif
#ifndef WP_XCD
#define WP_XCD 1
#endif
#ifndef WP_EVEN
#define WP_EVEN 1          // 1: the micro-ops dealt evenly over a row's twelve groups of four MFMAs | 0: front-loaded (the first v2 run)
#endif
#ifndef WP_AHEAD
#define WP_AHEAD 3         // B fragments read this many MFMA slots ahead
#endif

constexpr int PW = 68, ROWB = PW * 128, BLKB = 2 * ROWB;      // a raw block = two input rows of 66 (+ 2 unused) pixels = 17,408 bytes = 17 pieces
constexpr int OFF_V = 3 * BLKB;                                // 52,224: three raw blocks
constexpr int VROW = 4 * 4096;                                 // one V row: four positions x 32 tiles x 128 B
constexpr int OFF_DUMP = OFF_V + 6 * VROW;                     // 150,528: six V rows
constexpr int OFF_BIAS = OFF_DUMP + 1024;                      // [wave 4][hh 2][16] fp32
constexpr int LDS_BYTES = OFF_BIAS + 512;

struct Args {
    const half_t* in;        // [B][H][W][64]
    const half_t* wpk;       // [cout block 8][dy 3][pos 4][ks 4][lane 64][8]
    const float* bias;       // [256]
    const half_t* tailw;     // [256] (checksum weights)
    half_t* out;             // MODE 1: [B][H][W][256] fp16 (PReLU(conv + bias)), else nullptr
    float* chk;              // MODE 0: [grid][256] per-thread checksums
    float slope;
    int B, H, W;
};

template <int N> using ic = std::integral_constant<int, N>;

template <int MODE>
__global__ __launch_bounds__(256) void wino_kernel(Args a)
{
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hh = lane >> 5;
    const int H = a.H, W = a.W;
    const int px = W / 64;
#if WP_XCD
    // the two workgroups of a strip (channel halves) on ONE XCD -- workgroups go round-robin over the 8 XCDs, so blocks b and b + 8 share an L2: the partner's read of the
    // same input rows hits it instead of going out to memory a second time
    const int ch = (blockIdx.x >> 3) & 1;
    const int pair = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7);       // 0 .. grid / 2 - 1
#else
    const int ch = blockIdx.x & 1;
    const int pair = blockIdx.x >> 1;
#endif
    const int cb = ch * 4 + w4;                                 // this wave's block of 32 output channels
    constexpr unsigned kOOR = 0xFFFF0000u;

    // ---- weights: 48 A fragments (dy, position, k-slice) in AGPRs ---------------------------------------------------------------------------------------------------------------
    half8_t wf[12][4];
    {
        const half_t* wsrc = a.wpk + (long long)cb * (48 * 512);
#pragma unroll
        for (int p = 0; p < 12; ++p)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wf[p][ks] = *(const half8_t*)(wsrc + ((p * 4 + ks) * 64 + lane) * 8);
#pragma unroll
        for (int p = 0; p < 12; ++p)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+a"(wf[p][ks]));
    }
    // bias: the initial value of position 1's accumulator (A^T column 1 = (1, 1)); kept in LDS, reloaded where the chain starts
    if (tid < 128) {
        const int bw = tid >> 5, bh = (tid >> 4) & 1, br = tid & 15;
        *(float*)(smem + OFF_BIAS + tid * 4) = a.bias[(ch * 4 + bw) * 32 + 8 * (br >> 2) + 4 * bh + (br & 3)];
    }
    const unsigned bias_ad = lds0 + (unsigned)(OFF_BIAS + (w4 * 2 + hh) * 64);
    half2_t tw2[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { tw2[r][0] = a.tailw[cb * 32 + 8 * (r >> 1) + 4 * hh + 2 * (r & 1)]; tw2[r][1] = a.tailw[cb * 32 + 8 * (r >> 1) + 4 * hh + 2 * (r & 1) + 1]; }
    const half2_t slope2 = {(half_t)a.slope, (half_t)a.slope};
    unsigned neg1_u = 0xBC00BC00u;
    asm volatile("" : "+v"(neg1_u));
    const half2_t neg1_2 = __builtin_bit_cast(half2_t, neg1_u);
    const half8_t neg1_8 = __builtin_shufflevector(neg1_2, neg1_2, 0, 1, 0, 1, 0, 1, 0, 1);

    // ---- B fragment of (V row slot, position p, k-slice ks): tile n at n * 128, 16-byte slot (2 ks + hh) ^ ((n >> 1) & 7): one base per k-slice, the rest immediates ---------
    unsigned fa4[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fa4[ks] = (lds0 + (unsigned)OFF_V + (unsigned)(n * 128 + ((hh ^ ((n >> 1) & 7)) << 4))) ^ (unsigned)(ks << 5);
    // ---- transform thread: tile tx = tid >> 3, channels 8 c8 .. + 7; raw column cc at line cc ^ ((cc >> 1) & 1) (no bank conflicts at a stride of two pixels: wino_probe.hip) ----
    const int tx = tid >> 3, c8 = tid & 7;
    const unsigned va = lds0 + (unsigned)OFF_V + (unsigned)(tx * 128 + ((c8 ^ ((tx >> 1) & 7)) << 4));
    unsigned ra[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { const int cc = 2 * tx + c; ra[c] = lds0 + (unsigned)((cc ^ ((cc >> 1) & 1)) * 128 + (c8 << 4)); }

    // ---- input descriptor (shifted so that block origins are non-negative offsets) -------------------------------------------------------------------------------------------
    const unsigned in_pad = (unsigned)(4 * W + 1) * 128u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in - in_pad), 0, (unsigned)a.B * H * W * 128u + in_pad, 0x00020000);
    unsigned d_off[5];
    auto piece_offsets = [&](int x0) {
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            const unsigned q = (unsigned)((m < 4 ? w4 + 4 * m : 16) * 8 + (lane >> 3));
            const unsigned r = q / 68u, lc = q - 68u * r;
            const unsigned cc = lc ^ ((lc >> 1) & 1u);
            const unsigned sl = (unsigned)(lane & 7);
            const bool ok = cc < 66u && (unsigned)(x0 - 1 + (int)cc) < (unsigned)W && (m < 4 || w4 == 0);
            d_off[m] = ok ? (((r * (unsigned)W + cc) << 7) | (sl << 4)) : kOOR;
        }
    };
    float chk = 0.f;

    for (int item = pair; item < a.B * px; item += gridDim.x >> 1) {
        const int b = item / px, x0 = (item - b * px) * 64;
        piece_offsets(x0);
        // raw block k = input rows 2k, 2k + 1 (columns x0 - 1 .. x0 + 64) into raw slot (k + 3) % 3; blocks / columns outside the image: zeros
        auto dma_piece = [&](int k, auto M_) __attribute__((always_inline)) {
            constexpr int m = decltype(M_)::value;
            const unsigned org = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((b * H + 2 * k + 4) * W + x0) * 128u));
            const unsigned slot = (unsigned)__builtin_amdgcn_readfirstlane(((k + 3) % 3) * BLKB);
            const bool inside = k >= 0 && 2 * k < H;
            const bool mine = m < 4 || w4 == 0;
            const unsigned dsto = (unsigned)__builtin_amdgcn_readfirstlane((int)(m < 4 ? slot + (w4 + 4 * m) * 1024 : (mine ? slot + 16 * 1024 : OFF_DUMP)));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + dsto), 16, inside ? d_off[m] : kOOR, org, 0, 0);
        };
        auto dma_block = [&](int k) { dma_piece(k, ic<0>{}); dma_piece(k, ic<1>{}); dma_piece(k, ic<2>{}); dma_piece(k, ic<3>{}); dma_piece(k, ic<4>{}); };
        // V row y (slot (y + 6) % 6) from raw row y: the thread's tile reads raw columns 2 tx .. 2 tx + 3
        half8_t xd[4];
        auto x_load = [&](int rslot, int rrow) __attribute__((always_inline)) {
            const unsigned rb = (unsigned)(rslot * BLKB + rrow * ROWB);
#pragma unroll
            for (int c = 0; c < 4; ++c) xd[c] = *(lds_h8_t)(ra[c] + rb);
        };
        auto x_put = [&](int vslot) __attribute__((always_inline)) {
            const unsigned vb = va + (unsigned)(vslot * VROW);
            *(__attribute__((address_space(3))) half8_t*)(vb) = hsub8(xd[0], xd[2]);
            *(__attribute__((address_space(3))) half8_t*)(vb + 4096u) = xd[1] + xd[2];
            *(__attribute__((address_space(3))) half8_t*)(vb + 8192u) = hsub8(xd[2], xd[1]);
            *(__attribute__((address_space(3))) half8_t*)(vb + 12288u) = hsub8(xd[1], xd[3]);
        };

        // ---- prologue.  Double-step d computes output rows 2d - 1, 2d from V rows 2d - 2 .. 2d + 1 and transforms rows 2d + 2, 2d + 3 (raw block d + 1) meanwhile; the DMA
        // of block d + 2 is issued at its start and waited for at its end.  Before d = 0: V rows -2 .. 1 (raw blocks -1, 0), raw block 1 landed ---------------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // everybody has left the previous strip
        dma_block(-1); dma_block(0); dma_block(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        x_load(2, 0); x_put(4);                                 // row -2: block -1 (slot 2), row 0 -> V slot 4
        x_load(2, 1); x_put(5);
        x_load(0, 0); x_put(0);
        x_load(0, 1); x_put(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        float16_t Y0, Y1;                                       // the previous row's two outputs per tile, awaiting their epilogue
        const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        Y0 = Y1 = zero16;
        half8_t fr[WP_AHEAD + 1];
        unsigned pk[8];
        // epilogue of half hf of output j of row oe: PReLU in packed fp16, then the stand-in for the tail conv (checksum) or the store
        auto op_e = [&](int oe, auto J_, auto HF_) __attribute__((always_inline)) {
            constexpr int j = decltype(J_)::value, hf = decltype(HF_)::value;
            const float16_t& Yv = j ? Y1 : Y0;
#pragma unroll
            for (int r = 4 * hf; r < 4 * hf + 4; ++r) {
                const half2_t v = {(half_t)Yv[2 * r], (half_t)Yv[2 * r + 1]};
                const half2_t m = __builtin_elementwise_max(v, v * slope2);
                pk[r] = __builtin_bit_cast(unsigned, m);
                if (MODE == 0) chk = __builtin_amdgcn_fdot2(m, tw2[r], chk, false);
            }
            if (MODE == 1 && hf == 1) {
                const int ox = x0 + 2 * n + j;
                if (oe >= 0 && oe < H) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const u2_t v2 = {pk[2 * qd], pk[2 * qd + 1]};
                        *(u2_t*)(a.out + (((long long)(b * H + oe) * W + ox) * 256 + cb * 32 + 8 * qd + 4 * hh)) = v2;
                    }
                }
            }
        };

        // ---- one double-step: 2 x 48 MFMAs, the four position chains of a row interleaved (dy-major, then k-slices, then p: a dependent MFMA is the fourth one issued behind
        // its predecessor); ONE barrier.  What rides behind each group of four MFMAs: the output transform and the epilogue of the previous row, the transforms of rows
        // 2d + 2, 2d + 3, and first of all the DMA of block d + 2
        float16_t T[4];
        T[0] = T[1] = T[2] = T[3] = zero16;
        auto dstep = [&](int d, auto U_) __attribute__((always_inline)) {
            constexpr int u = decltype(U_)::value;              // d % 3: V row 2d + c lives in slot (2u + c + 6) % 6, raw block d + c in slot (u + c) % 3
            auto frag = [&](int s) {                            // MFMA slot s = 0..95 = (row half, dy, ks, p): row half 0 = output row 2d - 1, 1 = row 2d
                const int rh = s / 48, r = s % 48, dy = r / 16, ks = (r >> 2) & 3, pp = r & 3;
                const int vslot = (2 * u + rh - 2 + dy + 6) % 6;      // V row (2d - 1 + rh) - 1 + dy
                return *(lds_h8_t)(fa4[ks] + (unsigned)(vslot * VROW + pp * 4096));
            };
#pragma unroll
            for (int i = 0; i < WP_AHEAD; ++i) fr[i] = frag(i);
            float16_t acc[4];
            auto init_acc = [&]() __attribute__((always_inline)) {
                acc[0] = acc[2] = acc[3] = zero16;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4_t bq = *(const __attribute__((address_space(3))) float4_t*)(bias_ad + (unsigned)(q4 * 16));
                    acc[1][4 * q4] = bq[0]; acc[1][4 * q4 + 1] = bq[1]; acc[1][4 * q4 + 2] = bq[2]; acc[1][4 * q4 + 3] = bq[3];
                }
            };
            auto sub = [&](auto Q_) __attribute__((always_inline)) {      // Q = 0..23: (row half, dy, ks): 4 MFMAs, one per position
                constexpr int q = decltype(Q_)::value, rh = q / 12, ql = q % 12;
                if (ql == 0) init_acc();
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) {
                    const int sidx = q * 4 + pp;
                    acc[pp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[(ql >> 2) * 4 + pp][ql & 3], fr[sidx % (WP_AHEAD + 1)], acc[pp], 0, 0, 0);
                    if (sidx + WP_AHEAD < 96) fr[(sidx + WP_AHEAD) % (WP_AHEAD + 1)] = frag(sidx + WP_AHEAD);
                }
                // ---- what rides behind these four MFMAs ---------------------------------------------------------------------------------------------------------------------
                const int oprev = 2 * d - 2 + rh;                               // the row whose chains closed last
#if WP_EVEN
                // (one unit of ~16 VALU instructions behind each group of four MFMAs)
                if (ql == 0) { Y0 = T[0] + T[1]; }
                if (ql == 1) { Y0 = Y0 + T[2]; }
                if (ql == 2) { Y1 = T[1] - T[2]; }
                if (ql == 3) { Y1 = Y1 - T[3]; }
                if (ql == 4) op_e(oprev, ic<0>{}, ic<0>{});
                if (ql == 5) op_e(oprev, ic<0>{}, ic<1>{});
                if (ql == 6) op_e(oprev, ic<1>{}, ic<0>{});
                if (ql == 7) op_e(oprev, ic<1>{}, ic<1>{});
                if (ql == 8) x_load((u + 1) % 3, rh);                           // raw row 2d + 2 + rh = block d + 1, row rh
                if (ql == 9) x_put((2 * u + 2 + rh + 6) % 6);
                if (rh == 0 && ql >= 7) {                                       // block d + 2 into the raw slot of block d - 1 (transformed in double-step d - 2)
                    if (ql == 7) dma_piece(d + 2, ic<0>{});
                    if (ql == 8) dma_piece(d + 2, ic<1>{});
                    if (ql == 9) dma_piece(d + 2, ic<2>{});
                    if (ql == 10) dma_piece(d + 2, ic<3>{});
                    if (ql == 11) dma_piece(d + 2, ic<4>{});
                }
#else
                if (ql == 0) { Y0 = (T[0] + T[1]) + T[2]; }
                if (ql == 1) { Y1 = (T[1] - T[2]) - T[3]; }
                if (ql == 2) op_e(oprev, ic<0>{}, ic<0>{});
                if (ql == 3) op_e(oprev, ic<0>{}, ic<1>{});
                if (ql == 4) op_e(oprev, ic<1>{}, ic<0>{});
                if (ql == 5) op_e(oprev, ic<1>{}, ic<1>{});
                if (ql == 6) x_load((u + 1) % 3, rh);                           // raw row 2d + 2 + rh = block d + 1, row rh
                if (ql == 7) x_put((2 * u + 2 + rh + 6) % 6);
                if (rh == 0 && ql < 5) {                                        // block d + 2 into the raw slot of block d - 1 (transformed in double-step d - 2)
                    if (ql == 0) dma_piece(d + 2, ic<0>{});
                    if (ql == 1) dma_piece(d + 2, ic<1>{});
                    if (ql == 2) dma_piece(d + 2, ic<2>{});
                    if (ql == 3) dma_piece(d + 2, ic<3>{});
                    if (ql == 4) dma_piece(d + 2, ic<4>{});
                }
#endif
                if (ql == 11) { T[0] = acc[0]; T[1] = acc[1]; T[2] = acc[2]; T[3] = acc[3]; }
#if WP_PIN
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x080, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x006, WP_FILL, 0);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            };
            sub(ic<0>{}); sub(ic<1>{}); sub(ic<2>{}); sub(ic<3>{}); sub(ic<4>{}); sub(ic<5>{}); sub(ic<6>{}); sub(ic<7>{}); sub(ic<8>{}); sub(ic<9>{}); sub(ic<10>{}); sub(ic<11>{});
            sub(ic<12>{}); sub(ic<13>{}); sub(ic<14>{}); sub(ic<15>{}); sub(ic<16>{}); sub(ic<17>{}); sub(ic<18>{}); sub(ic<19>{}); sub(ic<20>{}); sub(ic<21>{}); sub(ic<22>{}); sub(ic<23>{});
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        const int nd = H / 2 + 1;                               // double-steps 0 .. H/2 (rows -1 and H are discarded)
        for (int d = 0; d < nd; d += 3) {
            dstep(d, ic<0>{});
            if (d + 1 < nd) dstep(d + 1, ic<1>{});
            if (d + 2 < nd) dstep(d + 2, ic<2>{});
        }
        Y0 = (T[0] + T[1]) + T[2];
        Y1 = (T[1] - T[2]) - T[3];
        op_e(H, ic<0>{}, ic<0>{}); op_e(H, ic<0>{}, ic<1>{}); op_e(H, ic<1>{}, ic<0>{}); op_e(H, ic<1>{}, ic<1>{});      // (row H: outside the image)
    }
    if (MODE == 0) a.chk[blockIdx.x * 256 + tid] = chk;
}

// ---- host ------------------------------------------------------------------------------------------------------------------------------------------------------------
static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.f * 2.f - 1.f; }

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 96, H = argc > 2 ? atoi(argv[2]) : 512, W = argc > 3 ? atoi(argv[3]) : 512, reps = argc > 4 ? atoi(argv[4]) : 20;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    hipFuncSetAttribute((const void*)wino_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)wino_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    // weights g[cout 256][cin 64][3][3], U = G g G^T in fp32, rounded to fp16, packed as A fragments
    unsigned seed = 12345;
    std::vector<float> g(256 * 64 * 9), bias(256);
    for (auto& v : g) v = frand(seed) * 0.06f;
    for (auto& v : bias) v = frand(seed) * 0.1f;
    std::vector<half_t> wpk((size_t)8 * 48 * 512), tailw(256);
    for (auto& v : tailw) v = (half_t)(frand(seed) * 0.1f);
    static const float G[4][3] = {{1, 0, 0}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0, 0, 1}};
    std::vector<float> U((size_t)256 * 64 * 12);                   // U[o][c][dy][p] = sum_k g[o][c][dy][k] G[p][k]
    for (int o = 0; o < 256; ++o)
        for (int c = 0; c < 64; ++c) {
            const float* gg = &g[((size_t)o * 64 + c) * 9];
            for (int dy = 0; dy < 3; ++dy) for (int p = 0; p < 4; ++p) U[((size_t)o * 64 + c) * 12 + dy * 4 + p] = G[p][0] * gg[dy * 3 + 0] + G[p][1] * gg[dy * 3 + 1] + G[p][2] * gg[dy * 3 + 2];
        }
    for (int cb = 0; cb < 8; ++cb)
        for (int p = 0; p < 12; ++p)
            for (int ks = 0; ks < 4; ++ks)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e)
                        wpk[((((size_t)cb * 12 + p) * 4 + ks) * 64 + l) * 8 + e] = (half_t)U[((size_t)(cb * 32 + (l & 31)) * 64 + ks * 16 + 8 * (l >> 5) + e) * 12 + p];
    half_t *d_w, *d_tw; float* d_bias;
    hipMalloc(&d_w, wpk.size() * 2); hipMalloc(&d_tw, 512); hipMalloc(&d_bias, 1024);
    hipMemcpy(d_w, wpk.data(), wpk.size() * 2, hipMemcpyHostToDevice); hipMemcpy(d_tw, tailw.data(), 512, hipMemcpyHostToDevice); hipMemcpy(d_bias, bias.data(), 1024, hipMemcpyHostToDevice);
    const float slope = 0.2f;

    // ---- validation: 2 planes of 24 x 64 against the direct convolution (fp32 on the fp16 inputs) ---------------------------------------------------------------------------
    {
        const int vB = 2, vH = 24, vW = 128;
        std::vector<half_t> x((size_t)vB * vH * vW * 64);
        for (auto& v : x) v = (half_t)frand(seed);
        half_t *d_x, *d_o;
        hipMalloc(&d_x, x.size() * 2 + 4096); hipMalloc(&d_o, (size_t)vB * vH * vW * 256 * 2);
        hipMemcpy(d_x, x.data(), x.size() * 2, hipMemcpyHostToDevice);
        hipMemset(d_o, 0, (size_t)vB * vH * vW * 256 * 2);
        Args a{d_x, d_w, d_bias, d_tw, d_o, nullptr, slope, vB, vH, vW};
        wino_kernel<1><<<dim3(16), dim3(256), LDS_BYTES>>>(a);
        if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "validation launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        std::vector<half_t> o((size_t)vB * vH * vW * 256);
        hipMemcpy(o.data(), d_o, o.size() * 2, hipMemcpyDeviceToHost);
        double worst = 0, big = 0;
        long long bad = 0;
        for (int b = 0; b < vB; ++b)
            for (int y = 0; y < vH; ++y)
                for (int xx = 0; xx < vW; ++xx)
                    for (int oc = 0; oc < 256; oc += 3) {
                        double s = bias[oc];
                        for (int dy = 0; dy < 3; ++dy)
                            for (int dx = 0; dx < 3; ++dx) {
                                const int iy = y + dy - 1, ix = xx + dx - 1;
                                if (iy < 0 || iy >= vH || ix < 0 || ix >= vW) continue;
                                const half_t* xp = &x[(((size_t)b * vH + iy) * vW + ix) * 64];
                                for (int c = 0; c < 64; ++c) s += (double)(float)xp[c] * g[((size_t)oc * 64 + c) * 9 + dy * 3 + dx];
                            }
                        const double want = s > 0 ? s : s * slope;
                        const double got = (float)o[(((size_t)b * vH + y) * vW + xx) * 256 + oc];
                        const double e = std::fabs(got - want);
                        if (e > worst) worst = e;
                        if (std::fabs(want) > big) big = std::fabs(want);
                        if (e > 0.02) ++bad;
                    }
        printf("validation (2 x 24 x 128, every third channel): max-abs error %.3e against the direct conv (|y| <= %.2f), %lld values off by more than 0.02 -> %s\n", worst, big, bad, bad == 0 ? "OK" : "WRONG");
        hipFree(d_x); hipFree(d_o);
        if (bad) return 2;
    }

    // ---- timing at the U-up1 shape --------------------------------------------------------------------------------------------------------------------------------------
    std::vector<half_t> x((size_t)B * H * W * 64);
    {
        unsigned s2 = 777;
        for (size_t i = 0; i < x.size(); ++i) x[i] = (half_t)(frand(s2) * 0.7f);
    }
    half_t* d_x; float* d_chk;
    hipMalloc(&d_x, x.size() * 2 + 4096); hipMalloc(&d_chk, (size_t)cus * 256 * 4);
    hipMemcpy(d_x, x.data(), x.size() * 2, hipMemcpyHostToDevice);
    Args a{d_x, d_w, d_bias, d_tw, nullptr, d_chk, slope, B, H, W};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) wino_kernel<0><<<dim3(cus), dim3(256), LDS_BYTES>>>(a);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) wino_kernel<0><<<dim3(cus), dim3(256), LDS_BYTES>>>(a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flop = 2.0 * B * H * W * 256.0 * 64 * 9;           // the conv's algorithmic FLOPs
    const double mfma = (double)B * (W / 64) * 2 * (H + 2) * 4 * 48;      // MFMAs executed (per wave: 48 a row, H + 2 rows)
    const double peak = cus * 4.0 * 1024 * prop.clockRate * 1e3 / 1e12;
    std::vector<float> hc((size_t)cus * 256);
    hipMemcpy(hc.data(), d_chk, hc.size() * 4, hipMemcpyDeviceToHost);
    double cs = 0; for (float v : hc) cs += v;
    printf("wino1d_probe: %d planes of %d x %d: %.3f ms per launch (%d reps) = %.0f TFLOP/s algorithmic = %.3f of the nominal fp16 peak %.0f; MFMA time at 2.4 GHz %.3f ms -> busy x clock/2.4 = %.3f; checksum %.6e\n",
           B, H, W, ms, reps, flop / ms / 1e9, flop / ms / 1e9 / peak, peak, mfma * 32 / (cus * 4.0) / 2.4e9 * 1e3 * (1.0), mfma * 32 / (cus * 4.0) / 2.4e9 * 1e3 / ms, cs);
    printf("reference: conv3x3_ps4<1> takes 5.30 ms for the same 96-plane launch (profiles/r05), the gate of VERDICT r05 item 1 is 0.8 x 5.30 = 4.24 ms\n");
    return 0;
}
