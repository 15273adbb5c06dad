// wino_probe.hip -- VERDICT r05 item 1(a): can a Winograd F(2x2, 3x3) form of the U branch's last up-conv (3x3, 64 -> 256 + bias, PReLU; python/models.py:29-36,145-154 of the
// reference) beat conv3x3_ps4<1> under the package power cap?  2.25x fewer MFMAs (16 position products per 2x2 output tile instead of 36 tap products), paid for with an
// input transform (fp16), an fp32 output transform and one B-fragment read per MFMA.  This is a MEASUREMENT PROBE (tools/, not product): random data at the U-up1 shape
// (96 planes of 512 x 512 x 64 channels), the conv + bias + PReLU computed in full and verified against a direct convolution on a small shape; the 64 -> 1 tail conv of
// the product kernel is represented by one v_dot2 per pair of outputs (its real form costs 4 MFMAs per 72 in conv3x3_ps4).
//
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 tools/micro/wino_probe.hip -o tools/micro/bin/wino_probe
//   tools/micro/bin/wino_probe [planes 96] [H 512] [W 512] [reps 20]
//
// Form.  One workgroup = 4 waves (one per SIMD, 512 registers each) = 128 of the 256 output channels (half `ch` = blockIdx & 1); wave w holds the transformed weights
// U[pos 16][cin 64] of ITS 32 output channels as 64 A fragments of v_mfma_f32_32x32x16_f16 = 256 registers (the whole AGPR half), loaded once.  (All 256 channels would
// need 512 KB of registers = a whole CU's file: two workgroups share an input strip, each transforms it -- the price of weights that never move.)
// A workgroup walks DOWN a 64-pixel column strip of a plane; a step = one row of 32 tiles of 2x2 outputs = output rows 2t-1, 2t, from input rows
// 2t-2 .. 2t+1 (x 66 columns).  Input rows arrive in blocks of two rows by LDS-DMA (17 one-KiB pieces, the swizzle of conv3x3_ps4.hip) into a ring of three blocks.
// V = B^T d B (fp16, v_pk_add_f16) is formed by the four waves together -- thread (tile, 8-channel group) -- and written to LDS as B-fragment images
// [position][tile][64 ch], in two halves (position rows {0,1} and {2,3}) that ping-pong: while the waves multiply the half of one position-row pair, they
// transform the next pair.  Per half-step and wave: 32 MFMAs (2 position rows x 4 positions x 4 k-slices), each with one ds_read_b128; the four products of a
// position row are column-transformed in fp32 (T0 = M0 + M1 + M2, T1 = M1 - M2 - M3) and accumulated into the 2x2 outputs (Y0 = T(0) + T(1) + T(2), Y1 = T(1) - T(2) - T(3)).
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

// a - b on packed halves: hipcc expands `fsub <2 x half>` into v_sub_f16 + v_sub_f16_sdwa + v_pack_b32_f16 (three instructions per pair) and folds fma(b, -1, a) back into
// that; with a -1 it cannot see through, the subtraction is ONE v_pk_fma_f16 per pair (same single rounding) -- and a real VALU instruction for sched_group_barrier,
// which inline assembly is not
#define hsub8(a_, b_) __builtin_elementwise_fma((b_), neg1_8, (a_))

#ifndef WP_PIN
#define WP_PIN 1
#endif
#ifndef WP_PK32
#define WP_PK32 0          // 1: the fp32 subtractions of the output transform as v_pk_fma_f32 with an opaque -1 (hipcc packs fp32 adds by itself, never subtractions)
#endif
#if WP_PK32
#define fsub16(a_, b_) __builtin_elementwise_fma((b_), neg1f_16, (a_))
#else
#define fsub16(a_, b_) ((a_) - (b_))
#endif
#ifndef WP_HALF
#define WP_HALF 12         // VALU / SALU instructions pinned between the two MFMAs of a pair (the rest follows the second)
#endif

constexpr int PW = 68, ROWB = PW * 128, BLKB = 2 * ROWB;      // a block = two input rows of 66 (+ 2 unused) pixels = 17,408 bytes = 17 pieces
constexpr int OFF_V = 3 * BLKB;                                // 52,224
constexpr int VHALF = 8 * 4096;                                // eight positions x 32 tiles x 128 B
constexpr int OFF_DUMP = OFF_V + 2 * VHALF;                    // 117,760
constexpr int OFF_BIAS = OFF_DUMP + 1024;                      // [wave 4][hh 2][16] fp32
constexpr int LDS_BYTES = OFF_BIAS + 512;

struct Args {
    const half_t* in;        // [B][H][W][64]
    const half_t* wpk;       // [cout block 8][pos 16][ks 4][lane 64][8]
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
    const int px = W / 64, nsteps = H / 2 + 1;
    const int ch = blockIdx.x & 1;
    const int cb = ch * 4 + w4;                                 // this wave's block of 32 output channels
    constexpr unsigned kOOR = 0xFFFF0000u;

    // ---- weights: 64 A fragments in AGPRs ------------------------------------------------------------------------------------------------------------
    half8_t wf[16][4];
    {
        const half_t* wsrc = a.wpk + (long long)cb * (64 * 512);
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wf[p][ks] = *(const half8_t*)(wsrc + ((p * 4 + ks) * 64 + lane) * 8);
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+a"(wf[p][ks]));
    }
    // bias of the lane's 16 channels (register 4q + e = MFMA row 8q + 4hh + e): the initial value of position (1, 1)'s accumulator -- A^T e11 A = all ones; kept in LDS,
    // reloaded where the chain starts (conv3x3_ps4.hip: as four 16-byte loads)
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
    typedef float float2_t __attribute__((ext_vector_type(2)));
    float2_t neg1f_2 = {-1.f, -1.f};
    asm volatile("" : "+v"(neg1f_2));
    const float16_t neg1f_16 = __builtin_shufflevector(neg1f_2, neg1f_2, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1);
    (void)neg1f_16;

    // ---- B fragment of (position pl of the half, k-slice ks): tile n at n * 128, 16-byte slot (2 ks + hh) ^ ((n >> 1) & 7): one base per k-slice, the rest immediates --------
    unsigned fa4[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fa4[ks] = (lds0 + (unsigned)OFF_V + (unsigned)(n * 128 + ((hh ^ ((n >> 1) & 7)) << 4))) ^ (unsigned)(ks << 5);
    // ---- transform thread: tile tx = tid >> 3, channels 8 c8 .. + 7 ----------------------------------------------------------------------------------------------------------
    const int tx = tid >> 3, c8 = tid & 7;
    const unsigned va = lds0 + (unsigned)OFF_V + (unsigned)(tx * 128 + ((c8 ^ ((tx >> 1) & 7)) << 4));
    unsigned ra[4];                                             // raw pixel column 2 tx + c of a ring row: + slot * BLKB + row * ROWB
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        // the raw ring is read with a stride of TWO pixels (tile tx -> column 2 tx + c): with pixels at their own 128-byte lines, the 16 lanes one LDS cycle serves -- four
        // tiles x four slots -- would all fall into one half of the 64 banks (2-way conflicts on every read: the first run of this probe counted 2.3e8 conflict cycles per
        // launch).  Column cc therefore lives at line cc ^ ((cc >> 1) & 1) (columns 4k+2 and 4k+3 swapped): tiles alternate between the bank halves; no slot swizzle
        const int cc = 2 * tx + c;
        ra[c] = lds0 + (unsigned)((cc ^ ((cc >> 1) & 1)) * 128 + (c8 << 4));
    }

    // ---- input descriptor (conv3x3_ps4.hip: shifted so that block origins are non-negative offsets) -----------------------------------------------------------------------------
    const unsigned in_pad = (unsigned)(4 * W + 1) * 128u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in - in_pad), 0, (unsigned)a.B * H * W * 128u + in_pad, 0x00020000);
    unsigned d_off[5];                                          // piece i = w4 + 4 m (m < 4) / 16 (m = 4): the lane's pixel of the 2 x 68 block, its logical 16-byte slot; kOOR where the
                                                                // pixel's column lies outside the image or the strip's 66 (set per strip); whole blocks outside the image are a uniform test
    auto piece_offsets = [&](int x0) {
#pragma unroll
        for (int m = 0; m < 5; ++m) {
            const unsigned q = (unsigned)((m < 4 ? w4 + 4 * m : 16) * 8 + (lane >> 3));
            const unsigned r = q / 68u, lc = q - 68u * r;
            const unsigned cc = lc ^ ((lc >> 1) & 1u);              // the image column (relative to x0 - 1) that lives at line lc of the row
            const unsigned sl = (unsigned)(lane & 7);
            const bool ok = cc < 66u && (unsigned)(x0 - 1 + (int)cc) < (unsigned)W && (m < 4 || w4 == 0);
            d_off[m] = ok ? (((r * (unsigned)W + cc) << 7) | (sl << 4)) : kOOR;
        }
    };
    float chk = 0.f;

    for (int item = blockIdx.x >> 1; item < a.B * px; item += gridDim.x >> 1) {
        const int b = item / px, x0 = (item - b * px) * 64;
        piece_offsets(x0);
        auto dma_block = [&](int k) {                           // input rows 2k, 2k+1, columns x0 - 1 .. x0 + 64, into ring slot (k + 1) % 3; rows / columns outside the image: zeros
            const int ya = 2 * k, xa = x0 - 1;
            const unsigned org = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((b * H + ya + 4) * W + xa + 1) * 128u));
            const unsigned slot = (unsigned)__builtin_amdgcn_readfirstlane(((k + 1) % 3) * BLKB);
            const bool inside = k >= 0 && 2 * k < H;
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                const bool mine = m < 4 || w4 == 0;
                const unsigned dsto = (unsigned)__builtin_amdgcn_readfirstlane((int)(m < 4 ? slot + (w4 + 4 * m) * 1024 : (mine ? slot + 16 * 1024 : OFF_DUMP)));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + dsto), 16, inside ? d_off[m] : kOOR, org, 0, 0);
            }
        };
        // Step t = output rows 2t - 1, 2t of the strip (32 tiles of 2 x 2) from input rows 2t - 2 .. 2t + 1 = blocks t - 1 (tile rows r = 0, 1) and t (r = 2, 3).
        // Transform of position rows {2 q, 2 q + 1} of step tt into V half vh (q = 0: tile rows 0 1 2, q = 1: rows 1 2 3), in pieces that ride in the MFMA stream:
        half8_t xd[3], xt0[4], xt1[4];
        unsigned x_row[3], x_base = 0;
        auto x_setup = [&](int tt, auto Q_, int vh) __attribute__((always_inline)) {
            constexpr int q = decltype(Q_)::value;
            const unsigned sA = (unsigned)__builtin_amdgcn_readfirstlane((tt % 3) * BLKB);            // slot of block tt - 1
            const unsigned sB = (unsigned)__builtin_amdgcn_readfirstlane(((tt + 1) % 3) * BLKB);      // slot of block tt
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) { const int r = rr + q; x_row[rr] = r < 2 ? sA + (unsigned)(r * ROWB) : sB + (unsigned)((r - 2) * ROWB); }
            x_base = va + (unsigned)(vh * VHALF);
        };
        auto x_load = [&](auto C_) __attribute__((always_inline)) {
            constexpr int c = decltype(C_)::value;
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) xd[rr] = *(lds_h8_t)(ra[c] + x_row[rr]);
        };
        auto x_rowpass = [&](auto C_, auto Q_) __attribute__((always_inline)) {
            constexpr int c = decltype(C_)::value, q = decltype(Q_)::value;
            if (q == 0) { xt0[c] = hsub8(xd[0], xd[2]); xt1[c] = xd[1] + xd[2]; }      // B^T rows 0, 1
            else { xt0[c] = hsub8(xd[1], xd[0]); xt1[c] = hsub8(xd[0], xd[2]); }       // rows 2 (d2 - d1), 3 (d1 - d3) with xd[0] = d1, xd[1] = d2, xd[2] = d3
        };
        auto x_put = [&](auto K_) __attribute__((always_inline)) {
            constexpr int k = decltype(K_)::value;                // k = 0..3: positions (2k, 2k + 1) of the half
            const half8_t* tt_ = k < 2 ? xt0 : xt1;
            half8_t v0, v1;
            if ((k & 1) == 0) { v0 = hsub8(tt_[0], tt_[2]); v1 = tt_[1] + tt_[2]; }
            else { v0 = hsub8(tt_[2], tt_[1]); v1 = hsub8(tt_[1], tt_[3]); }
            *(__attribute__((address_space(3))) half8_t*)(x_base + (unsigned)((2 * k) * 4096)) = v0;
            *(__attribute__((address_space(3))) half8_t*)(x_base + (unsigned)((2 * k + 1) * 4096)) = v1;
        };

        // ---- prologue: blocks -1, 0, 1; V half 0 = position rows {0, 1} of step 0 ---------------------------------------------------------------------------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // everybody has left the previous strip
        dma_block(-1); dma_block(0); dma_block(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        x_setup(0, ic<0>{}, 0);
        x_load(ic<0>{}); x_rowpass(ic<0>{}, ic<0>{}); x_load(ic<1>{}); x_rowpass(ic<1>{}, ic<0>{}); x_load(ic<2>{}); x_rowpass(ic<2>{}, ic<0>{}); x_load(ic<3>{}); x_rowpass(ic<3>{}, ic<0>{});
        x_put(ic<0>{}); x_put(ic<1>{}); x_put(ic<2>{}); x_put(ic<3>{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        float16_t Y[2][2];                                      // [i][j]: the 2x2 outputs of the lane's tile, 16 channels each
        float16_t M[4], T0, T1;
        const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        Y[0][0] = Y[0][1] = Y[1][0] = Y[1][1] = zero16; M[1] = M[2] = M[3] = T0 = T1 = zero16;
        half8_t fr[3];                                          // fragment of MFMA slot u in fr[u % 3], read two slots ahead

        // ---- micro-ops: units of ~16 VALU instructions that ride behind the MFMAs ---------------------------------------------------------------------------------------------
        // epilogue of half `hf` (registers 8 hf .. 8 hf + 7) of output (i, j) of step te: PReLU in packed fp16 (slope < 1: max(x, slope x)), then the stand-in for the tail conv
        unsigned pk[8];
        auto op_e = [&](int te, auto I_, auto J_, auto HF_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value, j = decltype(J_)::value, hf = decltype(HF_)::value;
#pragma unroll
            for (int r = 4 * hf; r < 4 * hf + 4; ++r) {
                const half2_t v = {(half_t)Y[i][j][2 * r], (half_t)Y[i][j][2 * r + 1]};
                const half2_t m = __builtin_elementwise_max(v, v * slope2);
                pk[r] = __builtin_bit_cast(unsigned, m);
                if (MODE == 0) chk = __builtin_amdgcn_fdot2(m, tw2[r], chk, false);
            }
            if (MODE == 1 && hf == 1) {
                const int oy = 2 * te - 1 + i, ox = x0 + 2 * n + j;
                if (oy >= 0 && oy < H) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {            // channels cb * 32 + 8 qd + 4 hh .. + 3
                        const u2_t v2 = {pk[2 * qd], pk[2 * qd + 1]};
                        *(u2_t*)(a.out + (((long long)(b * H + oy) * W + ox) * 256 + cb * 32 + 8 * qd + 4 * hh)) = v2;
                    }
                }
            }
        };
        enum : int { E00A, E00B, E01A, E01B, E10A, E10B, E11A, E11B, B1, B2, B3, CA, Y0A, Y0B, Y1A, Y1B, XS, XR0, XR1, XR2, XR3, XL0, XL1, XL2, XL3, XP0, XP1, XP2, XP3, DM0, DM1, DM2, DM3, DM4, NOPS };
        // the output transform of a position row whose four products M[0..3] are complete:  T0 = M0 + M1 + M2,  T1 = M1 - M2 - M3:  CA one pair of chains behind pair (0, 1);
        // B1 B2 B3 and the Y updates behind pair (2, 3), in the next position row's first chunk
        unsigned dma_org = 0, dma_slot = 0;
        bool dma_inside = false;
        auto run_op = [&](int t, auto HS_, auto S_, auto OP_) __attribute__((always_inline)) {
            constexpr int HS = decltype(HS_)::value, sidx = decltype(S_)::value, op = decltype(OP_)::value;
            constexpr int ppr = (2 * HS + (sidx >> 3) + 3) & 3;               // the position row whose tail rides in this chunk
            constexpr int q = 1 - HS;                                         // the transform in flight: HS 0: rows {2, 3} of step t -> half 1;  HS 1: rows {0, 1} of step t + 1 -> half 0
            if constexpr (op >= E00A && op <= E11B) op_e(t - 1, ic<((op - E00A) >> 2)>{}, ic<(((op - E00A) >> 1) & 1)>{}, ic<((op - E00A) & 1)>{});
            if constexpr (op == B1) T1 = fsub16(M[1], M[2]);
            if constexpr (op == B2) T1 = fsub16(T1, M[3]);
            if constexpr (op == B3) T0 = T0 + M[2];
            if constexpr (op == CA) T0 = M[0] + M[1];
            if constexpr (op == Y0A) { if (ppr == 0) Y[0][0] = T0; else if (ppr <= 2) Y[0][0] += T0; }
            if constexpr (op == Y0B) { if (ppr == 0) Y[0][1] = T1; else if (ppr <= 2) Y[0][1] += T1; }
            if constexpr (op == Y1A) { if (ppr == 1) Y[1][0] = T0; else if (ppr >= 2) Y[1][0] = fsub16(Y[1][0], T0); }
            if constexpr (op == Y1B) { if (ppr == 1) Y[1][1] = T1; else if (ppr >= 2) Y[1][1] = fsub16(Y[1][1], T1); }
            if constexpr (op == XS) x_setup(t + HS, ic<q>{}, 1 - HS);
            if constexpr (op >= XL0 && op <= XL3) x_load(ic<op - XL0>{});
            if constexpr (op >= XR0 && op <= XR3) x_rowpass(ic<op - XR0>{}, ic<q>{});
            if constexpr (op >= XP0 && op <= XP3) x_put(ic<op - XP0>{});
            if constexpr (op >= DM0 && op <= DM4) {
                constexpr int m = op - DM0;
                if (m == 0) {                                   // block t + 2: input rows 2k, 2k+1, columns x0 - 1 .. x0 + 64, into ring slot (k + 1) % 3 (blocks outside the image: zeros)
                    const int k = t + 2;
                    dma_org = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((b * H + 2 * k + 4) * W + x0) * 128u));
                    dma_slot = (unsigned)__builtin_amdgcn_readfirstlane(((k + 1) % 3) * BLKB);
                    dma_inside = 2 * k < H;
                }
                const bool mine = m < 4 || w4 == 0;
                const unsigned dsto = (unsigned)__builtin_amdgcn_readfirstlane((int)(m < 4 ? dma_slot + (w4 + 4 * m) * 1024 : (mine ? dma_slot + 16 * 1024 : OFF_DUMP)));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + dsto), 16, dma_inside ? d_off[m] : kOOR, dma_org, 0, 0);
            }
        };
        // what rides behind MFMA pair s (0..15) of a half-step -- see the dependency notes in the header of this file
        struct Sched { unsigned long long m[2][16]; };
        constexpr auto bit = [](int o) constexpr { return 1ull << o; };
        constexpr Sched SC = {{
            // (t, 0): epilogue of step t - 1 (Y[0] final since (t - 1, 1), Y[1] after the tail of row 3), tails of rows 3 and 0, transform of rows {2, 3} of step t
            {bit(E00A) | bit(XS) | bit(XL0), bit(E00B) | bit(B1), bit(E01A) | bit(B2), bit(E01B) | bit(B3),
             bit(Y1A) | bit(XR0) | bit(XL1), bit(Y1B) | bit(XR1) | bit(XL2), bit(CA) | bit(E10A), bit(E10B) | bit(XR2) | bit(XL3),
             bit(E11A) | bit(XR3), bit(E11B) | bit(B1), bit(B2) | bit(XP0), bit(B3) | bit(XP1),
             bit(Y0A) | bit(Y0B) | bit(XP2), bit(XP3), bit(CA), 0ull},
            // (t, 1): tails of rows 1 and 2, transform of rows {0, 1} of step t + 1, DMA of block t + 2
            {bit(XS) | bit(XL0) | bit(DM0) | bit(DM1), bit(B1) | bit(XR0) | bit(XL1) | bit(DM2), bit(B2) | bit(XR1) | bit(XL2) | bit(DM3), bit(B3) | bit(XR2) | bit(XL3) | bit(DM4),
             bit(Y0A) | bit(Y1A) | bit(XR3), bit(Y0B) | bit(Y1B), bit(CA) | bit(XP0), bit(XP1),
             bit(XP2), bit(B1) | bit(XP3), bit(B2), bit(B3),
             bit(Y0A) | bit(Y1A), bit(Y0B) | bit(Y1B), bit(CA), 0ull}}};

        // ---- one chunk = a PAIR of positions (prl, 2 pp), (prl, 2 pp + 1) of the half = 8 MFMAs, the two chains interleaved (a dependent MFMA is never the next one issued:
        // an instruction between two MFMAs on the SAME accumulator costs ~43 cycles, MI355X_MICROARCH.md); behind every pair of MFMAs its slice of the micro-ops, fenced:
        // program order IS the schedule.  HS: 0 = half-step (t, 0) [position rows 0, 1 from V half 0], 1 = (t, 1) [rows 2, 3 from half 1];  C = 0..3: prl = C >> 1, pp = C & 1
        auto chunk = [&](int t, auto HS_, auto C_) __attribute__((always_inline)) {
            constexpr int HS = decltype(HS_)::value, C = decltype(C_)::value;
            constexpr int prl = C >> 1, pp = C & 1, pr = 2 * HS + prl;
            float16_t acc[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (pr == 1 && 2 * pp + e == 1) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const float4_t bq = *(const __attribute__((address_space(3))) float4_t*)(bias_ad + (unsigned)(q4 * 16));
                        acc[e][4 * q4] = bq[0]; acc[e][4 * q4 + 1] = bq[1]; acc[e][4 * q4 + 2] = bq[2]; acc[e][4 * q4 + 3] = bq[3];
                    }
                } else acc[e] = zero16;
            }
            auto sub = [&](auto KS_) __attribute__((always_inline)) {
                constexpr int ks = decltype(KS_)::value, sidx = C * 4 + ks;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int u = C * 8 + ks * 2 + e;
                    acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[pr * 4 + 2 * pp + e][ks], fr[u % 3], acc[e], 0, 0, 0);
                    const int u2 = u + 2;                       // the fragment two slots ahead (the first two of a half-step are read behind the barrier)
                    if (u2 < 32) {
                        const int c2 = u2 >> 3, ks2 = (u2 >> 1) & 3, e2 = u2 & 1;
                        fr[u2 % 3] = *(lds_h8_t)(fa4[ks2] + (unsigned)(HS * VHALF + ((c2 >> 1) * 4 + 2 * (c2 & 1) + e2) * 4096));
                    }
                }
                constexpr unsigned long long mask = SC.m[HS][sidx];
                auto go = [&](auto OP_) __attribute__((always_inline)) { if constexpr ((mask >> decltype(OP_)::value) & 1ull) run_op(t, HS_, ic<sidx>{}, OP_); };
                go(ic<0>{}); go(ic<1>{}); go(ic<2>{}); go(ic<3>{}); go(ic<4>{}); go(ic<5>{}); go(ic<6>{}); go(ic<7>{}); go(ic<8>{}); go(ic<9>{}); go(ic<10>{}); go(ic<11>{});
                go(ic<12>{}); go(ic<13>{}); go(ic<14>{}); go(ic<15>{}); go(ic<16>{}); go(ic<17>{}); go(ic<18>{}); go(ic<19>{}); go(ic<20>{}); go(ic<21>{}); go(ic<22>{}); go(ic<23>{});
                go(ic<24>{}); go(ic<25>{}); go(ic<26>{}); go(ic<27>{}); go(ic<28>{}); go(ic<29>{}); go(ic<30>{}); go(ic<31>{}); go(ic<32>{}); go(ic<33>{});
#if WP_PIN
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x080, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x006, WP_HALF, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x080, 2, 0);
#endif
                __builtin_amdgcn_sched_barrier(0);
            };
            sub(ic<0>{}); sub(ic<1>{}); sub(ic<2>{}); sub(ic<3>{});
            M[2 * pp] = acc[0]; M[2 * pp + 1] = acc[1];
        };
        auto half_step = [&](int t, auto HS_) __attribute__((always_inline)) {
            constexpr int HS = decltype(HS_)::value;
            fr[0] = *(lds_h8_t)(fa4[0] + (unsigned)(HS * VHALF));
            fr[1] = *(lds_h8_t)(fa4[0] + (unsigned)(HS * VHALF + 4096));
            chunk(t, HS_, ic<0>{}); chunk(t, HS_, ic<1>{}); chunk(t, HS_, ic<2>{}); chunk(t, HS_, ic<3>{});
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        for (int t = 0; t < nsteps; ++t) {
            half_step(t, ic<0>{});
            half_step(t, ic<1>{});
        }
        // the last position row's tail and the last step's epilogue
        T1 = (M[1] - M[2]) - M[3]; T0 = T0 + M[2];
        Y[1][0] -= T0; Y[1][1] -= T1;
        op_e(nsteps - 1, ic<0>{}, ic<0>{}, ic<0>{}); op_e(nsteps - 1, ic<0>{}, ic<0>{}, ic<1>{}); op_e(nsteps - 1, ic<0>{}, ic<1>{}, ic<0>{}); op_e(nsteps - 1, ic<0>{}, ic<1>{}, ic<1>{});
        op_e(nsteps - 1, ic<1>{}, ic<0>{}, ic<0>{}); op_e(nsteps - 1, ic<1>{}, ic<0>{}, ic<1>{}); op_e(nsteps - 1, ic<1>{}, ic<1>{}, ic<0>{}); op_e(nsteps - 1, ic<1>{}, ic<1>{}, ic<1>{});
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
    std::vector<half_t> wpk((size_t)8 * 64 * 512), tailw(256);
    for (auto& v : tailw) v = (half_t)(frand(seed) * 0.1f);
    static const float G[4][3] = {{1, 0, 0}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0, 0, 1}};
    std::vector<float> U((size_t)256 * 64 * 16);
    for (int o = 0; o < 256; ++o)
        for (int c = 0; c < 64; ++c) {
            const float* gg = &g[((size_t)o * 64 + c) * 9];
            float t[4][3];
            for (int i = 0; i < 4; ++i) for (int k = 0; k < 3; ++k) t[i][k] = G[i][0] * gg[0 * 3 + k] + G[i][1] * gg[1 * 3 + k] + G[i][2] * gg[2 * 3 + k];
            for (int i = 0; i < 4; ++i) for (int l = 0; l < 4; ++l) U[((size_t)o * 64 + c) * 16 + i * 4 + l] = t[i][0] * G[l][0] + t[i][1] * G[l][1] + t[i][2] * G[l][2];
        }
    for (int cb = 0; cb < 8; ++cb)
        for (int p = 0; p < 16; ++p)
            for (int ks = 0; ks < 4; ++ks)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e)
                        wpk[((((size_t)cb * 16 + p) * 4 + ks) * 64 + l) * 8 + e] = (half_t)U[((size_t)(cb * 32 + (l & 31)) * 64 + ks * 16 + 8 * (l >> 5) + e) * 16 + p];
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
        wino_kernel<1><<<dim3(8), dim3(256), LDS_BYTES>>>(a);
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
    const double mfma = (double)B * (W / 64) * 2 * (H / 2 + 1) * 4 * 64;      // MFMAs executed (per wave: 64 a step)
    const double peak = cus * 4.0 * 1024 * prop.clockRate * 1e3 / 1e12;
    std::vector<float> hc((size_t)cus * 256);
    hipMemcpy(hc.data(), d_chk, hc.size() * 4, hipMemcpyDeviceToHost);
    double cs = 0; for (float v : hc) cs += v;
    printf("wino_probe: %d planes of %d x %d: %.3f ms per launch (%d reps) = %.0f TFLOP/s algorithmic = %.3f of the nominal fp16 peak %.0f; MFMA time at 2.4 GHz %.3f ms -> busy x clock/2.4 = %.3f; checksum %.6e\n",
           B, H, W, ms, reps, flop / ms / 1e9, flop / ms / 1e9 / peak, peak, mfma * 32 / (cus * 4.0) / 2.4e9 * 1e3 * (1.0), mfma * 32 / (cus * 4.0) / 2.4e9 * 1e3 / ms, cs);
    printf("reference: conv3x3_ps4<1> takes 5.30 ms for the same 96-plane launch (profiles/r05), the gate of VERDICT r05 item 1 is 0.8 x 5.30 = 4.24 ms\n");
    return 0;
}
