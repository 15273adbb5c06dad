// conv64_sq.hip -- conv64_q8.hip's layer (the split-operand 3x3 64->64 convolution of MOE_PREC_MIXED's exact layers, python/models.py:76-80, 108-123 of the
// reference, with its two correction products on fp8 operands) STREAMED down a 32-pixel column by a pair of SPECIALISED waves:
//
//     out = conv(w_hi, a_hi)  +  2^-11 * (conv(w_lo, a_hi) + conv(w_hi, a_lo))
//           --- wave H ----           ----------------- wave Q ----------------
//
// conv64_q8.hip walks 8 x 32 patches with four waves of 32 channels each: three workgroup barriers, two conversion phases and two passes per patch, row steps
// of 12-36 MFMAs; its cycle trace shows 18-20k cycles per patch for 9.2k of MFMA (profiles/r03/m_conv64_q8.txt), PMC 0.47-0.59 MFMA busy at 2.0 GHz, 3-4 TB/s.
// Here a workgroup is TWO waves that share one column and differ in what they hold:
//
//   wave H        the fp16 product for all 64 output channels: 72 A fragments of w_hi (288 registers: 256 AGPR + 32 VGPR).  Row step = the 12 fragments (dx, ks)
//                 of one a_hi row, each into 3 output rows x 2 channel halves: 72 MFMAs of 32 cycles, six per LDS read (conv64_q8: three)
//   wave Q        both fp8 products for all 64 channels: 18 + 18 A fragments of w_lo 2^8 and w_hi 2^8 (288 registers).  Row step = three fragments of the a_hi
//                 row's fp8 image and three of the a_lo row, each into 3 rows x 2 halves: 36 MFMAs of 64 cycles -- the same 2,304 cycles as wave H
//   hand-off      Q's sums ARE the correction (the E8M0 scales fold the 2^-11 in); 2^-11 of the result needs no more than fp16: a finished row crosses as
//                 64 bytes per lane through LDS (same lane, same MFMA C layout, no transposition), H adds it in fp32 and runs the epilogue (residual, PReLU,
//                 hi / lo split, fp8 low part, stores).  Q does the rest of the housekeeping: ALL LDS-DMA (a_hi rows, the fp8 a_lo rows, EPI 2: the residual's
//                 fp16 rows), and the fp8 image of each a_hi row as it arrives (one row per step: 3 units a lane, no conversion phase)
//   rows          streamed in two-row blocks, three blocks per ring (in use / landed / in flight); ONE barrier of the two waves per row step, in its middle:
//                 before it Q has written the hand-off of the row that completed in the step before, behind it H finishes that row.  Every row step is alike.
//   ranges        a workgroup streams a contiguous range of two-row blocks (column-major, as arsb_s.hip); a range [ya, yb) runs input rows ya-1 .. yb+2
//   LDS           a_hi 3 x 9,216 + a_lo8 3 x 5,120 + a_hi8 2 x 2,304 + hand-off 2 x 4,096 (+ EPI 2: residual 3 x 8,192) = 55,808 (80,384) bytes: two workgroups per CU
//
// Same operands, same products, same scales as conv64_q8.hip; the summation differs in where the correction joins (added once per row in fp32, from an
// fp16 word, instead of accumulated in the same registers): results agree to ~1e-7 relative, not bit for bit (tests/test_gpu_parity.py).
#include "common.h"
#include "rowtile.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#ifndef SQ_FILL
#define SQ_FILL 5
#endif
// timing experiments (tools/mk_variant.sh; results of such builds are WRONG by design): 1 no DMA waits, 2 no DMA, 4 no barriers, 8 no epilogue in wave H,
// 16 no fp8 image conversion, 64 no hand-off
#ifndef SQ_ABL
#define SQ_ABL 0
#endif
#ifndef SQ_LATE
#define SQ_LATE 1      // 1: a block's pieces are waited for at the e = 1 barrier of the block before it is used (two to three row steps behind their issue) and the
#endif                 // image of a block's first row is made behind that barrier; 0: at the e = 0 barrier (one to two row steps), image in the first half

namespace {

constexpr int RB = 2, TW = 32, XW = 36;        // rows per block; output columns; pixels of a ring row (34 used: a block of a_hi is nine 1-KiB pieces)
constexpr int ROWB = XW * 128, ROWB8 = XW * 64;   // 4,608 / 2,304
constexpr int BLKB = RB * ROWB;                // 9,216
constexpr int BLKB8 = 5 * 1024;                // 4,608 used: five 1-KiB pieces land
constexpr int RESROWB = TW * 128, RESBLKB = RB * RESROWB;      // 4,096 / 8,192
constexpr int NRING = 3;
constexpr int OFF_LO8 = NRING * BLKB;          // 27,648
constexpr int OFF_Q8 = OFF_LO8 + NRING * BLKB8;   // 43,008
constexpr int OFF_CORR = OFF_Q8 + 2 * ROWB8;   // 47,616
constexpr int OFF_RES = OFF_CORR + 2 * 4096;   // 55,808
constexpr int LDS_PLAIN = OFF_RES, LDS_RES = OFF_RES + NRING * RESBLKB;      // 55,808 / 80,384

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef int i8v_t __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;
typedef const __attribute__((address_space(3))) u4_t* lds_u4_t;

enum OpKind : int { OP_NONE = 0, OP_CORR, OP_RHI, OP_ADD, OP_ACT, OP_SPL, OP_ST, OP_XLO,      // wave H
                    OP_HND, OP_HWR, OP_CVR, OP_CVW, OP_DMA };                                  // wave Q
struct Op { int kind, a, b, c; };
struct OpList {
    int n = 0;
    Op op[64] = {};
    constexpr void push(int kind, int a = 0, int b = 0, int c = 0) { op[n] = Op{kind, a, b, c}; ++n; }
};
// wave H, second half of a row step (behind the barrier): the four 16-byte slots (channel half, slot) of the row that completed in the step before --
// correction word (and residual word) from LDS, sums, [PReLU], split, stores -- and LAST the residual's low words of the row two steps on (arsb_s.hip)
constexpr OpList h_ops(int epi, bool res)
{
    OpList r;
    for (int s = 0; s < 4; ++s) {
        r.push(OP_CORR, s);
        if (res) r.push(OP_RHI, s);
        r.push(OP_ADD, s, 0); r.push(OP_ADD, s, 2);
        if (epi == 1) { r.push(OP_ACT, s, 0); r.push(OP_ACT, s, 4); }
        r.push(OP_SPL, s, 0); r.push(OP_SPL, s, 2);
        r.push(OP_ST, s);
    }
    if (res) for (int s = 0; s < 4; ++s) r.push(OP_XLO, s);
    return r;
}
// wave Q, first half of a row step: the hand-off of the completed row (4 x: pack eight sums to fp16, one 16-byte LDS write) and the fp8 image of the next
// a_hi row (3 units a lane: two 16-byte reads, eight conversions, one write).  Second half: DMA pieces (e = 0: the first half of a block's, e = 1 first
// half: the rest -- so that every piece has a full row step behind it when the e = 0 barrier of the next block waits for it)
constexpr OpList q_ops_a(int e, int ndma)
{
    OpList r;
    if (!(SQ_ABL & 64)) for (int s = 0; s < 4; ++s) { r.push(OP_HND, s, 0); r.push(OP_HND, s, 2); r.push(OP_HWR, s); }
    if (!SQ_LATE && !(SQ_ABL & 16)) {
        for (int u = 0; u < 3; ++u) r.push(OP_CVR, u);
        for (int u = 0; u < 3; ++u) r.push(OP_CVW, u);
    }
    if (e == 1 && !(SQ_ABL & 2)) for (int m = ndma / 2; m < ndma; ++m) { r.push(OP_DMA, m, 0); r.push(OP_DMA, m, 1); }
    return r;
}
constexpr OpList q_ops_b(int e, int ndma)
{
    OpList r;
    if (e == 0 && !(SQ_ABL & 2)) for (int m = 0; m < ndma / 2; ++m) { r.push(OP_DMA, m, 0); r.push(OP_DMA, m, 1); }
    if (SQ_LATE && !(SQ_ABL & 16)) {
        for (int u = 0; u < 3; ++u) r.push(OP_CVR, u);
        for (int u = 0; u < 3; ++u) r.push(OP_CVW, u);
    }
    return r;
}

// EPI 0 plain | 1 PReLU (fp32, slope <= 1) | 2 + residual (hi + fp8 low word).  The input's (and the residual's) low part is the fp8 word of conv64_q8.hip's
// chain (ConvX3Args::in8); OUT8: so is the output's
template <int EPI, bool OUT8>
__global__ __launch_bounds__(128) void conv64_sq_kernel(ConvX3Args a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr unsigned kOOR = 0xFFFF0000u;
    constexpr bool RES = EPI == 2;
    constexpr int NDMA = 9 + 5 + (RES ? 8 : 0);      // 1-KiB pieces of a block: a_hi, a_lo8, residual
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\ts_nop 3" ::: "memory");      // MODE.FP16_OVFL: the fp8 conversions saturate (conv64_q8.hip)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wq_ = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0: wave H, 1: wave Q
    const int j = lane & 31, hh = lane >> 5;
    const int H = a.H, W = a.W;

    const int px = (W + TW - 1) / TW, nyb = H / RB;
    const int g = blockIdx.x, G = gridDim.x;
    const long long nitems = (long long)a.B * px * nyb;
    int item = (int)(nitems * g / G);
    const int item_end = (int)(nitems * (g + 1) / G);
    if (item >= item_end) return;

    const unsigned nbytes = (unsigned)a.B * H * W * 128u;
    const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int wi = lane & 31, wq4 = wi >> 3;
    const int src = (lane & 32) | (16 * (wq4 >> 1) + 8 * ((wi >> 2) & 1) + 4 * (wq4 & 1) + (wi & 3));      // MFMA row -> channel order (arsb32c.hip)

    if (wq_ == 0) {
        // =============================================================== wave H ===============================================================================
        half8_t w16[72];                       // fragment ((dy * 3 + dx) * 4 + ks) * 2 + channel half
#pragma unroll
        for (int f = 0; f < 72; ++f) w16[f] = *(const half8_t*)(a.wq_hi16 + (f * 64 + src) * 8);
#pragma unroll
        for (int f = 0; f < 64; ++f) asm volatile("" : "+a"(w16[f]));
#pragma unroll
        for (int f = 64; f < 72; ++f) asm volatile("" : "+v"(w16[f]));

        const __amdgpu_buffer_rsrc_t rrl = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? (const void*)a.res_lo : (const void*)a.in_hi), 0, nbytes / 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t ryh = __builtin_amdgcn_make_buffer_rsrc((void*)a.out_hi, 0, nbytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ryl = __builtin_amdgcn_make_buffer_rsrc((void*)a.out_lo, 0, OUT8 ? nbytes / 2 : nbytes, 0x00020000);
        // B fragment (dx, ks): pixel col at col * 128, 16-B slot s at s ^ ((col >> 1) & 7); lane (j, hh) reads slot 2 ks + hh of column j + dx
        unsigned fa[3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int cc = j + dx, z = (cc >> 1) & 7;
            fa[dx] = lds0 + (unsigned)(cc * 128 + ((z ^ hh) << 4));
            asm volatile("" : "+v"(fa[dx]));
        }
        // slot s = 2 c + o of this lane: channels 32 c + 16 o + 8 hh .. +7 of pixel column j.  Hand-off word s at OFF_CORR + s * 1024 + lane * 16; residual
        // word in the residual ring at pixel j, 16-B slot (4 c + 2 o + hh) ^ ((j >> 1) & 7)
        const unsigned ca = lds0 + (unsigned)OFF_CORR + (unsigned)(lane * 16);
        unsigned ra[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int sl = 4 * (s >> 1) + 2 * (s & 1) + hh;
            ra[s] = lds0 + (unsigned)OFF_RES + (unsigned)(j * 128 + ((sl ^ ((j >> 1) & 7)) << 4));
        }
        const unsigned lane_ob = (unsigned)(j * 128 + (8 * hh) * 2);      // byte offset of the lane's eight channels of slot 0 inside a row of the stream tensors; slot s: + (32 c + 16 o) * 2

        float16_t acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = zero16;
        half8_t fx[3];
        u4_t cw, rh;
        u2_t xl[2][4];                         // the residual's fp8 low words (eight channels each), requested two steps ahead (set = step parity)
        unsigned sh[4], sl[4];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int s = 0; s < 4; ++s) xl[p][s] = u2_t{0u, 0u};

        while (item < item_end) {
            const int s0 = item % nyb;
            const int t_ = item / nyb;
            const int pxi = t_ % px, b = t_ / px;
            const int s1 = min(nyb, s0 + (item_end - item));
            item += s1 - s0;
            const int x0 = pxi * TW;
            const int ya = RB * s0, yb = RB * s1;
            const int nblk = (yb - ya) / RB + 2;              // steps t = 0 .. yb - ya + 3: input rows ya - 1 .. yb + 2 (the last output row completes with row yb and is finished a step later)
            const unsigned vo = (x0 + j < W) ? lane_ob : kOOR;
            const unsigned vo8 = (x0 + j < W) ? lane_ob >> 1 : kOOR;

            __builtin_amdgcn_s_barrier();                     // (Q: the rings are free) ...
            __builtin_amdgcn_s_barrier();                     // ... the first two blocks have landed
            asm volatile("" ::: "memory");
            fx[0] = *(lds_h8_t)(fa[0]);
            fx[1] = *(lds_h8_t)(fa[0] ^ 32u);
            int xblk = 0;                                     // ring slot of the block

            auto block = [&](int k, auto BUF_) __attribute__((always_inline)) {
                constexpr int BUF = decltype(BUF_)::value;
                const int Rk = ya - 1 + RB * k;               // first input row of this block
                const int xnext = xblk + 1 == NRING ? 0 : xblk + 1;
                auto step = [&](auto E_) __attribute__((always_inline)) {
                    constexpr int e = decltype(E_)::value;
                    constexpr int T4 = 2 * BUF + e;
                    const int r = Rk + e;
                    const unsigned xo_cur = (unsigned)__builtin_amdgcn_readfirstlane(xblk * BLKB + e * ROWB);
                    const unsigned xo_nxt = (unsigned)__builtin_amdgcn_readfirstlane(e == 0 ? xblk * BLKB + ROWB : xnext * BLKB);
                    constexpr int S = T4 & 3;                 // accumulator slot of output row r - 2
                    const int orow = r - 2;
                    const bool ook = (orow >= ya) & (orow < yb);
                    const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)(ook ? (unsigned)((b * H + orow) * W + x0) * 128u : kOOR));
                    const bool ook2 = (orow + 2 >= ya) & (orow + 2 < yb);
                    const unsigned so2 = (unsigned)__builtin_amdgcn_readfirstlane((int)(ook2 ? (unsigned)((b * H + orow + 2) * W + x0) * 64u : kOOR));
                    const unsigned ro = (unsigned)__builtin_amdgcn_readfirstlane(xblk * RESBLKB + e * RESROWB);

                    auto op_corr = [&](auto S_) __attribute__((always_inline)) {
                        constexpr int s = decltype(S_)::value;
                        cw = *(lds_u4_t)(ca + (unsigned)((T4 & 1) * 4096 + s * 1024));
                    };
                    auto op_rhi = [&](auto S_) __attribute__((always_inline)) {
                        constexpr int s = decltype(S_)::value;
                        rh = *(lds_u4_t)(ra[s] + ro);
                    };
                    auto op_add = [&](auto S_, auto K0_) __attribute__((always_inline)) {      // acc += correction [+ res_hi + res_lo8 2^-9] for channel pairs k0, k0 + 1 of slot s
                        constexpr int s = decltype(S_)::value, k0 = decltype(K0_)::value, c = s >> 1, o = s & 1;
#pragma unroll
                        for (int k = k0; k < k0 + 2; ++k) {
                            float v0 = acc[S][c][8 * o + 2 * k], v1 = acc[S][c][8 * o + 2 * k + 1];
                            v0 = mix_lo(cw[k], 1.0f, v0); v1 = mix_hi(cw[k], 1.0f, v1);
                            if (RES) {
                                v0 = mix_lo(rh[k], 1.0f, v0); v1 = mix_hi(rh[k], 1.0f, v1);
                                const f2_t f = (k & 1) ? __builtin_amdgcn_cvt_pk_f32_fp8((int)xl[T4 & 1][s][k >> 1], true) : __builtin_amdgcn_cvt_pk_f32_fp8((int)xl[T4 & 1][s][k >> 1], false);
                                v0 = __builtin_fmaf(f[0], 0.001953125f, v0); v1 = __builtin_fmaf(f[1], 0.001953125f, v1);
                            }
                            acc[S][c][8 * o + 2 * k] = v0; acc[S][c][8 * o + 2 * k + 1] = v1;
                        }
                    };
                    auto op_act = [&](auto S_, auto E0_) __attribute__((always_inline)) {
                        constexpr int s = decltype(S_)::value, e0 = decltype(E0_)::value, c = s >> 1, o = s & 1;
#pragma unroll
                        for (int q = 8 * o + e0; q < 8 * o + e0 + 4; ++q) acc[S][c][q] = __builtin_fmaxf(acc[S][c][q], acc[S][c][q] * a.slope);
                    };
                    auto op_spl = [&](auto S_, auto K0_) __attribute__((always_inline)) {
                        constexpr int s = decltype(S_)::value, k0 = decltype(K0_)::value, c = s >> 1, o = s & 1;
#pragma unroll
                        for (int k = k0; k < k0 + 2; ++k) split2(acc[S][c][8 * o + 2 * k], acc[S][c][8 * o + 2 * k + 1], -2048.f, sh[k], sl[k]);
                    };
                    auto op_st = [&](auto S_) __attribute__((always_inline)) {
                        constexpr int s = decltype(S_)::value;
                        const u4_t dh = {sh[0], sh[1], sh[2], sh[3]};
                        __builtin_amdgcn_raw_buffer_store_b128(dh, ryh, vo + (unsigned)(s * 32), so, 0);
                        if (!OUT8) {
                            const u4_t dl = {sl[0], sl[1], sl[2], sl[3]};
                            __builtin_amdgcn_raw_buffer_store_b128(dl, ryl, vo + (unsigned)(s * 32), so, 0);
                        } else {
                            unsigned p0, p1;
                            const float quarter = 4.0f;
                            asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %2, %6\n\t"
                                         "v_cvt_scalef32_pk_fp8_f16 %1, %4, %6\n\t"
                                         "v_cvt_scalef32_pk_fp8_f16 %0, %3, %6 op_sel:[0,0,1]\n\t"
                                         "v_cvt_scalef32_pk_fp8_f16 %1, %5, %6 op_sel:[0,0,1]\n\t"
                                         "s_nop 0"
                                         : "=&v"(p0), "=&v"(p1) : "v"(sl[0]), "v"(sl[1]), "v"(sl[2]), "v"(sl[3]), "v"(quarter));
                            __builtin_amdgcn_raw_buffer_store_b64(u2_t{p0, p1}, ryl, vo8 + (unsigned)(s * 16), so >> 1, 0);
                        }
                    };
                    auto op_xlo = [&](auto S_) __attribute__((always_inline)) {
                        constexpr int s = decltype(S_)::value;
                        xl[T4 & 1][s] = __builtin_amdgcn_raw_buffer_load_b64(rrl, vo8 + (unsigned)(s * 16), so2, 0);
                    };

                    constexpr OpList L = h_ops(EPI, RES);
                    auto chunk = [&](auto A_) __attribute__((always_inline)) {
                        constexpr int ai = decltype(A_)::value;
                        constexpr int dx = ai / 4, ks = ai % 4;
                        if (ai == 6) {
                            if (!(SQ_ABL & 4)) __builtin_amdgcn_s_barrier();     // Q's hand-off of row r - 2 is in LDS (and the next block of the rings)
                            asm volatile("" ::: "memory");
                        }
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int dy = 0; dy < 3; ++dy) {
                                const int sl_ = (T4 + 3 - dy) & 3;      // output row r + 1 - dy
                                acc[sl_][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w16[((dy * 3 + dx) * 4 + ks) * 2 + c], fx[ai % 3], (dy == 0 && ai == 0) ? zero16 : acc[sl_][c], 0, 0, 0);
                            }
                        constexpr int a2 = (ai + 2) % 12;
                        fx[(ai + 2) % 3] = *(lds_h8_t)((fa[a2 / 4] ^ (unsigned)((a2 % 4) * 32)) + (ai + 2 < 12 ? xo_cur : xo_nxt));
                        if constexpr (ai == 6 && (SQ_ABL & 8) != 0) asm volatile("" : "+v"(acc[S][0]), "+v"(acc[S][1]));
                        if constexpr (ai >= 6 && (SQ_ABL & 8) == 0) {
                            constexpr int h = ai - 6;
                            constexpr int lo_ = h * L.n / 6, hi_ = (h + 1) * L.n / 6;
                            auto run = [&](auto I_) __attribute__((always_inline)) {
                                constexpr int I = decltype(I_)::value;
                                if constexpr (I >= lo_ && I < hi_) {
                                    constexpr Op o = L.op[I];
                                    if constexpr (o.kind == OP_CORR) op_corr(std::integral_constant<int, o.a>{});
                                    if constexpr (o.kind == OP_RHI) op_rhi(std::integral_constant<int, o.a>{});
                                    if constexpr (o.kind == OP_ADD) op_add(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                    if constexpr (o.kind == OP_ACT) op_act(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                    if constexpr (o.kind == OP_SPL) op_spl(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                    if constexpr (o.kind == OP_ST) op_st(std::integral_constant<int, o.a>{});
                                    if constexpr (o.kind == OP_XLO) op_xlo(std::integral_constant<int, o.a>{});
                                }
                            };
#define SQ_OP(I) run(std::integral_constant<int, I>{});
                            SQ_OP(0) SQ_OP(1) SQ_OP(2) SQ_OP(3) SQ_OP(4) SQ_OP(5) SQ_OP(6) SQ_OP(7) SQ_OP(8) SQ_OP(9) SQ_OP(10) SQ_OP(11) SQ_OP(12) SQ_OP(13) SQ_OP(14) SQ_OP(15)
                            SQ_OP(16) SQ_OP(17) SQ_OP(18) SQ_OP(19) SQ_OP(20) SQ_OP(21) SQ_OP(22) SQ_OP(23) SQ_OP(24) SQ_OP(25) SQ_OP(26) SQ_OP(27) SQ_OP(28) SQ_OP(29) SQ_OP(30) SQ_OP(31)
                            SQ_OP(32) SQ_OP(33) SQ_OP(34) SQ_OP(35) SQ_OP(36) SQ_OP(37) SQ_OP(38) SQ_OP(39) SQ_OP(40) SQ_OP(41) SQ_OP(42) SQ_OP(43) SQ_OP(44) SQ_OP(45) SQ_OP(46) SQ_OP(47)
#undef SQ_OP
                        }
#ifndef SQ_NOPIN
#pragma unroll
                        for (int i_ = 0; i_ < 6; ++i_) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (i_ == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x006, SQ_FILL, 0);
                        }
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    };
#define SQ_CHUNK(A) chunk(std::integral_constant<int, A>{});
                    SQ_CHUNK(0) SQ_CHUNK(1) SQ_CHUNK(2) SQ_CHUNK(3) SQ_CHUNK(4) SQ_CHUNK(5) SQ_CHUNK(6) SQ_CHUNK(7) SQ_CHUNK(8) SQ_CHUNK(9) SQ_CHUNK(10) SQ_CHUNK(11)
#undef SQ_CHUNK
                };
                step(std::integral_constant<int, 0>{});
                step(std::integral_constant<int, 1>{});
                xblk = xnext;
            };
            int k = 0;
            for (; k + 1 < nblk; k += 2) {
                block(k, std::integral_constant<int, 0>{});
                block(k + 1, std::integral_constant<int, 1>{});
            }
            if (k < nblk) block(k, std::integral_constant<int, 0>{});
        }
    } else {
        // =============================================================== wave Q ===============================================================================
        i8v_t wl8[18], wh8[18];                // fragment (dy * 3 + dx) * 2 + channel half
#pragma unroll
        for (int f = 0; f < 18; ++f) {
            wl8[f] = *(const i8v_t*)(a.wq_lo8 + (f * 64 + src) * 32);
            wh8[f] = *(const i8v_t*)(a.wq_hi8 + (f * 64 + src) * 32);
        }
#pragma unroll
        for (int f = 0; f < 18; ++f) asm volatile("" : "+a"(wl8[f]));
#pragma unroll
        for (int f = 0; f < 14; ++f) asm volatile("" : "+a"(wh8[f]));
#pragma unroll
        for (int f = 14; f < 18; ++f) asm volatile("" : "+v"(wh8[f]));
        int scale_a = 127 - 19, scale_b = 127 + 2;            // weights carry 2^8, activations 2^-2, the correction term 2^-11 (conv64_q8.hip)
        asm volatile("" : "+v"(scale_a), "+v"(scale_b));

        const unsigned in_pad = (unsigned)(RB * W + 2) * 128u;
        const __amdgpu_buffer_rsrc_t rhi = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in_hi - in_pad), 0, nbytes + in_pad, 0x00020000);
        const __amdgpu_buffer_rsrc_t rlo = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in_lo - in_pad / 2), 0, (nbytes + in_pad) / 2, 0x00020000);
        const unsigned res_pad = in_pad + (unsigned)W * 128u;      // (the residual's first rows lie one row further up than the input's)
        const __amdgpu_buffer_rsrc_t rrh = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(RES ? a.res_hi : a.in_hi) - res_pad), 0, nbytes + res_pad, 0x00020000);
        // ---- DMA pieces of a block whose first input row is yr, first column xa = x0 - 1: m 0..8 a_hi (8 pixels x 8 slots each), 9..13 a_lo8 (16 pixels x 4 slots),
        // 14..21 the residual's fp16 rows yr - 2, yr - 1 (the output rows finished in the block's two steps) of columns x0 .. x0 + 31 (8 pixels x 8 slots).  d_off: the lane's source offset relative to the block's origin
        unsigned d_off = 0, d_r = 0, d_cc = 0;
        auto piece_addr = [&](int m) {
            if (m < 9) {
                unsigned q = (unsigned)(m * 8 + (lane >> 3));
                asm volatile("" : "+v"(q));
                d_r = q >= (unsigned)XW ? 1u : 0u;
                d_cc = q - d_r * (unsigned)XW;
                const unsigned sl = (unsigned)(lane & 7) ^ ((d_cc >> 1) & 7u);
                d_off = ((__umul24(d_r, (unsigned)W) + d_cc) << 7) | (sl << 4);
            } else if (m < 14) {
                unsigned q = (unsigned)((m - 9) * 16 + (lane >> 2));
                asm volatile("" : "+v"(q));
                d_r = q >= (unsigned)XW ? 1u : 0u;
                d_cc = q - d_r * (unsigned)XW;
                d_r = q >= 2u * XW ? 2u : d_r;                // (the upper half of piece 13: nothing)
                const unsigned sl = (unsigned)(lane & 3) ^ ((d_cc >> 2) & 3u);
                d_off = ((__umul24(d_r, (unsigned)W) + d_cc) << 6) | (sl << 4);
            } else {
                unsigned q = (unsigned)((m - 14) * 8 + (lane >> 3));
                asm volatile("" : "+v"(q));
                d_r = q >> 5;
                d_cc = q & 31u;
                const unsigned sl = (unsigned)(lane & 7) ^ ((d_cc >> 1) & 7u);
                d_off = ((__umul24(d_r, (unsigned)W) + d_cc) << 7) | (sl << 4);
            }
        };
        auto piece_off = [&](int m, int yr, int x0, bool live) {
            bool ok;
            if (m < 14) ok = ((unsigned)(yr + (int)d_r) < (unsigned)H) & ((unsigned)(x0 - 1 + (int)d_cc) < (unsigned)W) & (d_cc < 34u) & (d_r < 2u) & live;
            else ok = ((unsigned)(yr - 2 + (int)d_r) < (unsigned)H) & ((unsigned)(x0 + (int)d_cc) < (unsigned)W) & live;      // (output rows lag the input rows by two)
            return ok ? d_off : kOOR;
        };
        // issue piece m of the block with ring slot `slot`, first input row yr
        auto piece_issue = [&](int m, int slot, int yr, int x0, int b, bool live) {
            const unsigned off = piece_off(m, yr, x0, live);
            if (m < 9) {
                const unsigned org = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((b * H + yr + RB) * W + x0 - 1 + 2) * 128u));
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(slot * BLKB + m * 1024);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rhi, (__attribute__((address_space(3))) void*)(smem + dst), 16, off, org, 0, 0);
            } else if (m < 14) {
                const unsigned org = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((b * H + yr + RB) * W + x0 - 1 + 2) * 64u));
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(OFF_LO8 + slot * BLKB8 + (m - 9) * 1024);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rlo, (__attribute__((address_space(3))) void*)(smem + dst), 16, off, org, 0, 0);
            } else {
                const unsigned org = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((b * H + yr - 2 + RB + 1) * W + x0 + 2) * 128u));
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(OFF_RES + slot * RESBLKB + (m - 14) * 1024);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rrh, (__attribute__((address_space(3))) void*)(smem + dst), 16, off, org, 0, 0);
            }
        };
        // fp8 B fragment dx of an image row: pixel at col * 64, 16-B slot s (16 channels) at s ^ ((col >> 2) & 3); lane (j, hh) reads slots 2 hh, 2 hh + 1
        unsigned fq[3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int cc = j + dx, z = (cc >> 2) & 3;
            fq[dx] = lds0 + (unsigned)(cc * 64 + (((2 * hh) ^ z) << 4));
            asm volatile("" : "+v"(fq[dx]));
        }
        auto read8 = [&](int dx, unsigned rowoff) {
            const unsigned ad = fq[dx] + rowoff;
            const u4_t lo4 = *(lds_u4_t)(ad), hi4 = *(lds_u4_t)(ad ^ 16u);
            return i8v_t{(int)lo4[0], (int)lo4[1], (int)lo4[2], (int)lo4[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
        };
        // fp8 image of an a_hi row: unit u of the lane = (pixel q, 16-channel group s8) with 64 u + lane = 4 q' + ... : 34 pixels x 4 groups = 136 units, three a lane
        // (units behind the last repeat unit 135: same data to the same place)
        unsigned cv_src[3], cv_dst[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int un = min(u * 64 + lane, 135);
            const int q = un % 34, s8 = un / 34;
            const int z16 = (q >> 1) & 7, z8 = (q >> 2) & 3;
            cv_src[u] = lds0 + (unsigned)(q * 128 + (((2 * s8) ^ z16) << 4));      // slot 2 s8; slot 2 s8 + 1 is this address ^ 16
            cv_dst[u] = lds0 + (unsigned)OFF_Q8 + (unsigned)(q * 64 + ((s8 ^ z8) << 4));
        }
        const float quarter = 4.0f;
        auto cvt4 = [&](unsigned a0, unsigned a1, unsigned b0, unsigned b1, unsigned& p0, unsigned& p1) __attribute__((always_inline)) {
            asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %2, %6\n\t"
                         "v_cvt_scalef32_pk_fp8_f16 %1, %4, %6\n\t"
                         "v_cvt_scalef32_pk_fp8_f16 %0, %3, %6 op_sel:[0,0,1]\n\t"
                         "v_cvt_scalef32_pk_fp8_f16 %1, %5, %6 op_sel:[0,0,1]\n\t"
                         "s_nop 0"
                         : "=&v"(p0), "=&v"(p1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(quarter));
        };
        auto cvt16 = [&](const u4_t& w0, const u4_t& w1) {
            u4_t d = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                unsigned p0, p1;
                cvt4(w0[2 * k], w0[2 * k + 1], w1[2 * k], w1[2 * k + 1], p0, p1);
                d[k] = p0; d[2 + k] = p1;
            }
            return d;
        };
        const unsigned ca = lds0 + (unsigned)OFF_CORR + (unsigned)(lane * 16);

        float16_t acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = zero16;
        i8v_t fr[2];
        u4_t cvw[3][2];
        unsigned hp[4];

        while (item < item_end) {
            const int s0 = item % nyb;
            const int t_ = item / nyb;
            const int pxi = t_ % px, b = t_ / px;
            const int s1 = min(nyb, s0 + (item_end - item));
            item += s1 - s0;
            const int x0 = pxi * TW;
            const int ya = RB * s0, yb = RB * s1;
            const int nblk = (yb - ya) / RB + 2;

            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // H has left the previous range: the rings are free
            asm volatile("" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int m = 0; m < NDMA; ++m) {
                    piece_addr(m);
                    piece_issue(m, kb, ya - 1 + RB * kb, x0, b, true);
                }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // the fp8 image of the first input row (row ya - 1: ring slot 0, row 0) -> image row (ya - 1) & 1 = 1
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const u4_t w0 = *(lds_u4_t)(cv_src[u]), w1 = *(lds_u4_t)(cv_src[u] ^ 16u);
                const u4_t d = cvt16(w0, w1);
                const unsigned adr = cv_dst[u];
                asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(adr), "v"(d), "n"(ROWB8) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            fr[0] = read8(0, (unsigned)(OFF_Q8 + ROWB8));
            int xblk = 0;

            auto block = [&](int k, auto BUF_) __attribute__((always_inline)) {
                constexpr int BUF = decltype(BUF_)::value;
                const int xnext = xblk + 1 == NRING ? 0 : xblk + 1;
                const int xnext2 = xnext + 1 == NRING ? 0 : xnext + 1;
                const bool live = RB * (k + 2) <= yb - ya + 3;           // the block after the next has rows somebody wants
                const int yrn = ya - 1 + RB * (k + 2);
                auto step = [&](auto E_) __attribute__((always_inline)) {
                    constexpr int e = decltype(E_)::value;
                    constexpr int T4 = 2 * BUF + e;
                    constexpr int S = T4 & 3;
                    // input row r = ya - 1 + 2 k + e: a_lo8 row at ring (xblk, e); its fp8 image at image row (r & 1) = 1 - e (ya even); the NEXT row's a_hi at
                    // ring (e == 0 ? (xblk, 1) : (xnext, 0)), its image goes to image row e
                    const unsigned lo_cur = (unsigned)__builtin_amdgcn_readfirstlane(OFF_LO8 + xblk * BLKB8 + e * ROWB8);
                    const unsigned lo_nxt = (unsigned)__builtin_amdgcn_readfirstlane(OFF_LO8 + (e == 0 ? xblk * BLKB8 + ROWB8 : xnext * BLKB8));
                    const unsigned hi_nxt = (unsigned)__builtin_amdgcn_readfirstlane(e == 0 ? xblk * BLKB + ROWB : xnext * BLKB);
                    constexpr unsigned q_cur = (unsigned)(OFF_Q8 + (1 - e) * ROWB8), q_nxt = (unsigned)(OFF_Q8 + e * ROWB8);

                    auto op_hnd = [&](auto S_, auto K0_) __attribute__((always_inline)) {      // eight sums of slot s -> four fp16 pairs
                        constexpr int s = decltype(S_)::value, k0 = decltype(K0_)::value, c = s >> 1, o = s & 1;
#pragma unroll
                        for (int kk = k0; kk < k0 + 2; ++kk) {
                            const h2_t pr = {(half_t)acc[S][c][8 * o + 2 * kk], (half_t)acc[S][c][8 * o + 2 * kk + 1]};
                            hp[kk] = __builtin_bit_cast(unsigned, pr);
                        }
                    };
                    auto op_hwr = [&](auto S_) __attribute__((always_inline)) {
                        constexpr int s = decltype(S_)::value;
                        const u4_t d = {hp[0], hp[1], hp[2], hp[3]};
                        const unsigned ad = ca;
                        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(ad), "v"(d), "n"((T4 & 1) * 4096 + s * 1024) : "memory");
                    };
                    auto op_cvr = [&](auto U_) __attribute__((always_inline)) {
                        constexpr int u = decltype(U_)::value;
                        cvw[u][0] = *(lds_u4_t)(cv_src[u] + hi_nxt);
                        cvw[u][1] = *(lds_u4_t)((cv_src[u] ^ 16u) + hi_nxt);
                    };
                    auto op_cvw = [&](auto U_) __attribute__((always_inline)) {
                        constexpr int u = decltype(U_)::value;
                        const u4_t d = cvt16(cvw[u][0], cvw[u][1]);
                        const unsigned adr = cv_dst[u];
                        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(adr), "v"(d), "n"(e * ROWB8) : "memory");
                    };
                    auto op_dma = [&](auto M_, auto HALF_) __attribute__((always_inline)) {
                        constexpr int m = decltype(M_)::value, half = decltype(HALF_)::value;
                        if constexpr (half == 0) piece_addr(m);
                        else piece_issue(m, xnext2, yrn, x0, b, live);
                    };
                    constexpr OpList LA = q_ops_a(e, NDMA), LB = q_ops_b(e, NDMA);
                    auto chunk = [&](auto N_) __attribute__((always_inline)) {
                        constexpr int n = decltype(N_)::value;            // 0..2: w_lo8 x a_hi8 (dx = n); 3..5: w_hi8 x a_lo8 (dx = n - 3)
                        constexpr int dx = n % 3;
                        if (n == 3) {
                            // the hand-off is written, and the pieces of the next block have landed (everything older than this block's own pieces: vmcnt retires in order)
                            if ((SQ_ABL & 3) || (SQ_LATE ? e == 0 : e == 1)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            else if (SQ_LATE) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA) : "memory");
                            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                            if (!(SQ_ABL & 4)) __builtin_amdgcn_s_barrier();
                            asm volatile("" ::: "memory");
                        }
                        if (n == 0) {
#pragma unroll
                            for (int c = 0; c < 2; ++c) acc[(T4 + 3) & 3][c] = zero16;      // (the scaled MFMA takes no literal as its C operand)
                        }
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int dy = 0; dy < 3; ++dy) {
                                const int sl_ = (T4 + 3 - dy) & 3;
                                const i8v_t wv = n < 3 ? wl8[(dy * 3 + dx) * 2 + c] : wh8[(dy * 3 + dx) * 2 + c];
                                acc[sl_][c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv, fr[n & 1], acc[sl_][c], 0, 0, 0, scale_a, 0, scale_b);
                            }
                        // the fragment of the next chunk: 0, 1 -> this row's image; 2, 3, 4 -> this row's a_lo8 (dx 0, 1, 2); 5 -> the next row's image (written in this step's first half)
                        constexpr int n1 = (n + 1) % 6;
                        if (n1 != 0) fr[(n + 1) & 1] = n1 < 3 ? read8(n1, q_cur) : read8(n1 - 3, lo_cur);
                        {
                            constexpr int h = n % 3;
                            constexpr int cnt = n < 3 ? LA.n : LB.n;
                            constexpr int lo_ = h * cnt / 3, hi_ = (h + 1) * cnt / 3;
                            auto run = [&](auto I_) __attribute__((always_inline)) {
                                constexpr int I = decltype(I_)::value;
                                if constexpr (I >= lo_ && I < hi_) {
                                    constexpr Op o = n < 3 ? LA.op[I] : LB.op[I];
                                    if constexpr (o.kind == OP_HND) op_hnd(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                    if constexpr (o.kind == OP_HWR) op_hwr(std::integral_constant<int, o.a>{});
                                    if constexpr (o.kind == OP_CVR) op_cvr(std::integral_constant<int, o.a>{});
                                    if constexpr (o.kind == OP_CVW) op_cvw(std::integral_constant<int, o.a>{});
                                    if constexpr (o.kind == OP_DMA) op_dma(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                }
                            };
#define SQ_OP(I) run(std::integral_constant<int, I>{});
                            SQ_OP(0) SQ_OP(1) SQ_OP(2) SQ_OP(3) SQ_OP(4) SQ_OP(5) SQ_OP(6) SQ_OP(7) SQ_OP(8) SQ_OP(9) SQ_OP(10) SQ_OP(11) SQ_OP(12) SQ_OP(13) SQ_OP(14) SQ_OP(15)
                            SQ_OP(16) SQ_OP(17) SQ_OP(18) SQ_OP(19) SQ_OP(20) SQ_OP(21) SQ_OP(22) SQ_OP(23) SQ_OP(24) SQ_OP(25) SQ_OP(26) SQ_OP(27) SQ_OP(28) SQ_OP(29) SQ_OP(30) SQ_OP(31)
                            SQ_OP(32) SQ_OP(33) SQ_OP(34) SQ_OP(35) SQ_OP(36) SQ_OP(37) SQ_OP(38) SQ_OP(39) SQ_OP(40) SQ_OP(41) SQ_OP(42) SQ_OP(43) SQ_OP(44) SQ_OP(45) SQ_OP(46) SQ_OP(47)
#undef SQ_OP
                        }
                        if (n1 == 0) fr[(n + 1) & 1] = read8(0, q_nxt);      // (behind the image's last write of this step)
#ifndef SQ_NOPIN
#pragma unroll
                        for (int i_ = 0; i_ < 6; ++i_) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (i_ == 0 && n1 != 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x006, 2 * SQ_FILL, 0);
                        }
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    };
#define SQ_CHUNK(N) chunk(std::integral_constant<int, N>{});
                    SQ_CHUNK(0) SQ_CHUNK(1) SQ_CHUNK(2) SQ_CHUNK(3) SQ_CHUNK(4) SQ_CHUNK(5)
#undef SQ_CHUNK
                };
                step(std::integral_constant<int, 0>{});
                step(std::integral_constant<int, 1>{});
                xblk = xnext;
            };
            int k = 0;
            for (; k + 1 < nblk; k += 2) {
                block(k, std::integral_constant<int, 0>{});
                block(k + 1, std::integral_constant<int, 1>{});
            }
            if (k < nblk) block(k, std::integral_constant<int, 0>{});
        }
    }
#endif
}

template <int EPI, bool OUT8>
hipError_t set_limit() { return hipFuncSetAttribute((const void*)conv64_sq_kernel<EPI, OUT8>, hipFuncAttributeMaxDynamicSharedMemorySize, EPI == 2 ? LDS_RES : LDS_PLAIN); }

}  // namespace

hipError_t conv64_sq_init()
{
    hipError_t e;
    if ((e = set_limit<0, true>()) != hipSuccess) return e;
    if ((e = set_limit<1, true>()) != hipSuccess) return e;
    if ((e = set_limit<2, true>()) != hipSuccess) return e;
    return set_limit<2, false>();
}

// The chain form of conv64_q8.hip's layer (fp8 low parts in; out: fp8, or fp16 behind the last layer).  false: not this kernel's (the caller uses conv64_q8)
bool launch_conv64_sq(ConvX3Args a, int max_groups, hipStream_t s)
{
    if (!(a.slope <= 1.f) || !a.in8 || a.pool) return false;
    if (a.H % RB != 0 || a.H < RB) return false;
    if ((long long)a.B * a.H * a.W * 128 + (long long)(RB * a.W + 2) * 128 >= (1ll << 32) - 65536) return false;
    if (!a.in_hi || !a.in_lo || !a.out_hi || !a.out_lo || !a.wq_hi16 || !a.wq_hi8 || !a.wq_lo8) return false;
    if ((a.res_hi == nullptr) != (a.res_lo == nullptr)) return false;
    if (a.res_hi && a.slope != 1.f) return false;
    const int epi = a.res_hi ? 2 : a.slope != 1.f ? 1 : 0;
    if (!a.out8 && epi != 2) return false;
    static const int epis = [] { const char* e = getenv("MOE_SQ_EPIS"); int m = 0; if (!e) return 7; for (; *e; ++e) if (*e >= '0' && *e <= '2') m |= 1 << (*e - '0'); return m; }();      // (diagnosis: the layer kinds this kernel takes)
    if (!(epis >> epi & 1)) return false;
    const int px = (a.W + TW - 1) / TW;
    const long long items = (long long)a.B * px * (a.H / RB);
    if (items >= (1ll << 31) / 4) return false;
    const int G = (int)std::min<long long>(items, 2ll * max_groups);
    const dim3 grid(G), blk(128);
    if (epi == 0) conv64_sq_kernel<0, true><<<grid, blk, LDS_PLAIN, s>>>(a);
    else if (epi == 1) conv64_sq_kernel<1, true><<<grid, blk, LDS_PLAIN, s>>>(a);
    else if (a.out8) conv64_sq_kernel<2, true><<<grid, blk, LDS_RES, s>>>(a);
    else conv64_sq_kernel<2, false><<<grid, blk, LDS_RES, s>>>(a);
    return true;
}
