// conv64_sq.hip -- conv64_q8.hip's layer (the split-operand 3x3 64->64 convolution of MOE_PREC_MIXED's exact layers, python/models.py:76-80, 108-123 of the
// reference, with its two correction products on fp8 operands) STREAMED down a 32-pixel column, the form conv3x3_ps4.hip / arsb_s.hip gave the other layers:
//
//     out = conv(w_hi, a_hi)  +  2^-11 * (conv(w_lo, a_hi) + conv(w_hi, a_lo))        all three products in ONE accumulator set (the E8M0 scales fold the 2^-11 in)
//
// conv64_q8.hip walks 8 x 32 patches with four waves: three workgroup barriers, two conversion phases and two passes per patch, row steps of 12-36 MFMAs; its
// cycle trace shows 18-20k cycles per patch for 9.2k of MFMA (profiles/r03/m_conv64_q8.txt), PMC 0.47-0.59 MFMA busy at 2.0 GHz.  Here
//
//   workgroup     TWO waves (c = 0, 1: output channels 32c .. 32c+31), two workgroups per CU; weights as in conv64_q8.hip: 36 fp16 A fragments of w_hi + 9 + 9
//                 fp8 ones of w_lo 2^8 and w_hi 2^8 = 288 registers (256 AGPR + 32 VGPR)
//   row step r    the 12 fp16 fragments (dx, ks) of a_hi row r into the output rows r-1, r, r+1 (36 MFMAs of 32 cycles); behind the k-slices 1 and 3 of a tap
//                 column its fp8 fragments -- of the row's fp8 IMAGE (x w_lo8) and of the a_lo row (x w_hi8) -- into the same three accumulators (18 MFMAs of
//                 64 cycles): 2,304 MFMA cycles, every row step alike.  Beside them: the epilogue of output row r-2 (residual, PReLU, hi / lo split, fp8 low
//                 word, stores), the fp8 image of a_hi row r+1 (the two waves share its 136 units: no conversion phase), and in every second step the LDS-DMA
//                 pieces of the block after the next (a_hi rows, fp8 a_lo rows, EPI 2: the residual's fp16 rows; the waves take alternate pieces)
//   rows          two-row blocks, three blocks per ring (in use / landed / in flight); ONE barrier of the two waves per row step (it publishes the image row, and
//                 in the first step of a block -- behind a counted vmcnt -- the block after it)
//   ranges        a workgroup streams a contiguous range of two-row blocks (column-major, as arsb_s.hip); a range [ya, yb) runs input rows ya-1 .. yb+2
//   LDS           a_hi 3 x 9,216 + a_lo8 3 x 5,120 + a_hi8 2 x 2,304 + 1,024 + DMA offset table 5,120 (+ EPI 2: residual 3 x 8,192) = 53,760 (78,336) bytes: two workgroups per CU
//
// A first form gave the fp16 product to one wave (64 channels: six MFMAs per fragment read) and both fp8 products to the other, with the correction handed over
// through LDS as fp16: correct, and only 3-10 % faster than conv64_q8.hip -- the fp16 wave carried the whole epilogue (~250 VALU instructions in the half step
// behind the barrier) while the fp8 wave idled (profiles/r04/k_conv64_sq_first_form_variants.txt).  Symmetric waves share every kind of work.
//
// Same operands, same products, same scales as conv64_q8.hip, one accumulator set; only the order of the sums inside a row differs: results agree to fp32
// rounding, and -- through the fp8 low words between the layers -- to a few 1e-5 at the net's output (tests/test_gpu_parity.py).
#include "common.h"
#include "rowtile.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#ifndef SQ_FILL
#define SQ_FILL 5
#endif
// timing experiments (tools/mk_variant.sh; results of such builds are WRONG by design): 1 no DMA waits, 2 no DMA, 8 no epilogue (accumulators kept live), 16 no fp8 image,
// 32 stores of a row as contiguous KiBs (what coalesced stores would cost), 64 no loads of the residual's low words, 128 stores issued and dropped
#ifndef SQ_ABL
#define SQ_ABL 0
#endif
// cache policy bits of the stores / the DMA loads (gfx942+: 1 sc0, 2 nt, 16 sc1): the tensors stream through once
#ifndef SQ_STAUX
#define SQ_STAUX 0
#endif
#ifndef SQ_LDAUX
#define SQ_LDAUX 0
#endif

namespace {

constexpr int RB = 2, TW = 32, XW = 36;        // rows per block; output columns; pixels of a ring row (34 used: a block of a_hi is nine 1-KiB pieces)
constexpr int ROWB = XW * 128, ROWB8 = XW * 64;   // 4,608 / 2,304
constexpr int BLKB = RB * ROWB;                // 9,216
constexpr int BLKB8 = 5 * 1024;                // 4,608 used: five 1-KiB pieces land
constexpr int RESROWB = TW * 128, RESBLKB = RB * RESROWB;      // 4,096 / 8,192
constexpr int NRING = 3;
constexpr int OFF_LO8 = NRING * BLKB;          // 27,648
constexpr int OFF_Q8 = OFF_LO8 + NRING * BLKB8;   // 43,008
constexpr int OFF_DUMP = OFF_Q8 + 2 * ROWB8;   // 47,616: where the pieces that do not exist land (wave 1's fifth a_hi piece, its third a_lo8 piece)
constexpr int OFF_TAB = OFF_DUMP + 1024;       // 48,640: the lanes' DMA source offsets of the range, [entry 10][thread 128] words
constexpr int OFF_RES = OFF_TAB + 10 * 512;    // 53,760
constexpr int LDS_PLAIN = OFF_RES, LDS_RES = OFF_RES + NRING * RESBLKB;      // 53,760 / 78,336

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef int i8v_t __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;
typedef const __attribute__((address_space(3))) u4_t* lds_u4_t;

enum OpKind : int { OP_NONE = 0, OP_DMA, OP_CVR, OP_CVW, OP_RHI, OP_ADD, OP_ACT, OP_SPL, OP_ST, OP_XLO };
struct Op { int kind, a, b, c; };
struct OpList {
    int n = 0;
    Op op[64] = {};
    constexpr void push(int kind, int a = 0, int b = 0, int c = 0) { op[n] = Op{kind, a, b, c}; ++n; }
};
// The side work of a row step, two lists dealt to its chunks proportionally.  dma_ops (e = 1, chunks 0..7): the wave's DMA pieces of the block after the next --
// half 0: the lane's table word, half 1: select + issue; the word of piece i + 1 is read in front of piece i's issue.  step_ops (chunks 0..11, behind the
// chunk's DMA ops): the wave's two units of the next row's fp8 image (in LDS before the barrier at chunk 10); the two 16-byte slots of output row r - 2 --
// [residual word from the ring, sums,] [PReLU,] split, stores -- and LAST the residual's low words of the row two steps on (arsb_s.hip: no store is issued
// behind them before they are consumed).  LDS reads sit ahead of their consumers with something else in between.
constexpr OpList dma_ops(int e, int np)
{
    OpList r;
    if (e == 1 && !(SQ_ABL & 2)) {
        for (int i = 0; i < np; ++i) { r.push(OP_DMA, i, 0); if (i >= 1) r.push(OP_DMA, i - 1, 1); }
        r.push(OP_DMA, np - 1, 1);
    }
    return r;
}
constexpr OpList step_ops(int epi)
{
    OpList r;
    if (!(SQ_ABL & 16)) r.push(OP_CVR, 0);
    if (!(SQ_ABL & 8)) {
        if (epi == 2) r.push(OP_RHI, 0);
        if (epi == 2) { r.push(OP_ADD, 0, 0); r.push(OP_ADD, 0, 2); }
        if (epi == 1) { r.push(OP_ACT, 0, 0); r.push(OP_ACT, 0, 4); }
        if (!(SQ_ABL & 16)) { r.push(OP_CVW, 0); r.push(OP_CVR, 1); }
        if (epi == 2) r.push(OP_RHI, 1);
        r.push(OP_SPL, 0, 0); r.push(OP_SPL, 0, 2);
        if (!(SQ_ABL & 16)) r.push(OP_CVW, 1);
        r.push(OP_ST, 0);
        if (epi == 2) { r.push(OP_ADD, 1, 0); r.push(OP_ADD, 1, 2); }
        if (epi == 1) { r.push(OP_ACT, 1, 0); r.push(OP_ACT, 1, 4); }
        r.push(OP_SPL, 1, 0); r.push(OP_SPL, 1, 2);
        r.push(OP_ST, 1);
        if (epi == 2 && !(SQ_ABL & 64)) { r.push(OP_XLO, 0); r.push(OP_XLO, 1); }
    } else if (!(SQ_ABL & 16)) { r.push(OP_CVW, 0); r.push(OP_CVR, 1); r.push(OP_CVW, 1); }
    return r;
}
// the image's writes are dealt to chunks in front of the barrier (chunk 10)
constexpr bool cvw_in_time(int epi)
{
    const OpList l = step_ops(epi);
    for (int i = 0; i < l.n; ++i) if (l.op[i].kind == OP_CVW && i >= 10 * l.n / 12) return false;
    return true;
}
constexpr int vm_of(int kind) { return kind == OP_ST ? 2 : kind == OP_XLO ? 1 : 0; }
// VM operations a wave issues behind its last DMA piece (chunk 7 of step e = 1) and in front of the barrier of the next step (e = 0, head of chunk 10): they
// may stay in flight at that barrier's counted wait -- vmcnt retires in order, so everything older, the pieces included, is then complete
constexpr int vm_behind(int epi)
{
    int n = 0;
    const OpList l = step_ops(epi);
    for (int i = 7 * l.n / 12; i < l.n; ++i) n += vm_of(l.op[i].kind);        // (the chunk's DMA ops run in front of its step ops)
    for (int i = 0; i < 10 * l.n / 12; ++i) n += vm_of(l.op[i].kind);
    return n;
}

// EPI 0 plain | 1 PReLU (fp32, slope <= 1) | 2 + residual (hi + fp8 low word).  The input's (and the residual's) low part is the fp8 word of conv64_q8.hip's
// chain (ConvX3Args::in8); OUT8: so is the output's
template <int EPI, bool OUT8>
__global__ __launch_bounds__(128) void conv64_sq_kernel(ConvX3Args a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr unsigned kOOR = 0xFFFF0000u;
    constexpr bool RES = EPI == 2;
    constexpr int NP = 5 + 3 + (RES ? 4 : 0);   // DMA pieces a wave issues per block: a_hi 2i + c (nine exist), a_lo8 2i + c (five exist), residual 2i + c (eight)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\ts_nop 3" ::: "memory");      // MODE.FP16_OVFL: the fp8 conversions saturate (conv64_q8.hip)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int c = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hh = lane >> 5;
    const int H = a.H, W = a.W;

    const int px = (W + TW - 1) / TW, nyb = H / RB;
    const int g = blockIdx.x, G = gridDim.x;
    const long long nitems = (long long)a.B * px * nyb;
    int item = (int)(nitems * g / G);
    const int item_end = (int)(nitems * (g + 1) / G);
    if (item >= item_end) return;

    // ---- weights (conv64_q8.hip: MFMA row i = 8q + 4h' + e is given channel 16 (q >> 1) + 8 h' + 4 (q & 1) + e: a lane's registers 8g .. 8g+7 are one 16-byte slot)
    half8_t w16[36];
    i8v_t wl8[9], wh8[9];
    {
        const int wi = lane & 31, wq = wi >> 3;
        const int src = (lane & 32) | (16 * (wq >> 1) + 8 * ((wi >> 2) & 1) + 4 * (wq & 1) + (wi & 3));
#pragma unroll
        for (int f = 0; f < 36; ++f) w16[f] = *(const half8_t*)(a.wq_hi16 + ((f * 2 + c) * 64 + src) * 8);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            wl8[t] = *(const i8v_t*)(a.wq_lo8 + ((t * 2 + c) * 64 + src) * 32);
            wh8[t] = *(const i8v_t*)(a.wq_hi8 + ((t * 2 + c) * 64 + src) * 32);
        }
#pragma unroll
        for (int f = 0; f < 36; ++f) asm volatile("" : "+a"(w16[f]));
#pragma unroll
        for (int t = 0; t < 9; ++t) asm volatile("" : "+a"(wl8[t]));
#pragma unroll
        for (int t = 0; t < 5; ++t) asm volatile("" : "+a"(wh8[t]));
#pragma unroll
        for (int t = 5; t < 9; ++t) asm volatile("" : "+v"(wh8[t]));
    }
    int scale_a = 127 - 19, scale_b = 127 + 2;                // weights carry 2^8, activations 2^-2, the correction term 2^-11 (conv64_q8.hip)
    asm volatile("" : "+v"(scale_a), "+v"(scale_b));

    const unsigned nbytes = (unsigned)a.B * H * W * 128u;
    const unsigned in_pad = (unsigned)(RB * W + 2) * 128u, res_pad = in_pad + (unsigned)W * 128u;      // (the residual's first rows lie one row further up than the input's)
    const __amdgpu_buffer_rsrc_t rhi = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in_hi - in_pad), 0, nbytes + in_pad, 0x00020000);
    const __amdgpu_buffer_rsrc_t rlo = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in_lo - in_pad / 2), 0, (nbytes + in_pad) / 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrh = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(RES ? a.res_hi : a.in_hi) - res_pad), 0, nbytes + res_pad, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrl = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? (const void*)a.res_lo : (const void*)a.in_hi), 0, nbytes / 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t ryh = __builtin_amdgcn_make_buffer_rsrc((void*)a.out_hi, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ryl = __builtin_amdgcn_make_buffer_rsrc((void*)a.out_lo, 0, OUT8 ? nbytes / 2 : nbytes, 0x00020000);

    // ---- DMA pieces of a block whose first input row is yr (columns x0 - 1 ..): slot i of the wave = a_hi piece 2 i + c (i < 5: 8 pixels x 8 slots), a_lo8 piece
    // 2 (i - 5) + c (i < 8: 16 pixels x 4 slots), the residual's fp16 rows yr - 2, yr - 1 -- the output rows finished in the block's two steps -- of columns x0 ..
    // x0 + 31, piece 2 (i - 8) + c (8 pixels x 8 slots).  d_off: the lane's source offset relative to the block's origin
    // Per range the lane's source offsets of the wave's pieces are formed ONCE (doff[i]: offset relative to the block's origin, bit 0 = the piece's row of the
    // block for this lane, 0xFFFF0000 where the lane fetches nothing: columns outside the image, ring padding, pieces that do not exist); a block then costs a
    // piece one select on its rows' validity instead of ~25 instructions of address arithmetic (the first PMC look: 4.6 VALU instructions per MFMA).
    constexpr int NT = RES ? 10 : 8;           // table entries: the residual's pieces 2, 3 of a wave are its pieces 0, 1 one row further down
    const unsigned tab = lds0 + (unsigned)OFF_TAB + (unsigned)(tid * 4);
    unsigned dt[2];
    auto piece_table = [&](int x0) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            unsigned d_off, d_r, d_cc;
            bool ok;
            if (i < 5) {
                const unsigned q = (unsigned)((2 * i + c) * 8 + (lane >> 3));
                d_r = q >= (unsigned)XW ? 1u : 0u;
                d_cc = q - d_r * (unsigned)XW;
                const unsigned sl = (unsigned)(lane & 7) ^ ((d_cc >> 1) & 7u);
                d_off = ((d_r * (unsigned)W + d_cc) << 7) | (sl << 4);
                ok = (q < 2u * XW) & (d_cc < 34u) & ((unsigned)(x0 - 1 + (int)d_cc) < (unsigned)W);
            } else if (i < 8) {
                const unsigned q = (unsigned)((2 * (i - 5) + c) * 16 + (lane >> 2));
                d_r = q >= (unsigned)XW ? 1u : 0u;
                d_cc = q - d_r * (unsigned)XW;
                const unsigned sl = (unsigned)(lane & 3) ^ ((d_cc >> 2) & 3u);
                d_off = ((d_r * (unsigned)W + d_cc) << 6) | (sl << 4);
                ok = (q < 2u * XW) & (d_cc < 34u) & ((unsigned)(x0 - 1 + (int)d_cc) < (unsigned)W);
            } else {
                const unsigned q = (unsigned)((2 * (i - 8) + c) * 8 + (lane >> 3));
                d_r = q >> 5;
                d_cc = q & 31u;
                const unsigned sl = (unsigned)(lane & 7) ^ ((d_cc >> 1) & 7u);
                d_off = ((d_r * (unsigned)W + d_cc) << 7) | (sl << 4);
                ok = (unsigned)(x0 + (int)d_cc) < (unsigned)W;
            }
            *(__attribute__((address_space(3))) unsigned*)(tab + (unsigned)(i * 512)) = ok ? (d_off | d_r) : kOOR;
        }
    };
    // piece i of the block with ring slot `slot` whose first input row is yr; ok0 / ok1: the block's two rows lie inside the image (and the block is wanted);
    // for the residual's pieces: the rows yr - 2, yr - 1
    auto piece_word = [&](int i) { return *(const __attribute__((address_space(3))) unsigned*)(tab + (unsigned)((i < 10 ? i : i - 2) * 512)); };
    auto piece_issue = [&](int i, unsigned d, int slot, int yr, int x0, int b, bool ok0, bool ok1, bool rk0, bool rk1) {
        const bool rowok = i >= 8 ? (i >= 10 ? rk1 : rk0) : (d & 1u) ? ok1 : ok0;
        const unsigned off = rowok ? (d & ~1u) : kOOR;
        if (i < 8) {
            const unsigned pix = (unsigned)((b * H + yr + RB) * W + x0 - 1 + 2);
            if (i < 5) {
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(2 * i + c < 9 ? slot * BLKB + (2 * i + c) * 1024 : OFF_DUMP);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rhi, (__attribute__((address_space(3))) void*)(smem + dst), 16, off, (unsigned)__builtin_amdgcn_readfirstlane((int)(pix * 128u)), 0, SQ_LDAUX);
            } else {
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(2 * (i - 5) + c < 5 ? OFF_LO8 + slot * BLKB8 + (2 * (i - 5) + c) * 1024 : OFF_DUMP);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rlo, (__attribute__((address_space(3))) void*)(smem + dst), 16, off, (unsigned)__builtin_amdgcn_readfirstlane((int)(pix * 64u)), 0, SQ_LDAUX);
            }
        } else {
            const unsigned org = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((b * H + yr - 2 + RB + 1 + (i >= 10 ? 1 : 0)) * W + x0 + 2) * 128u));
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(OFF_RES + slot * RESBLKB + (2 * (i - 8) + c) * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rrh, (__attribute__((address_space(3))) void*)(smem + dst), 16, off, org, 0, SQ_LDAUX);
        }
    };
    // ---- LDS addressing.  fp16 rows: pixel col at col * 128, 16-B slot s at s ^ ((col >> 1) & 7); B fragment (dx, ks): lane (j, hh) reads slot 2 ks + hh of column
    // j + dx.  fp8 rows: pixel at col * 64, slot s (16 channels) at s ^ ((col >> 2) & 3); B fragment dx: lane (j, hh) reads slots 2 hh, 2 hh + 1 of column j + dx
    unsigned fa[3], fq[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int cc = j + dx;
        fa[dx] = lds0 + (unsigned)(cc * 128 + ((((cc >> 1) & 7) ^ hh) << 4));
        fq[dx] = lds0 + (unsigned)(cc * 64 + (((2 * hh) ^ ((cc >> 2) & 3)) << 4));
        asm volatile("" : "+v"(fa[dx]), "+v"(fq[dx]));
    }
    auto read8 = [&](int dx, unsigned rowoff) {
        const unsigned ad = fq[dx] + rowoff;                  // (row offsets are multiples of 256: bit 4 is the address's own)
        const u4_t lo4 = *(lds_u4_t)(ad), hi4 = *(lds_u4_t)(ad ^ 16u);
        return i8v_t{(int)lo4[0], (int)lo4[1], (int)lo4[2], (int)lo4[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
    };
    // fp8 image of an a_hi row: 34 pixels x four 16-channel groups = 136 units, two per thread of the workgroup (units behind the last repeat unit 135)
    unsigned cv_src[2], cv_dst[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int un = min(u * 128 + tid, 135);
        const int q = un % 34, s8 = un / 34;
        cv_src[u] = lds0 + (unsigned)(q * 128 + (((2 * s8) ^ ((q >> 1) & 7)) << 4));      // slot 2 s8; slot 2 s8 + 1 is this address ^ 16
        cv_dst[u] = lds0 + (unsigned)OFF_Q8 + (unsigned)(q * 64 + ((s8 ^ ((q >> 2) & 3)) << 4));
    }
    const float quarter = 4.0f;                               // the source is DIVIDED by the scale operand
    // four packed fp16 pairs -> two words of four fp8 each (one block, the conversions of the two words alternating, a wait state at its end: conv64_q8.hip)
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
    // this lane's two 16-byte slots (channels 32 c + 16 o + 8 hh .. +7) of pixel column j: in the residual ring at slot (4 c + 2 o + hh) ^ ((j >> 1) & 7)
    unsigned ra[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) ra[o] = lds0 + (unsigned)OFF_RES + (unsigned)(j * 128 + (((4 * c + 2 * o + hh) ^ ((j >> 1) & 7)) << 4));
    const unsigned lane_ob = (unsigned)(j * 128 + (32 * c + 8 * hh) * 2);      // byte offset of slot o = 0 of output column j inside a row of the stream tensors
    const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    float16_t acc[4];                          // step t (input row ya - 1 + t): output row t + 1 - dy (relative to ya - 1) in acc[(t + 3 - dy) & 3]
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = zero16;
    half8_t fx[3];                             // fp16 fragment of chunk n in fx[n % 3], read two chunks ahead
    i8v_t f8[2];                               // [0]: fragment dx of the row's fp8 image, [1]: of the a_lo row
    u4_t rh, cvw[2];
    u2_t xl[2][2];                             // the residual's fp8 low words, requested two steps ahead (set = step parity)
    unsigned sh[4], sl[4];
    xl[0][0] = xl[0][1] = xl[1][0] = xl[1][1] = u2_t{0u, 0u};

    while (item < item_end) {
        const int s0 = item % nyb;
        const int t_ = item / nyb;
        const int pxi = t_ % px, b = t_ / px;
        const int s1 = min(nyb, s0 + (item_end - item));
        item += s1 - s0;
        const int x0 = pxi * TW;
        const int ya = RB * s0, yb = RB * s1;
        const int nblk = (yb - ya) / RB + 2;                  // steps t = 0 .. yb - ya + 3: input rows ya - 1 .. yb + 2 (the last output row completes with row yb and is finished a step later)
#if SQ_ABL & 128     // (timing experiment: every store is issued and dropped by the buffer's range check)
        const unsigned vo = kOOR, vo8 = kOOR;
#elif SQ_ABL & 32    // (timing experiment: every store instruction writes ONE contiguous KiB / half KiB of the row instead of 32 / 16-byte pieces 128 / 64 bytes apart)
        const unsigned vo = (unsigned)(c * 2048 + lane * 16) - 0u * lane_ob;
        const unsigned vo8 = (unsigned)(c * 1024 + lane * 8);
#else
        const unsigned vo = (x0 + j < W) ? lane_ob : kOOR;
        const unsigned vo8 = (x0 + j < W) ? lane_ob >> 1 : kOOR;
#endif

        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                         // both waves have left the previous range: the rings are free
        asm volatile("" ::: "memory");
        piece_table(x0);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int yr = ya - 1 + RB * kb;
#pragma unroll
            for (int i = 0; i < NP; ++i)
                piece_issue(i, piece_word(i), kb, yr, x0, b, (unsigned)yr < (unsigned)H, (unsigned)(yr + 1) < (unsigned)H, (unsigned)(yr - 2) < (unsigned)H, (unsigned)(yr - 1) < (unsigned)H);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // the fp8 image of the first input row (row ya - 1: ring slot 0, row 0) -> image row (ya - 1) & 1 = 1
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const u4_t w0 = *(lds_u4_t)(cv_src[u]), w1 = *(lds_u4_t)(cv_src[u] ^ 16u);
            const u4_t d = cvt16(w0, w1);
            const unsigned adr = cv_dst[u];
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(adr), "v"(d), "n"(ROWB8) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        fx[0] = *(lds_h8_t)(fa[0]);
        fx[1] = *(lds_h8_t)(fa[0] ^ 32u);
        f8[0] = read8(0, (unsigned)(OFF_Q8 + ROWB8));
        f8[1] = read8(0, (unsigned)OFF_LO8);
        int xblk = 0;                                         // ring slot of the block

        auto block = [&](int k, auto BUF_) __attribute__((always_inline)) {
            constexpr int BUF = decltype(BUF_)::value;        // k & 1: the accumulator slots repeat every four rows
            const int Rk = ya - 1 + RB * k;                   // first input row of this block
            const int xnext = xblk + 1 == NRING ? 0 : xblk + 1;
            const int xnext2 = xnext + 1 == NRING ? 0 : xnext + 1;
            const bool live = RB * (k + 2) <= yb - ya + 3;    // the block after the next has rows somebody wants
            const int yrn = Rk + 2 * RB;
            const bool nok0 = live & ((unsigned)yrn < (unsigned)H), nok1 = live & ((unsigned)(yrn + 1) < (unsigned)H);
            const bool nrk0 = live & ((unsigned)(yrn - 2) < (unsigned)H), nrk1 = live & ((unsigned)(yrn - 1) < (unsigned)H);

            auto step = [&](auto E_) __attribute__((always_inline)) {
                constexpr int e = decltype(E_)::value;
                constexpr int T4 = 2 * BUF + e;               // step index mod 4
                const int r = Rk + e;
                // input row r: a_hi / a_lo8 at ring (xblk, e), its fp8 image at image row r & 1 = 1 - e (ya is even); the NEXT row: ring (e == 0 ? (xblk, 1) : (xnext, 0)), image row e
                const unsigned xo_cur = (unsigned)__builtin_amdgcn_readfirstlane(xblk * BLKB + e * ROWB);
                const unsigned xo_nxt = (unsigned)__builtin_amdgcn_readfirstlane(e == 0 ? xblk * BLKB + ROWB : xnext * BLKB);
                const unsigned lo_cur = (unsigned)__builtin_amdgcn_readfirstlane(OFF_LO8 + xblk * BLKB8 + e * ROWB8);
                const unsigned lo_nxt = (unsigned)__builtin_amdgcn_readfirstlane(OFF_LO8 + (e == 0 ? xblk * BLKB8 + ROWB8 : xnext * BLKB8));
                constexpr unsigned q_cur = (unsigned)(OFF_Q8 + (1 - e) * ROWB8), q_nxt = (unsigned)(OFF_Q8 + e * ROWB8);
                constexpr int S = T4 & 3;                     // accumulator slot of output row r - 2
                const int orow = r - 2;
                const bool ook = (orow >= ya) & (orow < yb);
                const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)(ook ? (unsigned)((b * H + orow) * W + x0) * 128u : kOOR));
                const bool ook2 = (orow + 2 >= ya) & (orow + 2 < yb);
                const unsigned so2 = (unsigned)__builtin_amdgcn_readfirstlane((int)(ook2 ? (unsigned)((b * H + orow + 2) * W + x0) * 64u : kOOR));
                const unsigned ro = (unsigned)__builtin_amdgcn_readfirstlane(xblk * RESBLKB + e * RESROWB);

                auto op_dma = [&](auto I_, auto HALF_) __attribute__((always_inline)) {
                    constexpr int i = decltype(I_)::value, half = decltype(HALF_)::value;
                    if constexpr (half == 0) dt[i & 1] = piece_word(i);
                    else piece_issue(i, dt[i & 1], xnext2, yrn, x0, b, nok0, nok1, nrk0, nrk1);
                };
                auto op_cvr = [&](auto U_) __attribute__((always_inline)) {
                    constexpr int u = decltype(U_)::value;
                    cvw[0] = *(lds_u4_t)(cv_src[u] + xo_nxt);
                    cvw[1] = *(lds_u4_t)((cv_src[u] ^ 16u) + xo_nxt);
                };
                auto op_cvw = [&](auto U_) __attribute__((always_inline)) {
                    constexpr int u = decltype(U_)::value;
                    const u4_t d = cvt16(cvw[0], cvw[1]);
                    const unsigned adr = cv_dst[u];
                    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(adr), "v"(d), "n"(e * ROWB8) : "memory");
                };
                auto op_rhi = [&](auto O_) __attribute__((always_inline)) {
                    constexpr int o = decltype(O_)::value;
                    rh = *(lds_u4_t)(ra[o] + ro);
                };
                auto op_add = [&](auto O_, auto K0_) __attribute__((always_inline)) {      // acc += res_hi + res_lo8 2^-9 for channel pairs k0, k0 + 1 of slot o (conv64_q8.hip's arithmetic)
                    constexpr int o = decltype(O_)::value, k0 = decltype(K0_)::value;
#pragma unroll
                    for (int k = k0; k < k0 + 2; ++k) {
                        float v0 = acc[S][8 * o + 2 * k], v1 = acc[S][8 * o + 2 * k + 1];
                        v0 = mix_lo(rh[k], 1.0f, v0); v1 = mix_hi(rh[k], 1.0f, v1);
                        const f2_t f = (k & 1) ? __builtin_amdgcn_cvt_pk_f32_fp8((int)xl[T4 & 1][o][k >> 1], true) : __builtin_amdgcn_cvt_pk_f32_fp8((int)xl[T4 & 1][o][k >> 1], false);
                        v0 = __builtin_fmaf(f[0], 0.001953125f, v0); v1 = __builtin_fmaf(f[1], 0.001953125f, v1);
                        acc[S][8 * o + 2 * k] = v0; acc[S][8 * o + 2 * k + 1] = v1;
                    }
                };
                auto op_act = [&](auto O_, auto E0_) __attribute__((always_inline)) {
                    constexpr int o = decltype(O_)::value, e0 = decltype(E0_)::value;
#pragma unroll
                    for (int q = 8 * o + e0; q < 8 * o + e0 + 4; ++q) acc[S][q] = __builtin_fmaxf(acc[S][q], acc[S][q] * a.slope);
                };
                auto op_spl = [&](auto O_, auto K0_) __attribute__((always_inline)) {
                    constexpr int o = decltype(O_)::value, k0 = decltype(K0_)::value;
#pragma unroll
                    for (int k = k0; k < k0 + 2; ++k) split2(acc[S][8 * o + 2 * k], acc[S][8 * o + 2 * k + 1], -2048.f, sh[k], sl[k]);
                };
                auto op_st = [&](auto O_) __attribute__((always_inline)) {
                    constexpr int o = decltype(O_)::value;
                    const u4_t dh = {sh[0], sh[1], sh[2], sh[3]};
                    constexpr unsigned ostep = (SQ_ABL & 32) ? 1024u : 32u;
                    __builtin_amdgcn_raw_buffer_store_b128(dh, ryh, vo + (unsigned)o * ostep, so, SQ_STAUX);
                    if (!OUT8) {
                        const u4_t dl = {sl[0], sl[1], sl[2], sl[3]};
                        __builtin_amdgcn_raw_buffer_store_b128(dl, ryl, vo + (unsigned)o * ostep, so, SQ_STAUX);
                    } else {
                        unsigned p0, p1;
                        cvt4(sl[0], sl[1], sl[2], sl[3], p0, p1);
                        __builtin_amdgcn_raw_buffer_store_b64(u2_t{p0, p1}, ryl, vo8 + (unsigned)o * (ostep / 2), so >> 1, SQ_STAUX);
                    }
                };
                auto op_xlo = [&](auto O_) __attribute__((always_inline)) {
                    constexpr int o = decltype(O_)::value;
                    xl[T4 & 1][o] = __builtin_amdgcn_raw_buffer_load_b64(rrl, vo8 + (unsigned)(o * 16), so2, 0);
                };

                constexpr OpList L = step_ops(EPI), LD = dma_ops(e, NP);
                static_assert(cvw_in_time(EPI), "the image's writes must precede the barrier");
                auto chunk = [&](auto A_) __attribute__((always_inline)) {
                    constexpr int ai = decltype(A_)::value;               // fp16 fragment (dx, ks) = (ai / 4, ai % 4)
                    constexpr int dx = ai / 4, ks = ai % 4;
                    if (ai == 10) {
                        // the next row's fp8 image is written; e = 0: the block after this one has landed (its pieces and everything older; what the wave has issued
                        // behind its last piece may still be on its way)
                        if (e == 0 && !(SQ_ABL & 3)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(vm_behind(EPI)) : "memory");
                        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                    if constexpr (ai == 0 && (SQ_ABL & 8) != 0) asm volatile("" : "+v"(acc[S]));
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int sl_ = (T4 + 3 - dy) & 3;                // output row r + 1 - dy
                        acc[sl_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w16[(dy * 3 + dx) * 4 + ks], fx[ai % 3], (dy == 0 && ai == 0) ? zero16 : acc[sl_], 0, 0, 0);
                    }
                    constexpr int a2 = (ai + 2) % 12;
                    fx[(ai + 2) % 3] = *(lds_h8_t)((fa[a2 / 4] ^ (unsigned)((a2 % 4) * 32)) + (ai + 2 < 12 ? xo_cur : xo_nxt));
                    if constexpr (ks == 1 || ks == 3) {
                        // the tap column's fp8 products behind its k-slices 1 (image of the row x w_lo8) and 3 (a_lo row x w_hi8): between an fp16 and an fp8 MFMA on one
                        // accumulator the result is not forwarded (conv64_q8.hip) -- three accumulators in turn keep dependent MFMAs two issues apart
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int sl_ = (T4 + 3 - dy) & 3;
                            acc[sl_] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ks == 1 ? wl8[dy * 3 + dx] : wh8[dy * 3 + dx], f8[ks == 1 ? 0 : 1], acc[sl_], 0, 0, 0, scale_a, 0, scale_b);
                        }
                        // the set's next fragment: the next tap column of this row -- or, behind the last one, column 0 of the next row (its image: published by the barrier at
                        // chunk 10, so that read waits for chunk 11)
                        if constexpr (ks == 1 && dx < 2) f8[0] = read8(dx + 1, q_cur);
                        if constexpr (ks == 3 && dx < 2) f8[1] = read8(dx + 1, lo_cur);
                        if constexpr (ks == 3 && dx == 2) { f8[0] = read8(0, q_nxt); f8[1] = read8(0, lo_nxt); }
                    }
                    {
                        constexpr int dlo = ai < 8 ? ai * LD.n / 8 : LD.n, dhi = ai < 8 ? (ai + 1) * LD.n / 8 : LD.n;
                        auto rund = [&](auto I_) __attribute__((always_inline)) {
                            constexpr int I = decltype(I_)::value;
                            if constexpr (I >= dlo && I < dhi) {
                                constexpr Op o = LD.op[I];
                                op_dma(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                            }
                        };
#define SQ_OP(I) rund(std::integral_constant<int, I>{});
                        SQ_OP(0) SQ_OP(1) SQ_OP(2) SQ_OP(3) SQ_OP(4) SQ_OP(5) SQ_OP(6) SQ_OP(7) SQ_OP(8) SQ_OP(9) SQ_OP(10) SQ_OP(11) SQ_OP(12) SQ_OP(13) SQ_OP(14) SQ_OP(15)
                        SQ_OP(16) SQ_OP(17) SQ_OP(18) SQ_OP(19) SQ_OP(20) SQ_OP(21) SQ_OP(22) SQ_OP(23) SQ_OP(24) SQ_OP(25) SQ_OP(26) SQ_OP(27) SQ_OP(28) SQ_OP(29) SQ_OP(30) SQ_OP(31)
#undef SQ_OP
                        constexpr int lo_ = ai * L.n / 12, hi_ = (ai + 1) * L.n / 12;
                        auto run = [&](auto I_) __attribute__((always_inline)) {
                            constexpr int I = decltype(I_)::value;
                            if constexpr (I >= lo_ && I < hi_) {
                                constexpr Op o = L.op[I];
                                if constexpr (o.kind == OP_CVR) op_cvr(std::integral_constant<int, o.a>{});
                                if constexpr (o.kind == OP_CVW) op_cvw(std::integral_constant<int, o.a>{});
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
                    for (int i_ = 0; i_ < 3; ++i_) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i_ == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x006, SQ_FILL, 0);
                    }
                    if (ks == 1 || ks == 3) {
#pragma unroll
                        for (int i_ = 0; i_ < 3; ++i_) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (i_ == 0) __builtin_amdgcn_sched_group_barrier(0x100, (ks == 3 && dx == 2) ? 4 : 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x006, 2 * SQ_FILL, 0);
                        }
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
    if ((long long)a.B * a.H * a.W * 128 + (long long)((RB + 1) * a.W + 2) * 128 >= (1ll << 32) - 65536) return false;
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
    // Small launch sets lose here (a range start costs three barriers and an exposed DMA round trip: 28 against 17 us for the smallest set of an a2 frame,
    // profiles/r04/k_conv64_sq_frame_launches.txt) -- and still take this kernel: which form a layer runs on must not depend on how many tiles share its
    // launch, or a tile's bits would (the two forms differ in the order of their sums).  MOE_SQ_MIN_ITEMS = blocks per workgroup below which the patch form runs (experiments).
    static const long long min_items = [] { const char* e = getenv("MOE_SQ_MIN_ITEMS"); return e ? atoll(e) : 0ll; }();
    if (items < min_items * 2 * max_groups) return false;
    const int G = (int)std::min<long long>(items, 2ll * max_groups);
    const dim3 grid(G), blk(128);
    if (epi == 0) conv64_sq_kernel<0, true><<<grid, blk, LDS_PLAIN, s>>>(a);
    else if (epi == 1) conv64_sq_kernel<1, true><<<grid, blk, LDS_PLAIN, s>>>(a);
    else if (a.out8) conv64_sq_kernel<2, true><<<grid, blk, LDS_RES, s>>>(a);
    else conv64_sq_kernel<2, false><<<grid, blk, LDS_RES, s>>>(a);
    return true;
}
