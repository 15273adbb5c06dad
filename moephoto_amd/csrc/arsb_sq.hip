// arsb_sq.hip -- one EXACT ARSB  y = x + conv_2(PReLU(conv_1(x)))  of MOE_PREC_MIXED's leading blocks (python/models.py:53-80 of the reference; both convs with
// split operands: fp16 product + two fp8 correction products, conv64_q8.hip's arithmetic) in ONE launch: conv_1's rows never leave the CU.
//
// conv64_sq.hip runs these layers one conv per launch and is held by its bytes (stores dropped -> -21 %): per pixel conv_1 reads 192 and
// writes 192 bytes, conv_2 reads 192 + 192 (residual) and writes 192 -- 960.  Here 192 (x 36 / 30 for the column halo) in, 192 out.
//
//   workgroup     FOUR waves, one workgroup per CU: waves (P, c) hold conv_1's weights for the channels 32c .. 32c+31, waves (C, c) conv_2's -- 288 registers each,
//                 conv64_sq.hip's sets (36 fp16 A fragments of w_hi + 9 + 9 fp8 ones of w_lo 2^8, w_hi 2^8).  All four run conv64_sq.hip's row step (36 fp16 + 18
//                 fp8 MFMAs into three accumulators: 2,304 MFMA cycles), the P waves on the x rings, the C waves on the m rings
//   row step r    P: x row r into the m rows r-1, r, r+1; epilogue of m row r-2: PReLU, hi / lo split, and THREE LDS writes per 16-byte slot -- m_hi, the fp8 low
//                 word, the fp8 image of m_hi -- zeros outside the image (conv_2 pads with zeros); its share of the x row's fp8 image and of the DMA.
//                 C: m row r-3 into the output rows r-4, r-3, r-2; epilogue of output row r-5: + x_hi + x_lo8 (both from the x rings: no residual loads),
//                 split, stores
//   columns       30 outputs from 32 m columns from 34 x columns (arsb32c.hip): W = 256 takes nine columns (+12.5 % MFMAs against conv64_sq.hip's eight)
//   rings         x: six two-row blocks (rows r-5 .. r in use, one landed, one in flight) of a_hi (36 pixels x 128 B) and a_lo8 (x 64 B), the row's fp8 image
//                 (two rows); m: four rows of m_hi, m_lo8, m image.  ONE workgroup barrier per row step (chunk 10): it publishes m row r-2, the next x image
//                 row and -- first step of a block, behind a counted vmcnt -- the block after it
//   LDS           x 6 x (9,216 + 5,120) + 4,608 + m 4 x (4,608 + 2,304 + 2,304) + 1,024 + DMA offset table 5,120 = 133,632 bytes
//
// Arithmetic: per conv the operands, products and scales of conv64_q8.hip / conv64_sq.hip; m crosses as the same (fp16, fp8 low word) pair those kernels store.
// The results agree with the two-launch form to the fp32 rounding of a row's sums (tests/test_gpu_parity.py).
#include "common.h"
#include "rowtile.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#ifndef AQ_FILL
#define AQ_FILL 5
#endif

namespace {

constexpr int RB = 2, TW = 30, XW = 36;        // rows per block; output columns; pixels of a ring row (34 x columns used, 32 m columns)
constexpr int ROWB = XW * 128, ROWB8 = XW * 64;   // 4,608 / 2,304
constexpr int BLKB = RB * ROWB;                // 9,216
constexpr int BLKB8 = 5 * 1024;                // 4,608 used
constexpr int NB = 6;                          // x ring blocks
constexpr int OFF_LO8 = NB * BLKB;             // 55,296
constexpr int OFF_Q8 = OFF_LO8 + NB * BLKB8;   // 86,016: fp8 image of x rows (two)
constexpr int OFF_M = OFF_Q8 + 2 * ROWB8;      // 90,624: m_hi, four rows
constexpr int OFF_M8 = OFF_M + 4 * ROWB;       // 109,056: m_lo8
constexpr int OFF_MQ = OFF_M8 + 4 * ROWB8;     // 118,272: fp8 image of m_hi
constexpr int OFF_DUMP = OFF_MQ + 4 * ROWB8;   // 127,488
constexpr int OFF_TAB = OFF_DUMP + 1024;       // 128,512: [entry 5][thread 256] words
constexpr int NPW = 5;                         // DMA pieces a wave issues per block: a_hi 4 i + w (i < 3; nine exist), a_lo8 4 (i - 3) + w (five exist)
constexpr int LDS_BYTES = OFF_TAB + NPW * 1024;   // 133,632

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef int i8v_t __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;
typedef const __attribute__((address_space(3))) u4_t* lds_u4_t;
typedef const __attribute__((address_space(3))) u2_t* lds_u2_t;

enum OpKind : int { OP_NONE = 0, OP_DMA, OP_CVR, OP_CVW, OP_ACT, OP_SPL, OP_MW, OP_RHI, OP_ADD, OP_ST };
struct Op { int kind, a, b, c; };
struct OpList {
    int n = 0;
    Op op[48] = {};
    constexpr void push(int kind, int a = 0, int b = 0, int c = 0) { op[n] = Op{kind, a, b, c}; ++n; }
};
// DMA ops (e = 1, dealt to chunks 0..7): half 0 the lane's table word, half 1 select + issue, the word of piece i + 1 read in front of piece i's issue
constexpr OpList dma_ops(int e)
{
    OpList r;
    if (e == 1) {
        for (int i = 0; i < NPW; ++i) { r.push(OP_DMA, i, 0); if (i >= 1) r.push(OP_DMA, i - 1, 1); }
        r.push(OP_DMA, NPW - 1, 1);
    }
    return r;
}
// step ops of a wave (dealt to chunks 0..11 behind the chunk's DMA ops).  Both: the thread's unit of the next x row's fp8 image.  P: m row r - 2: PReLU, split, the three LDS
// writes per slot.  C: output row r - 5: residual words from the x rings, sums, split, stores.  Everything a P wave writes to LDS sits in front of chunk 10.
constexpr OpList step_ops(int role)
{
    OpList r;
    if (role == 0) {
        r.push(OP_CVR, 0);
        r.push(OP_ACT, 0, 0); r.push(OP_ACT, 0, 4);
        r.push(OP_CVW, 0);
        r.push(OP_SPL, 0, 0); r.push(OP_SPL, 0, 2); r.push(OP_MW, 0);
        r.push(OP_ACT, 1, 0); r.push(OP_ACT, 1, 4);
        r.push(OP_SPL, 1, 0); r.push(OP_SPL, 1, 2); r.push(OP_MW, 1);
    } else {
        r.push(OP_CVR, 0);
        r.push(OP_RHI, 0);
        r.push(OP_RHI, 1);
        r.push(OP_CVW, 0);
        r.push(OP_ADD, 0, 0); r.push(OP_ADD, 0, 2);
        r.push(OP_SPL, 0, 0); r.push(OP_SPL, 0, 2); r.push(OP_ST, 0);
        r.push(OP_ADD, 1, 0); r.push(OP_ADD, 1, 2);
        r.push(OP_SPL, 1, 0); r.push(OP_SPL, 1, 2); r.push(OP_ST, 1);
    }
    return r;
}
constexpr int ops_chunks(int role) { return role == 0 ? 10 : 12; }      // the P waves' ops are dealt to the chunks in front of the barrier
constexpr bool lds_writes_in_time()
{
    const OpList l = step_ops(1);
    for (int i = 0; i < l.n; ++i) if (l.op[i].kind == OP_CVW && i >= 10 * l.n / 12) return false;
    return true;
}
// VM operations a C wave issues behind its last DMA piece (chunk 7 of step e = 1) and in front of the barrier of the next step (head of chunk 10): conv64_sq.hip
constexpr int vm_behind(int role)
{
    int n = 0;
    const OpList l = step_ops(role);
    for (int i = 7 * l.n / 12; i < l.n; ++i) n += l.op[i].kind == OP_ST ? 2 : 0;
    for (int i = 0; i < 10 * l.n / 12; ++i) n += l.op[i].kind == OP_ST ? 2 : 0;
    return n;
}

struct ArsbSqArgsK {       // (the launcher's copy of ArsbSqArgs: common.h)
    const half_t* x_hi; const unsigned char* x_lo8;
    half_t* y_hi; void* y_lo;
    const half_t* w16[2]; const unsigned char* wh8[2]; const unsigned char* wl8[2];
    float slope;
    int B, H, W;
};

// OUT8: the output's low part is the fp8 word of the chain (else fp16: the last exact block feeds the fused single-pass ARSBs)
template <bool OUT8>
__global__ __launch_bounds__(256) void arsb_sq_kernel(ArsbSqArgsK a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr unsigned kOOR = 0xFFFF0000u;
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\ts_nop 3" ::: "memory");      // MODE.FP16_OVFL: the fp8 conversions saturate (conv64_q8.hip)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wv >> 1, c = wv & 1;      // role 0: conv_1 (P), 1: conv_2 (C)
    const int j = lane & 31, hh = lane >> 5;
    const int H = a.H, W = a.W;

    const int px = (W + TW - 1) / TW, nyb = H / RB;
    const int g = blockIdx.x, G = gridDim.x;
    const long long nitems = (long long)a.B * px * nyb;
    int item = (int)(nitems * g / G);
    const int item_end = (int)(nitems * (g + 1) / G);
    if (item >= item_end) return;

    // ---- weights of the wave's conv (conv64_sq.hip)
    half8_t w16[36];
    i8v_t wl8[9], wh8[9];
    {
        const int wi = lane & 31, wq = wi >> 3;
        const int src = (lane & 32) | (16 * (wq >> 1) + 8 * ((wi >> 2) & 1) + 4 * (wq & 1) + (wi & 3));
        const half_t* p16 = role ? a.w16[1] : a.w16[0];
        const unsigned char* pl = role ? a.wl8[1] : a.wl8[0];
        const unsigned char* ph = role ? a.wh8[1] : a.wh8[0];
#pragma unroll
        for (int f = 0; f < 36; ++f) w16[f] = *(const half8_t*)(p16 + ((f * 2 + c) * 64 + src) * 8);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            wl8[t] = *(const i8v_t*)(pl + ((t * 2 + c) * 64 + src) * 32);
            wh8[t] = *(const i8v_t*)(ph + ((t * 2 + c) * 64 + src) * 32);
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
    int scale_a = 127 - 19, scale_b = 127 + 2;
    asm volatile("" : "+v"(scale_a), "+v"(scale_b));

    const unsigned nbytes = (unsigned)a.B * H * W * 128u;
    const unsigned in_pad = (unsigned)(RB * W + 2) * 128u;
    const __amdgpu_buffer_rsrc_t rhi = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.x_hi - in_pad), 0, nbytes + in_pad, 0x00020000);
    const __amdgpu_buffer_rsrc_t rlo = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.x_lo8 - in_pad / 2), 0, (nbytes + in_pad) / 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t ryh = __builtin_amdgcn_make_buffer_rsrc((void*)a.y_hi, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ryl = __builtin_amdgcn_make_buffer_rsrc((void*)a.y_lo, 0, OUT8 ? nbytes / 2 : nbytes, 0x00020000);

    // ---- DMA source offsets of a range in an LDS table (conv64_sq.hip): entry i of the wave = a_hi piece 4 i + wv (i < 3), a_lo8 piece 4 (i - 3) + wv; ring column 0
    // is image column x0 - 2
    const unsigned tab = lds0 + (unsigned)OFF_TAB + (unsigned)(tid * 4);
    unsigned dt[2];
    auto piece_table = [&](int x0) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            unsigned d_off, d_r, d_cc;
            bool ok;
            if (i < 3) {
                const unsigned q = (unsigned)((4 * i + wv) * 8 + (lane >> 3));
                d_r = q >= (unsigned)XW ? 1u : 0u;
                d_cc = q - d_r * (unsigned)XW;
                const unsigned sl = (unsigned)(lane & 7) ^ ((d_cc >> 1) & 7u);
                d_off = ((d_r * (unsigned)W + d_cc) << 7) | (sl << 4);
                ok = (q < 2u * XW) & (d_cc < 34u) & ((unsigned)(x0 - 2 + (int)d_cc) < (unsigned)W);
            } else {
                const unsigned q = (unsigned)((4 * (i - 3) + wv) * 16 + (lane >> 2));
                d_r = q >= (unsigned)XW ? 1u : 0u;
                d_cc = q - d_r * (unsigned)XW;
                const unsigned sl = (unsigned)(lane & 3) ^ ((d_cc >> 2) & 3u);
                d_off = ((d_r * (unsigned)W + d_cc) << 6) | (sl << 4);
                ok = (q < 2u * XW) & (d_cc < 34u) & ((unsigned)(x0 - 2 + (int)d_cc) < (unsigned)W);
            }
            *(__attribute__((address_space(3))) unsigned*)(tab + (unsigned)(i * 1024)) = ok ? (d_off | d_r) : kOOR;
        }
    };
    auto piece_word = [&](int i) { return *(const __attribute__((address_space(3))) unsigned*)(tab + (unsigned)(i * 1024)); };
    auto piece_issue = [&](int i, unsigned d, int slot, int yr, int x0, int b, bool ok0, bool ok1) {
        const bool rowok = (d & 1u) ? ok1 : ok0;
        const unsigned off = rowok ? (d & ~1u) : kOOR;
        const unsigned pix = (unsigned)((b * H + yr + RB) * W + x0 - 2 + 2);
        if (i < 3) {
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(4 * i + wv < 9 ? slot * BLKB + (4 * i + wv) * 1024 : OFF_DUMP);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rhi, (__attribute__((address_space(3))) void*)(smem + dst), 16, off, (unsigned)__builtin_amdgcn_readfirstlane((int)(pix * 128u)), 0, 0);
        } else {
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(4 * (i - 3) + wv < 5 ? OFF_LO8 + slot * BLKB8 + (4 * (i - 3) + wv) * 1024 : OFF_DUMP);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rlo, (__attribute__((address_space(3))) void*)(smem + dst), 16, off, (unsigned)__builtin_amdgcn_readfirstlane((int)(pix * 64u)), 0, 0);
        }
    };
    // ---- LDS addressing (conv64_sq.hip): fp16 rows pixel col at col * 128, slot s at s ^ ((col >> 1) & 7); fp8 rows pixel at col * 64, 16-channel slot s at
    // s ^ ((col >> 2) & 3).  B fragments of column j + dx: the same formulas for the x rings (ring column 0 = image column x0 - 2, m column j = image column
    // x0 - 1 + j) and the m rings (m column 0 = image column x0 - 1, output column j = image column x0 + j)
    unsigned fa[3], fq[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int cc = j + dx;
        fa[dx] = lds0 + (unsigned)(cc * 128 + ((((cc >> 1) & 7) ^ hh) << 4));
        fq[dx] = lds0 + (unsigned)(cc * 64 + (((2 * hh) ^ ((cc >> 2) & 3)) << 4));
        asm volatile("" : "+v"(fa[dx]), "+v"(fq[dx]));
    }
    auto read8 = [&](int dx, unsigned rowoff) {
        const unsigned ad = fq[dx] + rowoff;
        const u4_t lo4 = *(lds_u4_t)(ad), hi4 = *(lds_u4_t)(ad ^ 16u);
        return i8v_t{(int)lo4[0], (int)lo4[1], (int)lo4[2], (int)lo4[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
    };
    // fp8 image of an x row: 34 pixels x four 16-channel groups = 136 units, one per thread (the threads behind the last repeat unit 135: same data, same place)
    unsigned cv_src, cv_dst;
    {
        const int un = min(tid, 135);
        const int q = un % 34, s8 = un / 34;
        cv_src = lds0 + (unsigned)(q * 128 + (((2 * s8) ^ ((q >> 1) & 7)) << 4));
        cv_dst = lds0 + (unsigned)OFF_Q8 + (unsigned)(q * 64 + ((s8 ^ ((q >> 2) & 3)) << 4));
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
    // this lane's two 16-byte slots (channels 32 c + 16 o + 8 hh .. +7): P writes m column j -- m_hi slot (4 c + 2 o + hh) ^ ((j >> 1) & 7); the fp8 rows' 16-channel
    // slot 2 c + o, bytes 8 hh .. 8 hh + 7.  C reads the residual at x ring column j + 2 in the same forms.
    unsigned mw[2], mw8[2], xa[2], xa8[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int s = 4 * c + 2 * o + hh, s8 = 2 * c + o, xc = j + 2;
        mw[o] = lds0 + (unsigned)OFF_M + (unsigned)(j * 128 + ((s ^ ((j >> 1) & 7)) << 4));
        mw8[o] = lds0 + (unsigned)OFF_M8 + (unsigned)(j * 64 + ((s8 ^ ((j >> 2) & 3)) << 4) + 8 * hh);
        xa[o] = lds0 + (unsigned)(xc * 128 + ((s ^ ((xc >> 1) & 7)) << 4));
        xa8[o] = lds0 + (unsigned)OFF_LO8 + (unsigned)(xc * 64 + ((s8 ^ ((xc >> 2) & 3)) << 4) + 8 * hh);
    }
    const unsigned lane_ob = (unsigned)(j * 128 + (32 * c + 8 * hh) * 2);
    const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    float16_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = zero16;
    half8_t fx[3];
    i8v_t f8[2];
    u4_t rh[2], cvw[2];
    u2_t rl[2];
    unsigned sh[4], sl[4];

    auto body = [&](auto ROLE_) __attribute__((always_inline)) {
        constexpr int ROLE = decltype(ROLE_)::value;
        while (item < item_end) {
            const int s0 = item % nyb;
            const int t_ = item / nyb;
            const int pxi = t_ % px, b = t_ / px;
            const int s1 = min(nyb, s0 + (item_end - item));
            item += s1 - s0;
            const int x0 = pxi * TW;
            const int ya = RB * s0, yb = RB * s1;
            const int nblk = (yb - ya) / RB + 4;              // steps t = 0 .. yb - ya + 7: x rows ya - 2 .. yb + 5 (the last output row, yb - 1, is finished at x row yb + 4)
            const bool mokx = (unsigned)(x0 - 1 + j) < (unsigned)W;
            const unsigned vo = ((j < TW) & (x0 + j < W)) ? lane_ob : kOOR;
            const unsigned vo8 = ((j < TW) & (x0 + j < W)) ? lane_ob >> 1 : kOOR;

            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                     // everybody has left the previous range: the rings are free
            asm volatile("" ::: "memory");
            piece_table(x0);
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int yr = ya - 2 + RB * kb;
#pragma unroll
                for (int i = 0; i < NPW; ++i) piece_issue(i, piece_word(i), kb, yr, x0, b, (unsigned)yr < (unsigned)H, (unsigned)(yr + 1) < (unsigned)H);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            {      // the fp8 image of the first x row (row ya - 2: ring block 0, row 0; even) -> image row 0
                const u4_t w0 = *(lds_u4_t)(cv_src), w1 = *(lds_u4_t)(cv_src ^ 16u);
                const u4_t d = cvt16(w0, w1);
                const unsigned adr = cv_dst;
                asm volatile("ds_write_b128 %0, %1" ::"v"(adr), "v"(d) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (ROLE == 0) {
                fx[0] = *(lds_h8_t)(fa[0]);
                fx[1] = *(lds_h8_t)(fa[0] ^ 32u);
                f8[0] = read8(0, (unsigned)OFF_Q8);
                f8[1] = read8(0, (unsigned)OFF_LO8);
            } else {      // (the first steps of conv_2 run on whatever the m rings hold: rows nobody reads)
                fx[0] = *(lds_h8_t)(fa[0] + (unsigned)(OFF_M + 1 * ROWB));
                fx[1] = *(lds_h8_t)((fa[0] ^ 32u) + (unsigned)(OFF_M + 1 * ROWB));
                f8[0] = read8(0, (unsigned)(OFF_MQ + 1 * ROWB8));
                f8[1] = read8(0, (unsigned)(OFF_M8 + 1 * ROWB8));
            }
            int xblk = 0;

            auto block = [&](int k, auto BUF_) __attribute__((always_inline)) {
                constexpr int BUF = decltype(BUF_)::value;
                const int Rk = ya - 2 + RB * k;               // first x row of this block
                const int xnext = xblk + 1 == NB ? 0 : xblk + 1;
                const int xnext2 = xnext + 1 == NB ? 0 : xnext + 1;
                const bool live = RB * (k + 2) <= yb - ya + 7;
                const int yrn = Rk + 2 * RB;
                const bool nok0 = live & ((unsigned)yrn < (unsigned)H), nok1 = live & ((unsigned)(yrn + 1) < (unsigned)H);

                auto step = [&](auto E_) __attribute__((always_inline)) {
                    constexpr int e = decltype(E_)::value;
                    constexpr int T4 = 2 * BUF + e;
                    const int r = Rk + e;
                    // P: x row r at ring (xblk, e), image row e (r is even with e); next row: ring (e == 0 ? (xblk, 1) : (xnext, 0)), image row 1 - e.
                    // C: m row r - 3 at m ring row (T4 + 1) & 3, next one (T4 + 2) & 3 -- the row the P waves write in this step
                    const unsigned xo_cur = (unsigned)__builtin_amdgcn_readfirstlane(ROLE == 0 ? xblk * BLKB + e * ROWB : OFF_M + ((T4 + 1) & 3) * ROWB);
                    const unsigned xo_nxt = (unsigned)__builtin_amdgcn_readfirstlane(ROLE == 0 ? (e == 0 ? xblk * BLKB + ROWB : xnext * BLKB) : OFF_M + ((T4 + 2) & 3) * ROWB);
                    const unsigned lo_cur = (unsigned)__builtin_amdgcn_readfirstlane(ROLE == 0 ? OFF_LO8 + xblk * BLKB8 + e * ROWB8 : OFF_M8 + ((T4 + 1) & 3) * ROWB8);
                    const unsigned lo_nxt = (unsigned)__builtin_amdgcn_readfirstlane(ROLE == 0 ? OFF_LO8 + (e == 0 ? xblk * BLKB8 + ROWB8 : xnext * BLKB8) : OFF_M8 + ((T4 + 2) & 3) * ROWB8);
                    const unsigned q_cur = (unsigned)(ROLE == 0 ? OFF_Q8 + e * ROWB8 : OFF_MQ + ((T4 + 1) & 3) * ROWB8);
                    const unsigned q_nxt = (unsigned)(ROLE == 0 ? OFF_Q8 + (1 - e) * ROWB8 : OFF_MQ + ((T4 + 2) & 3) * ROWB8);
                    constexpr int S = T4 & 3;                 // accumulator slot of the row finished in this step (P: m row r - 2, C: output row r - 5)
                    constexpr int MROW = (T4 + 2) & 3;        // m ring row of m row r - 2
                    const bool mok = mokx & ((unsigned)(r - 2) < (unsigned)H);
                    const int orow = r - 5;
                    const bool ook = (orow >= ya) & (orow < yb);
                    const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)(ook ? (unsigned)((b * H + orow) * W + x0) * 128u : kOOR));
                    // the residual's row r - 5: e = 0: block k - 3, row 1; e = 1: block k - 2, row 0
                    const int rb_ = xblk - (e == 0 ? 3 : 2);
                    const int rbs = rb_ < 0 ? rb_ + NB : rb_;
                    const unsigned ro = (unsigned)__builtin_amdgcn_readfirstlane(rbs * BLKB + (e == 0 ? ROWB : 0));
                    const unsigned ro8 = (unsigned)__builtin_amdgcn_readfirstlane(rbs * BLKB8 + (e == 0 ? ROWB8 : 0));

                    auto op_dma = [&](auto I_, auto HALF_) __attribute__((always_inline)) {
                        constexpr int i = decltype(I_)::value, half = decltype(HALF_)::value;
                        if constexpr (half == 0) dt[i & 1] = piece_word(i);
                        else piece_issue(i, dt[i & 1], xnext2, yrn, x0, b, nok0, nok1);
                    };
                    const unsigned xnx = (unsigned)__builtin_amdgcn_readfirstlane(e == 0 ? xblk * BLKB + ROWB : xnext * BLKB);      // the next x row (both roles convert it)
                    auto op_cvr = [&]() __attribute__((always_inline)) {
                        cvw[0] = *(lds_u4_t)(cv_src + xnx);
                        cvw[1] = *(lds_u4_t)((cv_src ^ 16u) + xnx);
                    };
                    auto op_cvw = [&]() __attribute__((always_inline)) {
                        const u4_t d = cvt16(cvw[0], cvw[1]);
                        const unsigned adr = cv_dst;
                        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(adr), "v"(d), "n"((1 - e) * ROWB8) : "memory");
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
                    auto op_mw = [&](auto O_) __attribute__((always_inline)) {      // m row r - 2, slot o: m_hi, its fp8 image, the fp8 low word (zeros outside the image)
                        constexpr int o = decltype(O_)::value;
                        unsigned p0, p1, i0, i1;
                        cvt4(sl[0], sl[1], sl[2], sl[3], p0, p1);
                        cvt4(sh[0], sh[1], sh[2], sh[3], i0, i1);
                        const u4_t dh = {mok ? sh[0] : 0u, mok ? sh[1] : 0u, mok ? sh[2] : 0u, mok ? sh[3] : 0u};
                        const u2_t dl = {mok ? p0 : 0u, mok ? p1 : 0u}, di = {mok ? i0 : 0u, mok ? i1 : 0u};
                        const unsigned a16 = mw[o], a8 = mw8[o];
                        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(a16), "v"(dh), "n"(MROW * ROWB) : "memory");      // (16-bit offset fields: the rings' bases sit in the registers)
                        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a8), "v"(dl), "n"(MROW * ROWB8) : "memory");
                        asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(a8), "v"(di), "n"(OFF_MQ - OFF_M8 + MROW * ROWB8) : "memory");
                    };
                    auto op_rhi = [&](auto O_) __attribute__((always_inline)) {
                        constexpr int o = decltype(O_)::value;
                        rh[o] = *(lds_u4_t)(xa[o] + ro);
                        rl[o] = *(lds_u2_t)(xa8[o] + ro8);
                    };
                    auto op_add = [&](auto O_, auto K0_) __attribute__((always_inline)) {      // acc += x_hi + x_lo8 2^-9 (conv64_q8.hip's arithmetic)
                        constexpr int o = decltype(O_)::value, k0 = decltype(K0_)::value;
#pragma unroll
                        for (int k = k0; k < k0 + 2; ++k) {
                            float v0 = acc[S][8 * o + 2 * k], v1 = acc[S][8 * o + 2 * k + 1];
                            v0 = mix_lo(rh[o][k], 1.0f, v0); v1 = mix_hi(rh[o][k], 1.0f, v1);
                            const f2_t f = (k & 1) ? __builtin_amdgcn_cvt_pk_f32_fp8((int)rl[o][k >> 1], true) : __builtin_amdgcn_cvt_pk_f32_fp8((int)rl[o][k >> 1], false);
                            v0 = __builtin_fmaf(f[0], 0.001953125f, v0); v1 = __builtin_fmaf(f[1], 0.001953125f, v1);
                            acc[S][8 * o + 2 * k] = v0; acc[S][8 * o + 2 * k + 1] = v1;
                        }
                    };
                    auto op_st = [&](auto O_) __attribute__((always_inline)) {
                        constexpr int o = decltype(O_)::value;
                        const u4_t dh = {sh[0], sh[1], sh[2], sh[3]};
                        __builtin_amdgcn_raw_buffer_store_b128(dh, ryh, vo + (unsigned)(o * 32), so, 0);
                        if (!OUT8) {
                            const u4_t dl = {sl[0], sl[1], sl[2], sl[3]};
                            __builtin_amdgcn_raw_buffer_store_b128(dl, ryl, vo + (unsigned)(o * 32), so, 0);
                        } else {
                            unsigned p0, p1;
                            cvt4(sl[0], sl[1], sl[2], sl[3], p0, p1);
                            __builtin_amdgcn_raw_buffer_store_b64(u2_t{p0, p1}, ryl, vo8 + (unsigned)(o * 16), so >> 1, 0);
                        }
                    };

                    constexpr OpList L = step_ops(ROLE), LD = dma_ops(e);
                    static_assert(lds_writes_in_time(), "what the P waves write to LDS must precede the barrier");
                    auto chunk = [&](auto A_) __attribute__((always_inline)) {
                        constexpr int ai = decltype(A_)::value;
                        constexpr int dx = ai / 4, ks = ai % 4;
                        if (ai == 10) {
                            // m row r - 2 and the next x image row are written; e = 0: the block after this one has landed (conv64_sq.hip's counted wait)
                            if (e == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(vm_behind(ROLE)) : "memory");
                            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();
                            asm volatile("" ::: "memory");
                        }
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int sl_ = (T4 + 3 - dy) & 3;
                            acc[sl_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w16[(dy * 3 + dx) * 4 + ks], fx[ai % 3], (dy == 0 && ai == 0) ? zero16 : acc[sl_], 0, 0, 0);
                        }
                        constexpr int a2 = (ai + 2) % 12;
                        fx[(ai + 2) % 3] = *(lds_h8_t)((fa[a2 / 4] ^ (unsigned)((a2 % 4) * 32)) + (ai + 2 < 12 ? xo_cur : xo_nxt));
                        if constexpr (ks == 1 || ks == 3) {
#pragma unroll
                            for (int dy = 0; dy < 3; ++dy) {
                                const int sl_ = (T4 + 3 - dy) & 3;
                                acc[sl_] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ks == 1 ? wl8[dy * 3 + dx] : wh8[dy * 3 + dx], f8[ks == 1 ? 0 : 1], acc[sl_], 0, 0, 0, scale_a, 0, scale_b);
                            }
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
#define AQ_OP(I) rund(std::integral_constant<int, I>{});
                            AQ_OP(0) AQ_OP(1) AQ_OP(2) AQ_OP(3) AQ_OP(4) AQ_OP(5) AQ_OP(6) AQ_OP(7) AQ_OP(8) AQ_OP(9) AQ_OP(10) AQ_OP(11) AQ_OP(12) AQ_OP(13) AQ_OP(14) AQ_OP(15)
#undef AQ_OP
                            constexpr int ND = ops_chunks(ROLE);
                            constexpr int lo_ = ai < ND ? ai * L.n / ND : L.n, hi_ = ai < ND ? (ai + 1) * L.n / ND : L.n;
                            auto run = [&](auto I_) __attribute__((always_inline)) {
                                constexpr int I = decltype(I_)::value;
                                if constexpr (I >= lo_ && I < hi_) {
                                    constexpr Op o = L.op[I];
                                    if constexpr (o.kind == OP_CVR) op_cvr();
                                    if constexpr (o.kind == OP_CVW) op_cvw();
                                    if constexpr (o.kind == OP_ACT) op_act(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                    if constexpr (o.kind == OP_SPL) op_spl(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                    if constexpr (o.kind == OP_MW) op_mw(std::integral_constant<int, o.a>{});
                                    if constexpr (o.kind == OP_RHI) op_rhi(std::integral_constant<int, o.a>{});
                                    if constexpr (o.kind == OP_ADD) op_add(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                    if constexpr (o.kind == OP_ST) op_st(std::integral_constant<int, o.a>{});
                                }
                            };
#define AQ_OP(I) run(std::integral_constant<int, I>{});
                            AQ_OP(0) AQ_OP(1) AQ_OP(2) AQ_OP(3) AQ_OP(4) AQ_OP(5) AQ_OP(6) AQ_OP(7) AQ_OP(8) AQ_OP(9) AQ_OP(10) AQ_OP(11) AQ_OP(12) AQ_OP(13) AQ_OP(14) AQ_OP(15)
                            AQ_OP(16) AQ_OP(17) AQ_OP(18) AQ_OP(19) AQ_OP(20) AQ_OP(21) AQ_OP(22) AQ_OP(23)
#undef AQ_OP
                        }
#ifndef AQ_NOPIN
#pragma unroll
                        for (int i_ = 0; i_ < 3; ++i_) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (i_ == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x006, AQ_FILL, 0);
                        }
                        if (ks == 1 || ks == 3) {
#pragma unroll
                            for (int i_ = 0; i_ < 3; ++i_) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                if (i_ == 0) __builtin_amdgcn_sched_group_barrier(0x100, (ks == 3 && dx == 2) ? 4 : 2, 0);
                                __builtin_amdgcn_sched_group_barrier(0x006, 2 * AQ_FILL, 0);
                            }
                        }
#endif
                        __builtin_amdgcn_sched_barrier(0);
                    };
#define AQ_CHUNK(A) chunk(std::integral_constant<int, A>{});
                    AQ_CHUNK(0) AQ_CHUNK(1) AQ_CHUNK(2) AQ_CHUNK(3) AQ_CHUNK(4) AQ_CHUNK(5) AQ_CHUNK(6) AQ_CHUNK(7) AQ_CHUNK(8) AQ_CHUNK(9) AQ_CHUNK(10) AQ_CHUNK(11)
#undef AQ_CHUNK
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
    };
    if (role == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
#endif
}

template <bool OUT8>
hipError_t set_limit() { return hipFuncSetAttribute((const void*)arsb_sq_kernel<OUT8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); }

}  // namespace

hipError_t arsb_sq_init()
{
    hipError_t e;
    if ((e = set_limit<true>()) != hipSuccess) return e;
    return set_limit<false>();
}

// One exact ARSB of a conv64_q8 / conv64_sq chain (fp8 low words in; out: fp8, or fp16 behind the last exact block); y must not alias x.
// false: not this kernel's (the caller runs the two convs)
bool launch_arsb_sq(const ArsbSqArgs& q, int max_groups, hipStream_t s)
{
    if (!(q.slope <= 1.f) || q.H % RB != 0 || q.H < RB) return false;
    if ((long long)q.B * q.H * q.W * 128 + (long long)(RB * q.W + 2) * 128 >= (1ll << 32) - 65536) return false;
    if (!q.x_hi || !q.x_lo8 || !q.y_hi || !q.y_lo || q.y_hi == q.x_hi) return false;
    for (int i = 0; i < 2; ++i) if (!q.w16[i] || !q.wh8[i] || !q.wl8[i]) return false;
    const int px = (q.W + TW - 1) / TW;
    const long long items = (long long)q.B * px * (q.H / RB);
    if (items >= (1ll << 31) / 4) return false;
    ArsbSqArgsK a{};
    a.x_hi = q.x_hi; a.x_lo8 = q.x_lo8; a.y_hi = q.y_hi; a.y_lo = q.y_lo;
    for (int i = 0; i < 2; ++i) { a.w16[i] = q.w16[i]; a.wh8[i] = q.wh8[i]; a.wl8[i] = q.wl8[i]; }
    a.slope = q.slope; a.B = q.B; a.H = q.H; a.W = q.W;
    const int G = (int)std::min<long long>(items, (long long)max_groups);
    const dim3 grid(G), blk(256);
    if (q.out8) arsb_sq_kernel<true><<<grid, blk, LDS_BYTES, s>>>(a);
    else arsb_sq_kernel<false><<<grid, blk, LDS_BYTES, s>>>(a);
    return true;
}
