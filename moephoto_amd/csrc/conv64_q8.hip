// conv64_q8.hip -- the split-operand 3x3 64->64 convolution of MOE_PREC_MIXED's exact layers (conv_input2 and the leading ARSBs; python/models.py:76-80,
// 108-123 of the reference) with its two CORRECTION products on fp8 operands:
//
//     out = conv(w_hi, a_hi)  +  2^-11 * (conv(w_lo, a_hi) + conv(w_hi, a_lo))        w = w_hi + w_lo 2^-11,  a = a_hi + a_lo 2^-11  (fp16 parts)
//           ------ fp16 -----            -------------- fp8 e4m3 ----------------
//
// The corrections carry 2^-11 of the result, so four mantissa bits are plenty for them (tests/emu_precision.py corr8: the same end-to-end error as the
// fp16 form on every model and input class, profiles/r03/j_fp8_corrections_emulation.txt).  v_mfma_scale_f32_32x32x64_f8f6f4 runs at twice the fp16 rate and
// its E8M0 scales fold the 2^-11 (and the operands' own power-of-two scalings) in, so ALL THREE products accumulate in one accumulator set and an exact
// layer costs 1 + 1/2 + 1/2 = 2 product-times instead of conv64_x3.hip's 3.  Operand layout of the scaled MFMA and the conversion's semantics:
// tools/micro/mfma_scale_probe.hip, cvt_scale_probe.hip, cvt_ovfl_probe.hip (32 consecutive k per lane; uniform scales in byte 0; the conversion divides by
// its scale operand and gives NaN beyond 464 unless MODE.FP16_OVFL is set -- the kernel runs with it: saturation at +-448).
//
// Drop-in for conv64_x3.hip (same tensors in and out) -- and, between two layers of this kernel, a cheaper interface: the low part of an activation tensor as
// the fp8 word the correction product reads anyway (ConvX3Args::in8 / out8, template parameters IN8 / OUT8; engine.cpp plans the chain: Act::lo8).  These
// layers are held by their bytes (profiles/r03/o_stream_bytes.txt): 384 / 576 instead of 512 / 768 bytes a pixel, and the a_lo image arrives by DMA.
//
//   wave (c, h)   output channels 32c .. 32c+31, output rows 4h .. 4h+3 of the 8 x 32 patch (arsb32.hip's conv_2 half)
//   weights       36 fp16 A fragments of w_hi (144 registers) + 9 + 9 fp8 A fragments (one per tap: 32 rows x 64 k) of w_lo 2^8 and w_hi 2^8 (144): 256 in
//                 AGPRs, the last four fp8 fragments in arch VGPRs
//   LDS           a_hi patch 10 x 34 x 128 B double buffered; a_lo: the fp16 patch single buffered (only live between its landing and its conversion) or
//                 -- IN8 -- two fp8 images (patches p & 1) in the same 45 KB, filled by DMA; ONE fp8 image 10 x 34 x 64 B that the conversions write
//                 (a_lo / 4 for the short pass unless IN8, then a_hi / 4 for the long pass); 1 KB that takes the two DMA pieces an fp8 image does not
//                 have: 3 x 45,056 + 22,528 + 1,024 = 158,720 B
//   patch p       counted wait + barrier | EPI 2: the residual words of all four rows are requested | [a_lo -> fp8 image | barrier] | SHORT pass: rows of
//                 the a_lo image (w_hi8 x a_lo8: 36 fp8 MFMAs per wave), the DMA of a_lo[p+1], EPI 2: the residual added to a row once its products of
//                 this pass are in | barrier | first fp16 fragments, a_hi -> fp8 image | barrier | LONG pass: rows of a_hi (144 fp16 MFMAs) and of its
//                 image (w_lo8 x a_hi8: 36 fp8 MFMAs); row steps 0..2 carry the DMA of a_hi[p+1], row steps 3..5 the epilogues and stores of output rows
//                 0..2 (a row is complete two steps after its first one), row 3 behind the pass.  The first form ran the long pass first and finished the
//                 rows in the short one: 2.3k cycles of MFMA cannot hide 700 instructions of epilogue (cycle trace: 4.3k for that pass, 23k per patch;
//                 profiles/r03/m_conv64_q8.txt, which also has the ablations -- what a patch spends where -- and the forms that were tried and dropped).
//   loads and stores in separate stretches (arsb32c.hip): DMA and residual loads before row step 3 of the long pass, stores from there on; the wait that
//   opens a patch counts (all but the sixteen stores behind the last DMA piece).
#include "common.h"
#include "rowtile.h"
#include <type_traits>
#include <vector>
#include <cstdio>

namespace {

constexpr int TW = 32, TH = 8;
constexpr int XW = 34, XH = 10;                // input patch (halo 1)
constexpr int NPIX = XW * XH;                  // 340
constexpr int NDMA_W = 11;                     // 1-KiB pieces per wave: 44 >= 340 / 8
constexpr int XBYTES = NDMA_W * 4 * 1024;      // 45,056
constexpr int ROWB = XW * 128, ROWB8 = XW * 64;
constexpr int Q8BYTES = 22 * 1024;             // fp8 image: 340 x 64 = 21,760 B
constexpr int NDMA8_W = 6;                     // IN8: 1-KiB pieces of the fp8 a_lo image per wave (22 of the 24 exist; the other two land in DUMP)
constexpr int DUMP = 3 * XBYTES + Q8BYTES;     // 1 KiB nobody reads
constexpr int LDS_BYTES = DUMP + 1024;         // 158,720

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef int i8v_t __attribute__((ext_vector_type(8)));
typedef _Float16 h2v_t __attribute__((ext_vector_type(2)));
typedef short s2v_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;
typedef const __attribute__((address_space(3))) u4_t* lds_u4_t;

enum OpKind : int { OP_NONE = 0, OP_DMA, OP_RLD, OP_RADD, OP_ACT, OP_SPL, OP_ST };
struct Op { int kind, a, b, c; };
struct OpList {
    int n = 0;
    Op op[40] = {};
    constexpr void push(int kind, int a = 0, int b = 0, int c = 0) { op[n] = Op{kind, a, b, c}; ++n; }
    constexpr void append(const OpList& o) { for (int i = 0; i < o.n; ++i) { op[n] = o.op[i]; ++n; } }
};
// OP_DMA(image, piece, half): image 0 = a_lo of this patch, 1 = a_hi of the next one; half 0 = address, 1 = validity + issue
constexpr OpList dma_ops(int img, int i0, int i1)
{
    OpList r;
    for (int i = i0; i < i1; ++i) { r.push(OP_DMA, img, i, 0); r.push(OP_DMA, img, i, 1); }
    return r;
}
// epilogue of output row i: [PReLU in fp32,] hi / lo split by channel pairs, stores, per 16-byte slot
constexpr OpList yrow_ops(int i, bool act)
{
    OpList r;
    for (int o = 0; o < 2; ++o) {
        if (act) { r.push(OP_ACT, i, o, 0); r.push(OP_ACT, i, o, 4); }
        r.push(OP_SPL, i, o, 0); r.push(OP_SPL, i, o, 2); r.push(OP_ST, i, o);
    }
    return r;
}
// SHORT pass, row step s (a_lo row 4h + s: 3, 6, 9, 9, 6, 3 fp8 MFMAs).  EPI 2: the residual words of all four output rows are requested at the head of the
// patch, ahead of the first conversion (64 registers nobody else wants until the long pass: a load takes 3-4k cycles at this kernel's 4 TB/s, requested
// inside this pass they were waited for) and added to a row's accumulator once its own products of this pass are in (row i: steps i .. i+2)
// -- and the pass carries the DMA of a_lo[p+1] (its buffer is free since the conversion that opened the patch)
constexpr OpList short_ops(int s, bool res, bool in8)
{
    OpList r;
    if (!in8) {
        if (s == 0) r = dma_ops(0, 0, 1);
        if (s == 1) r = dma_ops(0, 1, 3);
        if (s == 2) r = dma_ops(0, 3, 6);
        if (s == 3) r = dma_ops(0, 6, 9);
        if (s == 4) r = dma_ops(0, 9, 11);
    } else {                                   // the fp8 low part: six 1-KiB pieces per wave (16 pixels each)
        if (s == 2) r = dma_ops(0, 0, 2);      // (not in steps 0, 1: the pass opens ~300 cycles behind the last store of the patch before, and a request
        if (s == 3) r = dma_ops(0, 2, 4);      // issued behind stores in flight held the wave for ~1.4k cycles)
        if (s == 4) r = dma_ops(0, 4, 5);
        if (s == 5) r = dma_ops(0, 5, 6);
    }
    if (!res) return r;
    if (s == 3) { r.push(OP_RADD, 0, 0); r.push(OP_RADD, 0, 1); }
    if (s == 4) { r.push(OP_RADD, 1, 0); r.push(OP_RADD, 1, 1); }
    if (s == 5) { r.push(OP_RADD, 2, 0); r.push(OP_RADD, 2, 1); }
    return r;
}
// LONG pass, row step s (a_hi row 4h + s: 12, 24, 36, 36, 24, 12 fp16 + 3, 6, 9, 9, 6, 3 fp8 MFMAs)
constexpr OpList long_ops(int s, bool act)
{
    OpList r;
    if (s == 0) r = dma_ops(1, 0, 2);
    if (s == 1) r = dma_ops(1, 2, 6);
    if (s == 2) r = dma_ops(1, 6, 11);
    if (s == 3) r = yrow_ops(0, act);
    if (s == 4) r = yrow_ops(1, act);
    if (s == 5) r = yrow_ops(2, act);
    return r;
}
// stores issued behind the last DMA piece of a patch (four rows x two slots x hi / lo): what the wait that opens the next patch leaves in flight
constexpr int stores_per_patch = 16;

#ifndef Q8_DBG
#define Q8_DBG 0      // debug builds: 1 = without the w_lo8 x a_hi8 product, 2 = without w_hi8 x a_lo8
#endif

struct Item { int b, pyi, pxi; };

// experiment (tools/mk_variant.sh q8nolo conv64_q8.hip -DQ8_NOLO): the low-part tensors' descriptors cover nothing -- loads and DMA return zero, stores are dropped,
// the instruction stream is the same: what the launches would take without those bytes through HBM (profiles/r03/o_stream_bytes.txt)
#ifdef Q8_NOLO
#define Q8_LO_BYTES(n) 0u
#else
#define Q8_LO_BYTES(n) (n)
#endif

// cycle-level trace (tools/mk_variant.sh traceq8 conv64_q8.hip -DQ8_TRACE; tools/show_trace_q8.py): s_memtime stamps of patches 4 and 7 of the first eight
// workgroups, buffered in LDS and written to a.pool (repurposed: 8 x 2 x 4 x 32 x 8 bytes) when the workgroup is done
#if defined(Q8_TRACE) && !defined(Q8_STEPWAIT)
#define Q8_STEPWAIT      // the stamps are inline-asm LDS stores inside the passes: the compiler's counted lgkmcnt waits would be one short
#endif
#ifdef Q8_TRACE
constexpr int TRACE_LDS = 2 * 4 * 32 * 8 + 4 * 32 * 8;
#define Q8_STAMP(SLOT) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); const unsigned tb_ = tbase; asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(tb_), "v"(t_), "n"((SLOT) * 8) : "memory"); }      // (unconditional, to a dummy slot when off: a branch per stamp splits the pinned regions and spills; an LDS store: a flat one would drain vmcnt)
#else
constexpr int TRACE_LDS = 0;
#define Q8_STAMP(SLOT)
#endif

// EPI 0 plain | 1 PReLU (fp32, slope <= 1) | 2 + residual (hi + lo 2^-11)
// IN8 / OUT8: the low part of the input (and of the residual) / of the output is stored as fp8 e4m3 -- the word the correction product reads, (v - fp16(v)) 2^11 / 4
// (ConvX3Args::in8 / out8): 64 bytes a pixel instead of 128, and the a_lo image comes by DMA instead of through a conversion
template <int EPI, bool IN8, bool OUT8>
__global__ __launch_bounds__(256) void conv64_q8_kernel(ConvX3Args a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr unsigned kOOR = 0xFFFF0000u;
    constexpr bool RES = EPI == 2, ACT = EPI == 1;
    // MODE.FP16_OVFL for the whole kernel: with it the fp8 conversions saturate at +-448 (without: NaN beyond 464, i.e. for activations beyond 1856 --
    // tools/micro/cvt_ovfl_probe.hip).  The epilogue's fp16 conversions then clamp at 65504 where IEEE gives infinity: beyond the range either way.
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\ts_nop 3" ::: "memory");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const unsigned lbase = lds0 + 2u * XBYTES, qbase = lds0 + 3u * XBYTES;      // lbase: a_lo as fp16 (IN8: two fp8 a_lo images, patches p & 1), qbase: the fp8 image the conversions write
    const int tid = threadIdx.x, lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = w4 & 1, h = w4 >> 1;
    const int j = lane & 31, hh = lane >> 5;

    const int g = blockIdx.x, G = gridDim.x;
    const int nitems = a.B * a.py * a.px;
    const int K = (nitems - g + G - 1) / G;               // this workgroup's patches: g, g+G, ...
    if (K <= 0) return;
    auto decode = [&](int item) {
        Item it;
        it.pxi = item % a.px;
        const int t = item / a.px;
        it.pyi = t % a.py;
        it.b = t / a.py;
        return it;
    };
    const int Gx = G % a.px, Gy = (G / a.px) % a.py, Gb = G / (a.px * a.py);      // work items g, g+G, ... are walked with carries instead of divisions
    auto advance = [&](const Item& it) {
        Item n;
        int x = it.pxi + Gx;
        const int cx = x >= a.px;
        x -= cx ? a.px : 0;
        int y = it.pyi + Gy + cx;
        const int cy = y >= a.py;
        y -= cy ? a.py : 0;
        n.pxi = x; n.pyi = y; n.b = it.b + Gb + cy;
        return n;
    };

    // ---- weights (arsb32.hip: MFMA row i = 8q + 4h' + e is given channel 16 (q >> 1) + 8 h' + 4 (q & 1) + e, so that a lane's accumulator registers
    // 8g .. 8g+7 are eight consecutive channels = one 16-byte slot; the fp8 fragments use the same row order: the scaled MFMA's C/D map is the f16 one) ----
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
    // E8M0 scales of the fp8 products (byte 0; kept opaque: the compiler would read a constant as an f32 literal): weights carry 2^8, activations 2^-2,
    // the correction term 2^-11  ->  2^(-8 - 11) on the A side, 2^2 on the B side
    int scale_a = 127 - 19, scale_b = 127 + 2;
    asm volatile("" : "+v"(scale_a), "+v"(scale_b));

    // ---- patch DMA (arsb32.hip): the lane's source offset of piece i is formed when the piece is issued ---------------------------------------------------
    const unsigned in_pad = (unsigned)(a.W + 1) * 128u;
    const unsigned nbytes = (unsigned)a.B * a.H * a.W * 128u;
    const __amdgpu_buffer_rsrc_t rhi = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in_hi - in_pad), 0, nbytes + in_pad, 0x00020000);
    const unsigned in_pad_lo = IN8 ? in_pad / 2 : in_pad, nbytes_in_lo = IN8 ? nbytes / 2 : nbytes, nbytes_out_lo = OUT8 ? nbytes / 2 : nbytes;
    const __amdgpu_buffer_rsrc_t rlo = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in_lo - in_pad_lo), 0, Q8_LO_BYTES(nbytes_in_lo + in_pad_lo), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrh = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? a.res_hi : a.in_hi), 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrl = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? a.res_lo : a.in_hi), 0, Q8_LO_BYTES(nbytes_in_lo), 0x00020000);      // (the residual's format is the input's)
#ifdef Q8_NOST        // (experiment: what the stores cost)
    const __amdgpu_buffer_rsrc_t ryh = __builtin_amdgcn_make_buffer_rsrc((void*)a.out_hi, 0, 0u, 0x00020000);
#else
    const __amdgpu_buffer_rsrc_t ryh = __builtin_amdgcn_make_buffer_rsrc((void*)a.out_hi, 0, nbytes, 0x00020000);
#endif
    const __amdgpu_buffer_rsrc_t ryl = __builtin_amdgcn_make_buffer_rsrc((void*)a.out_lo, 0, Q8_LO_BYTES(nbytes_out_lo), 0x00020000);
    const int qlane = w4 * 8 + (lane >> 3);
    unsigned d_off = 0, d_r = 0, d_cc = 0;
    auto piece_addr = [&](int i) {
        unsigned q = (unsigned)(i * 32 + qlane);
        asm volatile("" : "+v"(q));
        d_r = __umul24(q, 241u) >> 13;                        // q / 34 for q < 442
        d_cc = (unsigned)(__mul24((int)d_r, -XW) + (int)q);
        const unsigned sl = (unsigned)(lane & 7) ^ ((d_cc >> 1) & 7u);      // logical 16-B slot behind this physical slot
        d_off = ((__umul24(d_r, (unsigned)a.W) + d_cc) << 7) | (sl << 4);
    };
    auto piece_off = [&](int i, int ya, int xa, bool live) {
        bool ok = ((unsigned)(ya + (int)d_r) < (unsigned)a.H) & ((unsigned)(xa + (int)d_cc) < (unsigned)a.W) & live;
        if (i * 32 + 31 >= NPIX) ok &= (i * 32 + qlane < NPIX);
        return ok ? d_off : kOOR;
    };
    auto origin = [&](const Item& it) { return (unsigned)((it.b * a.H + it.pyi * TH - 1) * a.W + it.pxi * TW - 1 + a.W + 1) * 128u; };
    // IN8: the a_lo image is fetched as it is read -- 64 bytes a pixel, 16 pixels per 1-KiB piece, piece i of wave w is piece 4 i + w of the image (22 pieces);
    // lane l writes physical slot l & 3 of pixel 16 (4 i + w) + (l >> 2) and fetches the logical slot behind it (fp8 image swizzle: s ^ ((col >> 2) & 3))
    const int qlane8 = w4 * 16 + (lane >> 2);
    auto piece8_addr = [&](int i) {
        unsigned q = (unsigned)(i * 64 + qlane8);
        asm volatile("" : "+v"(q));
        d_r = __umul24(q, 241u) >> 13;
        d_cc = (unsigned)(__mul24((int)d_r, -XW) + (int)q);
        const unsigned sl = (unsigned)(lane & 3) ^ ((d_cc >> 2) & 3u);
        d_off = ((__umul24(d_r, (unsigned)a.W) + d_cc) << 6) | (sl << 4);
    };
    auto piece8_off = [&](int i, int ya, int xa, bool live) {
        bool ok = ((unsigned)(ya + (int)d_r) < (unsigned)a.H) & ((unsigned)(xa + (int)d_cc) < (unsigned)a.W) & live;
        if (i * 64 + 63 >= NPIX) ok &= (i * 64 + qlane8 < NPIX);
        return ok ? d_off : kOOR;
    };

    // ---- LDS addressing.  fp16 image: pixel (row, col) at (row * 34 + col) * 128, 16-B slot s (8 channels) at s ^ ((col >> 1) & 7); B fragment (dx, ks):
    // lane (j, hh) reads slot 2 ks + hh of column j + dx.  fp8 image: pixel at (row * 34 + col) * 64, 16-B slot s (16 channels) at s ^ ((col >> 2) & 3); B
    // fragment dx of the scaled MFMA: lane (j, hh) reads slots 2 hh, 2 hh + 1 (k block hh: channels 32 hh .. 32 hh + 31) of column j + dx. ---------------
    unsigned fa3[3], fq[3];      // fp16 fragment (dx, ks): fa3[dx] ^ (ks << 5) -- the slot index (2 ks + hh) ^ z = (hh ^ z) ^ 2 ks sits in bits 4..6, everything else is a multiple of 128
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int cc = j + dx, z = (cc >> 1) & 7;
        fa3[dx] = (unsigned)(4 * h * ROWB) + (unsigned)(cc * 128 + ((hh ^ z) << 4));      // + the a_hi buffer of the patch
    }
    // (formed where it is used, behind an opaque copy: hoisted out of the patch loop the twelve addresses are twelve registers again)
    auto fa_of = [&](int f) { unsigned b = fa3[f >> 2]; asm volatile("" : "+v"(b)); return b ^ (unsigned)((f & 3) << 5); };
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int cc = j + dx, z = (cc >> 2) & 3;
        fq[dx] = qbase + (unsigned)(4 * h * ROWB8) + (unsigned)(cc * 64 + (((2 * hh) ^ z) << 4));      // slot 2 hh; slot 2 hh + 1 is this address ^ 16
    }
    const unsigned lane_ob = (unsigned)(j * 128 + (32 * c + 8 * hh) * 2);      // byte offset of slot o = 0 of output column j inside a patch row of the tensors
    const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // fp16 patch -> fp8 image (all 256 threads): unit u = (pixel q, 16-channel slot s8): two 16-byte reads, eight conversions (value / 4), one 16-byte
    // write; six units per thread, all twelve reads first.  Units behind the last pixel repeat unit (339, s8): same data to the same place, no branch.
    const float quarter = 4.0f;                               // the source is DIVIDED by the scale operand
    // four packed fp16 pairs -> two words of four fp8 each.  One block, the two words' conversions alternating, a wait state at its end: a conversion writes
    // HALF a register, and gfx950 wants one wait state between such a write and the next VALU use of the register (the half-word is not forwarded).  The
    // compiler's hazard recognizer places that state for its own instructions and cannot see into inline asm: issued back to back, the second conversion
    // of a word at times kept a stale first half -- results that changed from run to run (tools/diag_q8_batch.py).
    // (inline asm: through __builtin_amdgcn_cvt_scalef32_pk_fp8_f16 this compiler converted the first word of a slot four times and read nothing else of
    // it -- found with the constant-image experiment of tools/diag_q8.py)
    auto cvt4 = [&](unsigned a0, unsigned a1, unsigned b0, unsigned b1, unsigned& p0, unsigned& p1) __attribute__((always_inline)) {
        asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %2, %6\n\t"
                     "v_cvt_scalef32_pk_fp8_f16 %1, %4, %6\n\t"
                     "v_cvt_scalef32_pk_fp8_f16 %0, %3, %6 op_sel:[0,0,1]\n\t"
                     "v_cvt_scalef32_pk_fp8_f16 %1, %5, %6 op_sel:[0,0,1]\n\t"
                     "s_nop 0"
                     : "=&v"(p0), "=&v"(p1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(quarter));
    };
    auto to_fp8_image = [&](unsigned srcbase) {
        auto cvt16 = [&](const u4_t& w0, const u4_t& w1) {
            u4_t d = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                unsigned p0, p1;
                const unsigned a0 = w0[2 * k], a1 = w0[2 * k + 1], b0 = w1[2 * k], b1 = w1[2 * k + 1];
                cvt4(a0, a1, b0, b1, p0, p1);
                d[k] = p0; d[2 + k] = p1;
            }
#ifdef Q8_CONST_IMAGE
            d = u4_t{0x38383838u, 0x38383838u, 0x38383838u, 0x38383838u};      // debug: every image value 1.0
#endif
            return d;
        };
        // units 0..3: pixel tid (0..255), all four 16-channel slots -- consecutive lanes are consecutive pixels with the same logical slot, the access pattern
        // both images are swizzled for; the pixel's row / column arithmetic is done once
        {
            unsigned q = (unsigned)tid;
            asm volatile("" : "+v"(q));                       // (recomputed per use: hoisted out of the patch loop these addresses spill)
            const unsigned r = __umul24(q, 241u) >> 13, cc = q - r * XW;
            const unsigned z16 = (cc >> 1) & 7u, z8 = (cc >> 2) & 3u;
            const unsigned src = srcbase + q * 128u, dst = qbase + q * 64u;
            u4_t w[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) w[t] = *(lds_u4_t)(src + (((unsigned)t ^ z16) << 4));
#pragma unroll
            for (int s8 = 0; s8 < 4; ++s8) {
                const u4_t d = cvt16(w[2 * s8], w[2 * s8 + 1]);
                const unsigned adr = dst + (((unsigned)s8 ^ z8) << 4);
                asm volatile("ds_write_b128 %0, %1" ::"v"(adr), "v"(d) : "memory");
            }
        }
        // units 4, 5: the remaining 84 pixels x 4 slots, 84 consecutive lanes per slot (lanes behind the last unit repeat unit (339, 3): same data to the same place)
        {
            u4_t w0[2], w1[2];
            unsigned ad[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                unsigned t2 = (unsigned)(tid + i * 256);
                asm volatile("" : "+v"(t2));
                const unsigned s8 = min(__umul24(t2, 781u) >> 16, 3u);      // t2 / 84 for t2 < 512 (781 / 65536 = 1 / 83.9)
                const unsigned q = min(256u + t2 - s8 * 84u, (unsigned)(NPIX - 1));
                const unsigned r = __umul24(q, 241u) >> 13, cc = q - r * XW;
                const unsigned z16 = (cc >> 1) & 7u, z8 = (cc >> 2) & 3u;
                w0[i] = *(lds_u4_t)(srcbase + q * 128u + (((2 * s8) ^ z16) << 4));
                w1[i] = *(lds_u4_t)(srcbase + q * 128u + (((2 * s8 + 1) ^ z16) << 4));
                ad[i] = qbase + q * 64u + ((s8 ^ z8) << 4);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const u4_t d = cvt16(w0[i], w1[i]);
                const unsigned adr = ad[i];
                asm volatile("ds_write_b128 %0, %1" ::"v"(adr), "v"(d) : "memory");
            }
        }
    };

    float16_t acc[4];
    half8_t fr[13];           // fp16 fragment f of pass-1 row step t lives in fr[(f - t) mod 13]
    i8v_t fr8[4];             // fp8 fragment dx of row step t lives in fr8[(dx - t) mod 4]
    u4_t rl[4][4];            // EPI 2: residual words (slot 0 hi, slot 1 hi, slot 0 lo, slot 1 lo) of the four rows (short pass only)
    unsigned sh[4], sl[4];

    // ===== prologue: a_hi and a_lo of the first patch =========================================================================================================
    Item it_cur = decode(g);
    {
        const unsigned org = origin(it_cur);
#pragma unroll
        for (int i = 0; i < NDMA_W; ++i) {
            piece_addr(i);
            const unsigned off = piece_off(i, it_cur.pyi * TH - 1, it_cur.pxi * TW - 1, true);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rhi, (__attribute__((address_space(3))) void*)(smem + (i * 4 + w4) * 1024), 16, off, org, 0, 0);
            if (!IN8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rlo, (__attribute__((address_space(3))) void*)(smem + 2 * XBYTES + (i * 4 + w4) * 1024), 16, off, org, 0, 0);
        }
        if (IN8) {
#pragma unroll
            for (int i = 0; i < NDMA8_W; ++i) {
                piece8_addr(i);
                const unsigned off = piece8_off(i, it_cur.pyi * TH - 1, it_cur.pxi * TW - 1, true);
                const int dst = (i * 4 + w4 < 22) ? 2 * XBYTES + (i * 4 + w4) * 1024 : DUMP;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rlo, (__attribute__((address_space(3))) void*)(smem + dst), 16, off, org >> 1, 0, 0);
            }
        }
    }

    for (int p = 0; p < K; ++p) {
        const Item it = it_cur;
        const bool has_next = p + 1 < K;
        const Item itn = advance(it);
        const unsigned xcur = lds0 + (unsigned)((p & 1) * XBYTES);
        const unsigned orgn = (unsigned)__builtin_amdgcn_readfirstlane((int)origin(itn));
        const int yan = __builtin_amdgcn_readfirstlane(itn.pyi * TH - 1), xan = __builtin_amdgcn_readfirstlane(itn.pxi * TW - 1);
        const unsigned dlo = (unsigned)__builtin_amdgcn_readfirstlane(2 * XBYTES + (IN8 ? ((p + 1) & 1) * Q8BYTES : 0) + w4 * 1024), dnx = (unsigned)__builtin_amdgcn_readfirstlane(((p + 1) & 1) * XBYTES + w4 * 1024);
        const unsigned dlo5 = (unsigned)__builtin_amdgcn_readfirstlane(w4 < 2 ? (int)dlo + 5 * 4096 : DUMP);      // IN8: pieces 22, 23 of the image do not exist
        const unsigned qshort = IN8 ? (unsigned)(p & 1) * Q8BYTES - (unsigned)XBYTES : 0u;      // short pass: where its fp8 image lies relative to qbase
        const int y0 = it.pyi * TH, x0 = it.pxi * TW;
#ifdef Q8_TRACE
        const bool trace_on = a.pool && g < 8 && (p == 4 || p == 7);
        const unsigned tbase = lds0 + (unsigned)(LDS_BYTES + (trace_on ? ((p == 7) * 4 + w4) : (8 + w4)) * 32 * 8);
#endif
        // stream tensors: byte offset of (output row 4h, column x0) of the patch and the lane's column part
        const unsigned so0 = (unsigned)(((it.b * a.H + y0 + 4 * h) * a.W + x0) * 128);
        const unsigned vo = (x0 + j < a.W) ? lane_ob : kOOR;
        auto row_so = [&](int i) { return (y0 + 4 * h + i < a.H) ? so0 + (unsigned)(i * a.W * 128) : kOOR; };
        const unsigned vo8 = (x0 + j < a.W) ? lane_ob >> 1 : kOOR;      // the same for a tensor of 64 bytes a pixel (fp8 low parts)
        auto row_so8 = [&](int i) { return (y0 + 4 * h + i < a.H) ? (so0 >> 1) + (unsigned)(i * a.W * 64) : kOOR; };

        auto op_rld = [&](auto I_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value;
            const unsigned so = row_so(i);
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                rl[i][o] = __builtin_amdgcn_raw_buffer_load_b128(rrh, vo + (unsigned)(o * 32), so, 0);
                if (!IN8) rl[i][2 + o] = __builtin_amdgcn_raw_buffer_load_b128(rrl, vo + (unsigned)(o * 32), so, 0);
                else {
                    const u2_t t = __builtin_amdgcn_raw_buffer_load_b64(rrl, vo8 + (unsigned)(o * 16), row_so8(i), 0);
                    rl[i][2 + o][0] = t[0]; rl[i][2 + o][1] = t[1];
                }
            }
        };
        Q8_STAMP(0)
        // a_hi[p] and a_lo[p] have landed: everything but the stores issued behind the last DMA piece of the patch before (first patch: the prologue's pieces)
        if (p == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        static_assert(stores_per_patch == 16, "counted wait at the head of a patch");
        __builtin_amdgcn_s_barrier();                         // ... for every wave, and everybody has left the long pass of the patch before (the fp8 image is free)
        asm volatile("" ::: "memory");
        Q8_STAMP(1)
        if (RES) { op_rld(std::integral_constant<int, 0>{}); op_rld(std::integral_constant<int, 1>{}); op_rld(std::integral_constant<int, 2>{}); op_rld(std::integral_constant<int, 3>{}); }
        if (!IN8) {
            to_fp8_image(lbase);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            Q8_STAMP(2)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        Q8_STAMP(3)

        // ---- micro-ops ------------------------------------------------------------------------------------------------------------------------------------
        auto op_dma = [&](auto IMG_, auto I_, auto HALF_) __attribute__((always_inline)) {      // image 0: a_lo of the next patch, 1: its a_hi
            constexpr int img = decltype(IMG_)::value, i = decltype(I_)::value, half = decltype(HALF_)::value;
            if constexpr (img == 0 && IN8) {
                if constexpr (half == 0) piece8_addr(i);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rlo, (__attribute__((address_space(3))) void*)((char*)smem + (i == 5 ? dlo5 : dlo + i * 4096)), 16, piece8_off(i, yan, xan, has_next), orgn >> 1, 0, 0);
            } else if constexpr (half == 0) piece_addr(i);
            else if constexpr (img == 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rlo, (__attribute__((address_space(3))) void*)((char*)smem + dlo + i * 4096), 16, piece_off(i, yan, xan, has_next), orgn, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rhi, (__attribute__((address_space(3))) void*)((char*)smem + dnx + i * 4096), 16, piece_off(i, yan, xan, has_next), orgn, 0, 0);
        };
        auto op_radd = [&](auto I_, auto O_) __attribute__((always_inline)) {      // acc += res_hi + res_lo 2^-11 for the eight channels of slot o
            constexpr int i = decltype(I_)::value, o = decltype(O_)::value;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v0 = acc[i][8 * o + 2 * k], v1 = acc[i][8 * o + 2 * k + 1];
                v0 = mix_lo(rl[i][o][k], 1.0f, v0); v1 = mix_hi(rl[i][o][k], 1.0f, v1);
                if (!IN8) { v0 = mix_lo(rl[i][2 + o][k], 0.00048828125f, v0); v1 = mix_hi(rl[i][2 + o][k], 0.00048828125f, v1); }
                else {      // fp8 low part: the stored word is lo / 4, lo in units of 2^-11
                    const f2_t f = (k & 1) ? __builtin_amdgcn_cvt_pk_f32_fp8((int)rl[i][2 + o][k >> 1], true) : __builtin_amdgcn_cvt_pk_f32_fp8((int)rl[i][2 + o][k >> 1], false);
                    v0 = __builtin_fmaf(f[0], 0.001953125f, v0); v1 = __builtin_fmaf(f[1], 0.001953125f, v1);
                }
                acc[i][8 * o + 2 * k] = v0; acc[i][8 * o + 2 * k + 1] = v1;
            }
        };
        auto op_act = [&](auto I_, auto O_, auto E0_) __attribute__((always_inline)) {      // PReLU in fp32 (slope <= 1) on four values
            constexpr int i = decltype(I_)::value, o = decltype(O_)::value, e0 = decltype(E0_)::value;
#pragma unroll
            for (int e = 8 * o + e0; e < 8 * o + e0 + 4; ++e) acc[i][e] = __builtin_fmaxf(acc[i][e], acc[i][e] * a.slope);
        };
        auto op_spl = [&](auto I_, auto O_, auto K0_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value, o = decltype(O_)::value, k0 = decltype(K0_)::value;
#pragma unroll
            for (int k = k0; k < k0 + 2; ++k) split2(acc[i][8 * o + 2 * k], acc[i][8 * o + 2 * k + 1], -2048.f, sh[k], sl[k]);
        };
        auto op_st = [&](auto I_, auto O_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value, o = decltype(O_)::value;
            const unsigned so = row_so(i), vv = vo + (unsigned)(o * 32);
            const u4_t dh = {sh[0], sh[1], sh[2], sh[3]}, dl = {sl[0], sl[1], sl[2], sl[3]};
            __builtin_amdgcn_raw_buffer_store_b128(dh, ryh, vv, so, 0);
            if (!OUT8) __builtin_amdgcn_raw_buffer_store_b128(dl, ryl, vv, so, 0);
            else {
                unsigned p0, p1;
                cvt4(sl[0], sl[1], sl[2], sl[3], p0, p1);
                __builtin_amdgcn_raw_buffer_store_b64(u2_t{p0, p1}, ryl, vo8 + (unsigned)(o * 16), row_so8(i), 0);
            }
        };
        // the ops [f n / NCH, (f + 1) n / NCH) of a list: PH 1 the short pass (3 chunks), 2 the long pass (12 chunks), 3 row 3's epilogue behind it (all ops)
        auto run_ops = [&](auto PH_, auto S_, auto F_) __attribute__((always_inline)) {
            constexpr int PH = decltype(PH_)::value, S = decltype(S_)::value, F = decltype(F_)::value;
            constexpr OpList L = PH == 1 ? short_ops(S, RES, IN8) : PH == 2 ? long_ops(S, ACT) : yrow_ops(3, ACT);
            constexpr int NCH = PH == 1 ? 3 : 12;
            constexpr int lo = PH < 3 ? F * L.n / NCH : 0, hi = PH < 3 ? (F + 1) * L.n / NCH : L.n;
            auto run = [&](auto I_) __attribute__((always_inline)) {
                constexpr int I = decltype(I_)::value;
                if constexpr (I >= lo && I < hi) {
                    constexpr Op o = L.op[I];
                    if constexpr (o.kind == OP_DMA) op_dma(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{}, std::integral_constant<int, o.c>{});
                    if constexpr (o.kind == OP_RLD) op_rld(std::integral_constant<int, o.a>{});
                    if constexpr (o.kind == OP_RADD) op_radd(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                    if constexpr (o.kind == OP_ACT) op_act(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{}, std::integral_constant<int, o.c>{});
                    if constexpr (o.kind == OP_SPL) op_spl(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{}, std::integral_constant<int, o.c>{});
                    if constexpr (o.kind == OP_ST) op_st(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                }
            };
#define Q8_OP(I) run(std::integral_constant<int, I>{});
            Q8_OP(0) Q8_OP(1) Q8_OP(2) Q8_OP(3) Q8_OP(4) Q8_OP(5) Q8_OP(6) Q8_OP(7) Q8_OP(8) Q8_OP(9) Q8_OP(10) Q8_OP(11) Q8_OP(12) Q8_OP(13) Q8_OP(14) Q8_OP(15)
            Q8_OP(16) Q8_OP(17) Q8_OP(18) Q8_OP(19) Q8_OP(20) Q8_OP(21) Q8_OP(22) Q8_OP(23) Q8_OP(24) Q8_OP(25) Q8_OP(26) Q8_OP(27) Q8_OP(28) Q8_OP(29) Q8_OP(30) Q8_OP(31)
            Q8_OP(32) Q8_OP(33) Q8_OP(34) Q8_OP(35) Q8_OP(36) Q8_OP(37) Q8_OP(38) Q8_OP(39)
#undef Q8_OP
        };
        auto read_q8 = [&](int dx, int row, unsigned rel = 0u) {      // the fp8 B fragment dx of image row 4h + row (rel: the image's offset from qbase, a multiple of 32)
            unsigned ad = fq[dx];
            asm volatile("" : "+v"(ad));                      // (see fa_of)
            ad += (unsigned)(row * ROWB8) + rel;              // (image rows are 2,176 bytes: bit 4 of a row offset is clear)
            const u4_t lo4 = *(lds_u4_t)(ad), hi4 = *(lds_u4_t)(ad ^ 16u);
            return i8v_t{(int)lo4[0], (int)lo4[1], (int)lo4[2], (int)lo4[3], (int)hi4[0], (int)hi4[1], (int)hi4[2], (int)hi4[3]};
        };

        // ================= SHORT pass: a_lo rows 4h .. 4h+5: acc = w_hi8 a_lo8 (fp8) ===================================================================================
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) fr8[dx] = read_q8(dx, 0, qshort);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = zero16;      // (explicit: the scaled MFMA takes no literal as its C operand, a zero tuple of sixteen registers would be kept)
        auto step_s = [&](auto S_) __attribute__((always_inline)) {
            constexpr int s = decltype(S_)::value;
            constexpr int nm = (s <= 3 ? 1 : 0) + ((s >= 1 && s <= 4) ? 1 : 0) + ((s >= 2) ? 1 : 0);      // output rows this input row contributes to
            auto chunk = [&](auto DX_) __attribute__((always_inline)) {
                constexpr int dx = decltype(DX_)::value;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int i = s - dy;
                    if (i >= 0 && i < 4 && !(Q8_DBG & 2))
                        acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wh8[dy * 3 + dx], fr8[(dx + 4 - (s & 3)) & 3], acc[i], 0, 0, 0, scale_a, 0, scale_b);
                }
                if (s < 5) fr8[(dx + 3 - (s & 3)) & 3] = read_q8(dx, s + 1, qshort);
                run_ops(std::integral_constant<int, 1>{}, S_, DX_);
#pragma unroll
                for (int i_ = 0; i_ < 3; ++i_) {
                    if (i_ < nm) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i_ == 0 && s < 5) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    if (i_ < nm) __builtin_amdgcn_sched_group_barrier(0x006, 12, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            chunk(std::integral_constant<int, 0>{}); chunk(std::integral_constant<int, 1>{}); chunk(std::integral_constant<int, 2>{});
#ifdef Q8_STEPWAIT      // (conv3x3_sp.hip's one lgkmcnt(0) per row step instead of the compiler's counted waits: here the counted ones win, a2 15.21 -> 15.12 ms;
                        // no inline-asm LDS instruction inside the passes, so the compiler's count is exact)
            __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
            Q8_STAMP(4 + s)
        };
#define Q8_STEP(S) step_s(std::integral_constant<int, S>{});
        Q8_STEP(0) Q8_STEP(1) Q8_STEP(2) Q8_STEP(3) Q8_STEP(4) Q8_STEP(5)
#undef Q8_STEP
        if (RES) { op_radd(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{}); op_radd(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}); }
        Q8_STAMP(10)
        __builtin_amdgcn_s_barrier();                                    // nobody reads the fp8 image (a_lo) any more
        asm volatile("" ::: "memory");
        Q8_STAMP(11)
#pragma unroll
        for (int f = 0; f < 12; ++f) fr[f] = *(lds_h8_t)(xcur + fa_of(f));      // the long pass's first fp16 fragments: a_hi has been there since the head of the patch
#ifndef Q8_NOCVT      // (experiment: what the a_hi conversion costs)
        to_fp8_image(xcur);
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        Q8_STAMP(12)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        Q8_STAMP(13)

        // ================= LONG pass: a_hi rows 4h .. 4h+5: acc += w_hi a_hi (fp16) + w_lo8 a_hi8 (fp8); DMA of the next patch, then the rows' epilogues ============
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) fr8[dx] = read_q8(dx, 0);
        Q8_STAMP(14)
        auto step_l = [&](auto S_) __attribute__((always_inline)) {
            constexpr int s = decltype(S_)::value;
            constexpr int nm = (s <= 3 ? 1 : 0) + ((s >= 1 && s <= 4) ? 1 : 0) + ((s >= 2) ? 1 : 0);
            auto chunk = [&](auto F_) __attribute__((always_inline)) {
                constexpr int f = decltype(F_)::value;
                constexpr int dx = f >> 2, ks = f & 3;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int i = s - dy;
                    if (i >= 0 && i < 4)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w16[(dy * 3 + dx) * 4 + ks], fr[(f + 13 - s) % 13], acc[i], 0, 0, 0);
                }
                // the tap column's fp8 products (one MFMA covers all 64 input channels of a tap) behind its last fp16 k-slice.  Steps with ONE output row (0, 5)
                // issue their three behind the step's last fp16 MFMA instead: every product of such a step goes to the same accumulator, and between an
                // fp16 and an fp8 MFMA on one accumulator the result is not forwarded (8 + 2 resp. 16 + 2 passes of wait) -- one change of kind, not six
                constexpr bool late8 = nm == 1;
                if (!(Q8_DBG & 1) && (late8 ? f == 11 : ks == 3)) {
#pragma unroll
                    for (int d8 = (late8 ? 0 : dx); d8 <= dx; ++d8)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int i = s - dy;
                            if (i >= 0 && i < 4)
                                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wl8[dy * 3 + d8], fr8[(d8 + 4 - (s & 3)) & 3], acc[i], 0, 0, 0, scale_a, 0, scale_b);
                        }
                }
                if (s < 5) {
                    fr[(f + 12 - s) % 13] = *(lds_h8_t)(xcur + fa_of(f) + (unsigned)((s + 1) * ROWB));
                    if (!late8 && ks == 0) fr8[(dx + 3 - (s & 3)) & 3] = read_q8(dx, s + 1);      // (fragment dx of the next row into the set fragment dx - 1 of this row has left)
                    if (late8 && f == 11) {
#pragma unroll
                        for (int d8 = 0; d8 < 3; ++d8) fr8[(d8 + 3 - (s & 3)) & 3] = read_q8(d8, s + 1);
                    }
                }
                run_ops(std::integral_constant<int, 2>{}, S_, F_);
#pragma unroll
                for (int i_ = 0; i_ < 3; ++i_) {
                    if (i_ < nm) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i_ == 0 && s < 5) __builtin_amdgcn_sched_group_barrier(0x100, (ks == 0 && !late8) ? 3 : 1, 0);
                    if (i_ < nm) __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);
                }
                if (late8 && f == 11) {
#pragma unroll
                    for (int i_ = 0; i_ < 3; ++i_) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (s < 5) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x006, 10, 0);
                    }
                }
                if (!late8 && ks == 3) {
#pragma unroll
                    for (int i_ = 0; i_ < 3; ++i_) {
                        if (i_ < nm) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i_ < nm) __builtin_amdgcn_sched_group_barrier(0x006, 10, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
#define Q8_CHUNK(F) chunk(std::integral_constant<int, F>{});
            Q8_CHUNK(0) Q8_CHUNK(1) Q8_CHUNK(2) Q8_CHUNK(3) Q8_CHUNK(4) Q8_CHUNK(5) Q8_CHUNK(6) Q8_CHUNK(7) Q8_CHUNK(8) Q8_CHUNK(9) Q8_CHUNK(10) Q8_CHUNK(11)
#undef Q8_CHUNK
#ifdef Q8_STEPWAIT
            __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
            Q8_STAMP(15 + s)
        };
#define Q8_STEP(S) step_l(std::integral_constant<int, S>{});
        Q8_STEP(0) Q8_STEP(1) Q8_STEP(2) Q8_STEP(3) Q8_STEP(4) Q8_STEP(5)
#undef Q8_STEP
        run_ops(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});      // output row 3 of the wave
        Q8_STAMP(21)
        it_cur = itn;
    }
#ifdef Q8_TRACE
    if (a.pool && g < 8 && K > 8 && lane < 32) {
        unsigned long long* t = (unsigned long long*)a.pool;
        t[((g * 2 + 0) * 4 + w4) * 32 + lane] = *(unsigned long long*)(smem + LDS_BYTES + ((0 * 4 + w4) * 32 + lane) * 8);
        t[((g * 2 + 1) * 4 + w4) * 32 + lane] = *(unsigned long long*)(smem + LDS_BYTES + ((1 * 4 + w4) * 32 + lane) * 8);
    }
#endif
#endif
}

template <int EPI, bool IN8, bool OUT8>
hipError_t set_limit()
{
    return hipFuncSetAttribute((const void*)conv64_q8_kernel<EPI, IN8, OUT8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + TRACE_LDS);
}

}  // namespace

hipError_t conv64_q8_init()
{
    hipError_t e;
    if ((e = set_limit<0, false, false>()) != hipSuccess) return e;
    if ((e = set_limit<0, false, true>()) != hipSuccess) return e;
    if ((e = set_limit<0, true, true>()) != hipSuccess) return e;
    if ((e = set_limit<1, false, false>()) != hipSuccess) return e;
    if ((e = set_limit<1, true, true>()) != hipSuccess) return e;
    if ((e = set_limit<2, false, false>()) != hipSuccess) return e;
    if ((e = set_limit<2, true, true>()) != hipSuccess) return e;
    return set_limit<2, true, false>();
}

// false: the layer does not fit this kernel (caller uses conv64_x3)
bool launch_conv64_q8(ConvX3Args a, int max_groups, hipStream_t s)
{
    if (!(a.slope <= 1.f)) return false;
    if ((long long)a.B * a.H * a.W * 128 + (a.W + 1ll) * 128 >= (1ll << 32) - 65536) return false;
#ifdef Q8_TRACE
    static unsigned long long* tbuf = nullptr;
    static int tcount = 0;
    if (!tbuf) { (void)hipMalloc((void**)&tbuf, 8 * 2 * 4 * 32 * 8); (void)hipMemset(tbuf, 0, 8 * 2 * 4 * 32 * 8); }
    a.pool = nullptr;
    const bool big = (long long)a.B * ((a.W + TW - 1) / TW) * ((a.H + TH - 1) / TH) >= 256 * 12;
    if (big && ++tcount == 30) a.pool = (float*)tbuf;         // (one full-size launch deep inside a warmed-up run carries the stamps)
    if (big && tcount == 31) {
        std::vector<unsigned long long> host(8 * 2 * 4 * 32);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(host.data(), tbuf, host.size() * 8, hipMemcpyDeviceToHost);
        if (FILE* fp = fopen("/tmp/q8_trace.bin", "wb")) { fwrite(host.data(), 8, host.size(), fp); fclose(fp); }
    }
#endif
    if (!a.in_hi || !a.in_lo || !a.out_hi || !a.out_lo || !a.wq_hi16 || !a.wq_hi8 || !a.wq_lo8 || a.pool) {
#ifndef Q8_TRACE
        return false;
#endif
    }
    if ((a.res_hi == nullptr) != (a.res_lo == nullptr)) return false;
    if (a.res_hi && a.slope != 1.f) return false;
    a.px = (a.W + TW - 1) / TW;
    a.py = (a.H + TH - 1) / TH;
    const long long items = (long long)a.B * a.px * a.py;
    const int G = (int)std::min<long long>(items, max_groups);
    const dim3 grid(G), blk(256);
    // formats of the low parts (in8: input and residual, out8: output) -- the combinations a chain of these layers needs (engine.cpp, forward):
    //   conv_input2 fp8 (the stem writes it) or fp16 -> fp8 | conv_1 fp8 -> fp8 | conv_2 fp8 -> fp8, and fp8 -> fp16 for the last one (the fused ARSB kernels read fp16 low parts)
    const int epi = a.res_hi ? 2 : a.slope != 1.f ? 1 : 0, fmt = epi * 4 + (a.in8 ? 2 : 0) + (a.out8 ? 1 : 0);
    const size_t lds = LDS_BYTES + TRACE_LDS;
    switch (fmt) {
    case 0: conv64_q8_kernel<0, false, false><<<grid, blk, lds, s>>>(a); break;
    case 1: conv64_q8_kernel<0, false, true><<<grid, blk, lds, s>>>(a); break;
    case 3: conv64_q8_kernel<0, true, true><<<grid, blk, lds, s>>>(a); break;
    case 4: conv64_q8_kernel<1, false, false><<<grid, blk, lds, s>>>(a); break;
    case 7: conv64_q8_kernel<1, true, true><<<grid, blk, lds, s>>>(a); break;
    case 8: conv64_q8_kernel<2, false, false><<<grid, blk, lds, s>>>(a); break;
    case 11: conv64_q8_kernel<2, true, true><<<grid, blk, lds, s>>>(a); break;
    case 10: conv64_q8_kernel<2, true, false><<<grid, blk, lds, s>>>(a); break;
    default: return false;
    }
    return true;
}
