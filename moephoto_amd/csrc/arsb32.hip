// arsb32.hip -- one ARSB (python/models.py:76-80 of the reference),  y = x + s * conv_2(PReLU(conv_1(x))),  as ONE kernel on the
// 32x32x16 MFMA shape with all four waves in lock-step on the same conv (second form of arsb_fused.hip, VERDICT r02 task 2).
//
// Why a second form: arsb_fused.hip (v_mfma_f32_16x16x32_f16, wave = 16 output channels) is bound by instruction issue -- 648 MFMAs of 16
// cycles ride with 1,270 other instructions per patch and wave, about two per MFMA where two fit, PMC MFMA busy 60 %, and removing 11 % of its
// MFMAs changed the frame by 0.9 % (profiles/r03/d_arsb_continuation_ab.txt).  With v_mfma_f32_32x32x16_f16 the same FLOPs are 324 issues of 32
// cycles, each with room for ~5 other instructions, and a wave produces 32 channels of a pixel (two 16-byte slots) per row.
//
//   wave (c, h)   output channels 32c .. 32c+31 of BOTH convs; conv_1: m rows 5h .. 5h+4 of the patch's ten, conv_2: output rows 4h .. 4h+3 of its eight
//   weights       2 x 36 A fragments (tap, k-slice) = 288 registers per wave: conv_1's and 28 of conv_2's in AGPRs (256), the last eight fragments of
//                 conv_2 in arch VGPRs; accumulators in arch VGPRs (-amdgpu-mfma-vgpr-form: the epilogues read them without v_accvgpr_read)
//   rows stream   both convs stream their input rows (conv3x3_rw.hip): the twelve B fragments (dx, k-slice) of a row feed the up to three rows they
//                 touch.  THIRTEEN fragment registers rotate: fragment f of the next row is read into the register fragment f-1 of this row has
//                 just left (row step t keeps fragment f in register (f - t) mod 13; a patch has 7 + 6 = 13 row steps, so the rotation closes).
//                 Reading it into fragment f's own register instead -- one set, 48 registers -- made every ds_read wait for the MFMA that had just
//                 been issued on that register (measured: 24k cycles per patch instead of 10k of MFMA); two full sets do not fit the register file
//   LDS           x patch 12 x 34 pixels x 128 B (halo 2), double buffered, raw-buffer LDS-DMA (rejected offsets = zero padding); m patch 10 x 32
//                 (pitch 34): conv_1's epilogue writes the wave's 32 channels of a pixel as two ds_write_b128 in the B-fragment image, conv_2 reads
//                 all 64; 2 x 53,248 + 43,520 = 150,016 B
//   residual      x_hi comes from the x patch in LDS (two ds_read_b128 per row), x_lo (hi + lo 2^-11 stream, MOE_PREC_MIXED) from HBM by raw-buffer
//                 loads; y_hi / y_lo leave as 16-byte raw-buffer stores (rejected offsets drop the two halo columns and out-of-image rows)
//   sync          barrier A at the start of conv_1's row step 1 (every wave has finished conv_2 of the previous patch: m and the old x buffer may be
//                 overwritten; the DMA of patch p+1 is issued behind it), barrier B between the convs (m complete, x[p+1] landed)
//   epilogues     micro-op tables dealt out to the chunks of a row step (conv3x3_rw.hip): PReLU by channel pairs, image writes, residual adds and
//                 hi/lo splits by 16-byte slot, stores
//
// Same arithmetic as arsb_fused.hip up to the summation order inside a conv (operands, rounding points and the hi/lo stream are identical).
#include "common.h"
#include "rowtile.h"
#include <type_traits>

namespace {

constexpr int TW = 30, TH = 8;                 // stored outputs per patch
constexpr int XW = 34, XH = 12;                // x patch (halo 2); the m patch shares the pitch
constexpr int MH = 10;
constexpr int NPIX = XW * XH;                  // 408
constexpr int NDMA_W = 13;                     // 1-KiB pieces per wave: 52 >= 408 / 8
constexpr int XBYTES = NDMA_W * 4 * 1024;      // 53,248
constexpr int ROWB = XW * 128;                 // bytes of a patch row
constexpr int MBYTES = MH * ROWB;              // 43,520
constexpr int LDS_BYTES = 2 * XBYTES + MBYTES; // 150,016

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;

enum OpKind : int { OP_NONE = 0, OP_DMA, OP_P, OP_MW, OP_XLO, OP_XHI, OP_RES, OP_SPL, OP_ST };
struct Op { int kind, a, b, c; };
struct OpList {
    int n = 0;
    Op op[48] = {};
    constexpr void push(int kind, int a = 0, int b = 0, int c = 0) { op[n] = Op{kind, a, b, c}; ++n; }
    constexpr void append(const OpList& o) { for (int i = 0; i < o.n; ++i) { op[n] = o.op[i]; ++n; } }
};
constexpr OpList interleave(const OpList& x, const OpList& y)
{
    OpList r;
    int i = 0, k = 0;
    while (i < x.n || k < y.n) {
        if (k >= y.n || (i < x.n && (long long)i * y.n <= (long long)k * x.n)) { r.op[r.n] = x.op[i]; ++i; }
        else { r.op[r.n] = y.op[k]; ++k; }
        ++r.n;
    }
    return r;
}
// epilogue of m row i (of the wave's five): PReLU of its 16 values by two channel pairs, one image write per 16-byte slot
constexpr OpList mrow_ops(int i)
{
    OpList r;
    r.push(OP_P, i, 0, 2); r.push(OP_P, i, 2, 2); r.push(OP_MW, i, 0);
    r.push(OP_P, i, 4, 2); r.push(OP_P, i, 6, 2); r.push(OP_MW, i, 1);
    return r;
}
// epilogue of slot o (eight channels of the lane's pixel) of output row i (of the wave's four): residual add in fp32 by halves, [hi/lo split by channel
// pairs,] store(s).  The next row's x_hi word takes the register this row's has just released (ONE set of residual registers).
constexpr OpList yslot_ops(int i, int o, bool lo)
{
    OpList r;
    r.push(OP_RES, i, o, 0); r.push(OP_RES, i, o, 2);
    if (i < 3) r.push(OP_XHI, i + 1, o);
    if (lo) { r.push(OP_SPL, i, o, 0); r.push(OP_SPL, i, o, 2); }
    r.push(OP_ST, i, o);
    return r;
}
constexpr OpList yrow_ops(int i, bool lo) { OpList r = yslot_ops(i, 0, lo); r.append(yslot_ops(i, 1, lo)); return r; }      // (slot after slot: they share sh / sl)
// A row step has 12 chunks of 1, 2 or 3 MFMAs; an MFMA hides about five other instructions and an op is dealt to ONE chunk, so ops are kept to ~12
// instructions (a DMA piece = address half + validity/issue half) and the steps with 12 MFMAs get as little as the dependences allow:
//   * output rows 2 (slot 1) and 3 are finished by conv_2's last two row steps (24 + 12 MFMAs): their epilogues ride in row steps 0..2 of the NEXT patch's
//     conv_1 (acc[2] / acc[3] are first written there in steps 2 / 3; the last patch of a workgroup runs them after the loop);
//   * the x_lo words of a patch are fetched in conv_1's steps 4 and 5, thousands of cycles before conv_2's step 3 needs the first (loads and stores share
//     vmcnt IN ORDER: issued behind a store they would wait for its acknowledgement; issued a step or two ahead of their use the wait showed in the
//     trace as 1.3 - 2.2k cycles per patch);
//   * DMA pieces 0..7 (patch rows 0 .. 7.5 of the next patch) go out behind barrier A in conv_1's steps 1..3 and land by barrier B; pieces 8..12 ride in
//     conv_2's idle steps 0..2 and are published by the NEXT barrier A (conv_1 reads rows >= 7 from its step 1 on).
constexpr OpList dma_ops(int i0, int i1)
{
    OpList r;
    for (int i = i0; i < i1; ++i) { r.push(OP_DMA, i, 0); r.push(OP_DMA, i, 1); }
    return r;
}
constexpr OpList tail_ops(bool lo)             // what the next patch's conv_1 would have carried
{
    OpList r = yslot_ops(2, 1, lo);
    r.append(yslot_ops(3, 0, lo)); r.append(yslot_ops(3, 1, lo));
    return r;
}
constexpr OpList conv1_ops(int s, bool lo)     // row step s of conv_1 (x row 5h + s): 12, 24, 36, 36, 36, 24, 12 MFMAs
{
    OpList r, x;
    if (s == 0) r = yslot_ops(2, 1, lo);
    if (s == 1) r = interleave(dma_ops(0, 1), yslot_ops(3, 0, lo));      // (behind barrier A)
    if (s == 2) r = interleave(dma_ops(1, 5), yslot_ops(3, 1, lo));
    if (s == 3) r = interleave(dma_ops(5, 8), mrow_ops(0));
    if (s == 4) { if (lo) { x.push(OP_XLO, 0, 0); x.push(OP_XLO, 0, 1); x.push(OP_XLO, 1, 0); x.push(OP_XLO, 1, 1); } r = interleave(x, mrow_ops(1)); }
    if (s == 5) { if (lo) { x.push(OP_XLO, 2, 0); x.push(OP_XLO, 2, 1); x.push(OP_XLO, 3, 0); x.push(OP_XLO, 3, 1); } r = interleave(x, mrow_ops(2)); }
    if (s == 6) r = mrow_ops(3);
    return r;
}
constexpr OpList conv2_ops(int s, bool lo)     // row step s of conv_2 (m row 4h + s): 12, 24, 36, 36, 24, 12 MFMAs
{
    OpList r;
    if (s == 0) r = dma_ops(8, 9);
    if (s == 1) r = dma_ops(9, 11);
    if (s == 2) { r = dma_ops(11, 13); r.push(OP_XHI, 0, 0); r.push(OP_XHI, 0, 1); }
    if (s == 3) r = yrow_ops(0, lo);
    if (s == 4) r = yrow_ops(1, lo);
    if (s == 5) r = yslot_ops(2, 0, lo);
    return r;
}
// stores in flight behind the last DMA piece of conv_2 when the next barrier A is reached (rows 0, 1, slot 0 of row 2; slot 1 of row 2 in conv_1's step 0)
constexpr int stores_behind_dma(bool lo) { return lo ? 12 : 6; }

struct Item { int b, pyi, pxi; };

// cycle-level trace (tools/mk_variant.sh trace32 arsb32.hip -DA32_TRACE; tools/show_trace_a32.py): s_memtime at the phase boundaries and after every row step
#ifdef A32_TRACE
#define A32_STAMP(SLOT) if (a.trace && g < 8 && p < 16 && lane == 0) a.trace[((g * 16 + p) * 4 + w4) * 40 + (SLOT)] = __builtin_amdgcn_s_memtime();
#else
#define A32_STAMP(SLOT)
#endif

template <bool LO>
__global__ __launch_bounds__(256) void arsb32_kernel(ArsbArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr unsigned kOOR = 0xFFFF0000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const unsigned mbase = lds0 + 2u * XBYTES;
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

    // ---- weights: 36 + 36 A fragments of this wave's 32 output channels (pack_conv order; MFMA row i = 8q + 4h' + e is given channel
    // 16 (q >> 1) + 8 h' + 4 (q & 1) + e, so that a lane's registers 8g .. 8g+7 are eight consecutive channels = one 16-byte slot) --------------
    half8_t w1[36], w2[36];
    {
        const int wi = lane & 31, wq = wi >> 3;
        const int src = (lane & 32) | (16 * (wq >> 1) + 8 * ((wi >> 2) & 1) + 4 * (wq & 1) + (wi & 3));
#pragma unroll
        for (int f = 0; f < 36; ++f) {
            w1[f] = *(const half8_t*)(a.w1 + ((f * 2 + c) * 64 + src) * 8);
            w2[f] = *(const half8_t*)(a.w2 + ((f * 2 + c) * 64 + src) * 8);
        }
#pragma unroll
        for (int f = 0; f < 36; ++f) asm volatile("" : "+a"(w1[f]));
#pragma unroll
        for (int f = 0; f < 28; ++f) asm volatile("" : "+a"(w2[f]));
#pragma unroll
        for (int f = 28; f < 36; ++f) asm volatile("" : "+v"(w2[f]));
    }

    // ---- patch DMA (conv3x3_rw.hip): lane offsets rebuilt only when the border pattern changes, patch origin (row -2, column -2) in an SGPR -----
    const unsigned in_pad = (unsigned)(2 * a.W + 2) * 128u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.x_hi - in_pad), 0,
                                                                         (unsigned)a.B * a.H * a.W * 128u + in_pad, 0x00020000);
    const unsigned nbytes = (unsigned)a.B * a.H * a.W * 128u;
    const __amdgpu_buffer_rsrc_t rlo = __builtin_amdgcn_make_buffer_rsrc((void*)(LO ? a.x_lo : a.x_hi), 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ryh = __builtin_amdgcn_make_buffer_rsrc((void*)a.y_hi, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ryl = __builtin_amdgcn_make_buffer_rsrc((void*)(LO ? a.y_lo : a.y_hi), 0, nbytes, 0x00020000);
    // The lane's source offset of DMA piece i (its pixel of the 12 x 34 patch, its logical 16-byte slot) is formed when the piece is issued -- ~14 VALU that
    // ride in the MFMA stream.  A per-lane table (13 registers, rebuilt when the patch cuts the image border differently: conv3x3_rw.hip) did not fit:
    // with both convs' weights resident the rebuild block spilled to scratch and cost 8,000 cycles on 44 % of the patches.
    const int qlane = w4 * 8 + (lane >> 3);
    unsigned d_off = 0, d_r = 0, d_cc = 0;                    // DMA piece in the making: byte offset inside the patch, patch row / column of the lane's pixel
    auto piece_addr = [&](int i) {
        unsigned q = (unsigned)(i * 32 + qlane);
        asm volatile("" : "+v"(q));                           // (recomputed per piece: hoisted out of the patch loop, the row / column of 13 pieces spill)
        d_r = __umul24(q, 241u) >> 13;                        // q / 34 for q < 442   (24-bit multiplies: full rate, and cheap enough that the compiler
        d_cc = (unsigned)(__mul24((int)d_r, -XW) + (int)q);     //  selects instead of branching around the offset -- a branch splits the pinned schedule)
        const unsigned sl = (unsigned)(lane & 7) ^ ((d_cc >> 1) & 7u);      // logical 16-B slot behind this physical slot
        d_off = ((__umul24(d_r, (unsigned)a.W) + d_cc) << 7) | (sl << 4);
    };
    auto piece_off = [&](int i, int ya, int xa, bool live) {      // ya, xa: image row / column of the patch origin (wave-uniform)
        bool ok = ((unsigned)ya + d_r < (unsigned)a.H) & ((unsigned)xa + d_cc < (unsigned)a.W) & live;
        if (i * 32 + 31 >= NPIX) ok &= (i * 32 + qlane < NPIX);
        return ok ? d_off : kOOR;
    };
    auto origin = [&](const Item& it) {
        return (unsigned)((it.b * a.H + it.pyi * TH - 2) * a.W + it.pxi * TW - 2 + 2 * a.W + 2) * 128u;
    };

    // ---- LDS addressing.  B fragment f = (dx, ks) of a patch row: pixel (row, col) at (row * 34 + col) * 128, 16-B slot s at s ^ ((col >> 1) & 7);
    // lane (j, hh) reads slot 2 ks + hh of column j + dx.  Row offsets are immediates; the buffer / first-row part is added per phase. -------------
    // fa[f]: byte address of fragment f in the first row this wave reads next (x buffer + row 5h for conv_1, m + row 4h for conv_2): the twelve
    // lane parts are shifted by a wave-uniform delta at every phase change instead of being kept twice
    unsigned fa[12];
#pragma unroll
    for (int f = 0; f < 12; ++f) {
        const int dx = f >> 2, ks = f & 3;
        const int cc = j + dx, z = (cc >> 1) & 7;
        fa[f] = lds0 + (unsigned)(5 * h * ROWB) + (unsigned)(cc * 128 + (((2 * ks + hh) ^ z) << 4));      // conv_1 of the first patch: x buffer 0
    }
    // this lane's two 16-byte slots of a pixel it produces: channels 32c + 16 o + 8 hh .. +7 = slot 4c + 2o + hh.  m pixel column j; residual pixel column j + 2
    unsigned mw[2], xh[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int s = 4 * c + 2 * o + hh;
        mw[o] = mbase + (unsigned)(5 * h * ROWB + j * 128 + ((s ^ ((j >> 1) & 7)) << 4));
        xh[o] = (unsigned)((4 * h + 2) * ROWB + (j + 2) * 128 + ((s ^ (((j + 2) >> 1) & 7)) << 4));
    }
    const unsigned lane_ob = (unsigned)(j * 128 + (32 * c + 8 * hh) * 2);      // byte offset of slot o = 0 of output column j inside a patch row of the stream tensors
    unsigned slope2;
    {
        const half2_t s2 = {(half_t)a.slope, (half_t)a.slope};
        slope2 = __builtin_bit_cast(unsigned, s2);
    }
    const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    float16_t acc[4];
    half8_t fr[13];           // fragment f of row step t (0..6 conv_1, 7..12 conv_2) lives in fr[(f - t) mod 13]
    u4_t xlo[4][2], xhv[2];   // residual words in flight: x_lo of the four rows, x_hi of ONE row (slot o)
    unsigned sh[4], sl[4];
    // stream tensors: byte offset of (output row 4h, column x0) of the patch, the lane's column part; row 3 of the PREVIOUS patch (stored one patch late)
    unsigned so0 = 0, vo = kOOR, so2p = kOOR, so3p = kOOR, vop = kOOR, xprev = lds0;
    int yrow0 = 0;
    {
        const u4_t z4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int o = 0; o < 2; ++o) { xhv[o] = z4; xlo[3][o] = z4; }      // (the first patch runs the late epilogues of "the patch before" with their stores rejected)
        xlo[2][1] = z4;
        acc[2] = zero16; acc[3] = zero16;
    }
    auto row_so = [&](int i) { return (yrow0 + i < a.H) ? so0 + (unsigned)(i * a.W * 128) : kOOR; };

    // ===== prologue: the first patch ====================================================================================================================
    Item it_cur = decode(g);
    {
        const unsigned org = origin(it_cur);
#pragma unroll
        for (int i = 0; i < NDMA_W; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + (i * 4 + w4) * 1024), 16,
                                                     (piece_addr(i), piece_off(i, it_cur.pyi * TH - 2, it_cur.pxi * TW - 2, true)), org, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int f = 0; f < 12; ++f) fr[f] = *(lds_h8_t)(fa[f]);
    }

    for (int p = 0; p < K; ++p) {
        const Item it = it_cur;
        const bool has_next = p + 1 < K;
        const Item itn = advance(it);                         // (beyond the last patch: nothing is fetched, see patch_key)
        const unsigned xcur = lds0 + (unsigned)((p & 1) * XBYTES), xnxt = lds0 + (unsigned)(((p + 1) & 1) * XBYTES);
        // (pinned in SGPRs once: left to the compiler, the carries of advance() and the origin are re-derived in front of every DMA piece)
        const unsigned orgn = (unsigned)__builtin_amdgcn_readfirstlane((int)origin(itn));
        const int yan = __builtin_amdgcn_readfirstlane(itn.pyi * TH - 2), xan = __builtin_amdgcn_readfirstlane(itn.pxi * TW - 2);
        const unsigned dnxt = (unsigned)__builtin_amdgcn_readfirstlane(((p + 1) & 1) * XBYTES + w4 * 1024);
        const int y0 = it.pyi * TH, x0 = it.pxi * TW;
        A32_STAMP(0)
        so0 = (unsigned)(((it.b * a.H + y0 + 4 * h) * a.W + x0) * 128);
        vo = ((j < TW) & (x0 + j < a.W)) ? lane_ob : kOOR;
        yrow0 = y0 + 4 * h;
        unsigned hp[8];                                       // activated m row being written (packed halves: slot 0 | slot 1)

        // ---- micro-ops ---------------------------------------------------------------------------------------------------------------------------
        auto op_dma = [&](auto I_, auto HALF_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value, half = decltype(HALF_)::value;
            if constexpr (half == 0) piece_addr(i);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)((char*)smem + dnxt + i * 4096), 16,
                                                          piece_off(i, yan, xan, has_next), orgn, 0, 0);
        };
        auto op_p = [&](auto I_, auto K0_, auto N_) __attribute__((always_inline)) {      // PReLU on packed halves (slope <= 1) of channel pairs k0 .. k0+n-1 of m row i
            constexpr int i = decltype(I_)::value, k0 = decltype(K0_)::value, n = decltype(N_)::value;
#pragma unroll
            for (int k = k0; k < k0 + n; ++k) {
                const half2_t pr = {(half_t)acc[i & 3][2 * k], (half_t)acc[i & 3][2 * k + 1]};
                const half2_t t = pr * __builtin_bit_cast(half2_t, slope2);
                hp[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(pr, t));
            }
        };
        auto op_mw = [&](auto I_, auto O_) __attribute__((always_inline)) {      // (inline asm: a C++ store to LDS is ordered behind the LDS-DMA in flight with vmcnt(0))
            constexpr int i = decltype(I_)::value, o = decltype(O_)::value;
            const u4_t d = {hp[4 * o], hp[4 * o + 1], hp[4 * o + 2], hp[4 * o + 3]};
            const unsigned ad = mw[o];         // (a local: inline-asm operands inside a generic lambda do not capture)
            asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(ad), "v"(d), "n"(i * ROWB) : "memory");
        };
        auto op_xlo = [&](auto I_, auto O_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value, o = decltype(O_)::value;
            xlo[i][o] = __builtin_amdgcn_raw_buffer_load_b128(rlo, vo + (unsigned)(o * 32), row_so(i), 0);
        };
        auto op_xhi = [&](auto I_, auto O_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value, o = decltype(O_)::value;
            // (row 3, slot 1 follows slot 1 of row 2, which rides in the NEXT patch's conv_1 -- or in the tail of this one: xprev is the buffer of the patch it belongs to)
            xhv[o] = *(const __attribute__((address_space(3))) u4_t*)((i == 3 && o == 1 ? xprev : xcur) + xh[o] + (unsigned)(i * ROWB));
        };
        auto op_res = [&](auto I_, auto O_, auto K0_) __attribute__((always_inline)) {      // acc += x_hi [+ x_lo 2^-11] (in place) for channel pairs k0, k0+1 of slot o
            constexpr int i = decltype(I_)::value, o = decltype(O_)::value, k0 = decltype(K0_)::value;
#pragma unroll
            for (int k = k0; k < k0 + 2; ++k) {
                float v0 = acc[i][8 * o + 2 * k], v1 = acc[i][8 * o + 2 * k + 1];
                v0 = mix_lo(xhv[o][k], 1.0f, v0); v1 = mix_hi(xhv[o][k], 1.0f, v1);
                if (LO) { v0 = mix_lo(xlo[i][o][k], 0.00048828125f, v0); v1 = mix_hi(xlo[i][o][k], 0.00048828125f, v1); }
                acc[i][8 * o + 2 * k] = v0; acc[i][8 * o + 2 * k + 1] = v1;
            }
            if (!LO) {
#pragma unroll
                for (int k = k0; k < k0 + 2; ++k) {
                    const half2_t pr = {(half_t)acc[i][8 * o + 2 * k], (half_t)acc[i][8 * o + 2 * k + 1]};
                    sh[k] = __builtin_bit_cast(unsigned, pr);
                }
            }
        };
        auto op_spl = [&](auto I_, auto O_, auto K0_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value, o = decltype(O_)::value, k0 = decltype(K0_)::value;
#pragma unroll
            for (int k = k0; k < k0 + 2; ++k) split2(acc[i][8 * o + 2 * k], acc[i][8 * o + 2 * k + 1], -2048.f, sh[k], sl[k]);
        };
        auto op_st = [&](auto I_, auto O_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value, o = decltype(O_)::value;
            constexpr bool late = i == 3 || (i == 2 && o == 1);      // (runs one patch late: offsets of the patch it belongs to)
            const unsigned so = i == 3 ? so3p : late ? so2p : row_so(i);
            const unsigned vv = (late ? vop : vo) + (unsigned)(o * 32);
            const u4_t dh = {sh[0], sh[1], sh[2], sh[3]};
            __builtin_amdgcn_raw_buffer_store_b128(dh, ryh, vv, so, 0);
            if (LO) {
                const u4_t dl = {sl[0], sl[1], sl[2], sl[3]};
                __builtin_amdgcn_raw_buffer_store_b128(dl, ryl, vv, so, 0);
            }
        };
        // the ops [f n / 12, (f + 1) n / 12) of a list: PH 0 conv_1 row step S, 1 conv_2 row step S, 2 / 3 the tails (m row 4, output row 3; F = 0, all ops)
        auto run_ops = [&](auto PH_, auto S_, auto F_) __attribute__((always_inline)) {
            constexpr int PH = decltype(PH_)::value, S = decltype(S_)::value, F = decltype(F_)::value;
            constexpr OpList L = PH == 0 ? conv1_ops(S, LO) : PH == 1 ? conv2_ops(S, LO) : PH == 2 ? mrow_ops(4) : tail_ops(LO);
            constexpr int lo = PH < 2 ? F * L.n / 12 : 0, hi = PH < 2 ? (F + 1) * L.n / 12 : L.n;
            auto run = [&](auto I_) __attribute__((always_inline)) {
                constexpr int I = decltype(I_)::value;
                if constexpr (I >= lo && I < hi) {
                    constexpr Op o = L.op[I];
                    if constexpr (o.kind == OP_DMA) op_dma(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                    if constexpr (o.kind == OP_P) op_p(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{}, std::integral_constant<int, o.c>{});
                    if constexpr (o.kind == OP_MW) op_mw(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                    if constexpr (o.kind == OP_XLO) op_xlo(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                    if constexpr (o.kind == OP_XHI) op_xhi(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                    if constexpr (o.kind == OP_RES) op_res(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{}, std::integral_constant<int, o.c>{});
                    if constexpr (o.kind == OP_SPL) op_spl(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{}, std::integral_constant<int, o.c>{});
                    if constexpr (o.kind == OP_ST) op_st(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                }
            };
#define A32_OP(I) run(std::integral_constant<int, I>{});
            A32_OP(0) A32_OP(1) A32_OP(2) A32_OP(3) A32_OP(4) A32_OP(5) A32_OP(6) A32_OP(7) A32_OP(8) A32_OP(9) A32_OP(10) A32_OP(11) A32_OP(12) A32_OP(13) A32_OP(14) A32_OP(15)
            A32_OP(16) A32_OP(17) A32_OP(18) A32_OP(19) A32_OP(20) A32_OP(21) A32_OP(22) A32_OP(23) A32_OP(24) A32_OP(25) A32_OP(26) A32_OP(27) A32_OP(28) A32_OP(29) A32_OP(30) A32_OP(31)
#undef A32_OP
        };

        // ================= conv_1: m rows 5h .. 5h+4 from x rows 5h .. 5h+6 =================================================================================
        auto step1 = [&](auto S_) __attribute__((always_inline)) {
            constexpr int s = decltype(S_)::value;
            constexpr int nm = (s <= 4 ? 1 : 0) + ((s >= 1 && s <= 5) ? 1 : 0) + ((s >= 2) ? 1 : 0);      // rows this x row contributes to
            if (s == 1) {       // barrier A: nobody reads m or the old x buffer any more; DMA pieces 8..12 of this patch (issued in the previous conv_2) have landed
                if (LO) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                static_assert(stores_behind_dma(true) == 12 && stores_behind_dma(false) == 6, "counted wait of barrier A");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            auto chunk = [&](auto F_) __attribute__((always_inline)) {
                constexpr int f = decltype(F_)::value;
                constexpr int dx = f >> 2, ks = f & 3;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int i = s - dy;
                    if (i >= 0 && i < 5)
                        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[(dy * 3 + dx) * 4 + ks], fr[(f + 13 - s) % 13], (dy == 0 && f == 0) ? zero16 : acc[i & 3], 0, 0, 0);
                }
                if (s < 6) fr[(f + 12 - s) % 13] = *(lds_h8_t)(fa[f] + (unsigned)((s + 1) * ROWB));
                run_ops(std::integral_constant<int, 0>{}, S_, F_);
#pragma unroll
                for (int i_ = 0; i_ < 3; ++i_) {
                    if (i_ < nm) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i_ == 0 && s < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (i_ < nm) __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
#define A32_CHUNK(F) chunk(std::integral_constant<int, F>{});
            A32_CHUNK(0) A32_CHUNK(1) A32_CHUNK(2) A32_CHUNK(3) A32_CHUNK(4) A32_CHUNK(5) A32_CHUNK(6) A32_CHUNK(7) A32_CHUNK(8) A32_CHUNK(9) A32_CHUNK(10) A32_CHUNK(11)
#undef A32_CHUNK
            __builtin_amdgcn_s_waitcnt(0xC07F);
            A32_STAMP(1 + s)
        };
#define A32_STEP(S) step1(std::integral_constant<int, S>{});
        A32_STEP(0) A32_STEP(1) A32_STEP(2) A32_STEP(3) A32_STEP(4) A32_STEP(5) A32_STEP(6)
#undef A32_STEP
        run_ops(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});      // m row 4 of the wave has no MFMAs of its own conv left to ride in
        // conv_2 pads with ZEROS: m pixels outside the image must be 0, not conv_1 evaluated there (border patches only; in-order LDS: behind the row writes)
        if ((y0 == 0) | (y0 + TH >= a.H) | (x0 == 0) | (x0 + TW + 1 >= a.W)) {
            const u4_t z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const bool in = ((unsigned)(y0 - 1 + 5 * h + i) < (unsigned)a.H) & ((unsigned)(x0 - 1 + j) < (unsigned)a.W);
                if (!in) {
                    asm volatile("ds_write_b128 %0, %1" ::"v"(mw[0] + (unsigned)(i * ROWB)), "v"(z) : "memory");
                    asm volatile("ds_write_b128 %0, %1" ::"v"(mw[1] + (unsigned)(i * ROWB)), "v"(z) : "memory");
                }
            }
        }
        A32_STAMP(8)
        // DMA pieces 0..7 of patch p+1 and the late stores of patch p-1 are complete; the eight x_lo loads issued behind them may still fly
        if (LO) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        A32_STAMP(9)
        __builtin_amdgcn_s_barrier();                         // barrier B: m is complete, x[p+1] has landed for every wave
        asm volatile("" ::: "memory");
        A32_STAMP(10)

        // ================= conv_2: output rows 4h .. 4h+3 from m rows 4h .. 4h+5 =================================================================================
        {
            const unsigned d12 = mbase + (unsigned)(4 * h * ROWB) - (xcur + (unsigned)(5 * h * ROWB));      // x buffer, row 5h -> m, row 4h
#pragma unroll
            for (int f = 0; f < 12; ++f) fa[f] += d12;
        }
#pragma unroll
        for (int f = 0; f < 12; ++f) fr[(f + 13 - 7) % 13] = *(lds_h8_t)(fa[f]);
        auto step2 = [&](auto S_) __attribute__((always_inline)) {
            constexpr int s = decltype(S_)::value;
            constexpr int nm = (s <= 3 ? 1 : 0) + ((s >= 1 && s <= 4) ? 1 : 0) + ((s >= 2) ? 1 : 0);
            if (s == 5) {       // the last row's reads already fetch row 0 of patch p+1's conv_1: m, row 4h -> the other x buffer, row 5h
                const unsigned d21 = xnxt + (unsigned)(5 * h * ROWB) - (mbase + (unsigned)(4 * h * ROWB));
#pragma unroll
                for (int f = 0; f < 12; ++f) fa[f] += d21;
            }
            if (LO && s == 3) __builtin_amdgcn_s_waitcnt(0x0F75);      // vmcnt(5): the x_lo words (issued in conv_1's steps 4, 5; only DMA pieces 8..12 are younger) have landed
                                                                        // BEFORE the first store goes out -- loads and stores share the counter, a later counted wait would also
                                                                        // wait for store acknowledgements
            auto chunk = [&](auto F_) __attribute__((always_inline)) {
                constexpr int f = decltype(F_)::value;
                constexpr int dx = f >> 2, ks = f & 3;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int i = s - dy;
                    if (i >= 0 && i < 4)
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[(dy * 3 + dx) * 4 + ks], fr[(f + 13 - (7 + s)) % 13], (dy == 0 && f == 0) ? zero16 : acc[i], 0, 0, 0);
                }
                if (s < 5) fr[(f + 12 - (7 + s)) % 13] = *(lds_h8_t)(fa[f] + (unsigned)((s + 1) * ROWB));
                else fr[f] = *(lds_h8_t)(fa[f]);            // row 0 of patch p+1's conv_1 (landed and published by barrier B): (f - 13) mod 13 = f
                run_ops(std::integral_constant<int, 1>{}, S_, F_);
#pragma unroll
                for (int i_ = 0; i_ < 3; ++i_) {
                    if (i_ < nm) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i_ == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (i_ < nm) __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
#define A32_CHUNK(F) chunk(std::integral_constant<int, F>{});
            A32_CHUNK(0) A32_CHUNK(1) A32_CHUNK(2) A32_CHUNK(3) A32_CHUNK(4) A32_CHUNK(5) A32_CHUNK(6) A32_CHUNK(7) A32_CHUNK(8) A32_CHUNK(9) A32_CHUNK(10) A32_CHUNK(11)
#undef A32_CHUNK
            __builtin_amdgcn_s_waitcnt(0xC07F);
            A32_STAMP(11 + s)
        };
#define A32_STEP(S) step2(std::integral_constant<int, S>{});
        A32_STEP(0) A32_STEP(1) A32_STEP(2) A32_STEP(3) A32_STEP(4) A32_STEP(5)
#undef A32_STEP
        A32_STAMP(17)
        so2p = row_so(2);                                     // slot 1 of output row 2 and row 3 of the wave: their epilogues ride in the next patch's conv_1
        so3p = row_so(3);
        vop = vo;
        xprev = xcur;
        it_cur = itn;
        if (!has_next) run_ops(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    }
#endif
}

}  // namespace

hipError_t arsb32_init()
{
    hipError_t e = hipFuncSetAttribute((const void*)arsb32_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)arsb32_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}

// w1 / w2: packed A fragments in the conv3x3_sp / pack_conv order (ConvLayer::w_hi).  false: the layer does not fit this kernel
bool launch_arsb32(ArsbArgs a, int max_groups, hipStream_t s)
{
    if (!(a.slope <= 1.f)) return false;                                  // PReLU as max(x, slope * x)
    if ((long long)a.B * a.H * a.W * 128 + (2ll * a.W + 2) * 128 >= (1ll << 32) - 65536) return false;   // 32-bit byte offsets
    if ((a.x_lo == nullptr) != (a.y_lo == nullptr)) return false;
    a.px = (a.W + TW - 1) / TW;
    a.py = (a.H + TH - 1) / TH;
    const long long items = (long long)a.B * a.px * a.py;
    const int G = (int)std::min<long long>(items, max_groups);
    if (a.x_lo) arsb32_kernel<true><<<dim3(G), dim3(256), LDS_BYTES, s>>>(a);
    else arsb32_kernel<false><<<dim3(G), dim3(256), LDS_BYTES, s>>>(a);
    return true;
}
