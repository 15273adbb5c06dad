// arsb_fused.hip -- one ARSB (python/models.py:76-80 of the reference),  y = x + s * conv_2(PReLU(conv_1(x))),  as ONE kernel.
//
// Why: an un-fused 3x3 64->64 layer moves 256-640 B per pixel for 73.7 kFLOP -- 115-288 FLOP/B, below MI355X's ~312 FLOP/B
// ridge (profiles/r01, r02: the trunk convs ran at 0.16-0.38 of the MFMA peak, HBM-bound).  Fused, x is read once and y
// written once: conv_1's output m never leaves the CU.
//
// How it fits: the two weight sets are 144 KiB, the CU has 160 KiB of LDS -- so the weights live in REGISTERS.  Wave w owns
// output channels 16w .. 16w+15 of BOTH convs for every pixel of the patch: as A operands of v_mfma_f32_16x16x32_f16 that is
// 2 x 18 fragments of 4 VGPRs = 144 VGPRs held for the whole launch (one wave per SIMD has 512).  LDS carries only activations,
//
//     x patch  12 x 34 pixels x 128 B (halo 2), double buffered, filled by LDS-DMA (global_load_lds, 1 KiB per wave-instruction)
//     m patch  10 x 32 pixels (halo 1) at a pitch of 34, written by conv_1's epilogue, read by conv_2
//
// and each B fragment (one ds_read_b128: 16 pixels x 32 channels) feeds the three output rows it touches.  Of the 32 computed
// output columns 30 are stored (columns 30, 31 would need m columns conv_1 did not compute): 83 % of the MFMA work is useful,
// the price of the fusion's halo (conv_1 over 10 x 32 pixels for 8 x 30 outputs).  Per patch and wave: 648 MFMAs (16 cycles
// each), 264 ds_read_b128, 20 ds_write_b64, 13 DMA pieces, 16 + 16 16-byte global accesses.
//
// LDS image: pixel-major, 128 B per pixel, the eight 16-byte channel slots of a pixel stored at slot ^ ((column >> 1) & 3).  A
// 32-channel k-step reads slots {2kh, 2kh+4, 2kh+1, 2kh+5} (lane group kq = 0..3): the hardware serves a ds_read_b128 in four
// groups of 16 lanes, each group = 16 consecutive pixels, half of them with slot s, half with s ^ 4 -- with the 2-bit column
// key every group covers all 64 banks exactly once, for every alignment of the 16-pixel window (dx = 0, 1, 2).  The weights
// are packed in the same k order (engine.cpp, pack_arsb).
//
// Epilogues ride in the MFMA stream row by row: an m row is finished (PReLU on packed halves, zero outside the image, LDS write)
// as soon as its third input row has been multiplied; an output row likewise.  v_permlane16_swap pairs the two 16-column tiles
// of a row so that every global access is 16 bytes per lane (8 consecutive channels of one pixel).  The trunk stream is carried
// as hi + lo * 2^-11 (MOE_PREC_MIXED): LO = true reads x_lo, adds in fp32 and stores y_hi and y_lo; only the MFMA operand is
// the fp16 part.
#include "common.h"
#include "rowtile.h"

namespace {

constexpr int TW = 30, TH = 8;                 // stored outputs per patch
constexpr int XW = 34, XH = 12;                // x patch (halo 2)
constexpr int MH = 10, MP = 34;                // m patch: 10 rows, 32 computed columns, pitch 34 (conv_2's columns 30, 31 read the 2 pad columns)
constexpr int NPIECE_W = 13;                   // 1-KiB DMA pieces per wave: 4 x 13 = 52 >= 408 pixels / 8
constexpr int XBYTES = 4 * NPIECE_W * 1024;    // 53,248
constexpr int MBYTES = MH * MP * 128;          // 43,520
constexpr int LDS_BYTES = 2 * XBYTES + MBYTES; // 150,016
static_assert(XW == MP, "x and m patches share the row pitch (one set of read offsets)");

__device__ __forceinline__ void dma16(const half_t* src, char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct Item { int b, pyi, pxi; };

// cycle-level trace (tools/trace_arsb.sh builds a -DARSB_TRACE variant): s_memtime at the phase boundaries and after every input row
#ifndef ARSB_ABL
#define ARSB_ABL 0      // timing ablations for tools/trace_arsb.sh (results are wrong when set): 1 no DMA, 2 no epilogues, 4 no residual loads, 8 no LDS reads
#endif
#ifdef ARSB_TRACE
#define ARSB_STAMP(SLOT)                                                                                  \
    if (a.trace && g < 8 && p < 16 && lane == 0) a.trace[((g * 16 + p) * 4 + w4) * 40 + (SLOT)] = __builtin_amdgcn_s_memtime();
#else
#define ARSB_STAMP(SLOT)
#endif

template <bool LO>
__global__ __launch_bounds__(256) void arsb_fused_kernel(ArsbArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xbuf = smem;
    char* const mbuf = smem + 2 * XBYTES;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of the dynamic segment
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);      // this wave's 16 output channels: 16 w4 ..
    const int n = lane & 15, q = lane >> 4;

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

    // ---- weights: 2 x 18 A fragments (tap, 32-channel half), resident in registers: [wave][f = tap * 2 + kh][lane][8] ------------
    half8_t w1[18], w2[18];
#pragma unroll
    for (int f = 0; f < 18; ++f) {
        w1[f] = *(const half8_t*)(a.w1 + ((w4 * 18 + f) * 64 + lane) * 8);
        w2[f] = *(const half8_t*)(a.w2 + ((w4 * 18 + f) * 64 + lane) * 8);
    }
    // Park them in the accumulator half of the register file (an MFMA reads its A operand from AGPRs as well): the arch VGPRs then
    // hold the accumulators and the epilogues read them directly -- with the weights in VGPRs the compiler put the accumulators into
    // AGPRs and every epilogue value cost a v_accvgpr_read (a fifth of the VALU stream, which is what limits this kernel).
#pragma unroll
    for (int f = 0; f < 18; ++f) {
        asm volatile("" : "+a"(w1[f]));
        asm volatile("" : "+a"(w2[f]));
    }

    // ---- DMA of an x patch: piece i * 4 + w4 carries pixels 8n' .. 8n'+7 (raster order of the 12 x 34 patch) ---------------------
    const unsigned long long zsrc = (unsigned long long)(a.zero + (lane & 7) * 8);
    int prc[NPIECE_W];            // (row << 8 | column) of this lane's pixel in piece i, row = 255 for the padding pixels behind the patch
    unsigned poff[NPIECE_W];      // byte offset of this lane's 16 bytes relative to the patch origin (row -2, column -2)
#pragma unroll
    for (int i = 0; i < NPIECE_W; ++i) {
        const int qq = (i * 4 + w4) * 8 + (lane >> 3);
        const int r = (qq * 241) >> 13;                   // qq / 34 for qq < 442
        const int cc = qq - r * XW;
        const int sl = (lane & 7) ^ ((cc >> 1) & 3);      // logical slot behind this lane's physical slot
        prc[i] = ((qq < XW * XH ? r : 255) << 8) | cc;
        poff[i] = (unsigned)((r * a.W + cc) * 128 + sl * 16);
    }
    auto issue_piece = [&](const Item& it, int i, char* dstbuf, bool live) {
        const int ya = it.pyi * TH - 2, xa = it.pxi * TW - 2;                           // wave-uniform
        const unsigned base = (unsigned)(((it.b * a.H + ya) * a.W + xa) * 128);         // 32-bit byte offset (the launcher checks the range)
        const bool ok = ((unsigned)(ya + (prc[i] >> 8)) < (unsigned)a.H) & ((unsigned)(xa + (prc[i] & 255)) < (unsigned)a.W) & live;
        unsigned off = base + poff[i];
        asm volatile("" : "+v"(off));                     // pinned: the out-of-image case is a select of the zero page, not a divergent
        unsigned long long src = (unsigned long long)a.x_hi + off;   // branch around the address arithmetic (which would split the scheduling region)
        asm volatile("" : "+v"(src));
        src = ok ? src : zsrc;
        asm volatile("" : "+v"(src));
        dma16((const half_t*)src, dstbuf + (i * 4 + w4) * 1024);
    };

    // ---- LDS read addressing of a B fragment (16 pixels x 32 channels): lane (n, q) reads slot SL(kh, q) = (2kh + (q >> 1)) ^ 4(q & 1)
    // of pixel column 16cb + n + dx; physical slot = slot ^ ((column >> 1) & 3) ----------------------------------------------------
    int rd[2][3][2];                                      // [cb][dx][kh] byte offset inside a patch row
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int col = 16 * cb + n + dx;
            const int z = (col >> 1) & 3;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) rd[cb][dx][kh] = col * 128 + ((((2 * kh + (q >> 1)) ^ (4 * (q & 1))) ^ z) << 4);
        }
    // m write addressing: lane (n, q) holds channels 16w4 + 4q .. +3 of pixel column 16cb + n: slot 2w4 + (q >> 1), byte 8(q & 1)
    int mw[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int col = 16 * cb + n;
        mw[cb] = col * 128 + (((2 * w4 + (q >> 1)) ^ ((col >> 1) & 3)) << 4) + 8 * (q & 1);
    }

    unsigned slope2;
    {
        const h2_t s2 = {(half_t)a.slope, (half_t)a.slope};
        slope2 = __builtin_bit_cast(unsigned, s2);
    }
    const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};

    // global addressing of outputs / residual after the tile pairing: lane (n, q) = pixel column 16(q & 1) + n, channels 16w4 + 8(q >> 1) .. +7
    const int ocol = 16 * (q & 1) + n;
    const unsigned lane_ob = ((unsigned)ocol * 64u + (unsigned)(16 * w4 + 8 * (q >> 1))) * 2u;
    const unsigned trash_ob = (unsigned)a.B * a.H * a.W * 128u + lane * 16u;       // slack behind every activation buffer

    const RowConsts kc = {1.0f, 0.00048828125f, -2048.f};

    // Both convs stream their input rows: the twelve B fragments (dx, kh, cb) of row r+1 are read into the second register set while
    // the (up to) 36 MFMAs of row r run from the first; a sched_group_barrier pattern pins the interleave (one MFMA, at most one
    // LDS read, a couple of VALU of the riding epilogue) -- left alone the compiler reads each fragment right in front of its first use
    // and waits out the LDS latency 264 times per patch.
    half8_t fr[2][12];
    const char* pb[12];           // per-lane base of each of the twelve fragments in the buffer being read; rows are immediate offsets
#define MOE_SET_BASE(PTR)                                                                                \
    _Pragma("unroll") for (int dx = 0; dx < 3; ++dx) _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)    \
        _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) pb[(dx * 2 + kh) * 2 + cb] = (PTR) + rd[cb][dx][kh];
#define MOE_READ_ROW(BUF, ROW)                                                                           \
    _Pragma("unroll") for (int f_ = 0; f_ < 12; ++f_) fr[BUF][f_] = *(const half8_t*)(pb[f_] + (ROW) * (XW * 128));
#define MOE_PIN_ROW(NMFMA, NREAD, NVALU)                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < (NMFMA); ++i_) {                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                               \
        if (i_ < (NREAD)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                             \
        __builtin_amdgcn_sched_group_barrier(0x002, (NVALU), 0);                                         \
    }

    Item it_cur = decode(g);
    {   // prologue: the first patch, and the B fragments of its first row
        constexpr int p = 0;
#pragma unroll
        for (int i = 0; i < NPIECE_W; ++i) issue_piece(it_cur, i, xbuf, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        MOE_SET_BASE(xbuf)
        MOE_READ_ROW(0, 0)
        (void)p;
    }

    // Synchronisation per patch p (one wave per SIMD, four waves):
    //   conv_1 rows 0..2           read x[p] only                                   (its row-0 fragments were read at the end of patch p-1)
    //   barrier A                  every wave has finished conv_2 of patch p-1: the m buffer may be overwritten
    //   conv_1 rows 3..11          m rows written as they complete; DMA of x[p+1] and the residual loads ride along
    //   vmcnt(0), lgkmcnt(0), barrier B   m is complete, x[p+1] has landed for every wave
    //   conv_2 rows 0..9           output rows stored as they complete; then the row-0 fragments of x[p+1] are read
    for (int p = 0; p < K; ++p) {
        const Item it = it_cur;
        const bool has_next = p + 1 < K;
        const Item itn = has_next ? decode(g + (p + 1) * G) : it;
        const char* const xb = xbuf + (p & 1) * XBYTES;
        char* const xn = xbuf + ((p + 1) & 1) * XBYTES;
        const int y0 = it.pyi * TH, x0 = it.pxi * TW;
        ARSB_STAMP(0)

        // residual of the eight output rows (hi / lo part, in the 16-byte store layout): fetched while conv_1 runs, one row per input row,
        // so that every load has landed long before conv_2's first finished row needs it
        uint4 resw[8], sidew[8];
        const unsigned rowb = ((unsigned)(it.b * a.H + y0) * (unsigned)a.W + (unsigned)x0) * 128u;   // byte offset of output (row 0, col 0)
        const bool okx = (ocol < TW) & (x0 + ocol < a.W);
        auto row_off = [&](int o) {
            const bool ok = okx & (y0 + o < a.H);
            unsigned off = ok ? rowb + (unsigned)o * (unsigned)a.W * 128u + lane_ob : trash_ob;
            asm volatile("" : "+v"(off));             // keep the select (no divergent branches around the accesses)
            return off;
        };
        // ================= conv_1: m rows 0 .. 9 (32 columns) from x rows 0 .. 11 ===========================================
        {
            float4_t acc[10][2];
            // epilogue of a finished m row: PReLU on packed halves (slope <= 1), to LDS (the zero padding of conv_2 is applied below)
            auto m_row = [&](const float4_t (&ac)[2], int mr) {
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    unsigned hv[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const h2_t pr = {(half_t)ac[cb][2 * k], (half_t)ac[cb][2 * k + 1]};
                        unsigned u = __builtin_bit_cast(unsigned, pr), t;
                        asm("v_pk_mul_f16 %0, %1, %2" : "=v"(t) : "v"(u), "v"(slope2));
                        asm("v_pk_max_f16 %0, %1, %2" : "=v"(u) : "v"(u), "v"(t));
                        hv[k] = u;
                    }
                    // (inline asm: as a C++ store the compiler orders it behind the LDS-DMA pieces in flight -- both write LDS -- and drains
                    // vmcnt to 0 in front of every m row, i.e. waits out an HBM latency twelve times per patch.  The DMA targets the other
                    // x buffer; the explicit lgkmcnt(0) + barrier below publishes m.)
                    const unsigned long long pk = ((unsigned long long)hv[1] << 32) | hv[0];
                    asm volatile("ds_write_b64 %0, %1" ::"v"(lds0 + (unsigned)(mbuf - smem) + (unsigned)(mr * (MP * 128)) + (unsigned)mw[cb]), "v"(pk));
                }
            };
            MOE_SET_BASE(xb)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int xr = 0; xr < 12; ++xr) {
                if (xr == 3) {
                    ARSB_STAMP(1)
                    __builtin_amdgcn_s_barrier();          // barrier A
                    asm volatile("" ::: "memory");
                    ARSB_STAMP(2)
                }
                // (this row's fragments were read during the previous row: one lgkmcnt(0) the compiler's count pass can see replaces the counted
                // waits it otherwise puts between the MFMAs, conv3x3_sp.hip)
                __builtin_amdgcn_s_waitcnt(0xC07F);
                if (xr < 11 && !(ARSB_ABL & 8)) { MOE_READ_ROW((xr + 1) & 1, xr + 1) }
                if (xr < 11 && (ARSB_ABL & 8)) { _Pragma("unroll") for (int f_ = 0; f_ < 12; ++f_) fr[(xr + 1) & 1][f_] = fr[xr & 1][f_]; }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                            for (int dy = 0; dy < 3; ++dy) {
                                const int mr = xr - dy;
                                if (mr >= 0 && mr < 10) {
                                    const bool first = (dy == 0 && dx == 0 && kh == 0);     // the first product of m row mr = xr
                                    acc[mr][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[(dy * 3 + dx) * 2 + kh], fr[xr & 1][(dx * 2 + kh) * 2 + cb],
                                                                                         first ? zero4 : acc[mr][cb], 0, 0, 0);
                                }
                            }
                        }
                // the DMA of the next patch rides along in the first seven rows (2 + 2 + 2 + 2 + 2 + 2 + 1): early, so that the pieces have
                // landed -- and stopped competing with the m-row writes for the LDS write port -- well before barrier B
                if (!(ARSB_ABL & 1)) {
                    if (xr < 6) { issue_piece(itn, 2 * xr, xn, has_next); issue_piece(itn, 2 * xr + 1, xn, has_next); }
                    if (xr == 6) issue_piece(itn, 12, xn, has_next);
                }
                if (xr >= 3) { if (ARSB_ABL & 2) asm volatile("" ::"v"(acc[xr - 3][0]), "v"(acc[xr - 3][1])); else m_row(acc[xr - 3], xr - 3); }
                if (xr < 8 && !(ARSB_ABL & 4)) {      // (rows 0..7: every load has landed when the vmcnt(0) in front of barrier B is reached)
                    const unsigned off = row_off(xr);
                    resw[xr] = *(const uint4*)((const char*)a.x_hi + off);
                    if (LO) sidew[xr] = *(const uint4*)((const char*)a.x_lo + off);
                }
                MOE_PIN_ROW(((xr < 2 || xr > 9) ? (xr == 0 || xr == 11 ? 12 : 24) : 36), (xr < 11 ? 12 : 0), 2)
                __builtin_amdgcn_sched_barrier(0);
                ARSB_STAMP(3 + xr)
            }
            if (ARSB_ABL & 2) asm volatile("" ::"v"(acc[9][0]), "v"(acc[9][1])); else m_row(acc[9], 9);         // the last row has no MFMAs left to hide behind
        }
        // conv_2 pads with ZEROS: m pixels outside the image must be 0, not conv_1 evaluated there.  Only patches on the image border
        // have such pixels (wave-uniform test), so the fix-up is a separate pass over this wave's own channel slice instead of a
        // select in every m row (in-order LDS: these writes land behind the row writes above)
        if ((y0 == 0) | (y0 + TH >= a.H) | (x0 == 0) | (x0 + TW >= a.W)) {
#pragma unroll
            for (int mr = 0; mr < MH; ++mr)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const bool in = ((unsigned)(y0 - 1 + mr) < (unsigned)a.H) & ((unsigned)(x0 - 1 + 16 * cb + n) < (unsigned)a.W);
                    if (!in) {
                        const unsigned long long z64 = 0;
                        asm volatile("ds_write_b64 %0, %1" ::"v"(lds0 + (unsigned)(mbuf - smem) + (unsigned)(mr * (MP * 128)) + (unsigned)mw[cb]), "v"(z64));
                    }
                }
        }
        ARSB_STAMP(15)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        ARSB_STAMP(16)
        __builtin_amdgcn_s_barrier();                     // barrier B: m is complete, x[p+1] has landed for every wave
        asm volatile("" ::: "memory");
        ARSB_STAMP(17)

        // ================= conv_2: output rows 0 .. 7 (32 columns computed, 30 stored) from m rows 0 .. 9 ===================
        {
            float4_t acc[8][2];
            // epilogue of a finished output row o: + residual (hi [+ lo * 2^-11]) in fp32, hi [and lo] parts stored, 16 bytes per lane
            auto out_row = [&](const float4_t (&ac)[2], int o) {
                float v[2][4];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[cb][e] = ac[cb][e];
                finish_row<LO, true, LO>(v, sidew[o], resw[o], kc, (char*)a.y_hi, (char*)a.y_lo, row_off(o));
            };
            MOE_SET_BASE(mbuf)
            MOE_READ_ROW(0, 0)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int xr = 0; xr < 10; ++xr) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                if (xr < 9 && !(ARSB_ABL & 8)) { MOE_READ_ROW((xr + 1) & 1, xr + 1) }
                if (xr < 9 && (ARSB_ABL & 8)) { _Pragma("unroll") for (int f_ = 0; f_ < 12; ++f_) fr[(xr + 1) & 1][f_] = fr[xr & 1][f_]; }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                            for (int dy = 0; dy < 3; ++dy) {
                                const int o = xr - dy;
                                if (o >= 0 && o < 8) {
                                    const bool first = (dy == 0 && dx == 0 && kh == 0);
                                    acc[o][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w2[(dy * 3 + dx) * 2 + kh], fr[xr & 1][(dx * 2 + kh) * 2 + cb],
                                                                                        first ? zero4 : acc[o][cb], 0, 0, 0);
                                }
                            }
                        }
                if (xr >= 3) { if (ARSB_ABL & 2) asm volatile("" ::"v"(acc[xr - 3][0]), "v"(acc[xr - 3][1])); else out_row(acc[xr - 3], xr - 3); }
                if (xr == 9) {        // row 0 of the NEXT patch's x (landed and published by barrier B): no LDS round trip at the top of conv_1
                    MOE_SET_BASE(xn)
                    MOE_READ_ROW(0, 0)
                }
                MOE_PIN_ROW(((xr < 2 || xr > 7) ? (xr == 0 || xr == 9 ? 12 : 24) : 36), (xr < 9 ? 12 : (xr == 9 ? 12 : 0)), 2)
                __builtin_amdgcn_sched_barrier(0);
                ARSB_STAMP(18 + xr)
            }
            if (ARSB_ABL & 2) asm volatile("" ::"v"(acc[7][0]), "v"(acc[7][1])); else out_row(acc[7], 7);
            ARSB_STAMP(28)
        }
        it_cur = itn;
    }
#undef MOE_READ_ROW
#undef MOE_SET_BASE
#undef MOE_PIN_ROW
}

}  // namespace

hipError_t arsb_fused_init()
{
    hipError_t e = hipFuncSetAttribute((const void*)arsb_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)arsb_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}

// false: the layer does not fit this kernel (caller runs the two convs separately)
bool launch_arsb_fused(ArsbArgs a, int max_groups, hipStream_t s)
{
    if (!(a.slope <= 1.f)) return false;                                  // PReLU as max(x, slope * x)
    if ((long long)a.B * a.H * a.W * 128 >= (1ll << 32) - 65536) return false;   // 32-bit byte offsets
    if ((a.x_lo == nullptr) != (a.y_lo == nullptr)) return false;
    a.px = (a.W + TW - 1) / TW;
    a.py = (a.H + TH - 1) / TH;
    const long long items = (long long)a.B * a.px * a.py;
    const int G = (int)std::min<long long>(items, max_groups);
    if (a.x_lo) arsb_fused_kernel<true><<<dim3(G), dim3(256), LDS_BYTES, s>>>(a);
    else arsb_fused_kernel<false><<<dim3(G), dim3(256), LDS_BYTES, s>>>(a);
    return true;
}
