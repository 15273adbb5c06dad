// arsb32c.hip -- arsb32.hip with VERTICAL CONTINUATION: a workgroup walks a run of vertically adjacent patches and keeps the two last m rows of a
// patch (conv_1's output, python/models.py:76-80 of the reference: y = x + s * conv_2(PReLU(conv_1(x)))) in LDS for the patch below.
//
// arsb32.hip computes 10 m rows for 8 output rows: 10 x 36 + 8 x 36 = 648 MFMAs per 8 rows and channel half (81 per row), of which the two halo rows
// of m are computed twice by vertically adjacent patches (17 % of the trunk's MFMAs are halo recompute).  Here a patch is TEN output rows;
// its m patch has twelve rows of which rows 0, 1 are the rows 10, 11 of the patch above (copied inside LDS), conv_1 computes rows 2..11 (five per
// wave: the very row steps of arsb32.hip) and conv_2 mirrors it (five output rows per wave, seven row steps): 2 x 180 MFMAs per wave for ten
// rows = 72 per row (-11 %), twelve x rows fetched for ten rows instead of twelve for eight (-20 % of the fetched bytes).
//
//   wave (c, h)   output channels 32c .. 32c+31 of BOTH convs; conv_1: m rows 2+5h .. 6+5h from x rows 5h .. 5h+6 (x row q = image row y0 + q),
//                 conv_2: output rows 5h .. 5h+4 from m rows 5h .. 5h+6 (m row r = image row y0 - 1 + r)
//   run           work items are the patches in COLUMN-major order (plane, patch column, patch row); workgroup g takes the contiguous range
//                 [g N / G, (g+1) N / G).  The first patch of a range and every patch at the top of a column is FRESH: its m rows 0, 1 are computed
//                 by a short unpipelined pre-step (wave pair h computes m row h: 36 MFMAs), from the x rows y0-2, y0-1 -- fetched beside the patch
//                 into the (not yet written) m rows 10, 11 at the start of a range, outside the image at a column top -- and the patch's own rows
//                 0, 1.  Same MFMAs in the same order as the streamed form: the result does not depend on where the ranges are cut.
//   fragments     FOURTEEN registers rotate: fragment f of row step t (0..6 conv_1, 7..13 conv_2) lives in fr[(f - t) mod 14]
//   accumulators  m row i -> acc[(i + 1) & 3], output row i -> acc[i & 3]: output rows 3 (slot 1) and 4 are finished inside the next patch's conv_1
//                 (steps 0 .. 2), which writes acc[3] / acc[0] for the first time in its steps 2 / 3
//   everything else (weights resident in AGPR + VGPR, LDS images, raw-buffer DMA / loads / stores, micro-op tables, counted waits) as arsb32.hip.
#include "common.h"
#include "rowtile.h"
#include <type_traits>

// experiment (tools/mk_variant.sh nolo arsb32c.hip -DA32_NOLO): the low-part stream's descriptors cover nothing -- its loads return zero, its stores are dropped, the
// instruction stream is the same: what the launch would take without those bytes through HBM (profiles/r03/o_stream_bytes.txt)
#ifdef A32_NOLO
#define A32_LO_BYTES(n) 0u
#else
#define A32_LO_BYTES(n) (n)
#endif
// round 5, "where do the joules go" (tools/r05_b.sh; results WRONG by design, the instruction stream is the product's): A32_NOST -- the output descriptors cover nothing (every
// store is issued and dropped), A32_NODMA -- the input descriptor covers nothing (the LDS-DMA pieces are issued and deliver zeros: no fetch traffic, and conv_1's B operand stops
// toggling), A32_NOEPI -- the epilogues' arithmetic (PReLU, residual add, hi / lo split) is left out, their LDS writes and stores stay
#ifdef A32_NOST
#define A32_ST_BYTES(n) 0u
#else
#define A32_ST_BYTES(n) (n)
#endif
#ifdef A32_NODMA
#define A32_IN_BYTES(n) 0u
#else
#define A32_IN_BYTES(n) (n)
#endif

namespace {

constexpr int TW = 30, TH = 10;                // stored outputs per patch
constexpr int XW = 34, XH = 12;                // x patch: image rows y0 .. y0+11, columns x0-2 .. x0+31; the m patch shares the pitch
constexpr int MH = 12;                         // m patch: image rows y0-1 .. y0+10, columns x0-1 .. x0+30 (+2 unused)
constexpr int NPIX = XW * XH;                  // 408
constexpr int NDMA_W = 13;                     // 1-KiB pieces per wave: 52 >= 408 / 8
constexpr int XBYTES = NDMA_W * 4 * 1024;      // 53,248
constexpr int ROWB = XW * 128;                 // bytes of a patch row
constexpr int MBYTES = MH * ROWB;              // 52,224
constexpr int LDS_BYTES = 2 * XBYTES + MBYTES + 512;      // 159,232 (+512: the ninth 1-KiB piece of the two extra x rows of a range start ends 4 pixels behind m)

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;

enum OpKind : int { OP_NONE = 0, OP_DMA, OP_P, OP_MW, OP_XLO, OP_XHI, OP_RES, OP_SPL, OP_ST, OP_MCP };
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
// epilogue of slot o (eight channels of the lane's pixel) of output row i (of the wave's five): residual add in fp32 by halves, [hi/lo split by channel
// pairs,] store(s).  The next row's x_hi word takes the register this row's has just released (ONE set of residual registers).
constexpr OpList yslot_ops(int i, int o, bool lo)
{
    OpList r;
    r.push(OP_RES, i, o, 0); r.push(OP_RES, i, o, 2);
    if (i < 4) r.push(OP_XHI, i + 1, o);
    if (lo) { r.push(OP_SPL, i, o, 0); r.push(OP_SPL, i, o, 2); }
    r.push(OP_ST, i, o);
    return r;
}
constexpr OpList yrow_ops(int i, bool lo) { OpList r = yslot_ops(i, 0, lo); r.append(yslot_ops(i, 1, lo)); return r; }      // (slot after slot: they share sh / sl)
// ALL x_lo words of a patch are requested in conv_1's steps 4..6, where no store follows for three row steps.  (A load that misses holds the in-order
// L1 pipeline; the second to fourth store issued behind it then stalls the wave for the rest of the load's latency, 1.3 - 1.9k cycles: seen in the cycle
// trace whenever the words of row 4 were requested next to conv_2's stores, profiles/r03/k_arsb32c_trace.txt.)
// A row step has 12 chunks of 1, 2 or 3 MFMAs; an MFMA hides about five other instructions and an op is dealt to ONE chunk, so ops are kept to ~12
// instructions (a DMA piece = address half + validity/issue half) and the steps with 12 MFMAs get as little as the dependences allow (arsb32.hip).
constexpr OpList dma_ops(int i0, int i1)
{
    OpList r;
    for (int i = i0; i < i1; ++i) { r.push(OP_DMA, i, 0); r.push(OP_DMA, i, 1); }
    return r;
}
constexpr OpList tail_ops(bool lo)             // what the next patch's conv_1 would have carried
{
    OpList r = yslot_ops(3, 1, lo);
    r.append(yslot_ops(4, 0, lo)); r.append(yslot_ops(4, 1, lo));
    return r;
}
constexpr OpList conv1_ops(int s, bool lo)     // row step s of conv_1 (x row 5h + s): 12, 24, 36, 36, 36, 24, 12 MFMAs
{
    OpList r, x;
    if (s == 0) r = yslot_ops(3, 1, lo);
    if (s == 1) {                               // (behind barrier A)  the late output row 4 with its stores; the DMA pieces follow in steps 2, 3: a store issued
        x.push(OP_MCP, 0, 0); x.push(OP_MCP, 0, 1); x.push(OP_MCP, 1, 0); x.push(OP_MCP, 1, 1);      // behind loads in flight waits for them at issue (see above)
        r = interleave(x, yrow_ops(4, lo));
    }
    if (s == 2) r = dma_ops(0, 6);
    if (s == 3) r = interleave(dma_ops(6, 11), mrow_ops(0));
    if (s == 4) { if (lo) { x.push(OP_XLO, 0, 0); x.push(OP_XLO, 0, 1); x.push(OP_XLO, 1, 0); x.push(OP_XLO, 1, 1); } r = interleave(x, mrow_ops(1)); }
    if (s == 5) { if (lo) { x.push(OP_XLO, 2, 0); x.push(OP_XLO, 2, 1); x.push(OP_XLO, 3, 0); x.push(OP_XLO, 3, 1); } r = interleave(x, mrow_ops(2)); }
    if (s == 6) { if (lo) { x.push(OP_XLO, 4, 0); x.push(OP_XLO, 4, 1); } r = interleave(x, mrow_ops(3)); }
    return r;
}
constexpr OpList conv2_ops(int s, bool lo)     // row step s of conv_2 (m row 5h + s): 12, 24, 36, 36, 36, 24, 12 MFMAs
{
    OpList r;
    if (s == 0) r = dma_ops(11, 12);           // (pieces 11, 12 = patch rows 10.4 .. 12: two row steps ahead of the first store)
    if (s == 1) r = dma_ops(12, 13);
    if (s == 2) { r.push(OP_XHI, 0, 0); r.push(OP_XHI, 0, 1); }
    if (s == 3) r = yrow_ops(0, lo);
    if (s == 4) r = yrow_ops(1, lo);
    if (s == 5) r = yrow_ops(2, lo);
    if (s == 6) r = yslot_ops(3, 0, lo);
    return r;
}

// The counted vmcnt waits of the kernel are derived from these tables (one buffer instruction per DMA piece and x_lo word, one or two per ST):
constexpr int count_ops(const OpList& l, int kind, int b = -1) { int n = 0; for (int i = 0; i < l.n; ++i) n += l.op[i].kind == kind && (b < 0 || l.op[i].b == b); return n; }
constexpr int vm_ops(const OpList& l, bool lo) { return count_ops(l, OP_DMA, 1) + count_ops(l, OP_XLO) + count_ops(l, OP_ST) * (lo ? 2 : 1); }
// barrier A (conv_1, step 1): everything up to the last DMA piece issued in conv_2 is complete = all but the stores issued behind it (conv_2's later steps, conv_1's step 0)
constexpr int wait_a(bool lo)
{
    int n = 0, s = 6;
    for (; s >= 0 && count_ops(conv2_ops(s, lo), OP_DMA) == 0; --s) n += vm_ops(conv2_ops(s, lo), lo);
    return n + vm_ops(conv1_ops(0, lo), lo);
}
// barrier B: the DMA pieces issued in conv_1 are complete = all but the x_lo words requested behind them
constexpr int wait_b(bool lo) { int n = 0; for (int s = 4; s < 7; ++s) n += count_ops(conv1_ops(s, lo), OP_XLO); return n; }
// before conv_2's first store (step 3): the x_lo words are complete = all but the DMA pieces issued in conv_2
constexpr int wait_c(bool lo) { int n = 0; for (int s = 0; s < 3; ++s) n += count_ops(conv2_ops(s, lo), OP_DMA, 1); return n; }
static_assert(wait_a(true) == 16 && wait_a(false) == 8 && wait_b(true) == 10 && wait_b(false) == 0 && wait_c(true) == 2, "the s_waitcnt immediates in the kernel");
static_assert(count_ops(conv1_ops(2, true), OP_XLO) + count_ops(conv1_ops(3, true), OP_XLO) == 0 && count_ops(conv1_ops(4, true), OP_DMA) + count_ops(conv1_ops(5, true), OP_DMA) + count_ops(conv1_ops(6, true), OP_DMA) == 0,
              "conv_1: DMA pieces first, x_lo words behind them");
static_assert(vm_ops(conv2_ops(0, true), true) + vm_ops(conv2_ops(1, true), true) + vm_ops(conv2_ops(2, true), true) == wait_c(true), "conv_2, steps 0..2: DMA pieces only");

struct Item { int b, pxi, pyi; };

// cycle-level trace (tools/mk_variant.sh trace32 arsb32c.hip -DA32_TRACE; tools/show_trace_a32.py)
// (stamps of patches 4 and 7 are kept in LDS and written out when the workgroup is done: a global store per stamp would sit among the counted vmcnt waits)
#ifdef A32_TRACE
constexpr int TRACE_LDS = 2 * 4 * 40 * 8;
#define A32_STAMP(SLOT) if (a.trace && g < 8 && (p == 4 || p == 7) && lane == 0) *(unsigned long long*)(smem + LDS_BYTES + (((p == 7) * 4 + w4) * 40 + (SLOT)) * 8) = __builtin_amdgcn_s_memtime();
#else
constexpr int TRACE_LDS = 0;
#define A32_STAMP(SLOT)
#endif

// KS: 16-channel k-slices of the convs' INPUT that carry data -- 4, or 3 for the 48-channel nets (NetDN: channels 48..63 of every activation and
// the weights on them are zero; the fourth slice's fragments, MFMAs and reads are simply left out: -25 % MFMAs, bit-identical results)
// (Round 5 built and measured a third parameter, L8: the stream's low part as the chain's fp8 words in and out, 430 instead of 558 bytes a pixel.  Looped by itself the
// kernel went 0.973 -> 0.939 ms per 96 planes (1.48 -> 1.66 GHz under the power cap); inside the frame the trunk stayed at 5.15 ms, the frame moved 23.73 -> 23.65 ms and the
// error budget paid 0.5e-4 (a4) to 3.8e-4 (a2, single tiles): dropped again -- profiles/r05/c_stream8_ab.txt, source in the history at 0486050.)
template <bool LO, int KS>
__global__ __launch_bounds__(256) void arsb32c_kernel(ArsbArgs a)
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

    // ---- this workgroup's range of the column-major patch sequence ---------------------------------------------------------------------------------------
    const int g = blockIdx.x, G = gridDim.x;
    const long long nitems = (long long)a.B * a.py * a.px;
    const int first = (int)(nitems * g / G);
    const int K = (int)(nitems * (g + 1) / G) - first;
    if (K <= 0) return;
    auto decode = [&](int item) {
        Item it;
        it.pyi = item % a.py;
        const int t = item / a.py;
        it.pxi = t % a.px;
        it.b = t / a.px;
        return it;
    };
    auto advance = [&](const Item& it) {          // the patch below; at the bottom of a column the top of the next column (of the next plane)
        Item n;
        int y = it.pyi + 1;
        const int cy = y >= a.py;
        y = cy ? 0 : y;
        int x = it.pxi + cy;
        const int cx = x >= a.px;
        x = cx ? 0 : x;
        n.pyi = y; n.pxi = x; n.b = it.b + cx;
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

    // ---- patch DMA: the lane's source offset of piece i (its pixel of the 12 x 34 patch, its logical 16-byte slot) is formed when the piece is issued ------
    const unsigned in_pad = (unsigned)(2 * a.W + 2) * 128u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.x_hi - in_pad), 0,
                                                                         A32_IN_BYTES((unsigned)a.B * a.H * a.W * 128u + in_pad), 0x00020000);
    const unsigned nbytes = (unsigned)a.B * a.H * a.W * 128u;
    const __amdgpu_buffer_rsrc_t rlo = __builtin_amdgcn_make_buffer_rsrc((void*)(LO ? a.x_lo : a.x_hi), 0, A32_LO_BYTES(nbytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t ryh = __builtin_amdgcn_make_buffer_rsrc((void*)a.y_hi, 0, A32_ST_BYTES(nbytes), 0x00020000);
    // drop_lo: nobody reads this block's low part (the last ARSB of an SR net: the upsampler takes the fp16 part) -- its stores are issued against an empty range
    const __amdgpu_buffer_rsrc_t ryl = __builtin_amdgcn_make_buffer_rsrc((void*)(LO ? a.y_lo : a.y_hi), 0, a.drop_lo ? 0u : A32_ST_BYTES(A32_LO_BYTES(nbytes)), 0x00020000);
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
    auto piece_off = [&](int npix, int i, int ya, int xa, bool live) {      // ya, xa: image row / column of the piece set's origin (wave-uniform)
        bool ok = ((unsigned)(ya + (int)d_r) < (unsigned)a.H) & ((unsigned)(xa + (int)d_cc) < (unsigned)a.W) & live;
        if (i * 32 + 31 >= npix) ok &= (i * 32 + qlane < npix);
        return ok ? d_off : kOOR;
    };
    auto origin = [&](const Item& it, int dy) {              // byte offset (in rin) of image row y0 + dy, column x0 - 2 of the patch
        return (unsigned)((it.b * a.H + it.pyi * TH + dy) * a.W + it.pxi * TW - 2 + 2 * a.W + 2) * 128u;
    };

    // ---- LDS addressing.  B fragment f = (dx, ks) of a patch row: pixel (row, col) at (row * 34 + col) * 128, 16-B slot s at s ^ ((col >> 1) & 7);
    // lane (j, hh) reads slot 2 ks + hh of column j + dx.  Row offsets are immediates; the buffer / first-row part is added per phase. -------------
    // fa[f]: byte address of fragment f in the first row this wave reads next (x buffer / m, row 5h): shifted by a wave-uniform delta at every phase change
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
        mw[o] = mbase + (unsigned)((2 + 5 * h) * ROWB + j * 128 + ((s ^ ((j >> 1) & 7)) << 4));
        xh[o] = (unsigned)(5 * h * ROWB + (j + 2) * 128 + ((s ^ (((j + 2) >> 1) & 7)) << 4));
    }
    const unsigned lane_ob = (unsigned)(j * 128 + (32 * c + 8 * hh) * 2);      // byte offset of slot o = 0 of output column j inside a patch row of the stream tensors
    unsigned slope2;
    {
        const half2_t s2 = {(half_t)a.slope, (half_t)a.slope};
        slope2 = __builtin_bit_cast(unsigned, s2);
    }
    const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    float16_t acc[4];
    half8_t fr[14];           // fragment f of row step t (0..6 conv_1, 7..13 conv_2) lives in fr[(f - t) mod 14]
    u4_t xlo[5][2], xhv[2];   // residual words in flight: x_lo of the five rows, x_hi of ONE row (slot o)
    unsigned sh[4], sl[4];
    // stream tensors: byte offset of (output row 5h, column x0) of the patch, the lane's column part; rows 3, 4 of the PREVIOUS patch (stored one patch late)
    unsigned so0 = 0, vo = kOOR, so3p = kOOR, so4p = kOOR, vop = kOOR, xprev = lds0;
    int yrow0 = 0;
    {
        const u4_t z4 = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int o = 0; o < 2; ++o) { xhv[o] = z4; xlo[4][o] = z4; xlo[3][o] = z4; }      // (the first patch runs the late epilogues of "the patch before" with their stores rejected)
        acc[0] = zero16; acc[3] = zero16;
    }
    auto row_so = [&](int i) { return (yrow0 + i < a.H) ? so0 + (unsigned)(i * a.W * 128) : kOOR; };

    // ===== prologue: the first patch (and, inside a column, the two x rows above it into m rows 10, 11) ====================================================
    Item it_cur = decode(first);
    {
        const unsigned org = origin(it_cur, 0);
#pragma unroll
        for (int i = 0; i < NDMA_W; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + (i * 4 + w4) * 1024), 16,
                                                     (piece_addr(i), piece_off(NPIX, i, it_cur.pyi * TH, it_cur.pxi * TW - 2, true)), org, 0, 0);
        if (it_cur.pyi > 0) {
            const unsigned org2 = origin(it_cur, -2);
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if ((i * 4 + w4) * 8 < 2 * XW)             // (wave-uniform: nine pieces cover 68 pixels)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + 2 * XBYTES + 10 * ROWB + (i * 4 + w4) * 1024), 16,
                                                             (piece_addr(i), piece_off(2 * XW, i, it_cur.pyi * TH - 2, it_cur.pxi * TW - 2, true)), org2, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int f = 0; f < 12; ++f)
            if ((f & 3) < KS) fr[f] = *(lds_h8_t)(fa[f]);
    }

    for (int p = 0; p < K; ++p) {
        const Item it = it_cur;
        const bool has_next = p + 1 < K;
        const unsigned xcur = lds0 + (unsigned)((p & 1) * XBYTES), xnxt = lds0 + (unsigned)(((p + 1) & 1) * XBYTES);
        const int y0 = it.pyi * TH, x0 = it.pxi * TW;
        const bool fresh = (p == 0) | (it.pyi == 0);          // (workgroup-uniform) m rows 0, 1 are not the rows 10, 11 of the patch before
        A32_STAMP(0)
        so0 = (unsigned)(((it.b * a.H + y0 + 5 * h) * a.W + x0) * 128);
        vo = ((j < TW) & (x0 + j < a.W)) ? lane_ob : kOOR;
        yrow0 = y0 + 5 * h;
        unsigned hp[8];                                       // activated m row being written (packed halves: slot 0 | slot 1)

        // ===== fresh patch: m rows 0, 1 (image rows y0 - 1, y0) from image rows y0-2 .. y0+1; wave pair h computes m row h =========================================
        if (fresh) {
            __builtin_amdgcn_s_barrier();                     // (inside a range: every wave has left conv_2 of the patch before, which read m rows 0, 1)
            asm volatile("" ::: "memory");
            // acc[1] and acc[2] are free here (acc[3] / acc[0] hold the late output rows of the patch before): one as the accumulator, one as four fragments
            float16_t t;
            const half8_t zero8 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int ys = y0 - 2 + h + dy;               // image row of this tap row (wave-uniform); outside the image: zero padding
                const bool valid = (unsigned)ys < (unsigned)a.H;
                const int q = h + dy;                         // 0, 1: the extra rows (in m rows 10, 11); 2, 3: rows 0, 1 of the x patch
                const unsigned rowa = (q < 2 ? mbase + (unsigned)((10 + q) * ROWB) : xcur + (unsigned)((q - 2) * ROWB)) - (xcur + (unsigned)(5 * h * ROWB));
#pragma unroll
                for (int f0 = 0; f0 < 12; f0 += 4) {
                    half8_t b4[4];
#pragma unroll
                    for (int u = 0; u < KS; ++u) {
                        b4[u] = *(lds_h8_t)(fa[f0 + u] + rowa);
                        b4[u] = valid ? b4[u] : zero8;
                    }
#pragma unroll
                    for (int u = 0; u < KS; ++u) {
                        const int f = f0 + u, dx = f >> 2, ks = f & 3;
                        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[(dy * 3 + dx) * 4 + ks], b4[u], (dy == 0 && f == 0) ? zero16 : t, 0, 0, 0);
                    }
                }
            }
            const bool in = ((unsigned)(y0 - 1 + h) < (unsigned)a.H) & ((unsigned)(x0 - 1 + j) < (unsigned)a.W);      // conv_2 pads with ZEROS
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                u4_t d;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const half2_t pr = {(half_t)t[8 * o + 2 * k], (half_t)t[8 * o + 2 * k + 1]};
                    const half2_t ts = pr * __builtin_bit_cast(half2_t, slope2);
                    d[k] = in ? __builtin_bit_cast(unsigned, __builtin_elementwise_max(pr, ts)) : 0u;
                }
                const unsigned ad = mw[o] - (unsigned)((2 + 4 * h) * ROWB);      // m row h
                asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(d) : "memory");
            }
        }

        // (the next work item behind the rare block above, so that its scalar arithmetic falls into row step 0; pinned in SGPRs once: left to the compiler,
        // the carries of advance() and the origin are re-derived in front of every DMA piece)
        const Item itn = advance(it);                         // (beyond the last patch: nothing is fetched)
        const unsigned orgn = (unsigned)__builtin_amdgcn_readfirstlane((int)origin(itn, 0));
        const int yan = __builtin_amdgcn_readfirstlane(itn.pyi * TH), xan = __builtin_amdgcn_readfirstlane(itn.pxi * TW - 2);
        const unsigned dnxt = (unsigned)__builtin_amdgcn_readfirstlane(((p + 1) & 1) * XBYTES + w4 * 1024);

        // ---- micro-ops ---------------------------------------------------------------------------------------------------------------------------
        auto op_dma = [&](auto I_, auto HALF_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value, half = decltype(HALF_)::value;
            if constexpr (half == 0) piece_addr(i);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)((char*)smem + dnxt + i * 4096), 16,
                                                          piece_off(NPIX, i, yan, xan, has_next), orgn, 0, 0);
        };
        auto op_p = [&](auto I_, auto K0_, auto N_) __attribute__((always_inline)) {      // PReLU on packed halves (slope <= 1) of channel pairs k0 .. k0+n-1 of m row i
            constexpr int i = decltype(I_)::value, k0 = decltype(K0_)::value, n = decltype(N_)::value;
#ifdef A32_NOEPI
#pragma unroll
            for (int k = k0; k < k0 + n; ++k) { hp[k] = __builtin_bit_cast(unsigned, acc[(i + 1) & 3][2 * k]); asm volatile("" :: "v"(acc[(i + 1) & 3][2 * k + 1])); }
            return;
#endif
#pragma unroll
            for (int k = k0; k < k0 + n; ++k) {
                const half2_t pr = {(half_t)acc[(i + 1) & 3][2 * k], (half_t)acc[(i + 1) & 3][2 * k + 1]};
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
        // m rows 10, 11 of the patch above become rows 0, 1 of this one: every lane moves the two slots it wrote there (waves h = 1: their rows 3, 4), so
        // the copy is ordered before the lane's own new writes of those rows; waves h = 0 run the same instructions on their own rows 3, 4 -> 0, 1
        // (m rows 5, 6 -> 2, 3: dead data they overwrite themselves in row steps 3, 4).  Behind barrier A: conv_2 of the patch above is over.
        u4_t mc[2];
        auto op_mcp = [&](auto R_, auto W_) __attribute__((always_inline)) {      // W = 0: read the lane's two slots of row 3 + rr, W = 1: write them to row rr (a few chunks
            constexpr int rr = decltype(R_)::value, wr = decltype(W_)::value;     // later: the write then waits for its own read only, not for the fragment reads behind it)
            // (a fresh patch has computed its rows 0, 1 itself: there the rows are copied onto themselves)
            const unsigned hoff = fresh ? (unsigned)(-3 * ROWB) : h ? 7u * ROWB : 0u;
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                if (wr == 0) mc[o] = *(const __attribute__((address_space(3))) u4_t*)(mw[o] + (unsigned)((3 + rr) * ROWB));
                else {
                    const unsigned dst = mw[o] - hoff;
                    const u4_t d = mc[o];
                    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(dst), "v"(d), "n"(rr * ROWB) : "memory");
                }
            }
        };
        auto op_xlo = [&](auto I_, auto O_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value, o = decltype(O_)::value;
            xlo[i][o] = __builtin_amdgcn_raw_buffer_load_b128(rlo, vo + (unsigned)(o * 32), row_so(i), 0);
        };
        auto op_xhi = [&](auto I_, auto O_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value, o = decltype(O_)::value;
            // (row 4, slot 1 follows slot 1 of row 3, which rides in the NEXT patch's conv_1 -- or in the tail of this one: xprev is the buffer of the patch it belongs to)
            xhv[o] = *(const __attribute__((address_space(3))) u4_t*)((i == 4 && o == 1 ? xprev : xcur) + xh[o] + (unsigned)(i * ROWB));
        };
        auto op_res = [&](auto I_, auto O_, auto K0_) __attribute__((always_inline)) {      // acc += x_hi [+ x_lo 2^-11] (in place) for channel pairs k0, k0+1 of slot o
            constexpr int i = decltype(I_)::value, o = decltype(O_)::value, k0 = decltype(K0_)::value;
#ifdef A32_NOEPI
#pragma unroll
            for (int k = k0; k < k0 + 2; ++k) { sh[k] = __builtin_bit_cast(unsigned, acc[i & 3][8 * o + 2 * k]); sl[k] = __builtin_bit_cast(unsigned, acc[i & 3][8 * o + 2 * k + 1]); }
            asm volatile("" :: "v"(xhv[o]), "v"(xlo[i][o]));
            return;
#endif
#pragma unroll
            for (int k = k0; k < k0 + 2; ++k) {
                float v0 = acc[i & 3][8 * o + 2 * k], v1 = acc[i & 3][8 * o + 2 * k + 1];
                v0 = mix_lo(xhv[o][k], 1.0f, v0); v1 = mix_hi(xhv[o][k], 1.0f, v1);
                if (LO) { v0 = mix_lo(xlo[i][o][k], 0.00048828125f, v0); v1 = mix_hi(xlo[i][o][k], 0.00048828125f, v1); }
                acc[i & 3][8 * o + 2 * k] = v0; acc[i & 3][8 * o + 2 * k + 1] = v1;
            }
            if (!LO) {
#pragma unroll
                for (int k = k0; k < k0 + 2; ++k) {
                    const half2_t pr = {(half_t)acc[i & 3][8 * o + 2 * k], (half_t)acc[i & 3][8 * o + 2 * k + 1]};
                    sh[k] = __builtin_bit_cast(unsigned, pr);
                }
            }
        };
        auto op_spl = [&](auto I_, auto O_, auto K0_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value, o = decltype(O_)::value, k0 = decltype(K0_)::value;
#ifdef A32_NOEPI
            return;
#endif
#pragma unroll
            for (int k = k0; k < k0 + 2; ++k) split2(acc[i & 3][8 * o + 2 * k], acc[i & 3][8 * o + 2 * k + 1], -2048.f, sh[k], sl[k]);
        };
        auto op_st = [&](auto I_, auto O_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value, o = decltype(O_)::value;
            constexpr bool late = i == 4 || (i == 3 && o == 1);      // (runs one patch late: offsets of the patch it belongs to)
            const unsigned so = i == 4 ? so4p : late ? so3p : row_so(i);
            const unsigned vv = (late ? vop : vo) + (unsigned)(o * 32);
            const u4_t dh = {sh[0], sh[1], sh[2], sh[3]};
            __builtin_amdgcn_raw_buffer_store_b128(dh, ryh, vv, so, 0);
            if (LO) {
                const u4_t dl = {sl[0], sl[1], sl[2], sl[3]};
                __builtin_amdgcn_raw_buffer_store_b128(dl, ryl, vv, so, 0);
            }
        };
        // the ops [f n / 12, (f + 1) n / 12) of a list: PH 0 conv_1 row step S, 1 conv_2 row step S, 2 / 3 the tails (m row 4; the late output rows; F = 0, all ops)
        auto run_ops = [&](auto PH_, auto S_, auto F_) __attribute__((always_inline)) {
            constexpr int PH = decltype(PH_)::value, S = decltype(S_)::value, F = decltype(F_)::value;
            constexpr OpList L = PH == 0 ? conv1_ops(S, LO) : PH == 1 ? conv2_ops(S, LO) : PH == 2 ? mrow_ops(4) : tail_ops(LO);
            constexpr int NCH = 3 * KS;      // chunks of a row step that carry MFMAs (F: index among them)
            constexpr int lo = PH < 2 ? F * L.n / NCH : 0, hi = PH < 2 ? (F + 1) * L.n / NCH : L.n;
            auto run = [&](auto I_) __attribute__((always_inline)) {
                constexpr int I = decltype(I_)::value;
                if constexpr (I >= lo && I < hi) {
                    constexpr Op o = L.op[I];
                    if constexpr (o.kind == OP_DMA) op_dma(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                    if constexpr (o.kind == OP_P) op_p(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{}, std::integral_constant<int, o.c>{});
                    if constexpr (o.kind == OP_MW) op_mw(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                    if constexpr (o.kind == OP_MCP) op_mcp(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
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

        // ================= conv_1: m rows 2+5h .. 6+5h from x rows 5h .. 5h+6 ==============================================================================
        auto step1 = [&](auto S_) __attribute__((always_inline)) {
            constexpr int s = decltype(S_)::value;
            constexpr int nm = (s <= 4 ? 1 : 0) + ((s >= 1 && s <= 5) ? 1 : 0) + ((s >= 2) ? 1 : 0);      // rows this x row contributes to
            if (s == 1) {       // barrier A: nobody reads m or the old x buffer any more; the last DMA pieces of this patch (issued in the previous conv_2) have landed: behind
                                // them went the stores of output rows 0 .. 3 (16 with the lo stream, else 8)
                if (LO) __builtin_amdgcn_s_waitcnt(0x4F70);      // vmcnt(16)
                else __builtin_amdgcn_s_waitcnt(0x0F78);         // vmcnt(8)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            auto chunk = [&](auto F_) __attribute__((always_inline)) {
                constexpr int f = decltype(F_)::value;
                constexpr int dx = f >> 2, ks = f & 3;
                if constexpr (ks >= KS) return;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int i = s - dy;
                    if (i >= 0 && i < 5)
                        acc[(i + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[(dy * 3 + dx) * 4 + ks], fr[(f + 14 - s) % 14], (dy == 0 && f == 0) ? zero16 : acc[(i + 1) & 3], 0, 0, 0);
                }
#ifdef A32_NOFRAG      // (timing experiment, results WRONG: what the fragment reads of the row steps cost -- every second one skipped, the registers kept live)
                if (s < 6 && (f & 1)) asm volatile("" : "+v"(fr[(f + 13 - s) % 14])); else
#endif
                if (s < 6) fr[(f + 13 - s) % 14] = *(lds_h8_t)(fa[f] + (unsigned)((s + 1) * ROWB));
                run_ops(std::integral_constant<int, 0>{}, S_, std::integral_constant<int, dx * KS + ks>{});
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
        if ((y0 + TH + 1 > a.H) | (x0 == 0) | (x0 + TW + 1 >= a.W)) {
            const u4_t z = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const bool in = ((unsigned)(y0 + 1 + 5 * h + i) < (unsigned)a.H) & ((unsigned)(x0 - 1 + j) < (unsigned)a.W);
                if (!in) {
                    asm volatile("ds_write_b128 %0, %1" ::"v"(mw[0] + (unsigned)(i * ROWB)), "v"(z) : "memory");
                    asm volatile("ds_write_b128 %0, %1" ::"v"(mw[1] + (unsigned)(i * ROWB)), "v"(z) : "memory");
                }
            }
        }
        A32_STAMP(8)
        // the DMA pieces of patch p+1 issued in conv_1 and the late stores of patch p-1 are complete; the ten x_lo loads issued behind them may still fly
        if (LO) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        A32_STAMP(9)
        __builtin_amdgcn_s_barrier();                         // barrier B: m is complete, the pieces of x[p+1] issued in conv_1 have landed for every wave
        asm volatile("" ::: "memory");
        A32_STAMP(10)

        // ================= conv_2: output rows 5h .. 5h+4 from m rows 5h .. 5h+6 ==========================================================================
        {
            const unsigned d12 = mbase - xcur;                // x buffer, row 5h -> m, row 5h
#pragma unroll
            for (int f = 0; f < 12; ++f) fa[f] += d12;
        }
#pragma unroll
        for (int f = 0; f < 12; ++f)
            if ((f & 3) < KS) fr[(f + 14 - 7) % 14] = *(lds_h8_t)(fa[f]);
        auto step2 = [&](auto S_) __attribute__((always_inline)) {
            constexpr int s = decltype(S_)::value;
            constexpr int nm = (s <= 4 ? 1 : 0) + ((s >= 1 && s <= 5) ? 1 : 0) + ((s >= 2) ? 1 : 0);
            if (s == 6) {       // the last row's reads already fetch row 0 of patch p+1's conv_1: m, row 5h -> the other x buffer, row 5h
                const unsigned d21 = xnxt - mbase;
#pragma unroll
                for (int f = 0; f < 12; ++f) fa[f] += d21;
            }
            if (LO && s == 3) __builtin_amdgcn_s_waitcnt(0x0F72);      // vmcnt(2): the x_lo words (issued in conv_1's steps 4..6; only DMA pieces 11, 12 are younger) have
                                                                        // landed BEFORE the first store goes out -- loads and stores share the counter, a later counted
                                                                        // wait would also wait for store acknowledgements
            auto chunk = [&](auto F_) __attribute__((always_inline)) {
                constexpr int f = decltype(F_)::value;
                constexpr int dx = f >> 2, ks = f & 3;
                if constexpr (ks >= KS) return;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int i = s - dy;
                    if (i >= 0 && i < 5)
                        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[(dy * 3 + dx) * 4 + ks], fr[(f + 14 - (7 + s)) % 14], (dy == 0 && f == 0) ? zero16 : acc[i & 3], 0, 0, 0);
                }
#ifdef A32_NOFRAG
                if (s < 6 && (f & 1)) asm volatile("" : "+v"(fr[(f + 13 - (7 + s)) % 14])); else
#endif
                if (s < 6) fr[(f + 13 - (7 + s)) % 14] = *(lds_h8_t)(fa[f] + (unsigned)((s + 1) * ROWB));
                else fr[f] = *(lds_h8_t)(fa[f]);            // row 0 of patch p+1's conv_1 (landed and published by barrier B): (f - 14) mod 14 = f
                run_ops(std::integral_constant<int, 1>{}, S_, std::integral_constant<int, dx * KS + ks>{});
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
        A32_STEP(0) A32_STEP(1) A32_STEP(2) A32_STEP(3) A32_STEP(4) A32_STEP(5) A32_STEP(6)
#undef A32_STEP
        so3p = row_so(3);                                     // slot 1 of output row 3 and row 4 of the wave: their epilogues ride in the next patch's conv_1
        so4p = row_so(4);
        vop = vo;
        xprev = xcur;
        it_cur = itn;
        if (!has_next) run_ops(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    }
#ifdef A32_TRACE
    if (a.trace && g < 8 && K > 8 && lane < 40) {
        a.trace[((g * 16 + 4) * 4 + w4) * 40 + lane] = *(unsigned long long*)(smem + LDS_BYTES + ((0 * 4 + w4) * 40 + lane) * 8);
        a.trace[((g * 16 + 7) * 4 + w4) * 40 + lane] = *(unsigned long long*)(smem + LDS_BYTES + ((1 * 4 + w4) * 40 + lane) * 8);
        if (lane == 0) a.trace[((g * 16 + 8) * 4 + w4) * 40] = a.trace[((g * 16 + 7) * 4 + w4) * 40 + 17] + 1;      // (tools/show_trace_a32.py takes "loop" from the next patch's first stamp)
        if (lane == 0) a.trace[((g * 16 + 5) * 4 + w4) * 40] = a.trace[((g * 16 + 4) * 4 + w4) * 40 + 17] + 1;
    }
#endif
#endif
}

}  // namespace

hipError_t arsb32c_init()
{
    hipError_t e;
    if ((e = hipFuncSetAttribute((const void*)arsb32c_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + TRACE_LDS)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)arsb32c_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + TRACE_LDS)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute((const void*)arsb32c_kernel<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + TRACE_LDS)) != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)arsb32c_kernel<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + TRACE_LDS);
}

// w1 / w2: packed A fragments in the conv3x3_sp / pack_conv order (ConvLayer::w_hi).  false: the layer does not fit this kernel
bool launch_arsb32c(ArsbArgs a, int max_groups, hipStream_t s)
{
    if (!(a.slope <= 1.f)) return false;                                  // PReLU as max(x, slope * x)
    if ((long long)a.B * a.H * a.W * 128 + (2ll * a.W + 2) * 128 >= (1ll << 32) - 65536) return false;   // 32-bit byte offsets
    if ((a.x_lo == nullptr) != (a.y_lo == nullptr)) return false;
    a.px = (a.W + TW - 1) / TW;
    a.py = (a.H + TH - 1) / TH;
    const long long items = (long long)a.B * a.px * a.py;
    if (items >= (1ll << 31) / 256) return false;
    const int G = (int)std::min<long long>(items, max_groups);
    if (a.cin != 0 && a.cin != 48 && a.cin != 64) return false;
    const bool k3 = a.cin == 48;               // the fourth 16-channel k-slice carries zeros only
    if (a.x_lo) { if (k3) arsb32c_kernel<true, 3><<<dim3(G), dim3(256), LDS_BYTES + TRACE_LDS, s>>>(a); else arsb32c_kernel<true, 4><<<dim3(G), dim3(256), LDS_BYTES + TRACE_LDS, s>>>(a); }
    else { if (k3) arsb32c_kernel<false, 3><<<dim3(G), dim3(256), LDS_BYTES + TRACE_LDS, s>>>(a); else arsb32c_kernel<false, 4><<<dim3(G), dim3(256), LDS_BYTES + TRACE_LDS, s>>>(a); }
    return true;
}
