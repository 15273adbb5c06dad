// conv3x3_rw.hip -- the upsampler convolutions (3x3, 64 -> 64 r^2, + bias, PixelShuffle(r), PReLU: python/models.py:29-39 of the reference)
// with the weights RESIDENT IN REGISTERS, second form of the hot kernel (conv3x3_sp.hip is the first and stays the fallback):
//
//   EPI 1   PReLU, pixel shuffle folded into the 16-byte stores
//   EPI 4   = 1 + per-plane channel sums of the stored output (one chunk, r = 1: SEDN's rblock.2, whose sums feed the fused block tail)
//   EPI 3   PReLU + the fused 64 -> 1 tail conv (python/models.py:145-154), "phase-class sums" form: the nine per-tap products of every HR pixel
//           are summed INSIDE the workgroup into the four output parity classes of its pixel-shuffle phase (tests/tailsum_model.py), 4 fp32 values
//           per conv-input pixel and phase instead of 9 tap planes, plus the aprons of the patch border (r = 2 only)
//   EPI 7   = 3 with the tail conv's activation operand split as well (PReLU in fp32, hi + lo 2^-11: MOE_PREC_MIXED, R branch)
//
// What the PMC passes of round 2 said about conv3x3_sp (profiles/r02/i_*): 72 % MFMA busy, 4.0 other instructions per MFMA and 120
// ds_read_b128 per 152 MFMAs and wave -- the four waves keep the CU's one LDS pipe 64 % busy with fragment reads, 72 of the 120 for
// weights.  Here a wave owns 32 of the chunk's 64 output channels and half of the patch rows:
//
//   wave (c, h)   output channels 32c .. 32c+31, output rows 4h .. 4h+3 of the 8 x 32 patch; its 36 A fragments (9 taps x 4 k-slices of
//                 v_mfma_f32_32x32x16_f16) are loaded once per launch and parked in accumulator registers (144);
//   rows stream   input row r of the wave's six (one ds_read_b128 per (dx, k-slice): 12 reads) feeds the up to three output rows it
//                 touches: 72 reads for 144 MFMAs per patch and wave, nothing else is read from LDS in the matrix loop;
//   epilogue      output row o is complete after input row o+2; its epilogue rides in the MFMA stream of the following rows (rows 2, 3: in
//                 the first row steps of the next patch), in pieces of a few instructions per B fragment;
//   fused tail    the 64-channel contraction of the tail conv spans two waves: each forms the per-tap sums of its 32 channels (2 MFMAs per row, 4
//                 with the activation split), folds the low-order weight rows in and writes them into the patch's TAP IMAGE in LDS
//                 ([row 8][channel half 2][tap 9][32 pixels] fp32; the taps that are consumed one column over are stored at the consumer's
//                 column, the one that leaves the patch goes to an export slot).  Two patches later (one barrier per patch publishes the image) wave
//                 (c, h) finishes rows 4h+2c, 4h+2c+1: lane (row, class, pixel quad) adds up to four taps x two halves with 16-byte LDS reads
//                 and stores ONE 16-byte word of the class plane; the aprons (row / column / corner exports) are one more 4-byte pass.
//                 Three exchange buffers rotate (written in patch p / p+1, read in p+2).
//   DMA, barrier  halo'd 10 x 34 patches, double buffered, raw-buffer loads to LDS as in conv3x3_sp; the one barrier per patch sits at
//                 the start of the last row step (its fragments are in registers), behind it the first fragments of the next patch
//                 are read while that step's MFMAs run.
//
// LDS: 2 x 45,056 (patches) + 3 x 19,984 (tap images) = 150,064 B.
#include "common.h"
#include "rowtile.h"
#include <type_traits>

#ifndef RW_HOIST
#define RW_HOIST 1        // fused tail: the bookkeeping of patch p+2 (item, origin, border key) as a micro-op of row step 4 (0: at the end of the iteration)
#endif
#ifndef RW_FILL
#define RW_FILL 5         // VALU / SALU slots pinned behind each MFMA of a chunk
#endif

namespace {

constexpr int PW = kTileW + 2, PH = kTileH + 2, NPIX = PW * PH;   // 34 x 10 halo'd patch
constexpr int NDMA_W = 11;                                         // 1-KiB pieces per wave (44 >= 340 / 8)
constexpr int PATCH_BYTES = NDMA_W * 4 * 1024;                      // 45,056
constexpr int XREC = 9 * 128 + 16;                                 // tap image record of one (row, channel half): [tap 9][pixel slot 32] fp32 + export slots [dy 3] + pad
constexpr int XEXP = 9 * 128;                                      // byte offset of the export slots inside a record
constexpr int XZERO = 16 * XREC;                                   // byte offset of the zero area inside an exchange buffer (two XREC-spaced 128-byte rows)
constexpr int XCH_BYTES = 16 * XREC + XREC + 128;                  // 19,984
constexpr int LDS_BYTES = 2 * PATCH_BYTES + 3 * XCH_BYTES;         // 150,064

// ---- fused tail: the epilogue as micro-ops ------------------------------------------------------------------------------------------
// What rides in the MFMA stream of a patch (accumulator read-out, PReLU [+ hi/lo split], the tail GEMM and its writes into the tap image, the
// finalize of patch p-2, the DMA of patch p+1) is cut into micro-ops of a few instructions; row step R runs the list tail_ops(R) in order,
// chunk f the slice [f N / 12, (f + 1) N / 12) of it.  A lump of 30-40 VALU instructions in one chunk leaves the MFMA pipe idle for as many
// issue slots (the first version of this epilogue: 18 such gaps per patch, rw<7> 10 % slower than conv3x3_sp<7>); beside an MFMA only ~5 fit.
enum OpKind : int { OP_NONE = 0, OP_DMA, OP_RD, OP_P, OP_TM, OP_TW, OP_FF, OP_FA, OP_FS, OP_AF, OP_AS, OP_ADV, OP_ST };
struct Op { int kind, a, b, c; };
struct OpList {
    int n = 0;
    Op op[40] = {};
    constexpr void push(int kind, int a = 0, int b = 0, int c = 0) { op[n] = Op{kind, a, b, c}; ++n; }
    constexpr void append(const OpList& o) { for (int i = 0; i < o.n; ++i) { op[n] = o.op[i]; ++n; } }
};
// both lists in order, proportionally interleaved (Bresenham)
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
// PReLU (+ split) of one output row: EPI 7 one channel pair per op (9 instructions), EPI 3 two pairs (6)
constexpr OpList act_ops(int row, bool split)
{
    OpList r;
    if (split) for (int k = 0; k < 8; ++k) r.push(OP_P, row, k, 1);
    else for (int k = 0; k < 8; k += 2) r.push(OP_P, row, k, 2);
    return r;
}
constexpr OpList rd_ops(int row) { OpList r; for (int q = 0; q < 4; ++q) r.push(OP_RD, row, q); return r; }
constexpr OpList tw_ops(int row) { OpList r; for (int k = 0; k < 5; ++k) r.push(OP_TW, row, k); return r; }
// PReLU-only epilogues (EPI 1 / 4): read-out, PReLU on packed halves (two channel pairs per op), the row's two 16-byte stores
constexpr OpList plain_ops(int R)
{
    OpList r, x;
    if (R == 0) {
        for (int i = 0; i < 4; ++i) x.push(OP_DMA, i);
        r = interleave(x, rd_ops(3));
    } else if (R == 1) {
        for (int i = 4; i < 8; ++i) x.push(OP_DMA, i);
        r = interleave(x, act_ops(2, false));
        r.push(OP_ST, 2);
    } else if (R == 2) {
        for (int i = 8; i < 11; ++i) x.push(OP_DMA, i);
        r = interleave(x, act_ops(3, false));
        r.push(OP_ST, 3);
    } else if (R == 3) {
        r = rd_ops(0);
        r.append(act_ops(0, false));
        r.push(OP_ST, 0);
    } else if (R == 4) {
        r = rd_ops(1);
        r.append(act_ops(1, false));
        if (RW_HOIST) r.push(OP_ADV);
    } else {                 // behind the barrier: a global store issued shortly before the barrier's vmcnt(0) would make it wait for the acknowledgement
        r.push(OP_ST, 1);
        r.append(rd_ops(2));
    }
    return r;
}
// Row o of a patch is complete after row step o + 2.  Rows 2 and 3 are finished inside the NEXT patch (row 3 stays in its accumulators
// meanwhile); the tap image of patch p-2 is finalized in steps 0 / 2 / 3; every row's image record is written long before the barrier
// (step 5 of the next patch) that publishes it.
template <bool SPLIT>
constexpr OpList tail_ops(int R)
{
    OpList r;
    if (R == 0) {            // 12 MFMAs: little room -- DMA, read-out of row 3 of the previous patch, apron reads of patch p-2
        OpList x, y;
        for (int i = 0; i < 4; ++i) x.push(OP_DMA, i);
        y = rd_ops(3);
        r = interleave(x, y);
        r.push(OP_AF);
    } else if (R == 1) {     // 24 MFMAs: DMA, PReLU of row 2 (previous patch) and its tail GEMM
        OpList x;
        for (int i = 4; i < 8; ++i) x.push(OP_DMA, i);
        r = interleave(x, act_ops(2, SPLIT));
        r.push(OP_TM, 2);
    } else if (R == 2) {     // 36 MFMAs: row 2's image writes, the class sums of patch p-2, PReLU + tail GEMM of row 3
        OpList x;
        x.push(OP_DMA, 8); x.push(OP_FF, 0); x.push(OP_TW, 2, 0); x.push(OP_FF, 1); x.push(OP_TW, 2, 1); x.push(OP_FA, 0); x.push(OP_DMA, 9);
        x.push(OP_TW, 2, 2); x.push(OP_FF, 2); x.push(OP_FA, 1); x.push(OP_TW, 2, 3); x.push(OP_FF, 3); x.push(OP_FA, 2); x.push(OP_TW, 2, 4);
        x.push(OP_DMA, 10); x.push(OP_FA, 3);
        r = interleave(x, act_ops(3, SPLIT));
        r.push(OP_TM, 3);
    } else if (R == 3) {     // 36 MFMAs: row 3's image writes, stores of patch p-2, row 0 of this patch from read-out to image
        OpList x;
        x = interleave(tw_ops(3), rd_ops(0));
        x.push(OP_FS);
        OpList y = act_ops(0, SPLIT);
        OpList z;
        z.push(OP_AS);
        r = x;
        r.append(interleave(y, z));
        r.push(OP_TM, 0);
        r.append(tw_ops(0));
    } else if (R == 4) {     // 24 MFMAs: row 1: read-out, PReLU, tail GEMM
        r = rd_ops(1);
        OpList z;
        if (RW_HOIST) z.push(OP_ADV);
        r.append(interleave(act_ops(1, SPLIT), z));
        r.push(OP_TM, 1);
    } else {                 // 12 MFMAs behind the barrier: row 1's image writes, read-out of row 2
        r = interleave(tw_ops(1), rd_ops(2));
    }
    return r;
}

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;

struct Item { int b, pyi, pxi; };

// MASK (fused tail only): the image is ragged against the 8 x 32 patches (H % 8 or W % 32 != 0), so the products of patch pixels outside the
// image must be zeroed before they enter the tap image; for patch-aligned images (every launch of the 1080p x4 frame) the selects are compiled out
template <int EPI, bool MASK = true>
__global__ __launch_bounds__(256) void conv3x3_rw_kernel(ConvArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass only needs the launch stub)
    constexpr bool TAIL = EPI == 3 || EPI == 7, TAIL2 = EPI == 7, POOL = EPI == 4;
    constexpr bool PERM = !TAIL;         // EPI 1: channel order that makes registers 8g..8g+7 eight consecutive channels (16-byte stores)
    constexpr unsigned kOOR = 0xFFFF0000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = w4 & 1, h = w4 >> 1;
    const int j = lane & 31, hh = lane >> 5;

    // ---- workgroup -> (chunk, group): as conv3x3_sp (the nchunks workgroups that share an input patch sit on one XCD) ----------------
    const int bid = blockIdx.x;
    const int chunk = (bid >> 3) % a.nchunks;
    const int g = (bid & 7) + 8 * (bid / (8 * a.nchunks));
    const int G = a.G;
    if (g >= G) return;
    const int nitems = a.B * a.py * a.px;
    // EPI 4 (pooled sums): the work items are PATCH ROWS -- workgroup g walks the patches of row g, g + G, ... of the (plane, patch row) sequence from left to right -- so that
    // the plane changes once per row instead of once per patch: the flush of the per-lane sums (16 values x five cross-lane steps + the slab stores) ran after EVERY patch
    // with the strided patch order below (one patch per plane and workgroup at 256 patches a plane: 158 against 121 us for SEDN's rblock.2, round 5).  A slab still holds
    // a fixed set of patches summed in a fixed order whatever the launch's plane count (the engine picks G as a multiple or divisor of the rows per plane: pooled_groups).
    const int nrows = a.B * a.py;
    const int K = POOL ? ((nrows - g + G - 1) / G) * a.px : (nitems - g + G - 1) / G;
    if (K <= 0) return;
    const int Gx = POOL ? 1 : G % a.px, Gy = POOL ? G % a.py : (G / a.px) % a.py, Gb = POOL ? G / a.py : G / (a.px * a.py);
    auto advance = [&](const Item& it) {
        Item n;
        int x = it.pxi + Gx;
        const int cx = x >= a.px;
        x -= cx ? a.px : 0;
        int y = it.pyi + (POOL ? (cx ? Gy : 0) : Gy + cx);
        const int cy = y >= a.py;
        y -= cy ? a.py : 0;
        n.pxi = x; n.pyi = y; n.b = it.b + (POOL ? (cx ? Gb : 0) : Gb) + cy;
        return n;
    };

    // ---- weights: 36 A fragments (tap, k-slice) of this wave's 32 output channels, conv3x3_sp / pack_conv fragment order ----------------
    half8_t wf[36];
    {
        int src = lane;
        if (PERM) {      // MFMA row i = 8q + 4h' + e (register 4q + e of the lanes hh = h') is given channel 16 (q >> 1) + 8 h' + 4 (q & 1) + e
            const int wi = lane & 31, wq = wi >> 3;
            src = (lane & 32) | (16 * (wq >> 1) + 8 * ((wi >> 2) & 1) + 4 * (wq & 1) + (wi & 3));
        }
        const half_t* wsrc = a.wpk + (long long)chunk * (72 * 512);
#pragma unroll
        for (int f = 0; f < 36; ++f) wf[f] = *(const half8_t*)(wsrc + ((f * 2 + c) * 64 + src) * 8);
#pragma unroll
        for (int f = 0; f < 36; ++f) asm volatile("" : "+a"(wf[f]));
    }
    // bias of this lane's 16 channels: the C operand of the first MFMA of every output row
    float16_t biasv;
    {
        const float* bsrc = a.bias_img + chunk * 256 + 32 * c;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) biasv[4 * q + e] = bsrc[PERM ? 16 * (q >> 1) + 8 * hh + 4 * (q & 1) + e : 8 * q + 4 * hh + e];
    }

    // ---- patch DMA (see conv3x3_sp.hip): lane offsets rebuilt only when the border pattern changes, patch origin in an SGPR -------------
    const unsigned in_px = (unsigned)a.in_cs * 2u;
    const unsigned in_pad = (unsigned)(a.W + 1) * in_px;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in - in_pad), 0,
                                                                         (unsigned)a.B * a.H * a.W * in_px + in_pad, 0x00020000);
    unsigned vofs[NDMA_W];
    int vkey = -2;
    auto patch_key = [&](const Item& it, bool live) {      // how the halo'd patch cuts the image border (-1: nothing to fetch)
        const int y0 = it.pyi * kTileH - 1, x0 = it.pxi * kTileW - 1;
        const int xhi = max(0, x0 + PW - a.W), yhi = max(0, y0 + PH - a.H);
        const int k = (xhi * 16 + yhi) * 4 + (x0 < 0) * 2 + (y0 < 0);
        return k | -(int)!live;          // -1 when there is nothing to fetch (as arithmetic: a branch here would cut the pipelined row step in two)
    };
    auto prep_patch_k = [&](const Item& it, int key) {     // rebuild the lane offsets when the border pattern differs from the previous patch's
        if (__builtin_expect(key != vkey, 0)) {
            asm volatile("" ::: "memory");                              // (not speculatable: keeps the rebuild behind a real, rarely taken branch)
            const int y0 = it.pyi * kTileH - 1, x0 = it.pxi * kTileW - 1;
            const bool live = key >= 0;
            vkey = key;
#pragma unroll
            for (int i = 0; i < NDMA_W; ++i) {
                const int q = (i * 4 + w4) * 8 + (lane >> 3);
                const int r = (q * 241) >> 13;                        // q / 34 for q < 352
                const int cc = q - r * PW;
                const int sl = (lane & 7) ^ ((cc >> 1) & 7);          // logical 16-B slot behind this physical slot
                const bool ok = ((unsigned)(y0 + r) < (unsigned)a.H) & ((unsigned)(x0 + cc) < (unsigned)a.W) & (q < NPIX) & live;
                vofs[i] = ok ? (unsigned)(r * a.W + cc) * in_px + (unsigned)sl * 16u : kOOR;
            }
        }
    };
    auto prep_patch = [&](const Item& it, bool live) { prep_patch_k(it, patch_key(it, live)); };
    auto origin = [&](const Item& it) {
        return (unsigned)((it.b * a.H + it.pyi * kTileH - 1) * a.W + it.pxi * kTileW - 1 + a.W + 1) * in_px;
    };

    // ---- LDS read addressing of a B fragment f = (dx, ks): pixel (row, col) at (row * 34 + col) * 128, 16-B slot s at s ^ ((col >> 1) & 7);
    // lane (j, hh) reads slot 2 ks + hh of column j + dx; rows are immediate offsets.  fa[0]: patch buffer 0, fa[1]: buffer 1
    unsigned fa[2][12];
#pragma unroll
    for (int f = 0; f < 12; ++f) {
        const int dx = f >> 2, ks = f & 3;
        const int cc = j + dx, z = (cc >> 1) & 7;
        const unsigned o = (unsigned)((4 * h * PW + cc) * 128 + (((2 * ks + hh) ^ z) << 4));
        fa[0][f] = lds0 + o;
        fa[1][f] = lds0 + (unsigned)PATCH_BYTES + o;
        asm volatile("" : "+v"(fa[0][f]), "+v"(fa[1][f]));
    }
    const unsigned xch0 = lds0 + 2u * PATCH_BYTES;                  // exchange area
    unsigned xrot[3] = {xch0, xch0 + 2u * XCH_BYTES, xch0 + 1u * XCH_BYTES};     // buffers of patch p, p-1, p-2: (p % 3, (p + 2) % 3, (p + 1) % 3) at p = 0

    // ---- output addressing -----------------------------------------------------------------------------------------------------------
    const int r = a.r;
    const int si = (r > 1) ? chunk / r : 0, sj = (r > 1) ? chunk % r : 0;
    const int cout0 = (r > 1) ? 0 : chunk * kCB;
    const unsigned Wo = (unsigned)(a.W * r), Ho = (unsigned)(a.H * r);
    const unsigned out_px = (unsigned)a.out_cs * 2u;
    const TailSumLayout tl = tailsum_layout(a.B, a.H, a.W);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(TAIL ? (void*)a.tplanes : (void*)a.out, 0,
                                                                          TAIL ? tl.total * 4u : (unsigned)a.B * Ho * Wo * out_px, 0x00020000);
    unsigned slope2;
    {
        const half2_t s2 = {(half_t)a.slope, (half_t)a.slope};
        slope2 = __builtin_bit_cast(unsigned, s2);
    }
    // fused tail: A fragments of the 64 -> 1 conv for this wave's two 16-channel k-slices (rows = taps, rows 16..24 their rounding remainders);
    // tailw2 (EPI 7): the fp16 tail weights again, in rows 16..24, for the activations' low parts (engine.cpp, tail())
    half8_t tailw[2], tailw2[2];
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
        tailw[gp] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        tailw2[gp] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        if (TAIL) tailw[gp] = *(const half8_t*)(a.tail_w + (2 * c + gp) * 512 + lane * 8);
        if (TAIL2) tailw2[gp] = *(const half8_t*)(a.tail_w + (4 + 2 * c + gp) * 512 + lane * 8);
    }
    // ---- fused tail, phase-class sums (tests/tailsum_model.py): per-lane LDS addresses and store offsets, built once ------------------------
    // phase (si, sj): the taps of tap row dyn / column dxn stay on the pixel's own conv-input row / column, those of dyf / dxf are consumed one
    // row (sv) / column (sh) over
    const int sv = si == 0 ? 1 : -1, sh = sj == 0 ? 1 : -1;
    const int dyn = si == 0 ? 0 : 2, dyf = 2 - dyn, dxn = sj == 0 ? 0 : 2, dxf = 2 - dxn;
    const int r_exp = sv == 1 ? 0 : kTileH - 1, c_exp = sh == 1 ? 0 : kTileW - 1;       // the row / column whose far taps leave the patch
    unsigned wa[5];          // tail_partial: byte offset (inside an exchange buffer, for this wave's first row) of the slot each of the lane's five products goes to
    unsigned ta[4];          // finalize: byte offset of the (up to) four taps this lane adds, channel half 0 (half 1: + XREC)
    unsigned aa[2];          // apron pass: the two 4-byte terms of this lane
    unsigned fs_lane = kOOR, ap_lane = kOOR;     // lane parts of the class-plane / apron store offsets (bytes)
    const int f_row = lane >> 5, f_cls = (lane >> 3) & 3, f_q = lane & 7;               // finalize: (row of the wave's pair, class, pixel quad)
    const int f_rho = 4 * h + 2 * c + f_row;
#pragma unroll
    for (int k = 0; k < 5; ++k) wa[k] = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) ta[k] = 0;
    aa[0] = aa[1] = 0;
    if (TAIL) {
        const unsigned row0 = (unsigned)((4 * h * 2 + c) * XREC);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int t = k < 4 ? 4 * hh + k : 8;
            const int dy = t / 3, dx = t - 3 * dy;
            unsigned o = (unsigned)(t * 128 + j * 4);
            if (dx == dxf) o = j == c_exp ? (unsigned)(XEXP + dy * 4) : (unsigned)(t * 128 + (j - sh) * 4);
            if (k == 4 && hh == 1) o = (unsigned)(XEXP + 12);                              // (no tap 12: the record's pad word)
            wa[k] = row0 + o;
        }
        const int ci = f_cls >> 1, cj = f_cls & 1;
        auto tap_at = [&](int dy, int dx, int dr, bool used) {
            const int rr = f_rho + dr;
            return (used && rr >= 0 && rr < kTileH) ? (unsigned)(rr * 2 * XREC + (dy * 3 + dx) * 128 + f_q * 16) : (unsigned)XZERO;
        };
        ta[0] = tap_at(ci ? dyn : 1, cj ? dxn : 1, 0, true);
        ta[1] = tap_at(ci ? dyn : 1, dxf, 0, cj != 0);
        ta[2] = tap_at(dyf, cj ? dxn : 1, sv, ci != 0);
        ta[3] = tap_at(dyf, dxf, sv, ci != 0 && cj != 0);
        fs_lane = ((unsigned)f_cls * (unsigned)(a.B * a.H * a.W) + (unsigned)(f_row * a.W + 4 * f_q)) * 4u;
        // apron pass: wave 0 the row apron (lane = class column cj, pixel), wave 1 the column apron (lane = class row ci, row), wave 2 the corner
        aa[0] = aa[1] = (unsigned)XZERO;
        if (w4 == 0) {
            const int cjj = lane >> 5;
            aa[0] = (unsigned)(r_exp * 2 * XREC + (dyf * 3 + (cjj ? dxn : 1)) * 128 + j * 4);
            if (cjj) aa[1] = (unsigned)(r_exp * 2 * XREC + (dyf * 3 + dxf) * 128 + j * 4);
            ap_lane = ((unsigned)cjj * (unsigned)(a.B * tl.py * a.W) + (unsigned)j) * 4u;
        } else if (w4 == 1) {
            if (lane < 16) {
                const int cii = lane >> 3, rho = lane & 7;
                aa[0] = (unsigned)(rho * 2 * XREC + XEXP + (cii ? dyn : 1) * 4);
                if (cii && rho + sv >= 0 && rho + sv < kTileH) aa[1] = (unsigned)((rho + sv) * 2 * XREC + XEXP + dyf * 4);
                ap_lane = ((unsigned)cii * (unsigned)(a.B * a.H * tl.px) + (unsigned)(rho * tl.px)) * 4u;
            }
        } else if (w4 == 2) {
            if (lane == 0) { aa[0] = (unsigned)(r_exp * 2 * XREC + XEXP + dyf * 4); ap_lane = 0; }
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) asm volatile("" : "+v"(wa[k]));
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(ta[k]));
        asm volatile("" : "+v"(aa[0]), "+v"(aa[1]), "+v"(fs_lane), "+v"(ap_lane));
    }
    // wave-uniform parts of the store offsets (bytes): class planes S[ph][cls][b][y][x] (this wave's row pair), and the apron array of this wave --
    // wave 0: RA[ph][cj][b][pyi][x], wave 1: CA[ph][ci][b][y][pxi], wave 2: CO[ph][b][pyi][pxi] (common.h: TailSumLayout) -- as base + b sb + pyi sy + pxi sx
    const unsigned ph = (unsigned)(si * r + sj);
    const unsigned fs_base = (tl.S + ph * 4u * (unsigned)(a.B * a.H * a.W) + (unsigned)((4 * h + 2 * c) * a.W)) * 4u;
    const unsigned fs_sb = (unsigned)(a.H * a.W) * 4u, fs_sy = (unsigned)(kTileH * a.W) * 4u;
    unsigned ap_base = 0, ap_sb = 0, ap_sy = 0, ap_sx = 0;
    int ap_vx = 0, ap_vy = 0, ap_lim = 1, ap_lv = 0;     // validity of the lane's element: pxi vx + pyi vy + lv < lim
    if (w4 == 0) {
        ap_base = (tl.RA + ph * 2u * (unsigned)(a.B * tl.py * a.W)) * 4u; ap_sb = (unsigned)(tl.py * a.W) * 4u; ap_sy = (unsigned)a.W * 4u; ap_sx = kTileW * 4u;
        ap_vx = kTileW; ap_lim = a.W; ap_lv = j;
    } else if (w4 == 1) {
        ap_base = (tl.CA + ph * 2u * (unsigned)(a.B * a.H * tl.px)) * 4u; ap_sb = (unsigned)(a.H * tl.px) * 4u; ap_sy = (unsigned)(kTileH * tl.px) * 4u; ap_sx = 4u;
        ap_vy = kTileH; ap_lim = a.H; ap_lv = lane & 7;
    } else if (w4 == 2) {
        ap_base = (tl.CO + ph * (unsigned)(a.B * tl.py * tl.px)) * 4u; ap_sb = (unsigned)(tl.py * tl.px) * 4u; ap_sy = (unsigned)tl.px * 4u; ap_sx = 4u;
    }

    const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float16_t acc[5];                      // output rows 0..2 of the patch in slots 0..2, row 3 in slot 3 (even patches) / 4 (odd): it is read out during the next patch
    half8_t fr[2][12];
    float v2[16], v3[16];                  // output rows 2 and 3 of the previous patch, read out of their accumulators late in that patch
#pragma unroll
    for (int e = 0; e < 16; ++e) { v2[e] = 0.f; v3[e] = 0.f; }

    // ===== epilogue pieces ==============================================================================================================
    // PReLU on packed halves (slope <= 1): values v[8 gp .. 8 gp + 7] -> four half2 registers
    auto act8 = [&](const float* v, unsigned (&hv)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const half2_t pr = {(half_t)v[2 * k], (half_t)v[2 * k + 1]};
            const half2_t t = pr * __builtin_bit_cast(half2_t, slope2);
            hv[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(pr, t));
        }
    };
    // EPI 7: PReLU in fp32 (slope <= 1), then hi + lo 2^-11 (rowtile.h: split2) -- the tail conv then sees its activations to ~22 bits
    auto act8s = [&](const float* v, unsigned (&hv)[4], unsigned (&lv)[4]) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = __builtin_fmaxf(v[e], v[e] * a.slope);
#pragma unroll
        for (int k = 0; k < 4; ++k) split2(t[2 * k], t[2 * k + 1], -2048.f, hv[k], lv[k]);
    };
    // EPI 4: this lane's sums of the 16 channels it stores (the fp16 values, as a second pass over the tensor would read them), for the plane
    // `pool_b`; when a row of another plane arrives (and at the end) the 32 pixel lanes are added up and lane j = 0 stores the slab
    float psum[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) psum[e] = 0.f;
    int pool_b = -1;
    auto pool_flush = [&]() {
        if (pool_b >= 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float t = psum[e];
                t += __shfl_xor(t, 16); t += __shfl_xor(t, 8); t += __shfl_xor(t, 4); t += __shfl_xor(t, 2); t += __shfl_xor(t, 1);
                psum[e] = t;
            }
            if (j == 0) {
                // slab = the plane-relative patch row(s) this workgroup sums, g % py: the same place in every plane whatever the launch's plane count, so that the consumer's
                // fixed-order sum over the slabs associates the same way (G is a multiple or a divisor of py: pooled_groups)
                float* dst = a.pool + ((long long)pool_b * a.pool_slabs + 2 * (g % a.py) + h) * 64 + 32 * c + 8 * hh;
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    *(float4_t*)(dst + 16 * gp) = float4_t{psum[8 * gp], psum[8 * gp + 1], psum[8 * gp + 2], psum[8 * gp + 3]};
                    *(float4_t*)(dst + 16 * gp + 4) = float4_t{psum[8 * gp + 4], psum[8 * gp + 5], psum[8 * gp + 6], psum[8 * gp + 7]};
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) psum[e] = 0.f;
    };
    // EPI 1: the 16-byte stores of output row `orow` (0..7 in the patch) of tile `it`: channels cout0 + 32c + 16 gp + 8 hh .. +7 of pixel j
    auto store_row = [&](const Item& it, int orow, bool live, const unsigned (&h0)[4], const unsigned (&h1)[4]) {
        const int y = it.pyi * kTileH + orow, x = it.pxi * kTileW + j;
        if (POOL) {
            if (live && it.b != pool_b) { pool_flush(); pool_b = it.b; }       // (wave-uniform, once per plane)
            const float m = (live & (y < a.H) & (x < a.W)) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                psum[2 * k] = mix_lo(h0[k], m, psum[2 * k]); psum[2 * k + 1] = mix_hi(h0[k], m, psum[2 * k + 1]);
                psum[8 + 2 * k] = mix_lo(h1[k], m, psum[8 + 2 * k]); psum[8 + 2 * k + 1] = mix_hi(h1[k], m, psum[8 + 2 * k + 1]);
            }
        }
        const unsigned vo = (x < a.W) ? (unsigned)(j * r) * out_px + (unsigned)hh * 16u : kOOR;
        const unsigned so = (live & (y < a.H)) ? ((unsigned)(it.b * (int)Ho + y * r + si) * Wo + (unsigned)(it.pxi * kTileW * r + sj)) * out_px + (unsigned)(cout0 + 32 * c) * 2u : kOOR;
        const u4_t d0 = {h0[0], h0[1], h0[2], h0[3]}, d1 = {h1[0], h1[1], h1[2], h1[3]};
        __builtin_amdgcn_raw_buffer_store_b128(d0, rout, vo, so, 0);
        __builtin_amdgcn_raw_buffer_store_b128(d1, rout, vo + 32u, so, 0);
    };
    // EPI 3 / 7: per-tap sums of one row over this wave's 32 channels -> the patch's tap image (exchange buffer xb), row 4h + RC of the patch.
    // `ok`: this lane's pixel of that row lies inside the image (the conv also runs on the padding of ragged patches; those pixels do not
    // exist for the tail conv).  Five 4-byte LDS stores at the lane's precomputed slots (inline asm: a C++ store to LDS is ordered behind the
    // LDS-DMA in flight with vmcnt(0), arsb_fused.hip cost 1).
    unsigned lA[4], lB[4];                             // EPI 7: low parts of the row being finished
    auto tail_partial = [&](const unsigned (&h0)[4], const unsigned (&h1)[4], unsigned xbase, auto RC_, bool ok) __attribute__((always_inline)) {
        constexpr int RC = decltype(RC_)::value;
        const half8_t b0 = __builtin_bit_cast(half8_t, u4_t{h0[0], h0[1], h0[2], h0[3]});
        const half8_t b1 = __builtin_bit_cast(half8_t, u4_t{h1[0], h1[1], h1[2], h1[3]});
        float16_t G = __builtin_amdgcn_mfma_f32_32x32x16_f16(tailw[0], b0, zero16, 0, 0, 0);
        G = __builtin_amdgcn_mfma_f32_32x32x16_f16(tailw[1], b1, G, 0, 0, 0);
        if (TAIL2) {
            const half8_t l0 = __builtin_bit_cast(half8_t, u4_t{lA[0], lA[1], lA[2], lA[3]});
            const half8_t l1 = __builtin_bit_cast(half8_t, u4_t{lB[0], lB[1], lB[2], lB[3]});
            G = __builtin_amdgcn_mfma_f32_32x32x16_f16(tailw2[0], l0, G, 0, 0, 0);
            G = __builtin_amdgcn_mfma_f32_32x32x16_f16(tailw2[1], l1, G, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            float t = __builtin_fmaf(G[8 + k], 0.00048828125f, G[k]);     // low-order weight rows (units of 2^-11)
            t = ok ? t : 0.f;
            const unsigned ad = xbase + wa[k];      // (a local: inline-asm operands inside a generic lambda do not capture)
            asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(ad), "v"(t), "n"(RC * 2 * XREC) : "memory");
        }
    };
    // EPI 3 / 7: rows 4h + 2c, 4h + 2c + 1 of a patch whose tap image is complete (both channel halves of every row, published by a barrier), in
    // three pieces: (a) eight 16-byte LDS reads (the lane's four taps x two halves; unused terms read the buffer's zero rows), (b) the sum in
    // the order of tests/tailsum_model.py and the 16-byte store of one class plane, (c) the apron pass (four 4-byte reads, one 4-byte store)
    float4_t fT[4][2];
    float aT[2][2];
    auto fin_fetch = [&](unsigned xbase) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            fT[t][0] = *(const __attribute__((address_space(3))) float4_t*)(xbase + ta[t]);
            fT[t][1] = *(const __attribute__((address_space(3))) float4_t*)(xbase + ta[t] + (unsigned)XREC);
        }
    };
    auto fin_put = [&](const Item& it, bool live, const float4_t& sacc) {
        const int y = it.pyi * kTileH + f_rho, x0 = it.pxi * kTileW + 4 * f_q;
        const unsigned vo = (live & (y < a.H) & (x0 < a.W)) ? fs_lane : kOOR;
        const unsigned so = fs_base + (unsigned)it.b * fs_sb + (unsigned)it.pyi * fs_sy + (unsigned)it.pxi * (unsigned)(kTileW * 4);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, sacc), rout, vo, so, 0);
    };
    auto apron_fetch = [&](unsigned xbase) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            aT[t][0] = *(const __attribute__((address_space(3))) float*)(xbase + aa[t]);
            aT[t][1] = *(const __attribute__((address_space(3))) float*)(xbase + aa[t] + (unsigned)XREC);
        }
    };
    auto apron_store = [&](const Item& it, bool live) {
        const float v = (aT[0][0] + aT[0][1]) + (aT[1][0] + aT[1][1]);
        const unsigned so = ap_base + (unsigned)it.b * ap_sb + (unsigned)it.pyi * ap_sy + (unsigned)it.pxi * ap_sx;
        const bool ok = it.pxi * ap_vx + it.pyi * ap_vy + ap_lv < ap_lim;
        const unsigned vo = (live & ok) ? ap_lane : kOOR;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rout, vo, so, 0);
    };
    auto finalize_patch = [&](const Item& it, unsigned xbase, bool live) {
        fin_fetch(xbase);
        float4_t sacc = fT[0][0] + fT[0][1];
#pragma unroll
        for (int t = 1; t < 4; ++t) sacc = sacc + (fT[t][0] + fT[t][1]);
        fin_put(it, live, sacc);
        apron_fetch(xbase);
        apron_store(it, live);
    };

    // ===== prologue ====================================================================================================================
    Item it_cur{0, 0, 0};
    {
        it_cur.pxi = POOL ? 0 : g % a.px;
        const int t = POOL ? g : g / a.px;
        it_cur.pyi = t % a.py;
        it_cur.b = t / a.py;
    }
    Item it_prev = it_cur, it_pp = it_cur, it_next = advance(it_cur);
    prep_patch(it_cur, true);
    // per-iteration bookkeeping of patch p+1, formed INSIDE the MFMA stream of patch p-1 (micro-op / chunk `adv`): origin, border key, the item after it
    Item it_nn = advance(it_next);
    unsigned org_next = origin(it_next);
    int key_next = patch_key(it_next, 1 < K);
    if (TAIL) {      // the tap images start as zeros: their zero rows and the slots no lane ever writes (the consumer column beyond the patch) stay zero
        const u4_t z = {0u, 0u, 0u, 0u};
        for (unsigned o = (unsigned)tid * 16u; o < 3u * XCH_BYTES; o += 256u * 16u)
            asm volatile("ds_write_b128 %0, %1" ::"v"(xch0 + o), "v"(z) : "memory");
    }
    {
        const unsigned org = origin(it_cur);
#pragma unroll
        for (int i = 0; i < NDMA_W; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + (i * 4 + w4) * 1024), 16, vofs[i], org, 0, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int f = 0; f < 12; ++f) fr[0][f] = *(lds_h8_t)(fa[0][f]);
    }

    // ===== one patch.  Row step R multiplies input row R (fragments fr[R & 1]) into output rows R, R-1, R-2 and reads the fragments of
    // row R+1; every B fragment f is a CHUNK: its (up to) three MFMAs, one fragment read, one piece of the other work. ===================
    auto iteration = [&](int p, auto BUF_) __attribute__((always_inline)) {
        constexpr int BUF = decltype(BUF_)::value;
        const bool fetch = p + 1 < K;                     // patch p+1 exists
        const bool ep3 = p >= 1, fin = p >= 2;            // rows 2, 3 of patch p-1 / the rows of patch p-2 are pending
        const Item it = it_cur, itp = it_prev, itpp = it_pp, itn = it_next;
        const unsigned orgn = org_next;
        prep_patch_k(itn, key_next);
        const unsigned xb = xrot[0], xbp = xrot[1], xbpp = xrot[2];     // exchange buffers of patch p, p-1, p-2 (rotated at the end of the iteration)
        (void)fetch;
        // bookkeeping of the NEXT iteration (patch p+2's item, origin and border key), run by a chunk of row step 4 / at the end (no fused tail)
        Item nn2 = it_nn;
        unsigned org_n2 = 0;
        int key_n2 = -1;
        auto adv = [&]() __attribute__((always_inline)) {
            org_n2 = origin(it_nn);
            key_n2 = patch_key(it_nn, p + 2 < K);
            nn2 = advance(it_nn);
        };
        char* const nbuf = smem + (BUF ^ 1) * PATCH_BYTES;
        unsigned hA[4], hB[4];                             // activated halves of the row being finished
        float v[16];
        // fused tail: is this lane's pixel of a given row of patch p / p-1 inside the image?
        const bool okx = it.pxi * kTileW + j < a.W, okxp = itp.pxi * kTileW + j < a.W;
        const int yw = it.pyi * kTileH + 4 * h, ywp = itp.pyi * kTileH + 4 * h;
        // ---- fused tail micro-ops (tail_ops lists above) ---------------------------------------------------------------------------------
        float16_t Gp = zero16;                              // tail GEMM result of the row whose image writes are pending
        float4_t sacc = {0.f, 0.f, 0.f, 0.f};               // class sums of patch p-2 being added up
        auto op_dma = [&](auto I_) __attribute__((always_inline)) {
            constexpr int i = decltype(I_)::value;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(nbuf + (i * 4 + w4) * 1024), 16, vofs[i], orgn, 0, 0);
        };
        auto op_rd = [&](auto ROW_, auto Q_) __attribute__((always_inline)) {      // accumulator registers 4q .. 4q+3 of an output row -> its fp32 row buffer
            constexpr int row = decltype(ROW_)::value, q = decltype(Q_)::value;
#pragma unroll
            for (int e = 4 * q; e < 4 * q + 4; ++e) {
                if (row == 0) v[e] = acc[0][e];
                if (row == 1) v[e] = acc[1][e];
                if (row == 2) v2[e] = acc[2][e];
                if (row == 3) v3[e] = acc[3 + (BUF ^ 1)][e];
            }
        };
        auto op_p = [&](auto ROW_, auto K0_, auto N_) __attribute__((always_inline)) {      // PReLU (+ split) of channel pairs k0 .. k0 + n - 1 of a row
            constexpr int row = decltype(ROW_)::value, k0 = decltype(K0_)::value, n = decltype(N_)::value;
#pragma unroll
            for (int k = k0; k < k0 + n; ++k) {
                const float s0 = row < 2 ? v[2 * k] : row == 2 ? v2[2 * k] : v3[2 * k], s1 = row < 2 ? v[2 * k + 1] : row == 2 ? v2[2 * k + 1] : v3[2 * k + 1];
                unsigned hv, lv = 0;
                if (TAIL2) {
                    const float t0 = __builtin_fmaxf(s0, s0 * a.slope), t1 = __builtin_fmaxf(s1, s1 * a.slope);
                    split2(t0, t1, -2048.f, hv, lv);
                } else {
                    const half2_t pr = {(half_t)s0, (half_t)s1};
                    const half2_t t = pr * __builtin_bit_cast(half2_t, slope2);
                    hv = __builtin_bit_cast(unsigned, __builtin_elementwise_max(pr, t));
                }
                if (k < 4) { hA[k] = hv; lA[k] = lv; } else { hB[k - 4] = hv; lB[k - 4] = lv; }
            }
        };
        auto op_tm = [&]() __attribute__((always_inline)) {      // the tail GEMM of the row in hA / hB (/ lA / lB): per-tap sums over this wave's 32 channels
            const half8_t b0 = __builtin_bit_cast(half8_t, u4_t{hA[0], hA[1], hA[2], hA[3]});
            const half8_t b1 = __builtin_bit_cast(half8_t, u4_t{hB[0], hB[1], hB[2], hB[3]});
            float16_t G = __builtin_amdgcn_mfma_f32_32x32x16_f16(tailw[0], b0, zero16, 0, 0, 0);
            G = __builtin_amdgcn_mfma_f32_32x32x16_f16(tailw[1], b1, G, 0, 0, 0);
            if (TAIL2) {
                const half8_t l0 = __builtin_bit_cast(half8_t, u4_t{lA[0], lA[1], lA[2], lA[3]});
                const half8_t l1 = __builtin_bit_cast(half8_t, u4_t{lB[0], lB[1], lB[2], lB[3]});
                G = __builtin_amdgcn_mfma_f32_32x32x16_f16(tailw2[0], l0, G, 0, 0, 0);
                G = __builtin_amdgcn_mfma_f32_32x32x16_f16(tailw2[1], l1, G, 0, 0, 0);
            }
            Gp = G;
        };
        auto op_tw = [&](auto ROW_, auto K_) __attribute__((always_inline)) {      // product k of the lane -> its slot of the tap image (rows 0, 1: this patch, 2, 3: the previous one)
            constexpr int row = decltype(ROW_)::value, k = decltype(K_)::value;
            float t = __builtin_fmaf(Gp[8 + k], 0.00048828125f, Gp[k]);     // low-order weight rows (units of 2^-11)
            if (MASK) {
                const bool ok = row < 2 ? (okx & (yw + row < a.H)) : (okxp & (ywp + row < a.H));
                t = ok ? t : 0.f;
            }
            const unsigned ad = (row < 2 ? xb : xbp) + wa[k];
            asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(ad), "v"(t), "n"(row * 2 * XREC) : "memory");
        };
        auto op_ff = [&](auto T_) __attribute__((always_inline)) {
            constexpr int t = decltype(T_)::value;
            fT[t][0] = *(const __attribute__((address_space(3))) float4_t*)(xbpp + ta[t]);
            fT[t][1] = *(const __attribute__((address_space(3))) float4_t*)(xbpp + ta[t] + (unsigned)XREC);
        };
        auto op_fa = [&](auto T_) __attribute__((always_inline)) {
            constexpr int t = decltype(T_)::value;
            if (t == 0) sacc = fT[0][0] + fT[0][1];
            else sacc = sacc + (fT[t][0] + fT[t][1]);
        };

        auto step = [&](auto R_) __attribute__((always_inline)) {
            constexpr int R = decltype(R_)::value;
            constexpr int nm = (R <= 3 ? 1 : 0) + ((R >= 1 && R <= 4) ? 1 : 0) + (R >= 2 ? 1 : 0);     // MFMAs per chunk
            if (R == 5) {
                // Everybody's pieces of patch p+1 have landed, nobody reads patch p's buffer any more (row 5 is in registers), the exchange
                // records written so far are published.  (vmcnt(0) also waits for this patch's store acknowledgements: all issued rows ago.)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            auto chunk_f = [&](auto F_) __attribute__((always_inline)) {
                constexpr int f = decltype(F_)::value;
                constexpr int dx = f >> 2, ks = f & 3;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int o = R - dy;
                    if (o >= 0 && o < 4) {
                        const int sl = o == 3 ? 3 + BUF : o;
                        acc[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[(dy * 3 + dx) * 4 + ks], fr[R & 1][f], (dy == 0 && f == 0) ? biasv : acc[sl], 0, 0, 0);
                    }
                }
                if (R < 5) fr[(R + 1) & 1][f] = *(lds_h8_t)(fa[BUF][f] + (unsigned)((R + 1) * (PW * 128)));
                else fr[0][f] = *(lds_h8_t)(fa[BUF ^ 1][f]);            // row 0 of patch p+1 (behind the barrier)
                // ---- the piece of other work of this chunk.  Output row o of this patch is complete after row step o + 2:
                //   row 0   finished in step 3;   row 1   read out and activated in step 4, emitted in step 5 BEHIND the barrier (a global
                //   store issued shortly before the barrier's vmcnt(0) would make it wait for the acknowledgement);   row 2   read out in
                //   step 5, finished in step 1 of the next patch;   row 3   stays in its accumulators (slot 3 + BUF), read out in step 0 of
                //   the next patch, finished in its steps 1, 2;   DMA of patch p+1 in steps 0..2.
                constexpr OpList L = TAIL ? tail_ops<TAIL2>(R) : plain_ops(R);
                constexpr int o_lo = f * L.n / 12, o_hi = (f + 1) * L.n / 12;
                int ntm = 0;                         // tail-GEMM MFMAs issued by this chunk (they need slots of their own in the pattern below)
                {
                    auto run = [&](auto I_) __attribute__((always_inline)) {
                        constexpr int I = decltype(I_)::value;
                        if constexpr (I >= o_lo && I < o_hi) {
                            constexpr Op o = L.op[I];
                            if constexpr (o.kind == OP_DMA) op_dma(std::integral_constant<int, o.a>{});
                            if constexpr (o.kind == OP_RD) op_rd(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                            if constexpr (o.kind == OP_P) op_p(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{}, std::integral_constant<int, o.c>{});
                            if constexpr (o.kind == OP_TM) op_tm();
                            if constexpr (o.kind == OP_TW) op_tw(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                            if constexpr (o.kind == OP_FF) op_ff(std::integral_constant<int, o.a>{});
                            if constexpr (o.kind == OP_FA) op_fa(std::integral_constant<int, o.a>{});
                            if constexpr (o.kind == OP_FS) fin_put(itpp, fin, sacc);
                            if constexpr (o.kind == OP_AF) apron_fetch(xbpp);
                            if constexpr (o.kind == OP_AS) apron_store(itpp, fin);
                            if constexpr (o.kind == OP_ADV) adv();
                            if constexpr (o.kind == OP_ST) store_row(o.a < 2 ? it : itp, 4 * h + o.a, o.a < 2 ? true : ep3, hA, hB);
                        }
                    };
#define RW_OP(I) run(std::integral_constant<int, I>{});
                    RW_OP(0) RW_OP(1) RW_OP(2) RW_OP(3) RW_OP(4) RW_OP(5) RW_OP(6) RW_OP(7) RW_OP(8) RW_OP(9) RW_OP(10) RW_OP(11) RW_OP(12) RW_OP(13)
                    RW_OP(14) RW_OP(15) RW_OP(16) RW_OP(17) RW_OP(18) RW_OP(19) RW_OP(20) RW_OP(21) RW_OP(22) RW_OP(23) RW_OP(24) RW_OP(25) RW_OP(26)
                    RW_OP(27) RW_OP(28) RW_OP(29) RW_OP(30) RW_OP(31) RW_OP(32) RW_OP(33) RW_OP(34) RW_OP(35) RW_OP(36) RW_OP(37) RW_OP(38) RW_OP(39)
#undef RW_OP
                    for (int I = o_lo; I < o_hi; ++I) ntm += L.op[I].kind == OP_TM ? (TAIL2 ? 4 : 2) : 0;
                }
#ifndef RW_NOPIN
#pragma unroll
                for (int i_ = 0; i_ < 3; ++i_) {
                    if (i_ < nm) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i_ == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (i_ < nm) __builtin_amdgcn_sched_group_barrier(0x006, RW_FILL, 0);
                }
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) {      // the tail GEMM's MFMAs, each with fillers behind it
                    if (i_ < ntm) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x006, RW_FILL, 0); }
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            };
#define RW_CHUNK(F) chunk_f(std::integral_constant<int, F>{});
            RW_CHUNK(0) RW_CHUNK(1) RW_CHUNK(2) RW_CHUNK(3) RW_CHUNK(4) RW_CHUNK(5) RW_CHUNK(6) RW_CHUNK(7) RW_CHUNK(8) RW_CHUNK(9) RW_CHUNK(10) RW_CHUNK(11)
#undef RW_CHUNK
            // the fragments of the next row step have landed before it starts (one wait the compiler's count pass can see, see conv3x3_sp.hip)
            __builtin_amdgcn_s_waitcnt(0xC07F);
        };
#define RW_STEP(S) step(std::integral_constant<int, S>{});
        RW_STEP(0) RW_STEP(1) RW_STEP(2) RW_STEP(3) RW_STEP(4) RW_STEP(5)
#undef RW_STEP
        if (!RW_HOIST) adv();
        it_pp = it_prev; it_prev = it_cur; it_cur = it_next; it_next = it_nn; it_nn = nn2;
        org_next = org_n2; key_next = key_n2;
        { const unsigned t = xrot[2]; xrot[2] = xrot[1]; xrot[1] = xrot[0]; xrot[0] = t; }
    };

    int p = 0;
    for (; p + 1 < K; p += 2) {
        iteration(p, std::integral_constant<int, 0>{});
        iteration(p + 1, std::integral_constant<int, 1>{});
    }
    if (p < K) {
        iteration(p, std::integral_constant<int, 0>{});
        ++p;
    }
    // ===== drain: rows 2 and 3 of the last patch, then (fused tail) the rows of the last two patches =====================================
    {
        unsigned hA[4], hB[4];
        const unsigned xlast = xch0 + (unsigned)(((K + 2) % 3) * XCH_BYTES), xprev = xch0 + (unsigned)(((K + 1) % 3) * XCH_BYTES);     // buffers of patch K-1 / K-2
        const bool okxp = it_prev.pxi * kTileW + j < a.W;
        const int ywp = it_prev.pyi * kTileH + 4 * h;
        if (TAIL2) { act8s(v2, hA, lA); act8s(v2 + 8, hB, lB); } else { act8(v2, hA); act8(v2 + 8, hB); }
        if (TAIL) tail_partial(hA, hB, xlast, std::integral_constant<int, 2>{}, okxp & (ywp + 2 < a.H));
        else store_row(it_prev, 4 * h + 2, true, hA, hB);
        if (K & 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v3[e] = acc[3][e];
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) v3[e] = acc[4][e];
        }
        if (TAIL2) { act8s(v3, hA, lA); act8s(v3 + 8, hB, lB); } else { act8(v3, hA); act8(v3 + 8, hB); }
        if (TAIL) tail_partial(hA, hB, xlast, std::integral_constant<int, 3>{}, okxp & (ywp + 3 < a.H));
        else store_row(it_prev, 4 * h + 3, true, hA, hB);
        if (POOL) pool_flush();
        if (TAIL) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (K >= 2) finalize_patch(it_pp, xprev, true);
            finalize_patch(it_prev, xlast, true);
        }
    }
#endif
}

template <int EPI, bool MASK = true>
hipError_t set_limit()
{
    return hipFuncSetAttribute((const void*)conv3x3_rw_kernel<EPI, MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}

}  // namespace

hipError_t conv3x3_rw_init()
{
    hipError_t e;
    if ((e = set_limit<1>()) != hipSuccess) return e;
    if ((e = set_limit<4>()) != hipSuccess) return e;
    if ((e = set_limit<7>()) != hipSuccess) return e;
    if ((e = set_limit<7, false>()) != hipSuccess) return e;
    if ((e = set_limit<3, false>()) != hipSuccess) return e;
    return set_limit<3>();
}

// false: not one of the compiled epilogues / shapes (caller uses conv3x3_sp)
bool launch_conv3x3_rw(const ConvArgs& a, hipStream_t s)
{
    if (a.acc_mode != 0 || a.dbg || a.plane_w || a.res || a.out_lo || a.side16) return false;
    if (!(a.slope < 1.f) || a.scale != 1.f || !a.bias_img || a.in_cs != 64) return false;
    const bool tail = a.tplanes != nullptr;
    if (tail && (a.tail_form != 1 || a.r != 2 || a.nchunks != 4 || a.W % 4 != 0 || !tailsum_fits(a.B, a.H, a.W))) return false;     // (the nine-plane form lives in conv3x3_sp.hip)
    if (!tail && !a.out) return false;
    if (2ll * a.B * a.H * a.W * a.in_cs + 2ll * (a.W + 1) * a.in_cs >= (1ll << 32) - 65536) return false;
    if (!tail && 2ll * a.B * a.H * a.r * a.W * a.r * a.out_cs >= (1ll << 32) - 65536) return false;
    const int blocks = a.nchunks * ((a.G + 7) / 8) * 8;
    if (a.pool && (tail || a.nchunks != 1 || a.r != 1 || a.pool_slabs < 2 * std::min(a.G, a.py) || (a.G % a.py != 0 && a.py % a.G != 0))) return false;
    const bool ragged = a.H % kTileH != 0 || a.W % kTileW != 0;
    if (tail && a.tail_split) {
        if (ragged) conv3x3_rw_kernel<7, true><<<dim3(blocks), dim3(256), LDS_BYTES, s>>>(a);
        else conv3x3_rw_kernel<7, false><<<dim3(blocks), dim3(256), LDS_BYTES, s>>>(a);
    } else if (tail) {
        if (ragged) conv3x3_rw_kernel<3, true><<<dim3(blocks), dim3(256), LDS_BYTES, s>>>(a);
        else conv3x3_rw_kernel<3, false><<<dim3(blocks), dim3(256), LDS_BYTES, s>>>(a);
    }
    else if (a.pool) conv3x3_rw_kernel<4><<<dim3(blocks), dim3(256), LDS_BYTES, s>>>(a);
    else conv3x3_rw_kernel<1><<<dim3(blocks), dim3(256), LDS_BYTES, s>>>(a);
    return true;
}
