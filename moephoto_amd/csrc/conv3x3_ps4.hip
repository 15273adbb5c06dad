// conv3x3_ps4.hip -- the last upsampler stage of a x2 / x4 net with its 64 -> 1 tail conv, ALL FOUR pixel-shuffle phases in one workgroup:
//   conv 3x3 64 -> 256 + bias, PixelShuffle(2), PReLU (python/models.py:29-36 of the reference), then conv 3x3 64 -> 1 on the shuffled tensor
//   (models.py:145-154) -- the 64-channel tensor at the output resolution never exists.
//
// conv3x3_rw.hip (round 3) gives a workgroup ONE phase (64 of the 256 output channels): four workgroups fetch and fill the same input patch, a
// fragment read feeds two MFMAs on average, the tail conv's 64-channel contraction spans two waves, and the nine per-tap products of an output
// pixel are spread over four workgroups -- hence the phase-class sums (16 B per output pixel and branch through HBM) and the tapsum4 gather.
// PMC of round 3: 0.75 MFMA busy at 1.6 GHz under the package power cap, 5x the algorithmic bytes written.  Here:
//
//   wave w        = phase (i, j) = (w >> 1, w & 1): all 64 output channels of that phase, as two groups of 32 (cg) -- 72 A fragments of
//                 v_mfma_f32_32x32x16_f16 = 288 registers (256 AGPR + 32 VGPR, as arsb32c.hip), loaded once per launch;
//   rows stream   a workgroup walks DOWN a 32-pixel column of the image.  Input row r (one ds_read_b128 per (dx, k-slice): 12 reads) feeds the
//                 three output rows r-1, r, r+1 of BOTH channel groups: 72 MFMAs per row step and wave, SIX per fragment read (rw: two), every
//                 row step alike (no 12- / 24-MFMA steps at patch tops and bottoms), the input is fetched once for all four phases and has no
//                 vertical halo (rw: 10 rows for 8, times four);
//   input ring    eight rows of 34 pixels in LDS, filled four rows (17 one-KiB raw-buffer LDS-DMA pieces) at a time, one barrier per four rows;
//   epilogue      output row o is complete after row step o+1 and is finished during step o+2 (four accumulator slots rotate): PReLU [+ hi / lo
//                 split], then the tail GEMM of the wave's OWN 64 channels: 4 (8 with the split) MFMAs give the nine per-tap sums T[tap][pixel]
//                 complete over the channels (rw: two half sums per wave pair);
//   tap image     T goes to a ring of 16 rows in LDS, [row][phase 4][tap 9][32 pixels] fp32, taps that are consumed one column over stored at
//                 the consumer's column (rw's layout idea), the two that leave the 32-pixel column into per-row export slots;
//   finishing     an output (HR) pixel (2y + i', 2x + j') is the sum of nine T values of the four phases at conv rows y-1 .. y+1: one lane per
//                 (i', pixel pair) reads 18 x 8 bytes, adds in a fixed order and stores FOUR consecutive fp32 outputs (16 B) -- 4 B per HR pixel
//                 and branch through HBM instead of 16 B of class sums + aprons; one task per wave and four rows;
//   column aprons the terms an output pixel at the left / right edge of the column receives from the neighbouring column (another workgroup's)
//                 are exported as one fp32 value per HR row and side; tailadd_kernel adds both branches' planes and these aprons (fixed order:
//                 deterministic, no atomics) -- it replaces tapsum4 (8 B read per output pixel instead of 32 B);
//   no row aprons the rows above / below a workgroup's range are RECOMPUTED (one block of four rows each side; same operands, same order: results
//                 do not depend on how the column-major ranges are cut).
//
// Arithmetic: the conv sums are bit-identical to conv3x3_rw's (same MFMAs in the same order, bias as the accumulators' initial value); the tail's
// nine products per output pixel are the same, associated differently in fp32.
#include "common.h"
#include "rowtile.h"
#include "../../include/moephoto_amd.h"
#include <algorithm>
#include <type_traits>

#ifndef PS4_FILL
#define PS4_FILL 5        // VALU / SALU slots pinned behind each MFMA of a chunk
#endif
#ifndef PS4_ZSKIP
#define PS4_ZSKIP 1       // 1: the input blocks above / below the image (all zeros) run no MFMAs (round 6) | 0: every block alike (A/B)
#endif
#ifndef PS4_ABL
#define PS4_ABL 0         // timing ablations (tools/mk_variant.sh; results are wrong): 1 no row epilogue (PReLU, tail GEMM, image writes), 2 no finishing / DMA ops,
#endif                    // 4 no bias reload, 8 no fragment reads, 16 no barrier, 32 no tail MFMAs only, 64 no DMA of the next block, 128 (with 64) both ring halves filled with real data at the head of a strip

namespace {

constexpr int RB = 4;                           // rows per block: one DMA fill, one barrier, one finishing task per wave
constexpr int PW = kTileW + 2;                  // 34
constexpr int ROWB = PW * 128;                  // bytes of an input row in LDS
constexpr int BLKB = RB * ROWB;                 // 17,408 = 17 KiB
constexpr int NPIECE = BLKB / 1024;             // 17
constexpr int TREC = 36 * 128 + 64;             // one row of the tap image: [phase 4][tap 9][32 px] fp32 + exports [side 2][i 2][dy 3] + pad
constexpr int TEXP = 36 * 128;
constexpr int TROWS = 16;
constexpr int OFF_T = 2 * BLKB;                 // 34,816
constexpr int OFF_BIAS = OFF_T + TROWS * TREC;  // + 74,752
constexpr int OFF_TW = OFF_BIAS + 1024;         // bias: [wave 4][cg 2][hh 2][16] fp32
constexpr int OFF_DUMP = OFF_TW + 8 * 1024;     // tail weights: 8 A fragments;  dump: where the 17th DMA piece of waves 1..3 lands (nothing)
constexpr int LDS_BYTES = OFF_DUMP + 1024;      // 119,808

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;

// ---- what rides in the MFMA stream of a row step, as micro-ops (conv3x3_rw.hip, arsb32c.hip: an MFMA hides ~5 other instructions) --------------------
enum OpKind : int { OP_NONE = 0, OP_P, OP_WL, OP_TM, OP_TW, OP_BI, OP_DMA, OP_FA, OP_FR, OP_FS, OP_AR, OP_AS, OP_ST };
struct Op { int kind, a, b, c; };
struct OpList {
    int n = 0;
    Op op[64] = {};
    constexpr void push(int kind, int a = 0, int b = 0, int c = 0) { op[n] = Op{kind, a, b, c}; ++n; }
    constexpr void append(const OpList& o) { for (int i = 0; i < o.n; ++i) { op[n] = o.op[i]; ++n; } }
};
// The row epilogue: per 16 channels (cg, half q) four PReLU ops (one channel pair each with the split, two without), then the tail MFMA(s) of that
// slice.  Tail-weight fragments come from LDS into two rotating register sets (WL n -> set n & 1, TM n reads set n & 1).
constexpr OpList row_ops(int epi)      // 0: store (PReLU, pixel shuffle), 1: fused tail, 2: fused tail with split activations
{
    OpList r;
    if (epi == 0) {                     // per 16 channels (cg, g): PReLU of four channel pairs, then their 16-byte store
        for (int s = 0; s < 4; ++s) {
            r.push(OP_P, s >> 1, 4 * (s & 1), 2); r.push(OP_P, s >> 1, 4 * (s & 1) + 2, 2);
            r.push(OP_ST, s >> 1, s & 1);
        }
        return r;
    }
    const bool split = epi == 2;
    int nl = 0;
    auto wl = [&](int frag) { r.push(OP_WL, frag, nl & 1); ++nl; };
    int nt = 0;
    wl(0);
    wl(split ? 4 : 1);
    for (int s = 0; s < 4; ++s) {               // slice s = 2 cg + q: channels 16 s .. 16 s + 15 of the phase
        if (split) for (int k = 0; k < 4; ++k) r.push(OP_P, s >> 1, 4 * (s & 1) + k, 1);
        else for (int k = 0; k < 4; k += 2) r.push(OP_P, s >> 1, 4 * (s & 1) + k, 2);
        r.push(OP_TM, nt & 1, 0, nt == 0); ++nt;                 // a: register set, b: 0 hi / 1 lo operand, c: first (C = 0)
        if (split) {
            if (s < 3) wl(s + 1);
            r.push(OP_TM, nt & 1, 1, 0); ++nt;
            if (s < 3) wl(4 + s + 1);
        } else if (s + 2 < 4) wl(s + 2);
    }
    for (int k = 0; k < 5; ++k) r.push(OP_TW, k);
    return r;
}
// what else a block carries: step 0 the finishing task of four rows published by the last barrier (its stores go out before this block's loads), behind
// it and in step 1 the DMA pieces of the next block (address half + issue half): two row steps ahead of the barrier in front of chunk 10 of step 3
constexpr OpList extra_ops(int e, bool tail)
{
    OpList r;
    if (!tail) {                        // (store form: the row's four stores go out in chunks 0 .. 4 of a step, the pieces in its chunks 6 .. 11)
        if (e == 0) for (int m = 0; m < 3; ++m) { r.push(OP_DMA, m, 0); r.push(OP_DMA, m, 1); }
        if (e == 1) for (int m = 3; m < 5; ++m) { r.push(OP_DMA, m, 0); r.push(OP_DMA, m, 1); }
        return r;
    }
    if (e == 0) {
        r.push(OP_FA);
        for (int dy = 0; dy < 3; ++dy) { r.push(OP_FR, dy, 0); r.push(OP_FR, dy, 1); r.push(OP_FR, dy, 2); }
        r.push(OP_FS);
        r.push(OP_AR); r.push(OP_AS);
        r.push(OP_DMA, 0, 0); r.push(OP_DMA, 0, 1);
    }
    if (e == 1) for (int m = 1; m < 5; ++m) { r.push(OP_DMA, m, 0); r.push(OP_DMA, m, 1); }
    return r;
}
constexpr int count_kind(const OpList& l, int lo, int hi, int kind) { int n = 0; for (int i = lo; i < hi; ++i) n += l.op[i].kind == kind; return n; }

// EPI 0: PReLU + pixel-shuffle stores (the other upsampler stages), 1: fused tail, 2: fused tail with the activation operand split (hi + lo 2^-11)
template <int EPI, bool MASK>
__global__ __launch_bounds__(256) void conv3x3_ps4_kernel(Ps4Args a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool TAIL = EPI >= 1, SPLIT = EPI == 2;
    constexpr bool PERM = !TAIL;         // store form: channel order that makes a lane's registers 8g .. 8g+7 eight consecutive channels (16-byte stores, conv3x3_rw.hip)
    constexpr unsigned kOOR = 0xFFFF0000u;
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // (fragment addresses XOR their k-slice bits: the base must be 128-byte aligned)
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pi = w4 >> 1, pj = w4 & 1;                      // this wave's pixel-shuffle phase
    const int j = lane & 31, hh = lane >> 5;
    const int H = a.H, W = a.W;

    // ---- this workgroup's range of the column-major sequence of four-row blocks ------------------------------------------------------------------------
    const int px = (W + kTileW - 1) / kTileW, nyb = H / RB;
    const int g = blockIdx.x, G = gridDim.x;
    const long long nitems = (long long)a.B * px * nyb;
    int item = (int)(nitems * g / G);
    const int item_end = (int)(nitems * (g + 1) / G);
    if (item >= item_end) return;

    // ---- weights: 72 A fragments (tap, k-slice, channel group) of this phase, pack_conv order [chunk = phase][(tap 4 + ks) 2 + cg][lane][8] ---------------
    half8_t wf[2][36];
    {
        const half_t* wsrc = a.wpk + (long long)w4 * (72 * 512);
        int src = lane;
        if (PERM) {      // MFMA row i = 8q + 4h' + e (register 4q + e of the lanes hh = h') is given channel 16 (q >> 1) + 8 h' + 4 (q & 1) + e of its group
            const int wi = lane & 31, wq = wi >> 3;
            src = (lane & 32) | (16 * (wq >> 1) + 8 * ((wi >> 2) & 1) + 4 * (wq & 1) + (wi & 3));
        }
#pragma unroll
        for (int f = 0; f < 36; ++f) {
            wf[0][f] = *(const half8_t*)(wsrc + ((f * 2 + 0) * 64 + src) * 8);
            wf[1][f] = *(const half8_t*)(wsrc + ((f * 2 + 1) * 64 + src) * 8);
        }
#pragma unroll
        for (int f = 0; f < 36; ++f) asm volatile("" : "+a"(wf[0][f]));
#pragma unroll
        for (int f = 0; f < 28; ++f) asm volatile("" : "+a"(wf[1][f]));
#pragma unroll
        for (int f = 28; f < 36; ++f) asm volatile("" : "+v"(wf[1][f]));
    }
    // ---- LDS tables: bias as the accumulators' initial value ([wave][cg][hh][16]: register 4q + e of a lane is MFMA row 8q + 4hh + e), the tail
    // conv's A fragments (engine.cpp tail(): rows 0..8 fp16 weights, rows 16..24 their remainders; fragments 4..7 for the activations' low parts),
    // the tap-image ring as zeros (the slots no lane ever writes -- the consumer column beyond the 32 pixels -- stay zero)
    {
        const int bw = tid >> 6, bcg = (tid >> 5) & 1, bhh = (tid >> 4) & 1, bq = (tid >> 2) & 3, be = tid & 3;
        *(float*)(smem + OFF_BIAS + tid * 4) = a.bias[64 * bw + 32 * bcg + (PERM ? 16 * (bq >> 1) + 8 * bhh + 4 * (bq & 1) + be : 8 * bq + 4 * bhh + be)];
        if (TAIL) {
            const u4_t* tsrc = (const u4_t*)a.tail_w;
            *(u4_t*)(smem + OFF_TW + tid * 16) = tsrc[tid];
            *(u4_t*)(smem + OFF_TW + 4096 + tid * 16) = SPLIT ? tsrc[256 + tid] : u4_t{0u, 0u, 0u, 0u};
            const u4_t z = {0u, 0u, 0u, 0u};
            for (int o = tid * 16; o < TROWS * TREC; o += 256 * 16) *(u4_t*)(smem + OFF_T + o) = z;
        }
    }
    const unsigned bias_ad = lds0 + (unsigned)(OFF_BIAS + ((w4 * 2 + 0) * 2 + hh) * 64);      // (cg 1: + 128)
    const unsigned tw_ad = lds0 + (unsigned)(OFF_TW + lane * 16);

    // ---- input: raw-buffer descriptor shifted by four rows + one pixel so that every block origin is a non-negative offset --------------------------------
    const unsigned in_pad = (unsigned)(RB * W + 1) * 128u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in - in_pad), 0, (unsigned)a.B * H * W * 128u + in_pad, 0x00020000);
    const __amdgpu_buffer_rsrc_t rpl = TAIL ? __builtin_amdgcn_make_buffer_rsrc((void*)a.plane, 0, (unsigned)a.B * H * W * 16u, 0x00020000)
                                            : __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (unsigned)a.B * H * W * 512u, 0x00020000);      // (store form: [B][2H][2W][64] fp16)
    const __amdgpu_buffer_rsrc_t rap = __builtin_amdgcn_make_buffer_rsrc((void*)(TAIL ? (void*)a.apron : (void*)a.out), 0, TAIL ? (unsigned)a.B * px * H * 16u : 0u, 0x00020000);
    unsigned d_off = 0, d_r = 0, d_cc = 0;
    auto piece_addr = [&](int m) {                           // piece i = w4 + 4 m (m < 4) / 16 (m = 4): the lane's pixel of the 4 x 34 block, its logical 16-byte slot
        unsigned q = (unsigned)((m < 4 ? w4 + 4 * m : 16) * 8 + (lane >> 3));
        asm volatile("" : "+v"(q));
        d_r = __umul24(q, 241u) >> 13;                        // q / 34 (q < 352)
        d_cc = (unsigned)(__mul24((int)d_r, -PW) + (int)q);
        const unsigned sl = (unsigned)(lane & 7) ^ ((d_cc >> 1) & 7u);
        d_off = ((__umul24(d_r, (unsigned)W) + d_cc) << 7) | (sl << 4);
    };
    auto piece_off = [&](int ya, int xa, bool live) {        // ya, xa: image row / column of the block's first pixel
        const bool ok = ((unsigned)(ya + (int)d_r) < (unsigned)H) & ((unsigned)(xa + (int)d_cc) < (unsigned)W) & live;
        return ok ? d_off : kOOR;
    };

    // ---- B fragment f = (dx, ks) of an input row: pixel col at col * 128, 16-B slot s at s ^ ((col >> 1) & 7); lane (j, hh) reads slot 2 ks + hh of column
    // j + dx = (z ^ hh) ^ 2 ks: one address per dx, the k-slice is an XOR of bits 5, 6
    unsigned fa[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int cc = j + dx, z = (cc >> 1) & 7;
        fa[dx] = lds0 + (unsigned)(cc * 128 + ((z ^ hh) << 4));
        asm volatile("" : "+v"(fa[dx]));
    }
    // ---- tap image: byte offset (inside a row record) of the slot each of the lane's five T values goes to: value k = tap 4 hh + k (k < 4), tap 8 (k = 4,
    // hh = 0; hh = 1: the record's pad word).  Tap (dy, dx) of phase (pi, pj) at conv pixel x is consumed by output column class j' at conv pixel
    //   x + 1 (pj = 1, dx = 0),  x - 1 (pj = 0, dx = 2),  x otherwise:  stored at the consumer's column; columns 32 / -1 are the export slots
    unsigned wa[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int t = k < 4 ? 4 * hh + k : 8;
        const int dy = t / 3, dx = t - 3 * dy;
        int col = j;
        if (pj == 1 && dx == 0) col = j + 1;
        if (pj == 0 && dx == 2) col = j - 1;
        unsigned o = (unsigned)((w4 * 9 + t) * 128 + col * 4);
        if (col == kTileW) o = (unsigned)(TEXP + ((0 * 2 + pi) * 3 + dy) * 4);
        if (col < 0) o = (unsigned)(TEXP + ((1 * 2 + pi) * 3 + dy) * 4);
        if (k == 4 && hh == 1) o = (unsigned)(TEXP + 48 + w4 * 4);
        wa[k] = lds0 + (unsigned)OFF_T + o;
        asm volatile("" : "+v"(wa[k]));
    }
    // ---- finishing lane (i', pixel pair xp); lanes 32..63 repeat lanes 0..31 with their stores rejected, lanes 32..35 carry the aprons (side, i') ----------
    const int f_i = (lane >> 4) & 1, f_xp = lane & 15;
    const int a_side = (lane >> 1) & 1, a_i = lane & 1;
    unsigned slope2;
    {
        const half2_t s2 = {(half_t)a.slope, (half_t)a.slope};
        slope2 = __builtin_bit_cast(unsigned, s2);
    }
    const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    float16_t acc[4][2];      // out row o lives in slot o & 3
    half8_t fr[3];            // fragment of chunk f in fr[f % 3], read two chunks ahead
    half8_t twr[2];
    unsigned hX[4], lX[4];
    float16_t Gt = zero16;

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    while (item < item_end) {
        // ===== one strip: plane b, column pxi, blocks [s0, s1) = conv rows [4 s0, 4 s1) =====================================================================
        const int s0 = item % nyb;
        const int t_ = item / nyb;
        const int pxi = t_ % px, b = t_ / px;
        const int s1 = min(nyb, s0 + (item_end - item));
        item += s1 - s0;
        const int x0 = pxi * kTileW;
        const int nblk = s1 - s0 + 3;                         // input blocks s0 - 1 .. s1, then (fused tail) one more iteration for the last finishing tasks
        // the block above the image (s0 = 0) holds zeros only: it is not run at all (the accumulators start as the bias either way, and the tap rows -2, -1 are written as
        // zeros by the epilogues of the next block); the block below the image (s1 = nyb) runs its epilogues -- the last two rows of the image, the zero tap rows H, H + 1 --
        // without MFMAs and fragment reads.  Same bits: a product with a zero activation adds nothing.
        const int kfirst = (PS4_ZSKIP && s0 == 0) ? 1 : 0;
        const int kz = (PS4_ZSKIP && s1 == nyb) ? nblk - 2 : nblk - 1;      // blocks [kfirst, kz) run MFMAs
        const bool okx = x0 + j < W;
        const int ylo = RB * s0, yhi = RB * s1;

        // everybody has left the previous strip (its last fragment reads, its finishing reads)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        {
            const int ya = RB * (s0 - 1 + kfirst), xa = x0 - 1;
            const unsigned org = (unsigned)((b * H + ya + RB) * W + xa + 1) * 128u;
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                piece_addr(m);
                const bool mine = m < 4 || w4 == 0;
                char* dst = smem + (m < 4 ? (w4 + 4 * m) * 1024 : (mine ? 16 * 1024 : OFF_DUMP));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)dst, 16, piece_off(ya, xa, mine), org, 0, 0);
#if PS4_ABL & 128
                // (ablation: real data in BOTH ring halves once per strip, no DMA afterwards -- what the stream costs with the operands' bits toggling but nothing fetched)
                if (m < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(dst + BLKB), 16, piece_off(ya + 4 * RB, xa, mine), org + (unsigned)(4 * RB * W) * 128u, 0, 0);
#endif
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int cg = 0; cg < 2; ++cg) acc[s][cg] = *(const __attribute__((address_space(3))) float16_t*)(bias_ad + cg * 128);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            fr[0] = *(lds_h8_t)(fa[0]);
            fr[1] = *(lds_h8_t)(fa[0] ^ 32u);
        }

        auto block = [&](int k, auto BUF_, auto KIND_) __attribute__((always_inline)) {
            constexpr int BUF = decltype(BUF_)::value;
            constexpr bool LAST = decltype(KIND_)::value == 2;     // the iteration behind the last input block: only its finishing task is wanted
            constexpr bool ZERO = decltype(KIND_)::value == 1;     // an input block of zeros (below the image): epilogues, finishing and barrier, no MFMAs
            const int Rk = RB * (s0 - 1 + k);                 // first input row of this block
            // the next block's DMA
            const bool live = k + 1 < kz;                     // (the next block is one that reads its input)
            const int yan = Rk + RB, xan = x0 - 1;
            const unsigned orgn = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((b * H + yan + RB) * W + xan + 1) * 128u));
            // finishing task of this wave: conv row yf, published by the previous block's barrier
            const int yf = Rk - 7 + w4;
            const bool f_ok = (yf >= ylo) & (yf < yhi);
            unsigned fb[3];
            float2_t fs[2] = {{0.f, 0.f}, {0.f, 0.f}};
            float2_t fv[2];
            float av[3];

            auto step = [&](auto E_) __attribute__((always_inline)) {
                constexpr int e = decltype(E_)::value;
                const int orow = Rk + e - 2;                  // the conv row whose epilogue rides in this step
                const unsigned trow = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((orow + 64) & 15) * (unsigned)TREC));
                const bool rowok = (unsigned)orow < (unsigned)H;
                constexpr int SL = (e + 2) & 3;               // its accumulator slot

                auto op_p = [&](auto CG_, auto K0_, auto N_) __attribute__((always_inline)) {
                    constexpr int cg = decltype(CG_)::value, k0 = decltype(K0_)::value, n = decltype(N_)::value;
#pragma unroll
                    for (int k = k0; k < k0 + n; ++k) {
                        const float s0v = acc[SL][cg][2 * k], s1v = acc[SL][cg][2 * k + 1];
                        unsigned hv, lv = 0;
                        if (SPLIT) {
                            const float t0 = __builtin_fmaxf(s0v, s0v * a.slope), t1 = __builtin_fmaxf(s1v, s1v * a.slope);
                            split2(t0, t1, -2048.f, hv, lv);
                        } else {
                            const half2_t pr = {(half_t)s0v, (half_t)s1v};
                            const half2_t t = pr * __builtin_bit_cast(half2_t, slope2);
                            hv = __builtin_bit_cast(unsigned, __builtin_elementwise_max(pr, t));
                        }
                        hX[k & 3] = hv; lX[k & 3] = lv;
                    }
                };
                // store form: channels 32 cg + 16 g + 8 hh .. + 7 of HR pixel (2 orow + pi, 2 (x0 + j) + pj): one 16-byte word
                auto op_st = [&](auto CG_, auto G_) __attribute__((always_inline)) {
                    constexpr int cg = decltype(CG_)::value, gq = decltype(G_)::value;
                    const bool rok = (orow >= ylo) & (orow < yhi);
                    const unsigned vo = okx ? (unsigned)(2 * j) * 128u + (unsigned)hh * 16u : kOOR;
                    const unsigned so = rok ? ((unsigned)((b * 2 * H + 2 * orow + pi) * 2 * W + 2 * x0 + pj) * 128u + (unsigned)(64 * cg + 32 * gq)) : kOOR;
                    const u4_t d = {hX[0], hX[1], hX[2], hX[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(d, rpl, vo, so, 0);
                };
                auto op_wl = [&](auto F_, auto S_) __attribute__((always_inline)) {
                    constexpr int fg = decltype(F_)::value, st = decltype(S_)::value;
                    twr[st] = *(lds_h8_t)(tw_ad + (unsigned)(fg * 1024));
                };
                auto op_tm = [&](auto S_, auto LO_, auto FIRST_) __attribute__((always_inline)) {
                    constexpr int st = decltype(S_)::value, lo = decltype(LO_)::value, first = decltype(FIRST_)::value;
                    const half8_t bv = lo ? __builtin_bit_cast(half8_t, u4_t{lX[0], lX[1], lX[2], lX[3]}) : __builtin_bit_cast(half8_t, u4_t{hX[0], hX[1], hX[2], hX[3]});
                    Gt = __builtin_amdgcn_mfma_f32_32x32x16_f16(twr[st], bv, first ? zero16 : Gt, 0, 0, 0);
                };
                auto op_tw = [&](auto K_) __attribute__((always_inline)) {
                    constexpr int k = decltype(K_)::value;
                    float t = __builtin_fmaf(Gt[8 + k], 0.00048828125f, Gt[k]);      // low-order rows (units of 2^-11)
                    const bool ok = MASK ? (rowok & okx) : rowok;
                    t = ok ? t : 0.f;
                    const unsigned ad = wa[k] + trow;
                    asm volatile("ds_write_b32 %0, %1" ::"v"(ad), "v"(t) : "memory");
                };
                auto op_bi = [&](auto CG_) __attribute__((always_inline)) {           // the drained slot becomes the accumulator of conv row Rk + e + 2: bias in
                    constexpr int cg = decltype(CG_)::value;
                    // (four 16-byte loads: a 64-byte load is split by the compiler into pieces without memory operands, and in front of such a piece it
                    // waits for every LDS-DMA in flight with vmcnt(0) -- four full fetch latencies per block, 1.0 of 7.1 ms of the R branch)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4_t t = *(const __attribute__((address_space(3))) float4_t*)(bias_ad + (unsigned)(cg * 128 + q * 16));
                        acc[SL][cg][4 * q] = t[0]; acc[SL][cg][4 * q + 1] = t[1]; acc[SL][cg][4 * q + 2] = t[2]; acc[SL][cg][4 * q + 3] = t[3];
                    }
                };
                auto op_dma = [&](auto M_, auto HALF_) __attribute__((always_inline)) {
                    constexpr int m = decltype(M_)::value, half = decltype(HALF_)::value;
                    if constexpr (half == 0) piece_addr(m);
                    else {
                        const bool mine = m < 4 || w4 == 0;
                        const unsigned dsto = (unsigned)__builtin_amdgcn_readfirstlane((int)(m < 4 ? (BUF ^ 1) * BLKB + (w4 + 4 * m) * 1024 : (mine ? (BUF ^ 1) * BLKB + 16 * 1024 : OFF_DUMP)));
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + dsto), 16, piece_off(yan, xan, live & mine), orgn, 0, 0);
                    }
                };
                // finishing: per tap row dy the source phase row i and conv row y + d depend on the lane's output row class i':
                //   i' = 0:  dy 0 -> (i 1, y - 1), dy 1 -> (i 0, y), dy 2 -> (i 1, y);    i' = 1:  dy 0 -> (i 0, y), dy 1 -> (i 1, y), dy 2 -> (i 0, y + 1)
                auto op_fa = [&]() __attribute__((always_inline)) {
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int isrc = f_i == 0 ? (dy == 1 ? 0 : 1) : (dy == 1 ? 1 : 0);
                        const int dd = f_i == 0 ? (dy == 0 ? -1 : 0) : (dy == 2 ? 1 : 0);
                        fb[dy] = lds0 + (unsigned)OFF_T + (unsigned)((yf + dd + 64) & 15) * (unsigned)TREC + (unsigned)((2 * isrc * 9 + dy * 3) * 128 + f_xp * 8);
                    }
                };
                // output column class j' = 0 takes  dx 0 <- phase j 1 (stored one column over), dx 1 <- j 0, dx 2 <- j 1;   j' = 1:  dx 0 <- j 0, dx 1 <- j 1, dx 2 <- j 0
                auto op_fr = [&](auto DY_, auto DX_) __attribute__((always_inline)) {
                    constexpr int dy = decltype(DY_)::value, dx = decltype(DX_)::value;
                    constexpr int j0 = dx == 1 ? 0 : 1, j1 = dx == 1 ? 1 : 0;
                    if constexpr (dx > 0 || dy > 0) { fs[0] = fs[0] + fv[0]; fs[1] = fs[1] + fv[1]; }      // the pair read one op earlier
                    fv[0] = *(const __attribute__((address_space(3))) float2_t*)(fb[dy] + (unsigned)((j0 * 9 + dx) * 128));
                    fv[1] = *(const __attribute__((address_space(3))) float2_t*)(fb[dy] + (unsigned)((j1 * 9 + dx) * 128));
                };
                auto op_fs = [&]() __attribute__((always_inline)) {
                    fs[0] = fs[0] + fv[0]; fs[1] = fs[1] + fv[1];
                    const float4_t o4 = {fs[0][0], fs[1][0], fs[0][1], fs[1][1]};      // HR columns 2 x .. 2 x + 3 of the pair
                    const bool ok = f_ok & (lane < 32) & (x0 + 2 * f_xp < W);
                    const unsigned vo = ok ? (unsigned)(f_i * 2 * W + 4 * f_xp) * 4u : kOOR;
                    const unsigned so = (unsigned)(((b * 2 * H + 2 * yf) * 2 * W) + 2 * x0) * 4u;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, o4), rpl, vo, f_ok ? so : 0u, 0);
                };
                auto op_ar = [&]() __attribute__((always_inline)) {
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int isrc = a_i == 0 ? (dy == 1 ? 0 : 1) : (dy == 1 ? 1 : 0);
                        const int dd = a_i == 0 ? (dy == 0 ? -1 : 0) : (dy == 2 ? 1 : 0);
                        const unsigned ad = lds0 + (unsigned)OFF_T + (unsigned)((yf + dd + 64) & 15) * (unsigned)TREC + (unsigned)(TEXP + ((a_side * 2 + isrc) * 3 + dy) * 4);
                        av[dy] = *(const __attribute__((address_space(3))) float*)ad;
                    }
                };
                auto op_as = [&]() __attribute__((always_inline)) {                   // apron[side][b][pxi][HR row]
                    const float v = (av[0] + av[1]) + av[2];
                    const bool ok = f_ok & (lane >= 32) & (lane < 36);
                    const unsigned vo = ok ? (unsigned)(((a_side * a.B) * px) * 2 * H + a_i) * 4u : kOOR;
                    const unsigned so = (unsigned)((b * px + pxi) * 2 * H + 2 * yf) * 4u;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rap, vo, f_ok ? so : 0u, 0);
                };

                if constexpr (LAST) {
                    if constexpr (e == 0 && !(PS4_ABL & 2)) {
                        op_fa();
                        op_fr(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); op_fr(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); op_fr(std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{});
                        op_fr(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); op_fr(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}); op_fr(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{});
                        op_fr(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}); op_fr(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}); op_fr(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
                        op_fs();
                        op_ar(); op_as();
                    }
                    return;
                }
                auto chunk = [&](auto F_) __attribute__((always_inline)) {
                    constexpr int f = decltype(F_)::value;
                    constexpr int dx = f >> 2, ks = f & 3;
                    if (e == 3 && f == 10 && !(PS4_ABL & 16)) {
                        // this block's T rows are written, the next block's pieces have landed, nobody reads this block's input rows any more (the last
                        // fragments are in registers); vmcnt(0) also covers this block's stores, issued three row steps ago
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                    constexpr OpList LM = row_ops(EPI);
                    constexpr OpList LX = extra_ops(e, TAIL);
                    // A chunk is two halves of three MFMAs, each followed by its slice of the op lists: at most one tail MFMA per half, so that two of them (a chain
                    // on one accumulator) are always three conv MFMAs apart -- issued back to back, the second waits out the first (the eight of a row of the R
                    // branch cost 15 % of the kernel for 10 % of its MFMAs)
                    auto half = [&](auto HC_) __attribute__((always_inline)) {
                        constexpr int hc = decltype(HC_)::value;
                        if constexpr (!ZERO) {
#pragma unroll
                            for (int u = 3 * hc; u < 3 * hc + 3; ++u) {
                                const int dy = u >> 1, cg = u & 1;
                                const int sl = (e + 1 - dy + 4) & 3;
                                acc[sl][cg] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cg][(dy * 3 + dx) * 4 + ks], fr[f % 3], acc[sl][cg], 0, 0, 0);
                            }
                        }
                        if constexpr (hc == 0 && !ZERO) {      // the fragment of chunk f + 2
                            constexpr int f2 = (f + 2) % 12;
                            constexpr int rowsel = f + 2 < 12 ? BUF * RB + e : (e < 3 ? BUF * RB + e + 1 : (BUF ^ 1) * RB);
                            if (!(PS4_ABL & 8)) fr[(f + 2) % 3] = *(lds_h8_t)((fa[f2 >> 2] ^ (unsigned)((f2 & 3) * 32)) + (unsigned)(rowsel * ROWB));
                        }
                        constexpr int h = 2 * f + hc;
                        constexpr int MH = TAIL ? 20 : 10;      // half-chunks the row's op list is dealt to
                        constexpr int m_lo = h < MH ? h * LM.n / MH : LM.n, m_hi = h < MH ? (h + 1) * LM.n / MH : LM.n;
                        constexpr int x_lo = TAIL ? h * LX.n / 24 : (h < 12 ? 0 : (h - 12) * LX.n / 12), x_hi = TAIL ? (h + 1) * LX.n / 24 : (h < 12 ? 0 : (h - 11) * LX.n / 12);
                        auto runm = [&](auto I_) __attribute__((always_inline)) {
                            constexpr int I = decltype(I_)::value;
                            if constexpr (I >= m_lo && I < m_hi && !(PS4_ABL & 1)) {
                                constexpr Op o = LM.op[I];
                                if constexpr (o.kind == OP_P) op_p(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{}, std::integral_constant<int, o.c>{});
                                if constexpr (o.kind == OP_WL) op_wl(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                if constexpr (o.kind == OP_TM && !(PS4_ABL & 32)) op_tm(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{}, std::integral_constant<int, o.c>{});
                                if constexpr (o.kind == OP_TW) op_tw(std::integral_constant<int, o.a>{});
                                if constexpr (o.kind == OP_ST) op_st(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                            }
                        };
                        auto runx = [&](auto I_) __attribute__((always_inline)) {
                            constexpr int I = decltype(I_)::value;
                            if constexpr (I >= x_lo && I < x_hi) {
                                constexpr Op o = LX.op[I];
                                if constexpr (o.kind != OP_DMA && (PS4_ABL & 2)) return;
                                if constexpr (o.kind == OP_DMA && (PS4_ABL & 64)) return;
                                if constexpr (o.kind == OP_DMA) op_dma(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                if constexpr (o.kind == OP_FA) op_fa();
                                if constexpr (o.kind == OP_FR) op_fr(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                if constexpr (o.kind == OP_FS) op_fs();
                                if constexpr (o.kind == OP_AR) op_ar();
                                if constexpr (o.kind == OP_AS) op_as();
                            }
                        };
#define PS4_M(I) runm(std::integral_constant<int, I>{});
                        PS4_M(0) PS4_M(1) PS4_M(2) PS4_M(3) PS4_M(4) PS4_M(5) PS4_M(6) PS4_M(7) PS4_M(8) PS4_M(9) PS4_M(10) PS4_M(11) PS4_M(12) PS4_M(13) PS4_M(14) PS4_M(15)
                        PS4_M(16) PS4_M(17) PS4_M(18) PS4_M(19) PS4_M(20) PS4_M(21) PS4_M(22) PS4_M(23) PS4_M(24) PS4_M(25) PS4_M(26) PS4_M(27) PS4_M(28) PS4_M(29) PS4_M(30) PS4_M(31)
                        PS4_M(32) PS4_M(33) PS4_M(34) PS4_M(35) PS4_M(36) PS4_M(37) PS4_M(38) PS4_M(39) PS4_M(40) PS4_M(41) PS4_M(42) PS4_M(43) PS4_M(44) PS4_M(45) PS4_M(46) PS4_M(47)
#undef PS4_M
#define PS4_X(I) runx(std::integral_constant<int, I>{});
                        PS4_X(0) PS4_X(1) PS4_X(2) PS4_X(3) PS4_X(4) PS4_X(5) PS4_X(6) PS4_X(7) PS4_X(8) PS4_X(9) PS4_X(10) PS4_X(11) PS4_X(12) PS4_X(13) PS4_X(14) PS4_X(15)
#undef PS4_X
                        if (f == 10 && hc == 1 && !(PS4_ABL & 4)) op_bi(std::integral_constant<int, 0>{});
                        if (f == 11 && hc == 1 && !(PS4_ABL & 4)) op_bi(std::integral_constant<int, 1>{});
                    };
                    half(std::integral_constant<int, 0>{});
                    half(std::integral_constant<int, 1>{});
#ifndef PS4_NOPIN
#pragma unroll
                    for (int hc = 0; hc < (ZERO ? 0 : 2); ++hc) {
                        const int h = 2 * f + hc;
                        const int MH = TAIL ? 20 : 10;
                        const int m_lo = h < MH ? h * LM.n / MH : LM.n, m_hi = h < MH ? (h + 1) * LM.n / MH : LM.n;
                        const int ntm = (PS4_ABL & 33) ? 0 : count_kind(LM, m_lo, m_hi, OP_TM);
#pragma unroll
                        for (int i_ = 0; i_ < 3; ++i_) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (i_ == 0 && hc == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x006, PS4_FILL, 0);
                        }
#pragma unroll
                        for (int i_ = 0; i_ < 2; ++i_)
                            if (TAIL && i_ < ntm) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x006, PS4_FILL, 0); }
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                };
#if PS4_ABL
                if constexpr (!LAST) asm volatile("" ::"v"(acc[SL][0]), "v"(acc[SL][1]), "v"(Gt));      // (ablation builds: the row's results stay live without its epilogue)
#endif
#define PS4_CHUNK(F) chunk(std::integral_constant<int, F>{});
                PS4_CHUNK(0) PS4_CHUNK(1) PS4_CHUNK(2) PS4_CHUNK(3) PS4_CHUNK(4) PS4_CHUNK(5) PS4_CHUNK(6) PS4_CHUNK(7) PS4_CHUNK(8) PS4_CHUNK(9) PS4_CHUNK(10) PS4_CHUNK(11)
#undef PS4_CHUNK
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
        };

        typedef std::integral_constant<int, 0> Run;
        typedef std::integral_constant<int, 1> Zero;
        typedef std::integral_constant<int, 2> Last;
        int k = kfirst;
        for (; k + 1 < kz; k += 2) {
            block(k, std::integral_constant<int, 0>{}, Run{});
            block(k + 1, std::integral_constant<int, 1>{}, Run{});
        }
        if (k < kz) { block(k, std::integral_constant<int, 0>{}, Run{}); ++k; }
        if (k < nblk - 1) { block(k, std::integral_constant<int, 0>{}, Zero{}); ++k; }      // (reads no input: its ring half does not matter)
        if (TAIL) block(k, std::integral_constant<int, 0>{}, Last{});      // (its ring half is not used)
    }
#endif
}

// out = plane_R + plane_U + the column aprons of both branches (fixed order), cast to the caller's type: what tapsum4 was for the phase-class sums.
// One thread = 8 consecutive outputs of a row (W % 8 == 0: HR widths are multiples of 16 here).
template <bool VEC>
__global__ __launch_bounds__(256) void tailadd_kernel(TailAddArgs a)
{
    const int nq = a.W >> 3;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.B * a.H * nq) return;
    const int xq = (int)(idx % nq);
    const int Y = (int)((idx / nq) % a.H), b = (int)(idx / ((long long)nq * a.H));
    const int X0 = xq * 8;
    const long long po = ((long long)b * a.H + Y) * a.W + X0;
    float v[8];
    {
        const float4 r0 = *(const float4*)(a.p0 + po), r1 = *(const float4*)(a.p0 + po + 4);
        v[0] = r0.x; v[1] = r0.y; v[2] = r0.z; v[3] = r0.w; v[4] = r1.x; v[5] = r1.y; v[6] = r1.z; v[7] = r1.w;
        if (a.p1) {
            const float4 u0 = *(const float4*)(a.p1 + po), u1 = *(const float4*)(a.p1 + po + 4);
            v[0] += u0.x; v[1] += u0.y; v[2] += u0.z; v[3] += u0.w; v[4] += u1.x; v[5] += u1.y; v[6] += u1.z; v[7] += u1.w;
        }
    }
    // HR column X = 64 c (c >= 1) receives what conv column c - 1 exported to its right; X = 64 c + 63 (c + 1 < px) what column c + 1 exported to its left
    const int c = X0 >> 6, in = X0 & 63;
    const long long as = (long long)a.B * a.px * a.H;            // elements of one side's array
    if (in == 0 && c >= 1) {
        const long long o = ((long long)b * a.px + (c - 1)) * a.H + Y;
        float t = a.a0[o];
        if (a.a1) t += a.a1[o];
        v[0] += t;
    }
    if (in == 56 && c + 1 < a.px) {
        const long long o = as + ((long long)b * a.px + (c + 1)) * a.H + Y;
        float t = a.a0[o];
        if (a.a1) t += a.a1[o];
        v[7] += t;
    }
    const long long yo = (a.y_off ? a.y_off[b] : (long long)b * a.H * a.W) + (long long)Y * a.W + X0;
    if (a.y_dtype == MOE_F16) {
        if (VEC) {
            half8_t h;
#pragma unroll
            for (int k = 0; k < 8; ++k) h[k] = (half_t)v[k];
            *(half8_t*)((half_t*)a.y + yo) = h;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) ((half_t*)a.y)[yo + k] = (half_t)v[k];
        }
    } else {
        float* yp = (float*)a.y + yo;
        if (VEC) {
            *(float4*)yp = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(yp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) yp[k] = v[k];
        }
    }
}

template <int EPI, bool MASK>
hipError_t set_limit()
{
    return hipFuncSetAttribute((const void*)conv3x3_ps4_kernel<EPI, MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}

}  // namespace

hipError_t conv3x3_ps4_init()
{
    hipError_t e;
    if ((e = set_limit<0, true>()) != hipSuccess) return e;
    if ((e = set_limit<1, false>()) != hipSuccess) return e;
    if ((e = set_limit<1, true>()) != hipSuccess) return e;
    if ((e = set_limit<2, false>()) != hipSuccess) return e;
    return set_limit<2, true>();
}

// bytes of one branch's buffers: the fp32 plane [B][2H][2W] and the column aprons [side 2][B][px][2H]
size_t ps4_plane_bytes(int B, int H, int W) { return (size_t)B * H * W * 16; }
size_t ps4_apron_bytes(int B, int H, int W) { return (size_t)2 * B * ((W + kTileW - 1) / kTileW) * 2 * H * 4; }

bool ps4_applicable(int B, int H, int W)
{
    if (H % RB != 0 || W % 4 != 0 || H < RB) return false;
    if ((long long)B * H * W * 128 + (long long)(RB * W + 1) * 128 >= (1ll << 32) - 65536) return false;      // 32-bit byte offsets (input; the plane is 16 B a pixel)
    return (long long)B * ((W + kTileW - 1) / kTileW) * (H / RB) < (1ll << 31) / 4;
}

// false: not applicable (the caller keeps conv3x3_rw + tapsum4)
bool launch_conv3x3_ps4(const Ps4Args& a, int max_groups, hipStream_t s)
{
    if (!ps4_tail_applicable(a.B, a.H, a.W, a.slope)) return false;      // (the store form below has the same shape conditions)
    const int px = (a.W + kTileW - 1) / kTileW;
    const long long items = (long long)a.B * px * (a.H / RB);
    const int G = (int)std::min<long long>(items, max_groups);
    const bool ragged = a.W % kTileW != 0;
    if (a.out) {        // store form
        if (a.plane || (long long)a.B * a.H * a.W * 512 >= (1ll << 32) - 65536) return false;
        conv3x3_ps4_kernel<0, true><<<dim3(G), dim3(256), LDS_BYTES, s>>>(a);
        return true;
    }
    if (a.split) {
        if (ragged) conv3x3_ps4_kernel<2, true><<<dim3(G), dim3(256), LDS_BYTES, s>>>(a);
        else conv3x3_ps4_kernel<2, false><<<dim3(G), dim3(256), LDS_BYTES, s>>>(a);
    } else {
        if (ragged) conv3x3_ps4_kernel<1, true><<<dim3(G), dim3(256), LDS_BYTES, s>>>(a);
        else conv3x3_ps4_kernel<1, false><<<dim3(G), dim3(256), LDS_BYTES, s>>>(a);
    }
    return true;
}

void launch_tailadd(const TailAddArgs& a, hipStream_t s)
{
    const long long n = (long long)a.B * a.H * (a.W / 8);
    if (a.vec_ok) tailadd_kernel<true><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(a);
    else tailadd_kernel<false><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(a);
}
