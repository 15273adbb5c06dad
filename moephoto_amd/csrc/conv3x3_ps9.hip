// conv3x3_ps9.hip -- the upsampler stage of a x3 net with its 64 -> 1 tail conv (round 6):
//   conv 3x3 64 -> 576 + bias, PixelShuffle(3), PReLU (python/models.py:33-36 of the reference), then conv 3x3 64 -> 1 on the shuffled tensor
//   (models.py:125-143) -- the 64-channel tensor at the output resolution never exists.
//
// Round 2/3 ran this layer per phase on conv3x3_sp<3/7> (nine workgroups fetch and fill the same input patch, two MFMAs per fragment read) with NINE fp32 tap
// planes per branch through HBM (36 B per output pixel and branch written, read again by tapsum<3>): 4.9 + 4.5 + 1.5 ms of a 19.6-ms a3 frame, 0.36 of the
// MFMA peak where conv3x3_ps4.hip reaches 0.46 - 0.51.  This file is conv3x3_ps4's design for nine phases:
//
//   nine phases   do not fit one workgroup: a phase's 64 output channels are 288 weight registers of a wave (one wave per SIMD), a CU holds four.  A workgroup is
//                 THREE waves = the three phases (pi, 0..2) of ONE phase row pi; three workgroups (pi = 0, 1, 2) walk the same range of four-row blocks.  The
//                 fourth SIMD of the CU idles -- under the package power cap (every MFMA kernel runs at 1.4 - 1.75 of 2.4 GHz) the clock
//                 takes most of that back;
//   rows stream   exactly as in conv3x3_ps4: input row r feeds output rows r-1, r, r+1 of both channel groups: 72 MFMAs per row step and wave, six per fragment
//                 read; eight-row input ring filled four rows at a time by LDS-DMA (17 one-KiB pieces dealt to three waves), one barrier per four rows;
//   epilogue      PReLU [+ hi / lo split], then the tail GEMM of the wave's own 64 channels: nine per-tap sums T[tap][pixel] (4 / 8 MFMAs per row);
//   tap image     ring of 8 rows in LDS, [row][pj 3][tap 9][32 pixels] fp32, taps consumed one column over stored at the consumer's column, the two that
//                 leave the 32-pixel column in per-row export slots;
//   finishing     the tail conv's HORIZONTAL sums close inside the workgroup: an output (HR) pixel (3y + i', 3x + j') takes tap row dy from HR row
//                 3y + i' + dy - 1, i.e. from ONE phase row -- S[dy][HR row 3y + pi][HR column] = sum over dx of T[(pi, pj(dx))][tap (dy, dx)] is complete
//                 over this workgroup's three waves and needs no other conv row (conv3x3_ps4's finishing reads rows y-1 .. y+1: here no recomputed range
//                 ends for the tail, an 8-row ring).  One lane per (row, dy, pixel): nine 4-byte reads, a fixed-order sum, one 12-byte store: three fp32
//                 planes S[dy] per branch = 12 B per output pixel (tap planes: 36 B);
//   column aprons the term an S value at the left / right edge of the column receives from the neighbouring column: one fp32 per (HR row, dy, side);
//   tailadd3      out[Y][X] = sum over branches and dy of S[dy][Y + dy - 1][X] (+ aprons at X = 96 c, 96 c + 95), fixed order: deterministic, no atomics.
//
// Arithmetic: the conv sums are bit-identical to conv3x3_sp's / conv3x3_rw's (same MFMAs in the same order, bias as the accumulators' initial value); the tail's
// nine products per output pixel are the same, associated differently in fp32.
#include "common.h"
#include "rowtile.h"
#include "../../include/moephoto_amd.h"
#include <algorithm>
#include <type_traits>

#ifndef PS9_FILL
#define PS9_FILL 5        // VALU / SALU slots pinned behind each MFMA of a chunk
#endif
#ifndef PS9_XCD
#define PS9_XCD 1         // 1: the XCD-aware block -> (range, phase row) map for launches that fill the chip | 0: pi = blockIdx % 3 always (A/B)
#endif
#ifndef PS9_ZSKIP
#define PS9_ZSKIP 1       // 1: the input blocks above / below the image (all zeros) run no MFMAs | 0: every block alike (A/B)
#endif

namespace {

constexpr int NW = 3;                           // waves of a workgroup = phases pj of its phase row
constexpr int NT = NW * 64;
constexpr int RB = 4;                           // rows per block: one DMA fill, one barrier, two finishing rounds per wave
constexpr int PW = kTileW + 2;                  // 34
constexpr int ROWB = PW * 128;                  // bytes of an input row in LDS
constexpr int BLKB = RB * ROWB;                 // 17,408 = 17 KiB
constexpr int NPIECE = BLKB / 1024;             // 17
constexpr int NM = (NPIECE + NW - 1) / NW;      // 6 DMA slots per wave and block: piece w + 3 m (17 is nobody's: it lands in the dump)
constexpr int TEXP = 27 * 128;                  // one row of the tap image: [pj 3][tap 9][32 px] fp32, then the exports [side 2][dy 3] and the pad words
constexpr int TREC = TEXP + 64;
constexpr int TROWS = 8;
constexpr int OFF_T = 2 * BLKB;                 // 34,816
constexpr int OFF_BIAS = OFF_T + TROWS * TREC;  // + 28,160
constexpr int OFF_TW = OFF_BIAS + 1024;         // bias: [wave 3][cg 2][hh 2][16] fp32
constexpr int OFF_DUMP = OFF_TW + 8 * 1024;     // tail weights: 8 A fragments;  dump: where a piece that is nobody's lands (nothing)
constexpr int LDS_BYTES = OFF_DUMP + 1024;      // 73,216

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef unsigned u3_t __attribute__((ext_vector_type(3)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;

// ---- what rides in the MFMA stream of a row step, as micro-ops (conv3x3_ps4.hip: an MFMA hides ~5 other instructions) ----------------------------------
enum OpKind : int { OP_NONE = 0, OP_P, OP_WL, OP_TM, OP_TW, OP_DMA, OP_FA, OP_FR, OP_FS, OP_AR, OP_AS };
struct Op { int kind, a, b, c; };
struct OpList {
    int n = 0;
    Op op[64] = {};
    constexpr void push(int kind, int a = 0, int b = 0, int c = 0) { op[n] = Op{kind, a, b, c}; ++n; }
};
// The row epilogue: per 16 channels (cg, half q) the PReLU ops (one channel pair each with the split, two without), then the tail MFMA(s) of that slice.
// Tail-weight fragments come from LDS into two rotating register sets (WL n -> set n & 1, TM n reads set n & 1).
constexpr OpList row_ops(bool split)
{
    OpList r;
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
// what else a block carries: steps 0 and 1 one finishing round each (the four rows published by the last barrier: 12 (row, dy) tasks = 3 waves x 2 lane halves
// x 2 rounds) and the DMA pieces of the next block (address half + issue half): two row steps ahead of the barrier in front of chunk 10 of step 3
constexpr OpList extra_ops(int e)
{
    OpList r;
    if (e > 1) return r;
    r.push(OP_FA, e);
    r.push(OP_FR, e, 0); r.push(OP_FR, e, 1); r.push(OP_FR, e, 2);
    r.push(OP_FS, e);
    r.push(OP_AR, e); r.push(OP_AS, e);
    if (e == 0) for (int m = 0; m < 2; ++m) { r.push(OP_DMA, m, 0); r.push(OP_DMA, m, 1); }
    if (e == 1) for (int m = 2; m < NM; ++m) { r.push(OP_DMA, m, 0); r.push(OP_DMA, m, 1); }
    return r;
}
constexpr int count_kind(const OpList& l, int lo, int hi, int kind) { int n = 0; for (int i = lo; i < hi; ++i) n += l.op[i].kind == kind; return n; }

// SPLIT: the tail conv's activation operand as hi + lo 2^-11 (MOE_PREC_MIXED, R branch); MASK: ragged width (W % 32 != 0)
template <bool SPLIT, bool MASK>
__global__ __launch_bounds__(NT) void conv3x3_ps9_kernel(Ps9Args a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr unsigned kOOR = 0xFFFF0000u;
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // (fragment addresses XOR their k-slice bits: the base must be 128-byte aligned)
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w3 = __builtin_amdgcn_readfirstlane(tid >> 6);      // this wave's phase column pj
    const int j = lane & 31, hh = lane >> 5;
    const int H = a.H, W = a.W;

    // ---- this workgroup's phase row pi and its range of the column-major sequence of four-row blocks --------------------------------------------------
    const int px = (W + kTileW - 1) / kTileW, nyb = H / RB;
    // block -> (range g of G, phase row pi).  xcd_map: blocks go round-robin to the eight XCDs; the three workgroups of a range sit on ONE XCD wherever that divides (the
    // second and third fetch of its input rows hit that XCD's L2): an XCD's S = grid / 8 workgroups form S / 3 ranges, what is left over forms ranges across XCDs
    int pi, g, G;
    if (a.xcd_map) {
        const int S = (int)gridDim.x >> 3, q = S / 3, r = S - 3 * q, x = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3;
        G = 8 * q + (8 * r) / 3;
        if (i < 3 * q) { pi = i % 3; g = x * q + i / 3; }
        else {
            const int l = (i - 3 * q) * 8 + x;
            if (l >= 3 * ((8 * r) / 3)) return;
            pi = l % 3; g = 8 * q + l / 3;
        }
    } else { G = (int)gridDim.x / 3; pi = (int)blockIdx.x % 3; g = (int)blockIdx.x / 3; }
    const long long nitems = (long long)a.B * px * nyb;
    int item = (int)(nitems * g / G);
    const int item_end = (int)(nitems * (g + 1) / G);
    if (item >= item_end) return;
    const int phase = pi * 3 + w3;

    // ---- weights: 72 A fragments (tap, k-slice, channel group) of this phase, pack_conv order [chunk = phase][(tap 4 + ks) 2 + cg][lane][8] ---------------
    half8_t wf[2][36];
    {
        const half_t* wsrc = a.wpk + (long long)phase * (72 * 512);
#pragma unroll
        for (int f = 0; f < 36; ++f) {
            wf[0][f] = *(const half8_t*)(wsrc + ((f * 2 + 0) * 64 + lane) * 8);
            wf[1][f] = *(const half8_t*)(wsrc + ((f * 2 + 1) * 64 + lane) * 8);
        }
#pragma unroll
        for (int f = 0; f < 36; ++f) asm volatile("" : "+a"(wf[0][f]));
#pragma unroll
        for (int f = 0; f < 28; ++f) asm volatile("" : "+a"(wf[1][f]));
#pragma unroll
        for (int f = 28; f < 36; ++f) asm volatile("" : "+v"(wf[1][f]));
    }
    // ---- LDS tables: bias as the accumulators' initial value ([wave][cg][hh][16]: register 4q + e of a lane is MFMA row 8q + 4hh + e), the tail conv's A
    // fragments (engine.cpp tail(): rows 0..8 fp16 weights, rows 16..24 their remainders; fragments 4..7 for the activations' low parts), the tap-image ring
    // as zeros (the slots no lane ever writes -- the consumer column beyond the 32 pixels -- stay zero)
    {
        const int bw = tid >> 6, bcg = (tid >> 5) & 1, bhh = (tid >> 4) & 1, bq = (tid >> 2) & 3, be = tid & 3;
        *(float*)(smem + OFF_BIAS + tid * 4) = a.bias[64 * (pi * 3 + bw) + 32 * bcg + 8 * bq + 4 * bhh + be];
        const u4_t* tsrc = (const u4_t*)a.tail_w;
        const u4_t z = {0u, 0u, 0u, 0u};
        for (int i = tid; i < 256; i += NT) {
            *(u4_t*)(smem + OFF_TW + i * 16) = tsrc[i];
            *(u4_t*)(smem + OFF_TW + 4096 + i * 16) = SPLIT ? tsrc[256 + i] : z;
        }
        for (int o = tid * 16; o < TROWS * TREC; o += NT * 16) *(u4_t*)(smem + OFF_T + o) = z;
    }
    const unsigned bias_ad = lds0 + (unsigned)(OFF_BIAS + ((w3 * 2 + 0) * 2 + hh) * 64);      // (cg 1: + 128)
    const unsigned tw_ad = lds0 + (unsigned)(OFF_TW + lane * 16);

    // ---- input: raw-buffer descriptor shifted by four rows + one pixel so that every block origin is a non-negative offset --------------------------------
    const unsigned in_pad = (unsigned)(RB * W + 1) * 128u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in - in_pad), 0, (unsigned)a.B * H * W * 128u + in_pad, 0x00020000);
    const unsigned plane_b = (unsigned)a.B * H * W * 36u;      // bytes of one S plane [B][3H][3W] fp32
    const __amdgpu_buffer_rsrc_t rpl = __builtin_amdgcn_make_buffer_rsrc((void*)a.plane, 0, 3u * plane_b, 0x00020000);
    const unsigned apron_b = (unsigned)a.B * px * H * 12u;      // bytes of one (side, dy) apron array [B][px][3H] fp32
    const __amdgpu_buffer_rsrc_t rap = __builtin_amdgcn_make_buffer_rsrc((void*)a.apron, 0, 6u * apron_b, 0x00020000);
    unsigned d_off = 0, d_r = 0, d_cc = 0;
    auto piece_addr = [&](int m) {                           // piece i = w3 + 3 m: the lane's pixel of the 4 x 34 block, its logical 16-byte slot
        unsigned q = (unsigned)((w3 + 3 * m) * 8 + (lane >> 3));
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
    // hh = 0; hh = 1: the record's pad word).  Tap (dy, dx) of phase column pj at conv pixel x is a term of the HR column 3 x + pj - dx + 1:
    //   conv pixel x + 1 (pj = 2, dx = 0),  x - 1 (pj = 0, dx = 2),  x otherwise:  stored at the consumer's column; columns 32 / -1 are the export slots
    unsigned wa[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int t = k < 4 ? 4 * hh + k : 8;
        const int dy = t / 3, dx = t - 3 * dy;
        int col = j;
        if (w3 == 2 && dx == 0) col = j + 1;
        if (w3 == 0 && dx == 2) col = j - 1;
        unsigned o = (unsigned)((w3 * 9 + t) * 128 + col * 4);
        if (col == kTileW) o = (unsigned)(TEXP + (0 * 3 + dy) * 4);
        if (col < 0) o = (unsigned)(TEXP + (1 * 3 + dy) * 4);
        if (k == 4 && hh == 1) o = (unsigned)(TEXP + 32 + w3 * 4);
        wa[k] = lds0 + (unsigned)OFF_T + o;
        asm volatile("" : "+v"(wa[k]));
    }
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
        const int nblk = s1 - s0 + 3;                         // input blocks s0 - 1 .. s1, then one more iteration for the last finishing rounds
        // the block above the image (s0 = 0) holds zeros only: it is not run at all (the accumulators start as the bias either way); the block below the image
        // (s1 = nyb) runs its epilogues -- the last two rows of the image are finished there -- without MFMAs and fragment reads.  Same bits: a product with a zero
        // activation adds nothing.
        const int kfirst = (PS9_ZSKIP && s0 == 0) ? 1 : 0;
        const int kz = (PS9_ZSKIP && s1 == nyb) ? nblk - 2 : nblk - 1;      // blocks [kfirst, kz) run MFMAs
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
            for (int m = 0; m < NM; ++m) {
                piece_addr(m);
                const bool mine = w3 + 3 * m < NPIECE;
                char* dst = smem + (mine ? (w3 + 3 * m) * 1024 : OFF_DUMP);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)dst, 16, piece_off(ya, xa, mine), org, 0, 0);
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
            constexpr bool LAST = decltype(KIND_)::value == 2;     // the iteration behind the last input block: only its finishing rounds are wanted
            constexpr bool ZERO = decltype(KIND_)::value == 1;     // an input block of zeros (below the image): epilogues, finishing and barrier, no MFMAs
            const int Rk = RB * (s0 - 1 + k);                 // first input row of this block
            // the next block's DMA
            const bool live = k + 1 < kz;                     // (the next block is one that reads its input)
            const int yan = Rk + RB, xan = x0 - 1;
            const unsigned orgn = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((b * H + yan + RB) * W + xan + 1) * 128u));
            // finishing: the rows Rk - 6 .. Rk - 3 (their epilogues rode in the previous block, published by its barrier)
            unsigned fb = 0;
            float fv[3], fsum[3], av = 0.f;

            auto step = [&](auto E_) __attribute__((always_inline)) {
                constexpr int e = decltype(E_)::value;
                const int orow = Rk + e - 2;                  // the conv row whose epilogue rides in this step
                const unsigned trow = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((orow + 64) & (TROWS - 1)) * (unsigned)TREC));
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
                    // (four 16-byte loads: see conv3x3_ps4.hip -- a 64-byte load is split into pieces in front of which the compiler waits with vmcnt(0))
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
                        const bool mine = w3 + 3 * m < NPIECE;
                        const unsigned dsto = (unsigned)__builtin_amdgcn_readfirstlane((int)(mine ? (BUF ^ 1) * BLKB + (w3 + 3 * m) * 1024 : OFF_DUMP));
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + dsto), 16, piece_off(yan, xan, live & mine), orgn, 0, 0);
                    }
                };
                // finishing round t: this lane's task (row, dy) = divmod(6 t + 2 w3 + hh, 3) of the 12 of the four rows Rk - 6 .. Rk - 3.  Every op derives what it needs
                // from t again (a handful of VALU instructions in an MFMA gap) instead of keeping offsets live across two row steps: the split epilogue leaves no registers
                auto task = [&](int t, int& dy, int& yf, bool& rok) __attribute__((always_inline)) {
                    int hv = hh;
                    asm volatile("" : "+v"(hv));                                      // (opaque: nothing derived from it is hoisted out of the block loop into long-lived registers)
                    const int id = 6 * t + 2 * w3 + hv;
                    const int rho = (id * 11) >> 5;                                   // id / 3 (id < 12)
                    dy = id - 3 * rho;
                    yf = Rk - 6 + rho;
                    rok = (yf >= ylo) & (yf < yhi);
                };
                auto op_fa = [&](auto T_) __attribute__((always_inline)) {
                    int dy, yf; bool rok;
                    task(decltype(T_)::value, dy, yf, rok);
                    fb = lds0 + (unsigned)OFF_T + (unsigned)((yf + 64) & (TROWS - 1)) * (unsigned)TREC + (unsigned)(dy * 384 + j * 4);
                };
                // output column class j' takes tap column dx from phase column (j' + dx - 1) mod 3 (stored at the consumer's column)
                auto op_fr = [&](auto T_, auto JP_) __attribute__((always_inline)) {
                    constexpr int jp = decltype(JP_)::value;
                    if constexpr (jp > 0) fsum[jp - 1] = (fv[0] + fv[1]) + fv[2];      // the three read one op earlier
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int pjs = (jp + dx + 2) % 3;
                        fv[dx] = *(const __attribute__((address_space(3))) float*)(fb + (unsigned)((pjs * 9 + dx) * 128));
                    }
                };
                auto op_fs = [&](auto T_) __attribute__((always_inline)) {            // S[dy][b][3 yf + pi][3 (x0 + j) ..]
                    int dy, yf; bool rok;
                    task(decltype(T_)::value, dy, yf, rok);
                    fsum[2] = (fv[0] + fv[1]) + fv[2];
                    unsigned fo = (unsigned)dy * plane_b + (unsigned)(((b * 3 * H + 3 * yf + pi) * 3 * W) + 3 * (x0 + j)) * 4u;
                    asm volatile("" : "+v"(fo));                                      // (computed by every lane: no branch around it in the MFMA stream)
                    fo = (rok & okx) ? fo : kOOR;
                    const u3_t o3 = {__builtin_bit_cast(unsigned, fsum[0]), __builtin_bit_cast(unsigned, fsum[1]), __builtin_bit_cast(unsigned, fsum[2])};
                    __builtin_amdgcn_raw_buffer_store_b96(o3, rpl, fo, 0, 0);
                };
                auto op_ar = [&](auto T_) __attribute__((always_inline)) {            // the export slot (side = j & 1, dy) of the task's row
                    int dy, yf; bool rok;
                    task(decltype(T_)::value, dy, yf, rok);
                    const unsigned fe = lds0 + (unsigned)OFF_T + (unsigned)((yf + 64) & (TROWS - 1)) * (unsigned)TREC + (unsigned)(TEXP + ((j & 1) * 3 + dy) * 4);
                    av = *(const __attribute__((address_space(3))) float*)fe;
                };
                auto op_as = [&](auto T_) __attribute__((always_inline)) {            // apron[side = j][dy][b][pxi][3 yf + pi]  (lanes j < 2)
                    int dy, yf; bool rok;
                    task(decltype(T_)::value, dy, yf, rok);
                    unsigned ao = (unsigned)((j & 1) * 3 + dy) * apron_b + (unsigned)((b * px + pxi) * 3 * H + 3 * yf + pi) * 4u;
                    asm volatile("" : "+v"(ao));
                    ao = (rok & (j < 2)) ? ao : kOOR;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, av), rap, ao, 0, 0);
                };

                if constexpr (LAST) {
                    if constexpr (e < 2) {
                        op_fa(E_);
                        op_fr(E_, std::integral_constant<int, 0>{}); op_fr(E_, std::integral_constant<int, 1>{}); op_fr(E_, std::integral_constant<int, 2>{});
                        op_fs(E_);
                        op_ar(E_); op_as(E_);
                    }
                    return;
                }
                auto chunk = [&](auto F_) __attribute__((always_inline)) {
                    constexpr int f = decltype(F_)::value;
                    constexpr int dx = f >> 2, ks = f & 3;
                    if (e == 3 && f == 10) {
                        // this block's T rows are written, the next block's pieces have landed, nobody reads this block's input rows any more (the last
                        // fragments are in registers); vmcnt(0) also covers this block's stores
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                    constexpr OpList LM = row_ops(SPLIT);
                    constexpr OpList LX = extra_ops(e);
                    // A chunk is two halves of three MFMAs, each followed by its slice of the op lists: at most one tail MFMA per half (conv3x3_ps4.hip)
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
                            fr[(f + 2) % 3] = *(lds_h8_t)((fa[f2 >> 2] ^ (unsigned)((f2 & 3) * 32)) + (unsigned)(rowsel * ROWB));
                        }
                        constexpr int h = 2 * f + hc;
                        constexpr int MH = 20;      // half-chunks the row's op list is dealt to
                        constexpr int m_lo = h < MH ? h * LM.n / MH : LM.n, m_hi = h < MH ? (h + 1) * LM.n / MH : LM.n;
                        constexpr int x_lo = h * LX.n / 24, x_hi = (h + 1) * LX.n / 24;
                        auto runm = [&](auto I_) __attribute__((always_inline)) {
                            constexpr int I = decltype(I_)::value;
                            if constexpr (I >= m_lo && I < m_hi) {
                                constexpr Op o = LM.op[I];
                                if constexpr (o.kind == OP_P) op_p(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{}, std::integral_constant<int, o.c>{});
                                if constexpr (o.kind == OP_WL) op_wl(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                if constexpr (o.kind == OP_TM) op_tm(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{}, std::integral_constant<int, o.c>{});
                                if constexpr (o.kind == OP_TW) op_tw(std::integral_constant<int, o.a>{});
                            }
                        };
                        auto runx = [&](auto I_) __attribute__((always_inline)) {
                            constexpr int I = decltype(I_)::value;
                            if constexpr (I >= x_lo && I < x_hi) {
                                constexpr Op o = LX.op[I];
                                if constexpr (o.kind == OP_DMA) op_dma(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                if constexpr (o.kind == OP_FA) op_fa(std::integral_constant<int, o.a>{});
                                if constexpr (o.kind == OP_FR) op_fr(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                if constexpr (o.kind == OP_FS) op_fs(std::integral_constant<int, o.a>{});
                                if constexpr (o.kind == OP_AR) op_ar(std::integral_constant<int, o.a>{});
                                if constexpr (o.kind == OP_AS) op_as(std::integral_constant<int, o.a>{});
                            }
                        };
#define PS9_M(I) runm(std::integral_constant<int, I>{});
                        PS9_M(0) PS9_M(1) PS9_M(2) PS9_M(3) PS9_M(4) PS9_M(5) PS9_M(6) PS9_M(7) PS9_M(8) PS9_M(9) PS9_M(10) PS9_M(11) PS9_M(12) PS9_M(13) PS9_M(14) PS9_M(15)
                        PS9_M(16) PS9_M(17) PS9_M(18) PS9_M(19) PS9_M(20) PS9_M(21) PS9_M(22) PS9_M(23) PS9_M(24) PS9_M(25) PS9_M(26) PS9_M(27) PS9_M(28) PS9_M(29) PS9_M(30) PS9_M(31)
                        PS9_M(32) PS9_M(33) PS9_M(34) PS9_M(35) PS9_M(36) PS9_M(37) PS9_M(38) PS9_M(39) PS9_M(40) PS9_M(41) PS9_M(42) PS9_M(43) PS9_M(44) PS9_M(45) PS9_M(46) PS9_M(47)
#undef PS9_M
#define PS9_X(I) runx(std::integral_constant<int, I>{});
                        PS9_X(0) PS9_X(1) PS9_X(2) PS9_X(3) PS9_X(4) PS9_X(5) PS9_X(6) PS9_X(7) PS9_X(8) PS9_X(9) PS9_X(10) PS9_X(11) PS9_X(12) PS9_X(13) PS9_X(14) PS9_X(15)
#undef PS9_X
                        if (f == 10 && hc == 1) op_bi(std::integral_constant<int, 0>{});
                        if (f == 11 && hc == 1) op_bi(std::integral_constant<int, 1>{});
                    };
                    half(std::integral_constant<int, 0>{});
                    half(std::integral_constant<int, 1>{});
#ifndef PS9_NOPIN
#pragma unroll
                    for (int hc = 0; hc < (ZERO ? 0 : 2); ++hc) {
                        const int h = 2 * f + hc;
                        const int MH = 20;
                        const int m_lo = h < MH ? h * LM.n / MH : LM.n, m_hi = h < MH ? (h + 1) * LM.n / MH : LM.n;
                        const int ntm = count_kind(LM, m_lo, m_hi, OP_TM);
#pragma unroll
                        for (int i_ = 0; i_ < 3; ++i_) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (i_ == 0 && hc == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x006, PS9_FILL, 0);
                        }
#pragma unroll
                        for (int i_ = 0; i_ < 2; ++i_)
                            if (i_ < ntm) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x006, PS9_FILL, 0); }
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                };
#define PS9_CHUNK(F) chunk(std::integral_constant<int, F>{});
                PS9_CHUNK(0) PS9_CHUNK(1) PS9_CHUNK(2) PS9_CHUNK(3) PS9_CHUNK(4) PS9_CHUNK(5) PS9_CHUNK(6) PS9_CHUNK(7) PS9_CHUNK(8) PS9_CHUNK(9) PS9_CHUNK(10) PS9_CHUNK(11)
#undef PS9_CHUNK
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
        block(k, std::integral_constant<int, 0>{}, Last{});
    }
#endif
}

// out[b][Y][X] = sum over the branches of S0[Y - 1][X] + S1[Y][X] + S2[Y + 1][X] + the column aprons (fixed order), cast to the caller's type.
// One thread = 4 consecutive outputs of a row (HR widths are multiples of 12 here: W % 4 == 0).
template <bool VEC>
__global__ __launch_bounds__(256) void tailadd3_kernel(TailAdd3Args a)
{
    const int nq = a.W >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.B * a.H * nq) return;
    const int xq = (int)(idx % nq);
    const int Y = (int)((idx / nq) % a.H), b = (int)(idx / ((long long)nq * a.H));
    const int X0 = xq * 4;
    const long long ps = (long long)a.B * a.H * a.W;                 // elements of one S plane
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int br = 0; br < 2; ++br) {
        const float* p = br == 0 ? a.p0 : a.p1;
        if (!p) continue;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int Ys = Y + dy - 1;
            if ((unsigned)Ys >= (unsigned)a.H) continue;
            const float4 r = *(const float4*)(p + dy * ps + ((long long)b * a.H + Ys) * a.W + X0);
            v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
        }
    }
    // HR column X = 96 c (c >= 1) receives what conv column c - 1 exported to its right (side 0); X = 96 c + 95 (c + 1 < px) what column c + 1 exported to its left
    const int c = X0 / 96, in = X0 - 96 * c;
    const long long as = (long long)a.B * a.px * a.H;                // elements of one (side, dy) apron array
    const bool left = in == 0 && c >= 1, right = in == 92 && c + 1 < a.px;
    if (left || right) {
        const int side = left ? 0 : 1, cs = left ? c - 1 : c + 1;
        float t = 0.f;
#pragma unroll
        for (int br = 0; br < 2; ++br) {
            const float* q = br == 0 ? a.a0 : a.a1;
            if (!q) continue;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int Ys = Y + dy - 1;
                if ((unsigned)Ys >= (unsigned)a.H) continue;
                t += q[(side * 3 + dy) * as + ((long long)b * a.px + cs) * a.H + Ys];
            }
        }
        v[left ? 0 : 3] += t;
    }
    const long long yo = (a.y_off ? a.y_off[b] : (long long)b * a.H * a.W) + (long long)Y * a.W + X0;
    if (a.y_dtype == MOE_F16) {
        half_t* yp = (half_t*)a.y + yo;
        if (VEC) {
            typedef half_t half4_t __attribute__((ext_vector_type(4)));
            const half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            *(half4_t*)yp = h;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) yp[k] = (half_t)v[k];
        }
    } else {
        float* yp = (float*)a.y + yo;
        if (VEC) *(float4*)yp = make_float4(v[0], v[1], v[2], v[3]);
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k) yp[k] = v[k];
        }
    }
}

template <bool SPLIT, bool MASK>
hipError_t set_limit()
{
    return hipFuncSetAttribute((const void*)conv3x3_ps9_kernel<SPLIT, MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}

}  // namespace

hipError_t conv3x3_ps9_init()
{
    hipError_t e;
    if ((e = set_limit<false, false>()) != hipSuccess) return e;
    if ((e = set_limit<false, true>()) != hipSuccess) return e;
    if ((e = set_limit<true, false>()) != hipSuccess) return e;
    return set_limit<true, true>();
}

// bytes of one branch's buffers: three fp32 planes S[dy] [B][3H][3W] and the column aprons [side 2][dy 3][B][px][3H]
size_t ps9_plane_bytes(int B, int H, int W) { return (size_t)3 * B * H * W * 36; }
size_t ps9_apron_bytes(int B, int H, int W) { return (size_t)6 * B * ((W + kTileW - 1) / kTileW) * 3 * H * 4; }

bool ps9_applicable(int B, int H, int W)
{
    if (H % RB != 0 || W % 4 != 0 || H < RB) return false;
    if ((long long)B * H * W * 128 + (long long)(RB * W + 1) * 128 >= (1ll << 32) - 65536) return false;      // 32-bit byte offsets (input; the three S planes are 108 B a pixel)
    return (long long)B * ((W + kTileW - 1) / kTileW) * (H / RB) < (1ll << 31) / 4;
}

// false: not applicable (the caller keeps conv3x3_sp + the nine tap planes + tapsum<3>)
bool launch_conv3x3_ps9(const Ps9Args& a, int max_groups, hipStream_t s)
{
    if (!ps9_tail_applicable(a.B, a.H, a.W, a.slope) || max_groups < 3) return false;
    const int px = (a.W + kTileW - 1) / kTileW;
    const long long items = (long long)a.B * px * (a.H / RB);
    Ps9Args q = a;
    int grid;
    {
        const int S = max_groups / 8, nq = S / 3, r = S - 3 * nq, G = 8 * nq + (8 * r) / 3;      // the XCD-aware map (see the kernel): G ranges on a grid of 8 S workgroups
        if (PS9_XCD && nq >= 1 && items >= G) { q.xcd_map = 1; grid = 8 * S; }
        else { q.xcd_map = 0; grid = 3 * (int)std::min<long long>(items, max_groups / 3); }      // ranges; three workgroups (phase rows) each
    }
    const bool ragged = a.W % kTileW != 0;
    if (a.split) {
        if (ragged) conv3x3_ps9_kernel<true, true><<<dim3(grid), dim3(NT), LDS_BYTES, s>>>(q);
        else conv3x3_ps9_kernel<true, false><<<dim3(grid), dim3(NT), LDS_BYTES, s>>>(q);
    } else {
        if (ragged) conv3x3_ps9_kernel<false, true><<<dim3(grid), dim3(NT), LDS_BYTES, s>>>(q);
        else conv3x3_ps9_kernel<false, false><<<dim3(grid), dim3(NT), LDS_BYTES, s>>>(q);
    }
    return true;
}

void launch_tailadd3(const TailAdd3Args& a, hipStream_t s)
{
    const long long n = (long long)a.B * a.H * (a.W / 4);
    if (a.vec_ok) tailadd3_kernel<true><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(a);
    else tailadd3_kernel<false><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(a);
}
