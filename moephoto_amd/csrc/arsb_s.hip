// arsb_s.hip -- one ARSB  y = x + conv_2(PReLU(conv_1(x)))  (python/models.py:53-80 of the reference; conv_2's weights carry the ScaleLayer factor) STREAMED
// down a 30-pixel column, the trunk's form of conv3x3_ps4.hip's row streaming.
//
// arsb32c.hip (round 3) walks patches of ten rows: per patch seven row steps per conv of which four carry 12 or 24 MFMAs instead of 36, two workgroup
// barriers, a pre-step at every range start; PMC: 0.73 MFMA busy.  Here every row step is alike:
//
//   workgroup     TWO waves (c = 0, 1: output channels 32c .. 32c+31 of BOTH convs, 288 weight registers each: 256 AGPR + 32 VGPR), two workgroups per CU;
//                 a workgroup streams a contiguous range of four-row blocks of a 30-pixel column (column-major ranges, as arsb32c.hip)
//   row step r    conv_1: x row r (12 fragment reads) into the m rows r-1, r, r+1 (36 MFMAs);  conv_2: m row r-4 (12 reads) into the output rows r-5, r-4,
//                 r-3 (36 MFMAs);  E1: PReLU of m row r-2 -> the m ring in LDS;  E2: output row r-6: + x_hi (from the x ring, which keeps twelve rows for
//                 that) [+ x_lo 2^-11, requested from HBM / L2 at the end of the previous step], hi / lo split, stores
//   rings         x: twelve rows of 34 (+2) pixels, filled two rows (nine one-KiB LDS-DMA pieces, 5 + 4 per wave) at a time, two blocks ahead of conv_1; m: four rows; ONE barrier of the
//                 two waves per two rows: it publishes the block's m rows (conv_2 runs four rows behind conv_1) and the next x block
//   ranges        a range [ya, yb) runs the x rows ya - 2 .. yb + 5: eight row steps of extra work per range (conv_2 idles through the first five, conv_1
//                 through the last four -- on rows nobody reads) instead of a pre-step per patch column
//
// MEASURED (profiles/r04, one call, 48 planes of 256 x 256): bit-identical to arsb32c.hip on every shape tried, and 7 % SLOWER (2.39 vs 2.23 ms for the five
// ARSBs of a launch set).  PMC: MFMA busy 0.69 at 1.74 GHz against 0.73 at 1.65 GHz -- the same busy x clock product: both forms run at the package power
// cap, where the MFMA rate is what the power budget buys, and this one issues 9 % more MFMAs (the rows a range recomputes at its ends: 108 two-row blocks per
// workgroup and up to two range starts).  It is therefore NOT the default (option arsb_impl = s); it stays as the A/B that shows what bounds the trunk.
//
// Same MFMAs in the same order as arsb32c.hip (rows stream in both, dy = 0, 1, 2 by row step, fragments (dx, ks) inside), same epilogue arithmetic: the
// outputs are bit-identical (tests/test_gpu_parity.py), whatever the range cuts.
#include "common.h"
#include "rowtile.h"
#include <algorithm>
#include <type_traits>

#ifndef AS_FILL
#define AS_FILL 5
#endif

namespace {

constexpr int RB = 2, TW = 30, XW = 36;        // rows per block (one barrier of the two waves each); output columns; pixels of a ring row (34 + 2: a block is nine 1-KiB pieces)
constexpr int ROWB = XW * 128;                 // 4,608
constexpr int BLKB = RB * ROWB;                // 9,216 = 9 KiB
constexpr int XROWS = 12, MROWS = 4;
constexpr int OFF_M = XROWS * ROWB;            // 55,296
constexpr int OFF_DUMP = OFF_M + MROWS * ROWB; // + 18,432: where wave 1's fifth DMA piece lands (nothing)
constexpr int LDS_BYTES = OFF_DUMP + 1024;     // 74,752: two workgroups per CU

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;

enum OpKind : int { OP_NONE = 0, OP_P, OP_MW, OP_ST, OP_XLO, OP_XHI, OP_RES, OP_SPL, OP_DMA };
struct Op { int kind, a, b, c; };
struct OpList {
    int n = 0;
    Op op[48] = {};
    constexpr void push(int kind, int a = 0, int b = 0, int c = 0) { op[n] = Op{kind, a, b, c}; ++n; }
};
// The ops of a row step in issue order (dealt to the 2 x NCH half-chunks proportionally): the DMA pieces of the next block, E1, then E2 in the second half --
// x_hi words from the x ring, sums, split, the row's stores, and LAST the x_lo words of the next step's row: they have most of a row step to arrive, and no
// store is issued behind them before they have been consumed (arsb32c.hip: a store behind a load in flight waits for it at issue)
constexpr OpList step_ops(int e, bool lo)
{
    OpList r;
    if (e == 0) for (int m = 0; m < 3; ++m) { r.push(OP_DMA, m, 0); r.push(OP_DMA, m, 1); }
    if (e == 1) for (int m = 3; m < 5; ++m) { r.push(OP_DMA, m, 0); r.push(OP_DMA, m, 1); }
    r.push(OP_P, 0, 0, 2); r.push(OP_P, 0, 2, 2); r.push(OP_MW, 0); r.push(OP_P, 1, 0, 2); r.push(OP_P, 1, 2, 2); r.push(OP_MW, 1);
    for (int o = 0; o < 2; ++o) {
        r.push(OP_XHI, o);
        r.push(OP_RES, o, 0); r.push(OP_RES, o, 2);
        if (lo) { r.push(OP_SPL, o, 0); r.push(OP_SPL, o, 2); }
        r.push(OP_ST, o);
    }
    if (lo) { r.push(OP_XLO, 0); r.push(OP_XLO, 1); }
    return r;
}

// VM operations of step 1's list that are issued behind its last DMA piece and in front of the barrier (the head of chunk nch - 2): the barrier's counted wait
// lets them -- the row's first stores -- stay in flight (vmcnt retires in order: everything older, the pieces included, is complete)
constexpr int vm_behind_dma(bool lo, int nch)
{
    // VM operations of a block, other than its five DMA pieces, that are issued behind the block's FIRST piece and in front of the barrier (head of chunk
    // nch - 2 of step 1): with the pieces themselves they may stay in flight at the barrier's counted wait -- everything older (vmcnt retires in order), the
    // pieces of the previous block included, is then complete
    int n = 0;
    for (int e = 0; e < 2; ++e) {
        const OpList l = step_ops(e, lo);
        int first = l.n;
        for (int i = l.n - 1; i >= 0; --i) if (l.op[i].kind == OP_DMA) first = i;
        const int hend = e == 0 ? 2 * nch : 2 * (nch - 2);
        for (int h = 0; h < hend; ++h)
            for (int i = h * l.n / (2 * nch); i < (h + 1) * l.n / (2 * nch); ++i)
                if (e == 1 || i > first) n += l.op[i].kind == OP_ST ? (lo ? 2 : 1) : l.op[i].kind == OP_XLO ? 1 : 0;
    }
    return n;
}

// KS: 16-channel k-slices of the convs' input that carry data (3 for the 48-channel nets, arsb32c.hip)
template <bool LO, int KS>
__global__ __launch_bounds__(128) void arsb_s_kernel(ArsbArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr unsigned kOOR = 0xFFFF0000u;
    constexpr int NCH = 3 * KS;                // chunks of a row step: one x fragment + one m fragment each
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

    // ---- weights (arsb32c.hip: MFMA row i = 8q + 4h' + e is given channel 16 (q >> 1) + 8 h' + 4 (q & 1) + e: a lane's registers 8g .. 8g+7 are one 16-byte slot)
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

    const unsigned nbytes = (unsigned)a.B * H * W * 128u;
    const unsigned in_pad = (unsigned)(RB * W + 2) * 128u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.x_hi - in_pad), 0, nbytes + in_pad, 0x00020000);
    const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc((void*)(LO ? a.x_lo : a.x_hi), 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ryh = __builtin_amdgcn_make_buffer_rsrc((void*)a.y_hi, 0, nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ryl = __builtin_amdgcn_make_buffer_rsrc((void*)(LO ? a.y_lo : a.y_hi), 0, nbytes, 0x00020000);
    unsigned d_off = 0, d_r = 0, d_cc = 0;
    auto piece_addr = [&](int m) {                           // piece i = 2 m + c of the 2 x 36 block (nine of them; wave 1's fifth does not exist)
        unsigned q = (unsigned)((2 * m + c) * 8 + (lane >> 3));
        asm volatile("" : "+v"(q));
        d_r = q >= (unsigned)XW ? 1u : 0u;
        d_cc = q - d_r * (unsigned)XW;
        const unsigned sl = (unsigned)(lane & 7) ^ ((d_cc >> 1) & 7u);
        d_off = ((__umul24(d_r, (unsigned)W) + d_cc) << 7) | (sl << 4);
    };
    auto piece_off = [&](int ya, int xa, bool live) {        // (ring columns 34, 35 are padding: nothing is fetched into them)
        const bool ok = ((unsigned)(ya + (int)d_r) < (unsigned)H) & ((unsigned)(xa + (int)d_cc) < (unsigned)W) & (d_cc < 34u) & live;
        return ok ? d_off : kOOR;
    };
    // B fragment (dx, ks): pixel col at col * 128, 16-B slot s at s ^ ((col >> 1) & 7); lane (j, hh) reads slot 2 ks + hh of column j + dx (conv3x3_ps4.hip)
    unsigned fa[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int cc = j + dx, z = (cc >> 1) & 7;
        fa[dx] = lds0 + (unsigned)(cc * 128 + ((z ^ hh) << 4));
        asm volatile("" : "+v"(fa[dx]));
    }
    // this lane's two 16-byte slots (channels 32c + 16 o + 8 hh .. +7 = slot 4c + 2o + hh) of m pixel column j, and of x pixel column j + 2 (the residual)
    unsigned mw[2], xa_[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int s = 4 * c + 2 * o + hh;
        mw[o] = lds0 + (unsigned)OFF_M + (unsigned)(j * 128 + ((s ^ ((j >> 1) & 7)) << 4));
        xa_[o] = lds0 + (unsigned)((j + 2) * 128 + ((s ^ (((j + 2) >> 1) & 7)) << 4));
        asm volatile("" : "+v"(mw[o]), "+v"(xa_[o]));
    }
    const unsigned lane_ob = (unsigned)(j * 128 + (32 * c + 8 * hh) * 2);      // byte offset of slot o = 0 of output column j inside a row of the stream tensors
    unsigned slope2;
    {
        const half2_t s2 = {(half_t)a.slope, (half_t)a.slope};
        slope2 = __builtin_bit_cast(unsigned, s2);
    }
    const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    float16_t acc1[4], acc2[4];      // step t (x row ya - 2 + t): m row t-1+1-dy in acc1[(t + 3 - dy) & 3]; output rows likewise from m row t - 4
    half8_t fx[3], fm[3];            // fragments of chunk n in f?[n % 3], read two chunks ahead
    u4_t xh, xl[2][2];               // residual words: x_hi of the slot being finished (from the x ring); x_lo of a row is requested TWO steps ahead (set = step parity)
    unsigned sh[4], sl[4];           // the packed results of one 16-byte slot
    unsigned hp[4];
    xl[0][0] = xl[0][1] = xl[1][0] = xl[1][1] = u4_t{0u, 0u, 0u, 0u};

    while (item < item_end) {
        const int s0 = item % nyb;
        const int t_ = item / nyb;
        const int pxi = t_ % px, b = t_ / px;
        const int s1 = min(nyb, s0 + (item_end - item));
        item += s1 - s0;
        const int x0 = pxi * TW;
        const int ya = RB * s0, yb = RB * s1;
        const int nblk = (yb - ya) / RB + 4;                  // steps t = 0 .. yb - ya + 7: x rows ya - 2 .. yb + 5 (conv_1 wants them up to yb + 1, conv_2 lags four rows, E2 two more)
        const bool mokx = (unsigned)(x0 - 1 + j) < (unsigned)W;
        const unsigned vo = ((j < TW) & (x0 + j < W)) ? lane_ob : kOOR;

        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        {
            const int yr = ya - 2, xa = x0 - 2;
            const unsigned org = (unsigned)((b * H + yr + RB) * W + xa + 2) * 128u;
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                piece_addr(m);
                const bool mine = 2 * m + c <= 8;
                char* dst = smem + (mine ? (2 * m + c) * 1024 : OFF_DUMP);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)dst, 16, piece_off(yr, xa, mine), org, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(dst + (mine ? BLKB : 0)), 16, piece_off(yr + RB, xa, mine), org + (unsigned)(RB * W) * 128u, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            fx[0] = *(lds_h8_t)(fa[0]);
            fx[1] = *(lds_h8_t)(KS > 1 ? (fa[0] ^ 32u) : fa[1]);
        }
        int xrow = 0;                                         // ring row of the block's first x row (even, < 12)

        auto block = [&](int k, auto BUF_) __attribute__((always_inline)) {
            constexpr int BUF = decltype(BUF_)::value;        // k & 1: the accumulator slots and the m ring repeat every four rows
            const int Rk = ya - 2 + RB * k;                   // first x row of this block
            const bool live = RB * (k + 2) <= yb - ya + 3;    // conv_1 wants rows of the block after the next: its pieces go out now, two blocks (~10k cycles) ahead
            const int yan = Rk + 2 * RB, xan = x0 - 2;
            const unsigned orgn = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((b * H + yan + RB) * W + xan + 2) * 128u));
            const int xnext = xrow + RB == XROWS ? 0 : xrow + RB;      // ring row of the next block
            const int xnext2 = xnext + RB == XROWS ? 0 : xnext + RB;   // ... and of the one behind it (rows r - 10, r - 9: their last readers were the residual reads of block k - 1)

            auto step = [&](auto E_) __attribute__((always_inline)) {
                constexpr int e = decltype(E_)::value;
                constexpr int T4 = 2 * BUF + e;               // step index mod 4
                const int r = Rk + e;
                const unsigned xo_cur = (unsigned)__builtin_amdgcn_readfirstlane((xrow + e) * ROWB);
                const unsigned xo_nxt = (unsigned)__builtin_amdgcn_readfirstlane((e == 0 ? xrow + 1 : xnext) * ROWB);
                const int rres = xrow + e + 6;                // x row r - 6: six ring rows back = six on
                const unsigned xo_res = (unsigned)__builtin_amdgcn_readfirstlane((rres >= XROWS ? rres - XROWS : rres) * ROWB);
                constexpr int S1 = T4 & 3;                    // acc1 slot of m row r - 2
                constexpr int MROW = (T4 + 2) & 3;            // m ring row of m row r - 2
                const bool mok = mokx & ((unsigned)(r - 2) < (unsigned)H);
                constexpr int S2 = T4 & 3;                    // acc2 slot of output row r - 6
                const int orow = r - 6;
                const bool ook = (orow >= ya) & (orow < yb);
                const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)(ook ? (unsigned)((b * H + orow) * W + x0) * 128u : kOOR));
                const bool ook2 = (orow + 2 >= ya) & (orow + 2 < yb);
                const unsigned so2 = (unsigned)__builtin_amdgcn_readfirstlane((int)(ook2 ? (unsigned)((b * H + orow + 2) * W + x0) * 128u : kOOR));

                auto op_p = [&](auto O_, auto K0_, auto N_) __attribute__((always_inline)) {
                    constexpr int o = decltype(O_)::value, k0 = decltype(K0_)::value, n = decltype(N_)::value;
#pragma unroll
                    for (int k = k0; k < k0 + n; ++k) {
                        const half2_t pr = {(half_t)acc1[S1][8 * o + 2 * k], (half_t)acc1[S1][8 * o + 2 * k + 1]};
                        const half2_t t = pr * __builtin_bit_cast(half2_t, slope2);
                        hp[k] = mok ? __builtin_bit_cast(unsigned, __builtin_elementwise_max(pr, t)) : 0u;      // conv_2 pads with ZEROS: m outside the image is 0
                    }
                };
                auto op_mw = [&](auto O_) __attribute__((always_inline)) {
                    constexpr int o = decltype(O_)::value;
                    const u4_t d = {hp[0], hp[1], hp[2], hp[3]};
                    const unsigned ad = mw[o];
                    asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(ad), "v"(d), "n"(MROW * ROWB) : "memory");
                };
                auto op_st = [&](auto O_) __attribute__((always_inline)) {
                    constexpr int o = decltype(O_)::value;
                    const u4_t dh = {sh[0], sh[1], sh[2], sh[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(dh, ryh, vo + (unsigned)(o * 32), so, 0);
                    if (LO) {
                        const u4_t dl = {sl[0], sl[1], sl[2], sl[3]};
                        __builtin_amdgcn_raw_buffer_store_b128(dl, ryl, vo + (unsigned)(o * 32), so, 0);
                    }
                };
                auto op_xlo = [&](auto O_) __attribute__((always_inline)) {      // the row of the step after the next (a load from HBM: a row step and a half to arrive)
                    constexpr int o = decltype(O_)::value;
                    xl[T4 & 1][o] = __builtin_amdgcn_raw_buffer_load_b128(rxl, vo + (unsigned)(o * 32), so2, 0);
                };
                auto op_xhi = [&](auto O_) __attribute__((always_inline)) {
                    constexpr int o = decltype(O_)::value;
                    xh = *(const __attribute__((address_space(3))) u4_t*)(xa_[o] + xo_res);
                };
                auto op_res = [&](auto O_, auto K0_) __attribute__((always_inline)) {      // acc += x_hi [+ x_lo 2^-11] for channel pairs k0, k0+1 of slot o (arsb32c.hip's arithmetic)
                    constexpr int o = decltype(O_)::value, k0 = decltype(K0_)::value;
#pragma unroll
                    for (int k = k0; k < k0 + 2; ++k) {
                        float v0 = acc2[S2][8 * o + 2 * k], v1 = acc2[S2][8 * o + 2 * k + 1];
                        v0 = mix_lo(xh[k], 1.0f, v0); v1 = mix_hi(xh[k], 1.0f, v1);
                        if (LO) { v0 = mix_lo(xl[T4 & 1][o][k], 0.00048828125f, v0); v1 = mix_hi(xl[T4 & 1][o][k], 0.00048828125f, v1); }
                        if (LO) { acc2[S2][8 * o + 2 * k] = v0; acc2[S2][8 * o + 2 * k + 1] = v1; }
                        else {
                            const half2_t pr = {(half_t)v0, (half_t)v1};
                            sh[k] = __builtin_bit_cast(unsigned, pr);
                        }
                    }
                };
                auto op_spl = [&](auto O_, auto K0_) __attribute__((always_inline)) {
                    constexpr int o = decltype(O_)::value, k0 = decltype(K0_)::value;
#pragma unroll
                    for (int k = k0; k < k0 + 2; ++k) split2(acc2[S2][8 * o + 2 * k], acc2[S2][8 * o + 2 * k + 1], -2048.f, sh[k], sl[k]);
                };
                auto op_dma = [&](auto M_, auto HALF_) __attribute__((always_inline)) {
                    constexpr int m = decltype(M_)::value, half = decltype(HALF_)::value;
                    if constexpr (half == 0) piece_addr(m);
                    else {
                        const bool mine = 2 * m + c <= 8;
                        const unsigned dsto = (unsigned)__builtin_amdgcn_readfirstlane((int)(mine ? xnext2 * ROWB + (2 * m + c) * 1024 : OFF_DUMP));
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + dsto), 16, piece_off(yan, xan, live & mine), orgn, 0, 0);
                    }
                };

                constexpr OpList L = step_ops(e, LO);
                auto chunk = [&](auto A_) __attribute__((always_inline)) {
                    constexpr int ai = decltype(A_)::value;               // active chunk: fragment (dx, ks) = (ai / KS, ai % KS)
                    constexpr int dx = ai / KS, ks = ai % KS;
                    if (e == 1 && ai == NCH - 2) {
                        // the block's m rows are written, the next x block has landed, nobody reads this block's x rows as operands any more (the row's last
                        // fragments are in registers) nor the m rows conv_2 has consumed; the stores this step has issued so far may still be on their way (counted wait)
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(5 + vm_behind_dma(LO, NCH)) : "memory");      // (+ this block's five pieces, which are for the block after the next)
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                    auto half = [&](auto HC_) __attribute__((always_inline)) {
                        constexpr int hc = decltype(HC_)::value;
                        constexpr int a2 = (ai + 2) % NCH;
                        if constexpr (hc == 0) {
#pragma unroll
                            for (int dy = 0; dy < 3; ++dy) {
                                const int sl_ = (T4 + 3 - dy) & 3;                      // m row r + 1 - dy
                                acc1[sl_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[(dy * 3 + dx) * 4 + ks], fx[ai % 3], (dy == 0 && ai == 0) ? zero16 : acc1[sl_], 0, 0, 0);
                            }
                            fx[(ai + 2) % 3] = *(lds_h8_t)((fa[a2 / KS] ^ (unsigned)((a2 % KS) * 32)) + (ai + 2 < NCH ? xo_cur : xo_nxt));
                        } else {
#pragma unroll
                            for (int dy = 0; dy < 3; ++dy) {
                                const int sl_ = (T4 + 3 - dy) & 3;                      // output row (r - 4) + 1 - dy
                                acc2[sl_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2[(dy * 3 + dx) * 4 + ks], fm[ai % 3], (dy == 0 && ai == 0) ? zero16 : acc2[sl_], 0, 0, 0);
                            }
                            constexpr int mrow = ai + 2 < NCH ? (T4 & 3) : ((T4 + 1) & 3);      // m row r - 4 (this step) / r - 3 (the next one)
                            fm[(ai + 2) % 3] = *(lds_h8_t)((fa[a2 / KS] ^ (unsigned)((a2 % KS) * 32)) + (unsigned)(OFF_M + mrow * ROWB));
                        }
                        constexpr int h = 2 * ai + hc;
                        constexpr int lo_ = h * L.n / (2 * NCH), hi_ = (h + 1) * L.n / (2 * NCH);
                        auto run = [&](auto I_) __attribute__((always_inline)) {
                            constexpr int I = decltype(I_)::value;
                            if constexpr (I >= lo_ && I < hi_) {
                                constexpr Op o = L.op[I];
                                if constexpr (o.kind == OP_P) op_p(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{}, std::integral_constant<int, o.c>{});
                                if constexpr (o.kind == OP_MW) op_mw(std::integral_constant<int, o.a>{});
                                if constexpr (o.kind == OP_ST) op_st(std::integral_constant<int, o.a>{});
                                if constexpr (o.kind == OP_XLO) op_xlo(std::integral_constant<int, o.a>{});
                                if constexpr (o.kind == OP_XHI) op_xhi(std::integral_constant<int, o.a>{});
                                if constexpr (o.kind == OP_RES) op_res(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                if constexpr (o.kind == OP_SPL) op_spl(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                if constexpr (o.kind == OP_DMA) op_dma(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                            }
                        };
#define AS_OP(I) run(std::integral_constant<int, I>{});
                        AS_OP(0) AS_OP(1) AS_OP(2) AS_OP(3) AS_OP(4) AS_OP(5) AS_OP(6) AS_OP(7) AS_OP(8) AS_OP(9) AS_OP(10) AS_OP(11) AS_OP(12) AS_OP(13) AS_OP(14) AS_OP(15)
                        AS_OP(16) AS_OP(17) AS_OP(18) AS_OP(19) AS_OP(20) AS_OP(21) AS_OP(22) AS_OP(23) AS_OP(24) AS_OP(25) AS_OP(26) AS_OP(27) AS_OP(28) AS_OP(29) AS_OP(30) AS_OP(31)
#undef AS_OP
                    };
                    half(std::integral_constant<int, 0>{});
                    half(std::integral_constant<int, 1>{});
#ifndef AS_NOPIN
#pragma unroll
                    for (int hc = 0; hc < 2; ++hc)
#pragma unroll
                        for (int i_ = 0; i_ < 3; ++i_) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (i_ == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x006, AS_FILL, 0);
                        }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                };
#define AS_CHUNK(A) if constexpr (A < NCH) chunk(std::integral_constant<int, A>{});
                AS_CHUNK(0) AS_CHUNK(1) AS_CHUNK(2) AS_CHUNK(3) AS_CHUNK(4) AS_CHUNK(5) AS_CHUNK(6) AS_CHUNK(7) AS_CHUNK(8) AS_CHUNK(9) AS_CHUNK(10) AS_CHUNK(11)
#undef AS_CHUNK
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            xrow = xnext;
        };

        // (the first steps' conv_2 and the last steps' conv_1 work on rows nobody reads: 8 row steps of extra work per range; separate variants of the block body
        // for them made the kernel three times as long and the register allocator give up)
        int k = 0;
        for (; k + 1 < nblk; k += 2) {
            block(k, std::integral_constant<int, 0>{});
            block(k + 1, std::integral_constant<int, 1>{});
        }
        if (k < nblk) block(k, std::integral_constant<int, 0>{});
    }
#endif
}

template <bool LO, int KS>
hipError_t set_limit() { return hipFuncSetAttribute((const void*)arsb_s_kernel<LO, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES); }

}  // namespace

hipError_t arsb_s_init()
{
    hipError_t e;
#ifndef AS_ONLY_LO4
    if ((e = set_limit<false, 4>()) != hipSuccess) return e;
    if ((e = set_limit<false, 3>()) != hipSuccess) return e;
    if ((e = set_limit<true, 3>()) != hipSuccess) return e;
#endif
    if ((e = set_limit<true, 4>()) != hipSuccess) return e;
    return hipSuccess;
}

// w1 / w2: packed A fragments in the pack_conv order (ConvLayer::w_hi).  false: the layer does not fit this kernel (the caller tries arsb32c)
bool launch_arsb_s(ArsbArgs a, int max_groups, hipStream_t s)
{
    if (!(a.slope <= 1.f) || a.H % RB != 0 || a.H < RB) return false;
    if ((long long)a.B * a.H * a.W * 128 + (long long)(RB * a.W + 2) * 128 >= (1ll << 32) - 65536) return false;   // 32-bit byte offsets
    if ((a.x_lo == nullptr) != (a.y_lo == nullptr)) return false;
    const int px = (a.W + TW - 1) / TW;
    const long long items = (long long)a.B * px * (a.H / RB);
    if (items >= (1ll << 31) / 4) return false;
    const int G = (int)std::min<long long>(items, 2ll * max_groups);      // two workgroups of two waves per CU
    if (a.cin != 0 && a.cin != 48 && a.cin != 64) return false;
    const bool k3 = a.cin == 48;
#ifdef AS_ONLY_LO4
    if (!a.x_lo || k3) return false;
    arsb_s_kernel<true, 4><<<dim3(G), dim3(128), LDS_BYTES, s>>>(a);
    return true;
#else
    if (a.x_lo) { if (k3) arsb_s_kernel<true, 3><<<dim3(G), dim3(128), LDS_BYTES, s>>>(a); else arsb_s_kernel<true, 4><<<dim3(G), dim3(128), LDS_BYTES, s>>>(a); }
    else { if (k3) arsb_s_kernel<false, 3><<<dim3(G), dim3(128), LDS_BYTES, s>>>(a); else arsb_s_kernel<false, 4><<<dim3(G), dim3(128), LDS_BYTES, s>>>(a); }
    return true;
#endif
}
