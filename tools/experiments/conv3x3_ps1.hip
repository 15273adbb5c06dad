// conv3x3_ps1.hip -- a plain 3x3 64 -> 64 convolution + PReLU / LeakyReLU (SEDN's rblock.0 / rblock.2, python/models.py:198-213 of the reference) in the
// row-streaming form of conv3x3_ps4.hip (round 6).
//
// conv3x3_rw.hip runs these layers on 8 x 32 patches: ten input rows for eight output rows, two MFMAs per fragment read, 0.42 of the fp16 peak executed
// (config 3: 61 of 161 ms).  conv3x3_ps4's store form reaches 0.56 on the same arithmetic.  Here the layer has ONE 64-channel chunk, so a wave holds ALL of
// its weights (72 A fragments of v_mfma_f32_32x32x16_f16 = 288 registers, 256 AGPR + 32 VGPR) and the four waves of a workgroup take four neighbouring
// 32-pixel COLUMNS of a 128-pixel strip:
//
//   wave w        = pixels x0 + 32 w .. + 31 of the strip, all 64 output channels as two groups of 32;
//   rows stream   input row r (one ds_read_b128 per (dx, k-slice): 12 reads) feeds the output rows r-1, r, r+1 of both channel groups: 72 MFMAs per row step
//                 and wave, six per fragment read, no vertical halo;
//   input ring    eight rows of 130 pixels in LDS, filled four rows (65 one-KiB raw-buffer LDS-DMA pieces, 16-17 per wave) at a time, one barrier per four rows;
//   epilogue      output row o is complete after row step o+1 and leaves during step o+2: PReLU on packed fp16 (as conv3x3_rw's EPI 1), four 16-byte stores;
//   ranges        the column-major sequence of four-row blocks is cut into one range per workgroup; the block in front of a range is run for its last input row
//                 only (same MFMAs in the same order as everywhere else: the result does not depend on the cut); blocks above / below the image run no MFMAs.
//
// Arithmetic: the conv sums are conv3x3_rw's bit for bit (same MFMAs in the same order per output, bias as the accumulators' initial value), and so is the
// activation: small launch sets stay on conv3x3_rw (a range needs >= 24 blocks to pay for its lead-in) and give the same bits.
#include "common.h"
#include "rowtile.h"
#include <algorithm>
#include <type_traits>

#ifndef PS1_FILL
#define PS1_FILL 5        // VALU / SALU slots pinned behind each MFMA of a chunk
#endif

namespace {

constexpr int RB = 4;                           // rows per block: one DMA fill, one barrier
constexpr int CW = 4 * kTileW;                  // 128: pixels of a strip (four waves x 32)
constexpr int PW = CW + 2;                      // 130
constexpr int ROWB = PW * 128;                  // bytes of an input row in LDS: 16,640
constexpr int BLKB = RB * ROWB;                 // 66,560 = 65 KiB
constexpr int NPIECE = BLKB / 1024;             // 65
constexpr int NM = (NPIECE + 3) / 4;            // 17 DMA slots per wave and block: piece w + 4 m (pieces 65 .. 67 are nobody's: they land in the dump)
constexpr int OFF_BIAS = 2 * BLKB;              // 133,120: bias [cg 2][hh 2][16] fp32
constexpr int OFF_DUMP = OFF_BIAS + 1024;
constexpr int LDS_BYTES = OFF_DUMP + 1024;      // 135,168
static_assert(BLKB % 1024 == 0, "pieces");

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;

enum OpKind : int { OP_NONE = 0, OP_P, OP_ST, OP_DMA };
struct Op { int kind, a, b, c; };
struct OpList {
    int n = 0;
    Op op[64] = {};
    constexpr void push(int kind, int a = 0, int b = 0, int c = 0) { op[n] = Op{kind, a, b, c}; ++n; }
};
// The row epilogue: per 16 channels (cg, g): PReLU of four channel pairs, then their 16-byte store
constexpr OpList row_ops()
{
    OpList r;
    for (int s = 0; s < 4; ++s) {
        r.push(OP_P, s >> 1, 4 * (s & 1), 2); r.push(OP_P, s >> 1, 4 * (s & 1) + 2, 2);
        r.push(OP_ST, s >> 1, s & 1);
    }
    return r;
}
// the DMA pieces of the next block (address half + issue half): six / six / five in the steps 0 .. 2, behind the row's stores
constexpr OpList extra_ops(int e)
{
    OpList r;
    const int lo = e == 0 ? 0 : e == 1 ? 6 : e == 2 ? 12 : NM, hi = e == 0 ? 6 : e == 1 ? 12 : NM;
    for (int m = lo; m < hi; ++m) { r.push(OP_DMA, m, 0); r.push(OP_DMA, m, 1); }
    return r;
}

__global__ __launch_bounds__(256) void conv3x3_ps1_kernel(Ps1Args a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr unsigned kOOR = 0xFFFF0000u;
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // (fragment addresses XOR their k-slice bits: the base must be 128-byte aligned)
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);      // this wave's 32-pixel column of the strip
    const int j = lane & 31, hh = lane >> 5;
    const int H = a.H, W = a.W;

    // ---- this workgroup's range of the column-major sequence of four-row blocks ------------------------------------------------------------------------
    const int px = (W + CW - 1) / CW, nyb = H / RB;
    const int g = blockIdx.x, G = gridDim.x;
    const long long nitems = (long long)a.B * px * nyb;
    int item = (int)(nitems * g / G);
    const int item_end = (int)(nitems * (g + 1) / G);
    if (item >= item_end) return;

    // ---- weights: 72 A fragments (tap, k-slice, channel group), pack_conv order [(tap 4 + ks) 2 + cg][lane][8]; rows permuted so that a lane's registers
    // 8g .. 8g+7 are eight consecutive channels (conv3x3_rw.hip): MFMA row i = 8q + 4h' + e is given channel 16 (q >> 1) + 8 h' + 4 (q & 1) + e of its group
    half8_t wf[2][36];
    {
        const int wi = lane & 31, wq = wi >> 3;
        const int src = (lane & 32) | (16 * (wq >> 1) + 8 * ((wi >> 2) & 1) + 4 * (wq & 1) + (wi & 3));
#pragma unroll
        for (int f = 0; f < 36; ++f) {
            wf[0][f] = *(const half8_t*)(a.wpk + ((f * 2 + 0) * 64 + src) * 8);
            wf[1][f] = *(const half8_t*)(a.wpk + ((f * 2 + 1) * 64 + src) * 8);
        }
#pragma unroll
        for (int f = 0; f < 36; ++f) asm volatile("" : "+a"(wf[0][f]));
#pragma unroll
        for (int f = 0; f < 28; ++f) asm volatile("" : "+a"(wf[1][f]));
#pragma unroll
        for (int f = 28; f < 36; ++f) asm volatile("" : "+v"(wf[1][f]));
    }
    if (tid < 64) {      // bias as the accumulators' initial value ([cg][hh][16]: register 4q + e of a lane is MFMA row 8q + 4hh + e)
        const int bcg = (tid >> 5) & 1, bhh = (tid >> 4) & 1, bq = (tid >> 2) & 3, be = tid & 3;
        *(float*)(smem + OFF_BIAS + tid * 4) = a.bias[32 * bcg + 16 * (bq >> 1) + 8 * bhh + 4 * (bq & 1) + be];
    }
    const unsigned bias_ad = lds0 + (unsigned)(OFF_BIAS + hh * 64);      // (cg 1: + 128)

    // ---- input: raw-buffer descriptor shifted by four rows + one pixel so that every block origin is a non-negative offset --------------------------------
    const unsigned in_pad = (unsigned)(RB * W + 1) * 128u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in - in_pad), 0, (unsigned)a.B * H * W * 128u + in_pad, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, (unsigned)a.B * H * W * 128u, 0x00020000);
    unsigned d_off = 0, d_r = 0, d_cc = 0;
    auto piece_addr = [&](int m) {                           // piece i = w4 + 4 m: the lane's pixel of the 4 x 130 block, its logical 16-byte slot
        unsigned q = (unsigned)((w4 + 4 * m) * 8 + (lane >> 3));
        asm volatile("" : "+v"(q));
        d_r = __umul24(q, 505u) >> 16;                        // q / 130 (q < 600)
        d_cc = (unsigned)(__mul24((int)d_r, -PW) + (int)q);
        const unsigned sl = (unsigned)(lane & 7) ^ ((d_cc >> 1) & 7u);
        d_off = ((__umul24(d_r, (unsigned)W) + d_cc) << 7) | (sl << 4);
    };
    auto piece_off = [&](int ya, int xa, bool live) {        // ya, xa: image row / column of the block's first pixel
        const bool ok = ((unsigned)(ya + (int)d_r) < (unsigned)H) & ((unsigned)(xa + (int)d_cc) < (unsigned)W) & live;
        return ok ? d_off : kOOR;
    };

    // ---- B fragment f = (dx, ks) of an input row: pixel col at col * 128, 16-B slot s at s ^ ((col >> 1) & 7); lane (j, hh) reads slot 2 ks + hh of column
    // 32 w + j + dx: one address per dx, the k-slice is an XOR of bits 5, 6
    unsigned fa[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int cc = 32 * w4 + j + dx, z = (cc >> 1) & 7;
        fa[dx] = lds0 + (unsigned)(cc * 128 + ((z ^ hh) << 4));
        asm volatile("" : "+v"(fa[dx]));
    }
    unsigned slope2;
    {
        const half2_t s2 = {(half_t)a.slope, (half_t)a.slope};
        slope2 = __builtin_bit_cast(unsigned, s2);
    }

    float16_t acc[4][2];      // out row o lives in slot o & 3
    half8_t fr[3];            // fragment of chunk f in fr[f % 3], read two chunks ahead
    unsigned hX[4];

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    while (item < item_end) {
        // ===== one strip: plane b, column group pxi, blocks [s0, s1) = conv rows [4 s0, 4 s1) ================================================================
        const int s0 = item % nyb;
        const int t_ = item / nyb;
        const int pxi = t_ % px, b = t_ / px;
        const int s1 = min(nyb, s0 + (item_end - item));
        item += s1 - s0;
        const int x0 = pxi * CW;
        const int nblk = s1 - s0 + 2;                         // input blocks s0 - 1 .. s1
        const bool okx = x0 + 32 * w4 + j < W;
        const int ylo = RB * s0, yhi = RB * s1;
        // the block above the image (s0 = 0) holds zeros only: it is not run at all (the accumulators start as the bias either way); the block below the image
        // (s1 = nyb) runs its epilogues -- the last two rows of the image leave there -- without MFMAs and fragment reads.  Same bits.
        const int kfirst = s0 == 0 ? 1 : 0;
        const int kz = s1 == nyb ? nblk - 1 : nblk;           // blocks [kfirst, kz) run MFMAs

        // everybody has left the previous strip (its last fragment reads)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        {
            const int ya = RB * (s0 - 1 + kfirst), xa = x0 - 1;
            const unsigned org = (unsigned)((b * H + ya + RB) * W + xa + 1) * 128u;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                piece_addr(m);
                const bool mine = w4 + 4 * m < NPIECE;
                char* dst = smem + (mine ? (w4 + 4 * m) * 1024 : OFF_DUMP);
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

        auto block = [&](int k, auto BUF_, auto ZERO_) __attribute__((always_inline)) {
            constexpr int BUF = decltype(BUF_)::value;
            constexpr bool ZERO = decltype(ZERO_)::value;     // an input block of zeros (below the image): epilogues and barrier, no MFMAs
            const int Rk = RB * (s0 - 1 + k);                 // first input row of this block
            const bool live = k + 1 < kz;                     // (the next block is one that reads its input)
            const int yan = Rk + RB, xan = x0 - 1;
            const unsigned orgn = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((b * H + yan + RB) * W + xan + 1) * 128u));

            auto step = [&](auto E_) __attribute__((always_inline)) {
                constexpr int e = decltype(E_)::value;
                const int orow = Rk + e - 2;                  // the conv row whose epilogue rides in this step
                constexpr int SL = (e + 2) & 3;               // its accumulator slot

                auto op_p = [&](auto CG_, auto K0_, auto N_) __attribute__((always_inline)) {
                    constexpr int cg = decltype(CG_)::value, k0 = decltype(K0_)::value, n = decltype(N_)::value;
#pragma unroll
                    for (int k = k0; k < k0 + n; ++k) {
                        const half2_t pr = {(half_t)acc[SL][cg][2 * k], (half_t)acc[SL][cg][2 * k + 1]};
                        const half2_t t = pr * __builtin_bit_cast(half2_t, slope2);
                        hX[k & 3] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(pr, t));
                    }
                };
                // channels 32 cg + 16 g + 8 hh .. + 7 of pixel (orow, x0 + 32 w + j): one 16-byte word
                auto op_st = [&](auto CG_, auto G_) __attribute__((always_inline)) {
                    constexpr int cg = decltype(CG_)::value, gq = decltype(G_)::value;
                    const bool rok = (orow >= ylo) & (orow < yhi);
                    const unsigned vo = okx ? (unsigned)(32 * w4 + j) * 128u + (unsigned)hh * 16u : kOOR;
                    const unsigned so = rok ? ((unsigned)((b * H + orow) * W + x0) * 128u + (unsigned)(64 * cg + 32 * gq)) : kOOR;
                    const u4_t d = {hX[0], hX[1], hX[2], hX[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(d, rout, vo, so, 0);
                };
                auto op_bi = [&](auto CG_) __attribute__((always_inline)) {           // the drained slot becomes the accumulator of conv row Rk + e + 2: bias in
                    constexpr int cg = decltype(CG_)::value;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {      // (four 16-byte loads: see conv3x3_ps4.hip)
                        const float4_t t = *(const __attribute__((address_space(3))) float4_t*)(bias_ad + (unsigned)(cg * 128 + q * 16));
                        acc[SL][cg][4 * q] = t[0]; acc[SL][cg][4 * q + 1] = t[1]; acc[SL][cg][4 * q + 2] = t[2]; acc[SL][cg][4 * q + 3] = t[3];
                    }
                };
                auto op_dma = [&](auto M_, auto HALF_) __attribute__((always_inline)) {
                    constexpr int m = decltype(M_)::value, half = decltype(HALF_)::value;
                    if constexpr (half == 0) piece_addr(m);
                    else {
                        const bool mine = w4 + 4 * m < NPIECE;
                        const unsigned dsto = (unsigned)__builtin_amdgcn_readfirstlane((int)(mine ? (BUF ^ 1) * BLKB + (w4 + 4 * m) * 1024 : OFF_DUMP));
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + dsto), 16, piece_off(yan, xan, live & mine), orgn, 0, 0);
                    }
                };
                auto chunk = [&](auto F_) __attribute__((always_inline)) {
                    constexpr int f = decltype(F_)::value;
                    constexpr int dx = f >> 2, ks = f & 3;
                    if (e == 3 && f == 10) {
                        // the next block's pieces have landed, nobody reads this block's input rows any more (the last fragments are in registers); vmcnt(0) also covers
                        // this block's stores
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                    constexpr OpList LM = row_ops();
                    constexpr OpList LX = extra_ops(e);
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
                        constexpr int MH = 10;      // half-chunks the row's op list is dealt to; the DMA pieces go to the half-chunks behind them
                        constexpr int m_lo = h < MH ? h * LM.n / MH : LM.n, m_hi = h < MH ? (h + 1) * LM.n / MH : LM.n;
                        constexpr int x_lo = h < 12 ? 0 : (h - 12) * LX.n / 12, x_hi = h < 12 ? 0 : (h - 11) * LX.n / 12;
                        auto runm = [&](auto I_) __attribute__((always_inline)) {
                            constexpr int I = decltype(I_)::value;
                            if constexpr (I >= m_lo && I < m_hi) {
                                constexpr Op o = LM.op[I];
                                if constexpr (o.kind == OP_P) op_p(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{}, std::integral_constant<int, o.c>{});
                                if constexpr (o.kind == OP_ST) op_st(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                            }
                        };
                        auto runx = [&](auto I_) __attribute__((always_inline)) {
                            constexpr int I = decltype(I_)::value;
                            if constexpr (I >= x_lo && I < x_hi) {
                                constexpr Op o = LX.op[I];
                                if constexpr (o.kind == OP_DMA) op_dma(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                            }
                        };
#define PS1_M(I) runm(std::integral_constant<int, I>{});
                        PS1_M(0) PS1_M(1) PS1_M(2) PS1_M(3) PS1_M(4) PS1_M(5) PS1_M(6) PS1_M(7) PS1_M(8) PS1_M(9) PS1_M(10) PS1_M(11)
#undef PS1_M
#define PS1_X(I) runx(std::integral_constant<int, I>{});
                        PS1_X(0) PS1_X(1) PS1_X(2) PS1_X(3) PS1_X(4) PS1_X(5) PS1_X(6) PS1_X(7) PS1_X(8) PS1_X(9) PS1_X(10) PS1_X(11)
#undef PS1_X
                        if (f == 10 && hc == 1) op_bi(std::integral_constant<int, 0>{});
                        if (f == 11 && hc == 1) op_bi(std::integral_constant<int, 1>{});
                    };
                    half(std::integral_constant<int, 0>{});
                    half(std::integral_constant<int, 1>{});
#ifndef PS1_NOPIN
#pragma unroll
                    for (int hc = 0; hc < (ZERO ? 0 : 2); ++hc) {
#pragma unroll
                        for (int i_ = 0; i_ < 3; ++i_) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (i_ == 0 && hc == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x006, PS1_FILL, 0);
                        }
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                };
#define PS1_CHUNK(F) chunk(std::integral_constant<int, F>{});
                PS1_CHUNK(0) PS1_CHUNK(1) PS1_CHUNK(2) PS1_CHUNK(3) PS1_CHUNK(4) PS1_CHUNK(5) PS1_CHUNK(6) PS1_CHUNK(7) PS1_CHUNK(8) PS1_CHUNK(9) PS1_CHUNK(10) PS1_CHUNK(11)
#undef PS1_CHUNK
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
        };

        int k = kfirst;
        for (; k + 1 < kz; k += 2) {
            block(k, std::integral_constant<int, 0>{}, std::false_type{});
            block(k + 1, std::integral_constant<int, 1>{}, std::false_type{});
        }
        if (k < kz) { block(k, std::integral_constant<int, 0>{}, std::false_type{}); ++k; }
        if (k < nblk) block(k, std::integral_constant<int, 0>{}, std::true_type{});      // (reads no input: its ring half does not matter)
    }
#endif
}

}  // namespace

hipError_t conv3x3_ps1_init()
{
    return hipFuncSetAttribute((const void*)conv3x3_ps1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}

// the shape conditions of the launcher (the engine asks before it chooses this form: smaller launch sets stay on conv3x3_rw, with the same bits)
bool ps1_applicable(int B, int H, int W, int max_groups)
{
    if (H % RB != 0 || H < RB || W < 1) return false;
    if ((long long)B * H * W * 128 + (long long)(RB * W + 1) * 128 >= (1ll << 32) - 65536) return false;      // 32-bit byte offsets
    const long long items = (long long)B * ((W + CW - 1) / CW) * (H / RB);
    if (items >= (1ll << 31) / 4) return false;
    return items >= 24ll * std::max(1, max_groups);      // a range pays one lead-in block: at least 24 blocks per workgroup
}

bool launch_conv3x3_ps1(const Ps1Args& a, int max_groups, hipStream_t s)
{
    if (!(a.slope < 1.f) || !a.in || !a.out || !a.wpk || !a.bias || !ps1_applicable(a.B, a.H, a.W, max_groups)) return false;
    conv3x3_ps1_kernel<<<dim3(max_groups), dim3(256), LDS_BYTES, s>>>(a);
    return true;
}
