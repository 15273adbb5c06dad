// conv64_s.hip -- SEDN's fused block tail (python/models.py:186-224 of the reference: rblock.4 and the 1x1 `trans` folded, per plane, into one effective 3x3 64->64 conv,
// y = x + LeakyReLU(conv(W_eff[b], t)); engine.cpp, misc_kernels.hip: sedn_weff) STREAMED down a 32-pixel column: conv64_sq.hip without its fp8 half, with the
// plane's weights in REGISTERS.  conv3x3_sp<6> kept the per-plane weights in LDS and read 72 of its 120 fragments per patch for them: 166 us a launch; this form 145.
//
//   workgroup     TWO waves (c = 0, 1: output channels 32c .. 32c+31; 36 fp16 A fragments = 144 registers in AGPRs), two workgroups per CU
//   row step r    the 12 fragments (dx, ks) of input row r into the output rows r-1, r, r+1: 36 MFMAs; beside them the epilogue of output row r-2 -- LeakyReLU in fp32 + the
//                 residual's fp16 row from an LDS ring, one rounding (conv3x3_sp.hip's EPI 6) -- and in every second step the wave's DMA pieces of the block after the next
//   rows          two-row blocks, three per ring; ONE barrier of the two waves per block (first step, chunk 10, behind a counted vmcnt)
//   weights       PER PLANE ([B][72 fragments]): a range that enters another plane reloads its 144 registers
//   ranges        contiguous ranges of two-row blocks (column-major, conv64_sq.hip); rows at range ends are recomputed: results do not depend on the cuts
//   LDS           3 x 9,216 + 1,024 + table 4,096 + residual 3 x 8,192 = 57,344 bytes
//
// The template's EPI 1 (plain conv + LeakyReLU on packed halves, conv3x3_rw<1>'s bits) was built and measured too: 119.1 us against conv3x3_rw<1>'s 118.8 -- a single
// 64->64 conv moves 256 bytes for 73,728 FLOP per pixel, 4.2 TB/s at that speed: it sits at the machine's balance point and no schedule helps; SEDN's rblock.0 stays
// on conv3x3_rw.  What would help there is fewer bytes: rblock.0 + rblock.2 in one kernel.
#include "common.h"
#include "rowtile.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#ifndef S64_FILL
#define S64_FILL 5
#endif

namespace {

constexpr int RB = 2, TW = 32, XW = 36;
constexpr int ROWB = XW * 128;                 // 4,608
constexpr int BLKB = RB * ROWB;                // 9,216
constexpr int RESROWB = TW * 128, RESBLKB = RB * RESROWB;      // 4,096 / 8,192
constexpr int NRING = 3;
constexpr int OFF_DUMP = NRING * BLKB;         // 27,648
constexpr int OFF_TAB = OFF_DUMP + 1024;       // 28,672: [entry 7][thread 128] words
constexpr int OFF_RES = OFF_TAB + 8 * 512;     // 32,768
constexpr int LDS_PLAIN = OFF_RES, LDS_RES = OFF_RES + NRING * RESBLKB;      // 32,768 / 57,344

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;
typedef const __attribute__((address_space(3))) u4_t* lds_u4_t;

enum OpKind : int { OP_NONE = 0, OP_DMA, OP_RHI, OP_ACT, OP_ST };
struct Op { int kind, a, b, c; };
struct OpList {
    int n = 0;
    Op op[48] = {};
    constexpr void push(int kind, int a = 0, int b = 0, int c = 0) { op[n] = Op{kind, a, b, c}; ++n; }
};
constexpr OpList dma_ops(int e, int np)      // (conv64_sq.hip: half 0 the lane's table word, half 1 select + issue)
{
    OpList r;
    if (e == 1) {
        for (int i = 0; i < np; ++i) { r.push(OP_DMA, i, 0); if (i >= 1) r.push(OP_DMA, i - 1, 1); }
        r.push(OP_DMA, np - 1, 1);
    }
    return r;
}
constexpr OpList step_ops(bool res)
{
    OpList r;
    if (res) { r.push(OP_RHI, 0); r.push(OP_RHI, 1); }
    r.push(OP_ACT, 0, 0); r.push(OP_ACT, 0, 2); r.push(OP_ST, 0);
    r.push(OP_ACT, 1, 0); r.push(OP_ACT, 1, 2); r.push(OP_ST, 1);
    return r;
}
// stores a wave issues behind its last DMA piece (chunk 7 of step e = 1) and in front of the barrier of the next step (head of chunk 10)
constexpr int vm_behind(bool res)
{
    int n = 0;
    const OpList l = step_ops(res);
    for (int i = 7 * l.n / 12; i < l.n; ++i) n += l.op[i].kind == OP_ST ? 1 : 0;
    for (int i = 0; i < 10 * l.n / 12; ++i) n += l.op[i].kind == OP_ST ? 1 : 0;
    return n;
}

// EPI 1: out = LeakyReLU(conv(in)) on packed halves | 6: out = res + LeakyReLU(conv(W[b], in)) in fp32, per-plane weights (out may alias res)
template <int EPI>
__global__ __launch_bounds__(128) void conv64_s_kernel(ConvArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr unsigned kOOR = 0xFFFF0000u;
    constexpr bool RES = EPI == 6, PLANEW = EPI == 6;
    constexpr int NP = 5 + (RES ? 4 : 0);       // DMA pieces a wave issues per block: a_hi 2i + c (nine exist), residual 2i + c (eight)
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

    half8_t w16[36];
    const int wi = lane & 31, wq = wi >> 3;
    const int src = (lane & 32) | (16 * (wq >> 1) + 8 * ((wi >> 2) & 1) + 4 * (wq & 1) + (wi & 3));      // MFMA row -> channel order (conv3x3_rw.hip)
    auto load_weights = [&](int b) {
        const half_t* wp = a.wpk + (PLANEW ? (long long)b * 72 * 512 : 0ll);
#pragma unroll
        for (int f = 0; f < 36; ++f) w16[f] = *(const half8_t*)(wp + ((f * 2 + c) * 64 + src) * 8);
#pragma unroll
        for (int f = 0; f < 36; ++f) asm volatile("" : "+a"(w16[f]));
    };
    int wplane = -1;
    if (!PLANEW) load_weights(0);

    const unsigned nbytes = (unsigned)a.B * H * W * 128u;
    const unsigned in_pad = (unsigned)(RB * W + 2) * 128u, res_pad = in_pad + (unsigned)W * 128u;
    const __amdgpu_buffer_rsrc_t rhi = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in - in_pad), 0, nbytes + in_pad, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrh = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(RES ? a.res : a.in) - res_pad), 0, nbytes + res_pad, 0x00020000);
    const __amdgpu_buffer_rsrc_t ryh = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, nbytes, 0x00020000);

    // ---- DMA source offsets of a range in an LDS table (conv64_sq.hip): entry i = a_hi piece 2 i + c (i < 5), residual piece 2 (i - 5) + c for i = 5, 6 (pieces 2, 3 of
    // the wave are its pieces 0, 1 one row further down)
    constexpr int NT = RES ? 7 : 5;
    const unsigned tab = lds0 + (unsigned)OFF_TAB + (unsigned)(tid * 4);
    unsigned dt[2];
    auto piece_table = [&](int x0) {
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            unsigned d_off, d_r, d_cc;
            bool ok;
            if (i < 5) {
                const unsigned q = (unsigned)((2 * i + c) * 8 + (lane >> 3));
                d_r = q >= (unsigned)XW ? 1u : 0u;
                d_cc = q - d_r * (unsigned)XW;
                const unsigned sl = (unsigned)(lane & 7) ^ ((d_cc >> 1) & 7u);
                d_off = ((d_r * (unsigned)W + d_cc) << 7) | (sl << 4);
                ok = (q < 2u * XW) & (d_cc < 34u) & ((unsigned)(x0 - 1 + (int)d_cc) < (unsigned)W);
            } else {
                const unsigned q = (unsigned)((2 * (i - 5) + c) * 8 + (lane >> 3));
                d_r = q >> 5;
                d_cc = q & 31u;
                const unsigned sl = (unsigned)(lane & 7) ^ ((d_cc >> 1) & 7u);
                d_off = ((d_r * (unsigned)W + d_cc) << 7) | (sl << 4);
                ok = (unsigned)(x0 + (int)d_cc) < (unsigned)W;
            }
            *(__attribute__((address_space(3))) unsigned*)(tab + (unsigned)(i * 512)) = ok ? (d_off | d_r) : kOOR;
        }
    };
    auto piece_word = [&](int i) { return *(const __attribute__((address_space(3))) unsigned*)(tab + (unsigned)((i < 7 ? i : i - 2) * 512)); };
    auto piece_issue = [&](int i, unsigned d, int slot, int yr, int x0, int b, bool ok0, bool ok1, bool rk0, bool rk1) {
        const bool rowok = i >= 5 ? (i >= 7 ? rk1 : rk0) : (d & 1u) ? ok1 : ok0;
        const unsigned off = rowok ? (d & ~1u) : kOOR;
        if (i < 5) {
            const unsigned pix = (unsigned)((b * H + yr + RB) * W + x0 - 1 + 2);
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(2 * i + c < 9 ? slot * BLKB + (2 * i + c) * 1024 : OFF_DUMP);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rhi, (__attribute__((address_space(3))) void*)(smem + dst), 16, off, (unsigned)__builtin_amdgcn_readfirstlane((int)(pix * 128u)), 0, 0);
        } else {
            const unsigned org = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((b * H + yr - 2 + RB + 1 + (i >= 7 ? 1 : 0)) * W + x0 + 2) * 128u));
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(OFF_RES + slot * RESBLKB + (2 * (i - 5) + c) * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rrh, (__attribute__((address_space(3))) void*)(smem + dst), 16, off, org, 0, 0);
        }
    };
    unsigned fa[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int cc = j + dx;
        fa[dx] = lds0 + (unsigned)(cc * 128 + ((((cc >> 1) & 7) ^ hh) << 4));
        asm volatile("" : "+v"(fa[dx]));
    }
    unsigned ra[2];
#pragma unroll
    for (int o = 0; o < 2; ++o) ra[o] = lds0 + (unsigned)OFF_RES + (unsigned)(j * 128 + (((4 * c + 2 * o + hh) ^ ((j >> 1) & 7)) << 4));
    const unsigned lane_ob = (unsigned)(j * 128 + (32 * c + 8 * hh) * 2);
    unsigned slope2;
    {
        const half2_t s2 = {(half_t)a.slope, (half_t)a.slope};
        slope2 = __builtin_bit_cast(unsigned, s2);
    }
    const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    float16_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = zero16;
    half8_t fx[3];
    u4_t rh[2];
    unsigned sh[4];

    while (item < item_end) {
        const int s0 = item % nyb;
        const int t_ = item / nyb;
        const int pxi = t_ % px, b = t_ / px;
        const int s1 = min(nyb, s0 + (item_end - item));
        item += s1 - s0;
        const int x0 = pxi * TW;
        const int ya = RB * s0, yb = RB * s1;
        const int nblk = (yb - ya) / RB + 2;
        const unsigned vo = (x0 + j < W) ? lane_ob : kOOR;

        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (PLANEW && b != wplane) { load_weights(b); wplane = b; }
        piece_table(x0);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int yr = ya - 1 + RB * kb;
#pragma unroll
            for (int i = 0; i < NP; ++i)
                piece_issue(i, piece_word(i), kb, yr, x0, b, (unsigned)yr < (unsigned)H, (unsigned)(yr + 1) < (unsigned)H, (unsigned)(yr - 2) < (unsigned)H, (unsigned)(yr - 1) < (unsigned)H);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        fx[0] = *(lds_h8_t)(fa[0]);
        fx[1] = *(lds_h8_t)(fa[0] ^ 32u);
        int xblk = 0;

        auto block = [&](int k, auto BUF_) __attribute__((always_inline)) {
            constexpr int BUF = decltype(BUF_)::value;
            const int Rk = ya - 1 + RB * k;
            const int xnext = xblk + 1 == NRING ? 0 : xblk + 1;
            const int xnext2 = xnext + 1 == NRING ? 0 : xnext + 1;
            const bool live = RB * (k + 2) <= yb - ya + 3;
            const int yrn = Rk + 2 * RB;
            const bool nok0 = live & ((unsigned)yrn < (unsigned)H), nok1 = live & ((unsigned)(yrn + 1) < (unsigned)H);
            const bool nrk0 = live & ((unsigned)(yrn - 2) < (unsigned)H), nrk1 = live & ((unsigned)(yrn - 1) < (unsigned)H);

            auto step = [&](auto E_) __attribute__((always_inline)) {
                constexpr int e = decltype(E_)::value;
                constexpr int T4 = 2 * BUF + e;
                const int r = Rk + e;
                const unsigned xo_cur = (unsigned)__builtin_amdgcn_readfirstlane(xblk * BLKB + e * ROWB);
                const unsigned xo_nxt = (unsigned)__builtin_amdgcn_readfirstlane(e == 0 ? xblk * BLKB + ROWB : xnext * BLKB);
                constexpr int S = T4 & 3;
                const int orow = r - 2;
                const bool ook = (orow >= ya) & (orow < yb);
                const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)(ook ? (unsigned)((b * H + orow) * W + x0) * 128u : kOOR));
                const unsigned ro = (unsigned)__builtin_amdgcn_readfirstlane(xblk * RESBLKB + e * RESROWB);

                auto op_dma = [&](auto I_, auto HALF_) __attribute__((always_inline)) {
                    constexpr int i = decltype(I_)::value, half = decltype(HALF_)::value;
                    if constexpr (half == 0) dt[i & 1] = piece_word(i);
                    else piece_issue(i, dt[i & 1], xnext2, yrn, x0, b, nok0, nok1, nrk0, nrk1);
                };
                auto op_rhi = [&](auto O_) __attribute__((always_inline)) {
                    constexpr int o = decltype(O_)::value;
                    rh[o] = *(lds_u4_t)(ra[o] + ro);
                };
                auto op_act = [&](auto O_, auto K0_) __attribute__((always_inline)) {      // channel pairs k0, k0 + 1 of slot o -> sh[k0], sh[k0 + 1]
                    constexpr int o = decltype(O_)::value, k0 = decltype(K0_)::value;
#pragma unroll
                    for (int k = k0; k < k0 + 2; ++k) {
                        if (!RES) {      // conv3x3_rw.hip EPI 1: LeakyReLU on packed halves
                            const half2_t pr = {(half_t)acc[S][8 * o + 2 * k], (half_t)acc[S][8 * o + 2 * k + 1]};
                            const half2_t t = pr * __builtin_bit_cast(half2_t, slope2);
                            sh[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(pr, t));
                        } else {         // conv3x3_sp.hip EPI 6: LeakyReLU in fp32, + the residual, one rounding
                            float v0 = acc[S][8 * o + 2 * k], v1 = acc[S][8 * o + 2 * k + 1];
                            v0 = __builtin_fmaxf(v0, v0 * a.slope); v1 = __builtin_fmaxf(v1, v1 * a.slope);
                            v0 = mix_lo(rh[o][k], 1.0f, v0); v1 = mix_hi(rh[o][k], 1.0f, v1);
                            const half2_t pr = {(half_t)v0, (half_t)v1};
                            sh[k] = __builtin_bit_cast(unsigned, pr);
                        }
                    }
                };
                auto op_st = [&](auto O_) __attribute__((always_inline)) {
                    constexpr int o = decltype(O_)::value;
                    const u4_t dh = {sh[0], sh[1], sh[2], sh[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(dh, ryh, vo + (unsigned)(o * 32), so, 0);
                };

                constexpr OpList L = step_ops(RES), LD = dma_ops(e, NP);
                auto chunk = [&](auto A_) __attribute__((always_inline)) {
                    constexpr int ai = decltype(A_)::value;
                    constexpr int dx = ai / 4, ks = ai % 4;
                    if (ai == 10 && e == 0) {
                        // the block after this one has landed (its pieces and everything older; the stores behind the last piece may still be on their way)
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(vm_behind(RES)) : "memory");
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
                    {
                        constexpr int dlo = ai < 8 ? ai * LD.n / 8 : LD.n, dhi = ai < 8 ? (ai + 1) * LD.n / 8 : LD.n;
                        auto rund = [&](auto I_) __attribute__((always_inline)) {
                            constexpr int I = decltype(I_)::value;
                            if constexpr (I >= dlo && I < dhi) {
                                constexpr Op o = LD.op[I];
                                op_dma(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                            }
                        };
#define S64_OP(I) rund(std::integral_constant<int, I>{});
                        S64_OP(0) S64_OP(1) S64_OP(2) S64_OP(3) S64_OP(4) S64_OP(5) S64_OP(6) S64_OP(7) S64_OP(8) S64_OP(9) S64_OP(10) S64_OP(11) S64_OP(12) S64_OP(13) S64_OP(14) S64_OP(15)
                        S64_OP(16) S64_OP(17) S64_OP(18) S64_OP(19)
#undef S64_OP
                        constexpr int lo_ = ai * L.n / 12, hi_ = (ai + 1) * L.n / 12;
                        auto run = [&](auto I_) __attribute__((always_inline)) {
                            constexpr int I = decltype(I_)::value;
                            if constexpr (I >= lo_ && I < hi_) {
                                constexpr Op o = L.op[I];
                                if constexpr (o.kind == OP_RHI) op_rhi(std::integral_constant<int, o.a>{});
                                if constexpr (o.kind == OP_ACT) op_act(std::integral_constant<int, o.a>{}, std::integral_constant<int, o.b>{});
                                if constexpr (o.kind == OP_ST) op_st(std::integral_constant<int, o.a>{});
                            }
                        };
#define S64_OP(I) run(std::integral_constant<int, I>{});
                        S64_OP(0) S64_OP(1) S64_OP(2) S64_OP(3) S64_OP(4) S64_OP(5) S64_OP(6) S64_OP(7) S64_OP(8) S64_OP(9) S64_OP(10) S64_OP(11)
#undef S64_OP
                    }
#ifndef S64_NOPIN
#pragma unroll
                    for (int i_ = 0; i_ < 3; ++i_) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i_ == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x006, S64_FILL, 0);
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                };
#define S64_CHUNK(A) chunk(std::integral_constant<int, A>{});
                S64_CHUNK(0) S64_CHUNK(1) S64_CHUNK(2) S64_CHUNK(3) S64_CHUNK(4) S64_CHUNK(5) S64_CHUNK(6) S64_CHUNK(7) S64_CHUNK(8) S64_CHUNK(9) S64_CHUNK(10) S64_CHUNK(11)
#undef S64_CHUNK
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
#endif
}

template <int EPI>
hipError_t set_limit() { return hipFuncSetAttribute((const void*)conv64_s_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, EPI == 6 ? LDS_RES : LDS_PLAIN); }

}  // namespace

hipError_t conv64_s_init() { return set_limit<6>(); }

// false: not this kernel's layer (the caller uses conv3x3_sp<6>)
bool launch_conv64_s(const ConvArgs& a, int max_groups, hipStream_t s)
{
    if (a.r != 1 || a.in_cs != 64 || a.out_cs != 64 || a.out_lo || a.in_lo || a.res_lo || a.acc_mode != 0 || a.tplanes || a.tail1_w || a.pool || a.scale != 1.f || a.dbg) return false;
    if (!(a.slope <= 1.f) || a.H % RB != 0 || a.H < RB) return false;
    if ((long long)a.B * a.H * a.W * 128 + (long long)((RB + 1) * a.W + 2) * 128 >= (1ll << 32) - 65536) return false;
    if (!a.plane_w || !a.res || a.nchunks != a.B) return false;      // (EPI 1 -- shared weights, no residual -- is not instantiated: see the head of the file)
    if (!a.in || !a.out || !a.wpk || a.in == a.out) return false;
    const int px = (a.W + TW - 1) / TW;
    const long long items = (long long)a.B * px * (a.H / RB);
    if (items >= (1ll << 31) / 4) return false;
    const int G = (int)std::min<long long>(items, 2ll * max_groups);
    conv64_s_kernel<6><<<dim3(G), dim3(128), LDS_RES, s>>>(a);
    return true;
}
