// conv64_x3.hip -- a 3x3 64->64 convolution with SPLIT OPERANDS (MOE_PREC_MIXED's "exact" layers: conv_input2 and the leading
// ARSBs; python/models.py:76-80,108-123 of the reference) in ONE launch:
//
//     out = conv(w_hi, a_hi) + 2^-11 * (conv(w_lo, a_hi) + conv(w_hi, a_lo))        w = w_hi + w_lo 2^-11,  a = a_hi + a_lo 2^-11
//
// i.e. ~22-bit operands from three fp16 MFMA products.  Round 1 ran the three products as three launches of the one-pass kernel
// (side buffers through HBM: 0.23-0.27 ms per layer and 4-tile batch, 21 % of the benchmark frame for three layers).  Here the
// framework of arsb_fused.hip is reused: wave w owns output channels 16w .. 16w+15 and keeps BOTH weight parts as A fragments of
// v_mfma_f32_16x16x32_f16 in registers (2 x 18 fragments = 144 registers), LDS carries only activations in the same
// bank-conflict-free image, B fragments are row-streamed (one ds_read_b128 feeds the three output rows it touches).
//
//   pass 1  streams the ten rows of the a_hi patch:  acc_hi += w_hi a_hi,  acc_lo += w_lo a_hi      (72 MFMAs per row)
//   pass 2  streams the ten rows of the a_lo patch:  acc_lo += w_hi a_lo                             (36 MFMAs per row)
//           and finishes each output row as soon as its third a_lo row is in: v = acc_hi + acc_lo 2^-11 in fp32, then
//           EPI 0 plain | 1 PReLU (fp32, slope <= 1) | 2 + residual (hi + lo 2^-11) | 3 plain + pooled channel sums | 4 PReLU + pooled | 5 gate[plane][channel] * conv +
//           residual (lite's LB with the FRM gate known BEFORE conv_2 runs: engine.cpp, frm_pre_kernel); hi and lo parts stored, 16 bytes per lane.
//
// LDS: a_hi double buffered (its DMA for patch p+1 rides in pass 2 of patch p), a_lo single buffered: it is only live during pass 2,
// so its DMA for patch p is issued behind the barrier that opens patch p and lands while pass 1 runs.  2 x 45,056 + 45,056 B.
// A patch is 8 x 32 outputs (no halo recompute: one conv), 864 MFMAs per wave.
#include "common.h"
#include "rowtile.h"

namespace {

constexpr int TW = 32, TH = 8;
constexpr int XW = 34, XH = 10;                // input patch (halo 1)
constexpr int NPIECE_W = 11;                   // 1-KiB DMA pieces per wave: 4 x 11 = 44 >= 340 pixels / 8
constexpr int XBYTES = 4 * NPIECE_W * 1024;    // 45,056
constexpr int LDS_BYTES = 3 * XBYTES;          // 135,168

__device__ __forceinline__ void dma16(const half_t* src, char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct Item { int b, pyi, pxi; };

template <int EPI>
__global__ __launch_bounds__(256) void conv64_x3_kernel(ConvX3Args a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const hbuf = smem;                   // 2 x XBYTES: a_hi patches
    char* const lbuf = smem + 2 * XBYTES;      // a_lo patch
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;

    const int g = blockIdx.x, G = gridDim.x;
    const int nitems = a.B * a.py * a.px;
    const int K = (nitems - g + G - 1) / G;
    if (K <= 0) return;
    auto decode = [&](int item) {
        Item it;
        it.pxi = item % a.px;
        const int t = item / a.px;
        it.pyi = t % a.py;
        it.b = t / a.py;
        return it;
    };

    // ---- weights (arsb order, engine.cpp pack_arsb): hi and lo parts, 2 x 18 A fragments, parked in the accumulator registers --------
    half8_t wh[18], wl[18];
#pragma unroll
    for (int f = 0; f < 18; ++f) {
        wh[f] = *(const half8_t*)(a.w_hi + ((w4 * 18 + f) * 64 + lane) * 8);
        wl[f] = *(const half8_t*)(a.w_lo + ((w4 * 18 + f) * 64 + lane) * 8);
    }
#pragma unroll
    for (int f = 0; f < 18; ++f) {
        asm volatile("" : "+a"(wh[f]));
        asm volatile("" : "+a"(wl[f]));
    }

    // ---- DMA of a 10 x 34 patch: piece i * 4 + w4 carries pixels 8n' .. 8n'+7 in raster order ---------------------------------------
    const unsigned long long zsrc = (unsigned long long)(a.zero + (lane & 7) * 8);
    int prc[NPIECE_W];
    unsigned poff[NPIECE_W];
#pragma unroll
    for (int i = 0; i < NPIECE_W; ++i) {
        const int qq = (i * 4 + w4) * 8 + (lane >> 3);
        const int r = (qq * 241) >> 13;                   // qq / 34 for qq < 442
        const int cc = qq - r * XW;
        const int sl = (lane & 7) ^ ((cc >> 1) & 3);
        prc[i] = ((qq < XW * XH ? r : 255) << 8) | cc;
        poff[i] = (unsigned)((r * a.W + cc) * 128 + sl * 16);
    }
    auto issue_piece = [&](const half_t* tensor, const Item& it, int i, char* dstbuf, bool live) {
        const int ya = it.pyi * TH - 1, xa = it.pxi * TW - 1;
        const unsigned base = (unsigned)(((it.b * a.H + ya) * a.W + xa) * 128);
        const bool ok = ((unsigned)(ya + (prc[i] >> 8)) < (unsigned)a.H) & ((unsigned)(xa + (prc[i] & 255)) < (unsigned)a.W) & live;
        unsigned off = base + poff[i];
        asm volatile("" : "+v"(off));
        unsigned long long src = (unsigned long long)tensor + off;
        asm volatile("" : "+v"(src));
        src = ok ? src : zsrc;
        asm volatile("" : "+v"(src));
        dma16((const half_t*)src, dstbuf + (i * 4 + w4) * 1024);
    };

    // ---- LDS read addressing (same image as arsb_fused.hip): lane (n, q) reads slot (2kh + (q >> 1)) ^ 4(q & 1) of column 16cb + n + dx
    int rd[2][3][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int col = 16 * cb + n + dx;
            const int z = (col >> 1) & 3;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) rd[cb][dx][kh] = col * 128 + ((((2 * kh + (q >> 1)) ^ (4 * (q & 1))) ^ z) << 4);
        }
    const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    const RowConsts kc = {1.0f, 0.00048828125f, -2048.f};
    const int ocol = 16 * (q & 1) + n;
    const unsigned lane_ob = ((unsigned)ocol * 64u + (unsigned)(16 * w4 + 8 * (q >> 1))) * 2u;
    const unsigned trash_ob = (unsigned)a.B * a.H * a.W * 128u + lane * 16u;

    Item it_cur = decode(g);
#pragma unroll
    for (int i = 0; i < NPIECE_W; ++i) issue_piece(a.in_hi, it_cur, i, hbuf, true);    // prologue: a_hi of the first patch

    // EPI 3 = plain + pooled: this lane's partial channel sums of the plane being processed; when the workgroup moves on to another
    // plane (and at the end) the 16 pixel lanes of a channel group are added up and lane n = 0 stores the workgroup's slab
    constexpr bool POOL = EPI == 3 || EPI == 4, PRELU = EPI == 1 || EPI == 4, RES = EPI == 2 || EPI == 5, GATE = EPI == 5;
    // GATE: out = g (conv) + x = g (conv + x / g): the residual words are added into the accumulators scaled by 1 / g (gate[B*64 + ..]), the finished row is multiplied by g
    float4_t g4 = {1.f, 1.f, 1.f, 1.f}, rg4 = {1.f, 1.f, 1.f, 1.f}, rgl4 = {0.00048828125f, 0.00048828125f, 0.00048828125f, 0.00048828125f};
    float psum[4] = {0.f, 0.f, 0.f, 0.f};
    int pool_b = it_cur.b;
    auto pool_flush = [&](int b) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = psum[e];
            t += __shfl_xor(t, 8); t += __shfl_xor(t, 4); t += __shfl_xor(t, 2); t += __shfl_xor(t, 1);
            psum[e] = t;
        }
        if (n == 0) *(float4_t*)(a.pool + ((long long)b * a.pool_slabs + g) * 64 + 16 * w4 + 4 * q) = float4_t{psum[0], psum[1], psum[2], psum[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) psum[e] = 0.f;
    };

    for (int p = 0; p < K; ++p) {
        const Item it = it_cur;
        const bool has_next = p + 1 < K;
        const Item itn = has_next ? decode(g + (p + 1) * G) : it;
        const char* const hb = hbuf + (p & 1) * XBYTES;
        char* const hn = hbuf + ((p + 1) & 1) * XBYTES;
        const int y0 = it.pyi * TH, x0 = it.pxi * TW;
        if (POOL && it.b != pool_b) { pool_flush(pool_b); pool_b = it.b; }

        // a_hi[p] has landed: its eleven pieces went out in pass 2 of the previous patch, in front of the stores of that pass's output rows 2..7
        // (twelve 16-byte stores).  vmcnt counts loads and stores IN ORDER, so a plain vmcnt(0) here also waits for the acknowledgement of stores
        // issued a few hundred cycles ago -- every patch.  (First patch: the prologue's pieces, nothing behind them.)
        if (p == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // every wave is done with a_lo[p-1]
        asm volatile("" ::: "memory");
        if (GATE) {      // this plane's gate, channels 16 w4 + 4 q ..: requested here, OLDER than every load of pass 1 -- the counted waits there cover it
            g4 = *(const float4_t*)(a.gate + (long long)it.b * 64 + 16 * w4 + 4 * q);
            rg4 = *(const float4_t*)(a.gate + ((long long)a.B + it.b) * 64 + 16 * w4 + 4 * q);
        }

        half8_t fr[2][12];
        const char* pb[12];
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
        // EPI 2: the residual (hi + lo 2^-11) is added to ah[o] INSIDE PASS 1, three rows after the row's words were requested and one row after
        // its last w_hi a_hi product: pass 1 has no stores in flight, so the counted waits for these loads wait for loads only.  (Round 2 fetched
        // the words in pass 2 between the stores of finished rows: each wait then also waited for the stores -- and the next patch's DMA --
        // issued in front of the load; EPI 2 ran 28 % slower than EPI 0 / 1.)
        typedef unsigned u4v_t __attribute__((ext_vector_type(4)));
        u4v_t resw[4], sidew[4];      // residual hi / lo of the rows in flight (a ring of four)
        auto add_res = [&](float4_t (&h4)[2], const u4v_t& lo_w, const u4v_t& hi_w) {
            const auto l0 = __builtin_amdgcn_permlane16_swap(lo_w.x, lo_w.z, false, false), l1 = __builtin_amdgcn_permlane16_swap(lo_w.y, lo_w.w, false, false);
            const auto r0 = __builtin_amdgcn_permlane16_swap(hi_w.x, hi_w.z, false, false), r1 = __builtin_amdgcn_permlane16_swap(hi_w.y, hi_w.w, false, false);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                float t0 = h4[cb][0], t1 = h4[cb][1], t2 = h4[cb][2], t3 = h4[cb][3];
                if (GATE) {
                    t0 = mix_lo(l0[cb], rgl4[0], t0); t1 = mix_hi(l0[cb], rgl4[1], t1); t2 = mix_lo(l1[cb], rgl4[2], t2); t3 = mix_hi(l1[cb], rgl4[3], t3);
                    t0 = mix_lo(r0[cb], rg4[0], t0); t1 = mix_hi(r0[cb], rg4[1], t1); t2 = mix_lo(r1[cb], rg4[2], t2); t3 = mix_hi(r1[cb], rg4[3], t3);
                } else {
                    t0 = mix_lo(l0[cb], kc.lowscale, t0); t1 = mix_hi(l0[cb], kc.lowscale, t1); t2 = mix_lo(l1[cb], kc.lowscale, t2); t3 = mix_hi(l1[cb], kc.lowscale, t3);
                    t0 = mix_lo(r0[cb], kc.one, t0); t1 = mix_hi(r0[cb], kc.one, t1); t2 = mix_lo(r1[cb], kc.one, t2); t3 = mix_hi(r1[cb], kc.one, t3);
                }
                h4[cb] = float4_t{t0, t1, t2, t3};
            }
        };
        const unsigned rowb = ((unsigned)(it.b * a.H + y0) * (unsigned)a.W + (unsigned)x0) * 128u;
        const bool okx = x0 + ocol < a.W;
        auto row_off = [&](int o) {
            const bool ok = okx & (y0 + o < a.H);
            unsigned off = ok ? rowb + (unsigned)o * (unsigned)a.W * 128u + lane_ob : trash_ob;
            asm volatile("" : "+v"(off));
            return off;
        };
        float4_t ah[8][2], al[8][2];

        // ================= pass 1: a_hi rows 0 .. 9:  ah += w_hi a_hi,  al += w_lo a_hi =========================================
        MOE_SET_BASE(hb)
        MOE_READ_ROW(0, 0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int xr = 0; xr < 10; ++xr) {
            // this row's fragments (read during the previous row) have landed: no counted waits between the MFMAs.  EPI 2, rows 3..9: also the residual
            // words of output row xr-3, requested three row steps ago -- a COUNTED wait (younger: the DMA pieces and words of the two rows between),
            // placed here behind the previous row's sched_barrier: a wait in mid-row does not keep the scheduler from hoisting the words' first use
            // above it, and for a use it finds unprotected the compiler inserts vmcnt(0) while LDS-DMA pieces are in flight
            if (!RES || xr < 3) __builtin_amdgcn_s_waitcnt(0xC07F);
            else if (xr <= 5) __builtin_amdgcn_s_waitcnt(0x0078);     // vmcnt(8) lgkmcnt(0)
            else if (xr == 6) __builtin_amdgcn_s_waitcnt(0x0077);     // 7
            else if (xr == 7) __builtin_amdgcn_s_waitcnt(0x0075);     // 5
            else if (xr == 8) __builtin_amdgcn_s_waitcnt(0x0074);     // 4
            else __builtin_amdgcn_s_waitcnt(0x0072);                  // 2
            if (xr < 9) { MOE_READ_ROW((xr + 1) & 1, xr + 1) }
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
                                const half8_t bf = fr[xr & 1][(dx * 2 + kh) * 2 + cb];
                                ah[o][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(dy * 3 + dx) * 2 + kh], bf, first ? zero4 : ah[o][cb], 0, 0, 0);
                                al[o][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[(dy * 3 + dx) * 2 + kh], bf, first ? zero4 : al[o][cb], 0, 0, 0);
                            }
                        }
                    }
            if (RES && xr >= 3) {        // rows 0..6 (ah[o] is complete since row step o + 2)
                if (GATE && xr == 3) rgl4 = rg4 * 0.00048828125f;
                add_res(ah[xr - 3], sidew[(xr - 3) & 3], resw[(xr - 3) & 3]);
            }
            // a_lo of THIS patch (its buffer was freed by the barrier above): 11 pieces over the first six rows
            if (xr < 5) { issue_piece(a.in_lo, it, 2 * xr, lbuf, true); issue_piece(a.in_lo, it, 2 * xr + 1, lbuf, true); }
            if (xr == 5) issue_piece(a.in_lo, it, 10, lbuf, true);
            if (RES && xr < 8) {         // words of output row xr: their ring slot was released by the add above
                const unsigned off = row_off(xr);
                resw[xr & 3] = *(const u4v_t*)((const char*)a.res_hi + off);
                sidew[xr & 3] = *(const u4v_t*)((const char*)a.res_lo + off);
            }
            MOE_PIN_ROW(((xr < 2 || xr > 7) ? (xr == 0 || xr == 9 ? 24 : 48) : 72), (xr < 9 ? 12 : 0), 1)
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // a_lo[p] has landed for every wave
        asm volatile("" ::: "memory");

        // ================= pass 2: a_lo rows 0 .. 9:  al += w_hi a_lo;  rows finished and stored as they complete ==============
        {
            auto out_row = [&](const float4_t (&h4)[2], const float4_t (&l4)[2], int o) {
                float v[2][4];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[cb][e] = __builtin_fmaf(l4[cb][e], 0.00048828125f, h4[cb][e]);
                if (GATE) {
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[cb][e] *= g4[e];
                }
                if (PRELU) {
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = v[cb][e] * a.slope;
                            asm("v_max_f32 %0, %1, %2" : "=v"(v[cb][e]) : "v"(v[cb][e]), "v"(t));
                        }
                }
                if (POOL) {      // channel sums over the pixels inside the image (v[cb][e]: pixel x0 + 16 cb + n, channel 16 w4 + 4 q + e)
                    const bool oky = y0 + o < a.H;
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        const float m = (oky & (x0 + 16 * cb + n < a.W)) ? 1.f : 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) psum[e] = __builtin_fmaf(v[cb][e], m, psum[e]);
                    }
                }
                finish_row<false, false, true>(v, uint4{}, uint4{}, kc, (char*)a.out_hi, (char*)a.out_lo, row_off(o));
            };
            MOE_SET_BASE(lbuf)
            MOE_READ_ROW(0, 0)
            if (RES) add_res(ah[7], sidew[3], resw[3]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int xr = 0; xr < 10; ++xr) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                if (xr < 9) { MOE_READ_ROW((xr + 1) & 1, xr + 1) }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                        for (int cb = 0; cb < 2; ++cb) {
#pragma unroll
                            for (int dy = 0; dy < 3; ++dy) {
                                const int o = xr - dy;
                                if (o >= 0 && o < 8)
                                    al[o][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[(dy * 3 + dx) * 2 + kh], fr[xr & 1][(dx * 2 + kh) * 2 + cb], al[o][cb], 0, 0, 0);
                            }
                        }
                // a_hi of the NEXT patch rides here: 11 pieces over the first six rows
                if (xr < 5) { issue_piece(a.in_hi, itn, 2 * xr, hn, has_next); issue_piece(a.in_hi, itn, 2 * xr + 1, hn, has_next); }
                if (xr == 5) issue_piece(a.in_hi, itn, 10, hn, has_next);
                if (xr >= 3) out_row(ah[xr - 3], al[xr - 3], xr - 3);     // complete since the end of the previous row
                MOE_PIN_ROW(((xr < 2 || xr > 7) ? (xr == 0 || xr == 9 ? 12 : 24) : 36), (xr < 9 ? 12 : 0), 3)
                __builtin_amdgcn_sched_barrier(0);
            }
            out_row(ah[7], al[7], 7);
        }
#undef MOE_READ_ROW
#undef MOE_SET_BASE
#undef MOE_PIN_ROW
        it_cur = itn;
    }
    if (POOL) pool_flush(pool_b);
}

template <int EPI>
hipError_t set_limit()
{
    return hipFuncSetAttribute((const void*)conv64_x3_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}

}  // namespace

hipError_t conv64_x3_init()
{
    hipError_t e;
    if ((e = set_limit<0>()) != hipSuccess) return e;
    if ((e = set_limit<1>()) != hipSuccess) return e;
    if ((e = set_limit<3>()) != hipSuccess) return e;
    if ((e = set_limit<4>()) != hipSuccess) return e;
    if ((e = set_limit<5>()) != hipSuccess) return e;
    return set_limit<2>();
}

// false: the layer does not fit this kernel (caller uses the three-launch form)
bool launch_conv64_x3(ConvX3Args a, int max_groups, hipStream_t s)
{
    if (!(a.slope <= 1.f)) return false;
    if ((long long)a.B * a.H * a.W * 128 >= (1ll << 32) - 65536) return false;
    if (!a.in_hi || !a.in_lo || !a.out_hi || !a.out_lo || !a.w_hi || !a.w_lo) return false;
    if ((a.res_hi == nullptr) != (a.res_lo == nullptr)) return false;
    if (a.res_hi && a.slope != 1.f) return false;
    a.px = (a.W + TW - 1) / TW;
    a.py = (a.H + TH - 1) / TH;
    const long long items = (long long)a.B * a.px * a.py;
    const int G = a.pool ? pooled_groups((long long)a.px * a.py, items, max_groups) : (int)std::min<long long>(items, max_groups);
    const dim3 grid(G), blk(256);
    if (a.pool && (a.res_hi || a.pool_slabs < G)) return false;
    if (a.gate && !a.res_hi) return false;
    if (a.pool && a.slope != 1.f) conv64_x3_kernel<4><<<grid, blk, LDS_BYTES, s>>>(a);
    else if (a.pool) conv64_x3_kernel<3><<<grid, blk, LDS_BYTES, s>>>(a);
    else if (a.gate) conv64_x3_kernel<5><<<grid, blk, LDS_BYTES, s>>>(a);
    else if (a.res_hi) conv64_x3_kernel<2><<<grid, blk, LDS_BYTES, s>>>(a);
    else if (a.slope != 1.f) conv64_x3_kernel<1><<<grid, blk, LDS_BYTES, s>>>(a);
    else conv64_x3_kernel<0><<<grid, blk, LDS_BYTES, s>>>(a);
    return true;
}
