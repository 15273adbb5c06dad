// arsb_pc.hip -- one ARSB (python/models.py:76-80 of the reference),  y = x + s * conv_2(PReLU(conv_1(x))),  as ONE kernel, second form:
// the two convs run on DIFFERENT waves of a workgroup, pipelined over patches ("producer / consumer").
//
// arsb_fused.hip (first form) gives every wave 16 output channels of both convs; its weights fit the registers only as A fragments of
// v_mfma_f32_16x16x32_f16, and beside those 16-cycle MFMAs a single wave per SIMD hides only ~2 other instructions: measured 53 % MFMA
// busy, 19-20 k cycles per patch against 10.4 k of matrix work.  Here
//
//   waves 0, 1  (P)  own conv_1: m channels 32c .. 32c+31 for every pixel of the patch          36 A fragments of
//   waves 2, 3  (C)  own conv_2: output channels 32c .. 32c+31, plus residual and stores         v_mfma_f32_32x32x16_f16 = 144 registers
//
// so every MFMA is the 32-cycle shape (7 issue slots beside it) and there are half as many of them.  Both roles stream input rows: a B
// fragment (16 channels x 32 pixels, one ds_read_b128) feeds the three output rows it touches.  The workgroup advances in lock-step
// STEPS, one input row per step and role, twelve steps per patch, one s_barrier per step:
//
//   step (p, s)   P: x_p row s  ->  m_p rows s, s-1, s-2 (36 MFMAs);  m_p row s-3 finished (PReLU, zero padding, fp16) into the m ring
//                 C: m_q row j -> out_q rows j, j-1, j-2 (36 MFMAs),  j = (s + 7) mod 12, q = p for s >= 5 else p - 1;  out row j-3 finished
//                    (+ residual hi [+ lo 2^-11] in fp32, hi [and lo] stored);  residual of row j fetched;  DMA of x_{p+1} row s issued
//
// C lags P by five steps, so the m ring needs ONE slot per row: m_p[j] is written at step (p, j+3) and read at (p, j+4); the slot's previous
// content m_{p-1}[j] was read eleven steps earlier.  The x ring likewise: P reads x_p[s] into registers during step s-1, the DMA of
// x_{p+1}[s] is issued at step s and needed at step (p+1, s-1).  LDS: 12 x 40 pixels (a row = five 1-KiB DMA pieces) + 10 x 34 pixels
// + a dummy piece = 105 KiB.  P issues no vector-memory operation at all; C never writes LDS -- so the compiler's conservative
// ordering of LDS stores behind LDS-DMA (arsb_fused.hip, cost 1) cannot arise.  Before each barrier C waits with a hand-counted
// vmcnt (every step issues the same number of operations, live or not) until everything older than two steps has completed -- the DMA
// pieces P is about to read among them -- without ever waiting for the DMA it has just issued.
#include "common.h"
#include "rowtile.h"

#ifndef PC_DBG
#define PC_DBG 0      // fault bisection builds: 1 no residual loads / stores, 2 no in-loop DMA, 4 no prologue DMA, 8 no m writes
#endif

namespace {

constexpr int TW = 30, TH = 8;                 // stored outputs per patch
constexpr int XROWS = 12, XPITCH = 40;         // x ring: rows -2 .. 9 of the patch, 40-pixel pitch (columns -2 .. 31 used)
constexpr int MROWS = 10, MPITCH = 34;         // m ring: rows -1 .. 8, columns -1 .. 30 (+ 2 pad columns read by conv_2's discarded columns 30, 31)
constexpr int XBYTES = XROWS * XPITCH * 128;   // 61,440
constexpr int MBYTES = MROWS * MPITCH * 128;   // 43,520
constexpr int LDS_BYTES = XBYTES + MBYTES + 1024;   // + one dummy DMA piece: 105,984

__device__ __forceinline__ void dma16(unsigned long long src, char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct Item { int b, pyi, pxi; };

// vector-memory operations a C wave issues in step s (the same whether the step is live or not): 3 DMA pieces, 2 (+2) residual loads while
// rows 0..7 are fetched, 2 (+2) stores while rows 0..7 are drained
template <bool LO> constexpr int c_ops(int s)
{
    const int j = (s + 7) % 12;
    return 3 + ((j <= 7) ? (LO ? 4 : 2) : 0) + ((j >= 3 && j <= 10) ? (LO ? 4 : 2) : 0);
}

// s_waitcnt vmcnt(keep) lgkmcnt(0) with `keep` an immediate: the value is a compile-time constant at every call site after unrolling
__device__ __forceinline__ void wait_keep(int keep)
{
    switch (keep) {
#define PC_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory"); break;
        PC_W(6) PC_W(8) PC_W(10) PC_W(12) PC_W(14) PC_W(18) PC_W(22)
#undef PC_W
        default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
    }
}

template <bool LO>
__global__ __launch_bounds__(256) void arsb_pc_kernel(ArsbArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xring = smem;
    char* const mring = smem + XBYTES;
    char* const dummy = smem + XBYTES + MBYTES;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = w4 >> 1, c = w4 & 1;                 // 0 = P (conv_1), 1 = C (conv_2); channel half
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

    // ---- weights: this wave's 32 output channels of its conv, 36 A fragments (tap, k-step), resident in (accumulator) registers.
    // Engine order (pack_conv): fragment (tap * 4 + ks) * 2 + nblk, lane l = W[32 nblk + (l & 31)][16 ks + 8 (l >> 5) + e][tap]
    half8_t wf[36];
    {
        const half_t* wsrc = role == 0 ? a.w1 : a.w2;
#pragma unroll
        for (int f = 0; f < 36; ++f) wf[f] = *(const half8_t*)(wsrc + ((f * 2 + c) * 64 + lane) * 8);
#pragma unroll
        for (int f = 0; f < 36; ++f) asm volatile("" : "+a"(wf[f]));
    }

    // ---- LDS read addressing of a B fragment (16 channels x 32 pixels): pixel (row, col) at (row * pitch + col) * 128, 16-byte slot s of a
    // pixel stored at s ^ ((col >> 1) & 7); lane (j, hh) reads slot 2 ks + hh of column j + dx
    int Ad[3], Zd[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int cc = j + dx;
        const int z = (cc >> 1) & 7;
        Ad[dx] = cc * 128 + ((hh ^ (z & 1)) << 4);
        Zd[dx] = (z >> 1) << 5;
    }
    const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    half8_t fr[2][12];
#define PC_READ_ROW(BUF, BASE, PITCH, ROW)                                                               \
    _Pragma("unroll") for (int f_ = 0; f_ < 12; ++f_)                                                    \
        fr[BUF][f_] = *(const half8_t*)((BASE) + (ROW) * ((PITCH) * 128) + Ad[f_ >> 2] + (((f_ & 3) << 5) ^ Zd[f_ >> 2]));
#define PC_PIN(NMFMA, NREAD, NVALU)                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < (NMFMA); ++i_) {                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                               \
        if (i_ < (NREAD)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                             \
        __builtin_amdgcn_sched_group_barrier(0x002, (NVALU), 0);                                         \
    }

    // ---- DMA of one x-ring row: five 1-KiB pieces, piece k = columns 8k .. 8k+7 of the 40-pixel row (34 used) -----------------------
    const unsigned long long zsrc = (unsigned long long)(a.zero + (lane & 7) * 8);
    auto issue_piece = [&](const Item& it, int row, int k, bool live) {
        const int cc = 8 * k + (lane >> 3);                            // column in the row, image column = x0 - 2 + cc
        const int y = it.pyi * TH - 2 + row, x = it.pxi * TW - 2 + cc;
        const int sl = (lane & 7) ^ ((cc >> 1) & 7);                   // logical slot behind this lane's physical slot
        const bool ok = ((unsigned)y < (unsigned)a.H) & ((unsigned)x < (unsigned)a.W) & (cc < 34) & live;
        unsigned off = (unsigned)(((it.b * a.H + y) * a.W + x) * 128 + sl * 16);        // 32-bit byte offset (range checked by the launcher)
        asm volatile("" : "+v"(off));
        unsigned long long src = (unsigned long long)a.x_hi + off;
        asm volatile("" : "+v"(src));
        src = ok ? src : zsrc;
        asm volatile("" : "+v"(src));
        dma16(src, xring + (row * XPITCH + 8 * k) * 128);
    };

    {   // prologue: all of x_0 (60 pieces, 15 per wave), published by a barrier
        const Item it0 = decode(g);
#pragma unroll
        for (int i = 0; i < ((PC_DBG & 4) ? 0 : 15); ++i) {
            const int n = i * 4 + w4;                                  // 0 .. 59
            issue_piece(it0, n / 5, n % 5, true);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    if (role == 0) {
        // =================================== P: conv_1 =====================================================================
        const int zj = (j >> 1) & 7;
        int mw[4];                    // m write addressing: registers 4gq .. 4gq+3 = channels 32c + 8gq + 4hh .. +3 of pixel j: slot 4c + gq, byte 8hh
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) mw[gq] = j * 128 + (((4 * c + gq) ^ zj) << 4) + 8 * hh;
        unsigned slope2;
        {
            const h2_t s2 = {(half_t)a.slope, (half_t)a.slope};
            slope2 = __builtin_bit_cast(unsigned, s2);
        }
        // epilogue of a finished m row: PReLU on packed halves (slope <= 1), zeros outside the image (conv_2's padding), four 8-byte LDS writes
        auto m_row = [&](const float16_t& ac, int mr, int y0, bool inx) {
            const bool in = inx & ((unsigned)(y0 - 1 + mr) < (unsigned)a.H);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                unsigned hv[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const h2_t pr = {(half_t)ac[4 * gq + 2 * k], (half_t)ac[4 * gq + 2 * k + 1]};
                    unsigned u = __builtin_bit_cast(unsigned, pr), t;
                    asm("v_pk_mul_f16 %0, %1, %2" : "=v"(t) : "v"(u), "v"(slope2));
                    asm("v_pk_max_f16 %0, %1, %2" : "=v"(u) : "v"(u), "v"(t));
                    hv[k] = in ? u : 0u;
                }
                const unsigned long long pk = ((unsigned long long)hv[1] << 32) | hv[0];
                if (!(PC_DBG & 8)) asm volatile("ds_write_b64 %0, %1" ::"v"(lds0 + (unsigned)XBYTES + (unsigned)(mr * (MPITCH * 128)) + (unsigned)mw[gq]), "v"(pk));
            }
        };
        float16_t acc[10];
        int y0_prev = 0;
        bool inx_prev = false;
        PC_READ_ROW(0, xring, XPITCH, 0)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // row 0 is in registers before the consumers may refill its slot
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        for (int p = 0; p < K; ++p) {
            const Item it = decode(g + p * G);
            const int y0 = it.pyi * TH, x0 = it.pxi * TW;
            const bool inx = (unsigned)(x0 - 1 + j) < (unsigned)a.W;
#pragma unroll
            for (int s = 0; s < 12; ++s) {
                PC_READ_ROW((s + 1) & 1, xring, XPITCH, (s + 1) % 12)     // x_p row s+1; at s = 11: row 0 of x_{p+1} (landed eleven steps ago)
#pragma unroll
                for (int f = 0; f < 12; ++f) {
                    const int dx = f >> 2, ks = f & 3;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int mr = s - dy;
                        if (mr >= 0 && mr < 10)
                            acc[mr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[(dy * 3 + dx) * 4 + ks], fr[s & 1][f], (dy == 0 && f == 0) ? zero16 : acc[mr], 0, 0, 0);
                    }
                }
                if (s >= 3) m_row(acc[s - 3], s - 3, y0, inx);             // complete since the end of the previous step
                if (s == 0) m_row(acc[9], 9, y0_prev, inx_prev);           // row 9 of the previous patch (first period: a dummy, overwritten at (1, 0))
                PC_PIN((s == 0 || s == 11) ? 12 : ((s == 1 || s == 10) ? 24 : 36), 12, 2)
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this step's fragment reads and m writes are done
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            y0_prev = y0; inx_prev = inx;
        }
        // drain period: row 9 of the last patch, then keep the consumers' barriers company
        m_row(acc[9], 9, y0_prev, inx_prev);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    } else {
        // =================================== C: conv_2 + residual + stores; DMA of the next patch ==================================
        const RowConsts kc = {1.0f, 0.00048828125f, -2048.f};
        // global addressing in the 16-byte store layout: lane (j, 0) channels 32c + 16gp .. +7, lane (j, 1) +8 .. +15
        const unsigned lane_ob = ((unsigned)j * 64u + (unsigned)(32 * c + 8 * hh)) * 2u;
        const unsigned trash_ob = (unsigned)a.B * a.H * a.W * 128u + lane * 16u;
        float16_t acc[8];
        uint4 resw[4][2], sidew[4][2];        // residual ring: rows j & 3, slices gp
        Item itq = decode(g), itprev = itq;   // patch q of the steps s >= 5 (= p) and of the steps s < 5 (= p - 1)
        __builtin_amdgcn_s_barrier();         // (pairs with the producers' barrier behind their first fragment reads)
        asm volatile("" ::: "memory");
        for (int p = 0; p <= K; ++p) {
            const bool live_p = p < K, live_prev = p >= 1;
            const Item itn = (p + 1 < K) ? decode(g + (p + 1) * G) : itq;
            const bool live_n = p + 1 < K;
#pragma unroll
            for (int s = 0; s < 12; ++s) {
                const int jr = (s + 7) % 12;                         // m row this step multiplies (10, 11: none)
                const Item& it = s >= 5 ? itq : itprev;
                const bool live = s >= 5 ? live_p : live_prev;
                const int y0 = it.pyi * TH, x0 = it.pxi * TW;
                const unsigned rowb = ((unsigned)(it.b * a.H + y0) * (unsigned)a.W + (unsigned)x0) * 128u;
                const bool okx = (j < TW) & (x0 + j < a.W) & live;
                auto row_off = [&](int o, int gp) {
                    const bool ok = okx & (y0 + o < a.H);
                    unsigned off = ok ? rowb + (unsigned)o * (unsigned)a.W * 128u + lane_ob + (unsigned)gp * 32u : trash_ob;
                    asm volatile("" : "+v"(off));
                    return off;
                };
                // fragments of the next m row (row 0 during the idle step jr = 11; published by the previous barrier)
                if (jr <= 8) { PC_READ_ROW((jr + 1) & 1, mring, MPITCH, jr + 1) }
                if (jr == 11) { PC_READ_ROW(0, mring, MPITCH, 0) }
                if (jr <= 9) {
#pragma unroll
                    for (int f = 0; f < 12; ++f) {
                        const int dx = f >> 2, ks = f & 3;
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int o = jr - dy;
                            if (o >= 0 && o < 8)
                                acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[(dy * 3 + dx) * 4 + ks], fr[jr & 1][f], (dy == 0 && f == 0) ? zero16 : acc[o], 0, 0, 0);
                        }
                    }
                }
                // DMA of x_{p+1} row s (its slot was read into P's registers during the previous step): three pieces per wave, the sixth is a dummy
#pragma unroll
                for (int k3 = 0; k3 < ((PC_DBG & 2) ? 0 : 3); ++k3) {
                    const int k = c * 3 + k3;
                    if (k < 5) issue_piece(itn, s, k, live_n);
                    else dma16(zsrc, dummy);
                }
                if (jr <= 7 && !(PC_DBG & 1)) {        // residual of output row jr (drained three steps from now)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const unsigned off = row_off(jr, gp);
                        // (ordinary loads: an inline-asm load hides the in-flight destination registers from the register allocator -- it
                        // re-used them for addresses and the late data turned stores into wild writes)
                        resw[jr & 3][gp] = *(const uint4*)((const char*)a.x_hi + off);
                        if (LO) sidew[jr & 3][gp] = *(const uint4*)((const char*)a.x_lo + off);
                    }
                }
                if (jr >= 3 && jr <= 10 && !(PC_DBG & 1)) {      // output row o = jr - 3 is complete since the end of the previous step
                    const int o = jr - 3;
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        float v[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = acc[o][gp * 8 + e];     // channels 32c + 16gp + {0..3 | 8..11} + 4hh
                        if (LO) {
                            const uint4 w = sidew[o & 3][gp];
                            const auto qx = __builtin_amdgcn_permlane32_swap(w.x, w.z, false, false);
                            const auto qy = __builtin_amdgcn_permlane32_swap(w.y, w.w, false, false);
                            v[0] = mix_lo(qx[0], kc.lowscale, v[0]); v[1] = mix_hi(qx[0], kc.lowscale, v[1]);
                            v[2] = mix_lo(qy[0], kc.lowscale, v[2]); v[3] = mix_hi(qy[0], kc.lowscale, v[3]);
                            v[4] = mix_lo(qx[1], kc.lowscale, v[4]); v[5] = mix_hi(qx[1], kc.lowscale, v[5]);
                            v[6] = mix_lo(qy[1], kc.lowscale, v[6]); v[7] = mix_hi(qy[1], kc.lowscale, v[7]);
                        }
                        {
                            const uint4 w = resw[o & 3][gp];
                            const auto rx = __builtin_amdgcn_permlane32_swap(w.x, w.z, false, false);
                            const auto ry = __builtin_amdgcn_permlane32_swap(w.y, w.w, false, false);
                            v[0] = mix_lo(rx[0], kc.one, v[0]); v[1] = mix_hi(rx[0], kc.one, v[1]);
                            v[2] = mix_lo(ry[0], kc.one, v[2]); v[3] = mix_hi(ry[0], kc.one, v[3]);
                            v[4] = mix_lo(rx[1], kc.one, v[4]); v[5] = mix_hi(rx[1], kc.one, v[5]);
                            v[6] = mix_lo(ry[1], kc.one, v[6]); v[7] = mix_hi(ry[1], kc.one, v[7]);
                        }
                        unsigned h0, h1, h2, h3, l0 = 0, l1 = 0, l2 = 0, l3 = 0;
                        if (LO) {
                            split2(v[0], v[1], kc.neg2048, h0, l0); split2(v[2], v[3], kc.neg2048, h1, l1);
                            split2(v[4], v[5], kc.neg2048, h2, l2); split2(v[6], v[7], kc.neg2048, h3, l3);
                        } else {
                            const h2_t p0 = {(half_t)v[0], (half_t)v[1]}, p1 = {(half_t)v[2], (half_t)v[3]}, p2 = {(half_t)v[4], (half_t)v[5]}, p3 = {(half_t)v[6], (half_t)v[7]};
                            h0 = __builtin_bit_cast(unsigned, p0); h1 = __builtin_bit_cast(unsigned, p1); h2 = __builtin_bit_cast(unsigned, p2); h3 = __builtin_bit_cast(unsigned, p3);
                        }
                        const unsigned off = row_off(o, gp);
                        {
                            const auto sx = __builtin_amdgcn_permlane32_swap(h0, h2, false, false);
                            const auto sy = __builtin_amdgcn_permlane32_swap(h1, h3, false, false);
                            *(uint4*)((char*)a.y_hi + off) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                        }
                        if (LO) {
                            const auto sx = __builtin_amdgcn_permlane32_swap(l0, l2, false, false);
                            const auto sy = __builtin_amdgcn_permlane32_swap(l1, l3, false, false);
                            *(uint4*)((char*)a.y_lo + off) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                        }
                    }
                }
                PC_PIN((jr <= 9) ? ((jr == 0 || jr == 9) ? 12 : ((jr == 1 || jr == 8) ? 24 : 36)) : 0, (jr <= 8 || jr == 11) ? 12 : 0, 4)
                __builtin_amdgcn_sched_barrier(0);
                // everything issued before the previous step has completed: residual words (used from the next step on) and DMA pieces
                // (read by P eleven steps after their issue); what this step and the previous one issued may still be in flight
                wait_keep(c_ops<LO>(s) + c_ops<LO>((s + 11) % 12));       // (a constant once the step loop is unrolled)
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            itprev = itq; itq = itn;
        }
    }
#undef PC_READ_ROW
#undef PC_PIN
}

}  // namespace

hipError_t arsb_pc_init()
{
    hipError_t e = hipFuncSetAttribute((const void*)arsb_pc_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)arsb_pc_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}

// false: the layer does not fit this kernel (caller uses arsb_fused / the two-launch form).  a.w1 / a.w2: the conv3x3_sp fragment order (w_hi).
bool launch_arsb_pc(ArsbArgs a, int max_groups, hipStream_t s)
{
    if (!(a.slope <= 1.f)) return false;                                  // PReLU as max(x, slope * x)
    if ((long long)a.B * a.H * a.W * 128 >= (1ll << 32) - 65536) return false;   // 32-bit byte offsets
    if ((a.x_lo == nullptr) != (a.y_lo == nullptr)) return false;
    a.px = (a.W + TW - 1) / TW;
    a.py = (a.H + TH - 1) / TH;
    const long long items = (long long)a.B * a.px * a.py;
    const int G = (int)std::min<long long>(items, max_groups);
    if (a.x_lo) arsb_pc_kernel<true><<<dim3(G), dim3(256), LDS_BYTES, s>>>(a);
    else arsb_pc_kernel<false><<<dim3(G), dim3(256), LDS_BYTES, s>>>(a);
    return true;
}
