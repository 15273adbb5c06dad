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
// + a dummy piece + a 32-KiB ring of residual words (four rows; fetched by DMA three steps ahead of their use) = 136 KiB.  P issues no vector-memory operation at all; C never writes LDS -- so the compiler's conservative
// ordering of LDS stores behind LDS-DMA (arsb_fused.hip, cost 1) cannot arise.  Before each barrier C waits with a hand-counted
// vmcnt (every step issues the same number of operations, live or not) until everything older than two steps has completed -- the DMA
// pieces P is about to read among them -- without ever waiting for the DMA it has just issued.
#include "common.h"
#include "rowtile.h"
#include <type_traits>

#ifndef PC_DBG
#define PC_DBG 0      // fault bisection builds: 1 no residual loads / stores, 2 no in-loop DMA, 4 no prologue DMA, 8 no m writes
#endif

// cycle-level trace (-DPC_TRACE builds, tools/trace_pc.sh): s_memtime before and after every step barrier
#ifdef PC_TRACE
#define PC_STAMP(SLOT)                                                                                    \
    if (a.trace && g < 8 && p < 16 && lane == 0) a.trace[((g * 16 + p) * 4 + w4) * 40 + (SLOT)] = __builtin_amdgcn_s_memtime();
#define PC_TRACE_OPS 4
// with -DPC_SUBSTEP=<s>: a stamp behind every chunk of the consumer's step s (slots 24 .. 35)
#ifndef PC_SUBSTEP
#define PC_SUBSTEP -1
#endif
#define PC_SUB(N) if (s == PC_SUBSTEP) { PC_STAMP(24 + (N)) __builtin_amdgcn_sched_barrier(0); }
#else
#define PC_SUB(N)
#define PC_STAMP(SLOT)
#define PC_TRACE_OPS 0
#endif

namespace {

constexpr int TW = 30, TH = 8;                 // stored outputs per patch
constexpr int XROWS = 12, XPITCH = 40;         // x ring: rows -2 .. 9 of the patch, 40-pixel pitch (columns -2 .. 31 used)
constexpr int MROWS = 10, MPITCH = 34;         // m ring: rows -1 .. 8, columns -1 .. 30 (+ 2 pad columns read by conv_2's discarded columns 30, 31)
constexpr int XBYTES = XROWS * XPITCH * 128;   // 61,440
constexpr int MBYTES = MROWS * MPITCH * 128;   // 43,520
constexpr int RBYTES = 4 * 2 * 2 * 2 * 1024;   // residual ring: 4 rows x 2 consumer waves x 2 slices x (hi, lo) x one DMA piece: 32,768
constexpr int LDS_BYTES = XBYTES + MBYTES + 1024 + RBYTES;   // + one dummy DMA piece: 138,752

__device__ __forceinline__ void dma16(unsigned long long src, char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct Item { int b, pyi, pxi; };

// LOADS (LDS-DMA pieces) a C wave issues in step s, the same whether the step is live or not: 3 pieces of x, 2 (+2) of residual words while
// rows 0..7 are fetched.  The stores are deliberately not counted: loads complete in order among themselves but not relative to stores, so
// an allowance that included stores could be used up by loads still in flight.
template <bool LO> constexpr int c_ops(int s)
{
    const int j = (s + 7) % 12;
    return 3 + ((j <= 7) ? (LO ? 4 : 2) : 0);
}

// s_waitcnt vmcnt(keep) lgkmcnt(0) with `keep` an immediate: the value is a compile-time constant at every call site after unrolling
__device__ __forceinline__ void wait_keep(int keep)
{
    switch (keep) {
#define PC_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)" ::: "memory"); break;
        PC_W(5) PC_W(6) PC_W(7) PC_W(8) PC_W(9) PC_W(10) PC_W(11) PC_W(12) PC_W(14) PC_W(16) PC_W(18)
#undef PC_W
        default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
    }
}

template <bool LO>
__global__ __launch_bounds__(256) void arsb_pc_kernel(ArsbArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass only needs the launch stub; it drops the whole instantiation over the gfx950 builtins below)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xring = smem;
    char* const mring = smem + XBYTES;
    char* const dummy = smem + XBYTES + MBYTES;
    char* const rring = smem + XBYTES + MBYTES + 1024;
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
        // MFMA row i = 8q + 4hh' + e of the 32x32 result lands in register 4q + e of the lanes hh = hh'.  Row i is given output channel
        // 16 (q >> 1) + 8 hh' + 4 (q & 1) + e, so that registers 8g .. 8g+7 of lane (j, hh) are the eight CONSECUTIVE channels 16g + 8hh ..:
        // one 16-byte LDS write / global store / residual fetch per lane and g, no lane pairing (v_permlane32_swap) anywhere.
        const int wi = lane & 31, wq = wi >> 3;
        const int wl = (lane & 32) | (16 * (wq >> 1) + 8 * ((wi >> 2) & 1) + 4 * (wq & 1) + (wi & 3));
#pragma unroll
        for (int f = 0; f < 36; ++f) wf[f] = *(const half8_t*)(wsrc + ((f * 2 + c) * 64 + wl) * 8);
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
    int fa[12];                       // per fragment f = (dx, ks): lane offset inside a row; the row itself is an immediate offset of the read
#pragma unroll
    for (int f = 0; f < 12; ++f) {
        fa[f] = (int)lds0 + (role == 0 ? 0 : XBYTES) + Ad[f >> 2] + (((f & 3) << 5) ^ Zd[f >> 2]);    // (LDS byte address in this role's ring)
        asm volatile("" : "+v"(fa[f]));
    }
    typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;
    const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    half8_t fr[2][12];
#define PC_READ_ROW(BUF, BASE, PITCH, ROW)                                                               \
    _Pragma("unroll") for (int f_ = 0; f_ < 12; ++f_)                                                    \
        fr[BUF][f_] = *(lds_h8_t)(unsigned)(fa[f_] + (ROW) * ((PITCH) * 128));
// (constant trip count: with a bound that only becomes constant once the step loop is unrolled, the consumer's copy survived as an empty
// 36-iteration scalar loop -- three taken branches per iteration, 3,200 cycles per step)
#define PC_PIN(NMFMA, NREAD, NVALU)                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 36; ++i_) {                                                  \
        if (i_ < (NMFMA)) {                                                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                           \
            if (i_ < (NREAD)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                         \
            __builtin_amdgcn_sched_group_barrier(0x002, (NVALU), 0);                                     \
        }                                                                                                \
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
        int mw[2];                    // m write addressing: registers 8g .. 8g+7 = channels 32c + 16g + 8hh .. +7 of pixel j: slot 4c + 2g + hh
#pragma unroll
        for (int gq = 0; gq < 2; ++gq) mw[gq] = j * 128 + (((4 * c + 2 * gq + hh) ^ zj) << 4);
        unsigned slope2;
        {
            const h2_t s2 = {(half_t)a.slope, (half_t)a.slope};
            slope2 = __builtin_bit_cast(unsigned, s2);
        }
        // epilogue of a finished m row: PReLU on packed halves (slope <= 1), zeros outside the image (conv_2's padding), two 16-byte LDS writes
        // (inline asm: a C++ store to LDS is ordered behind the LDS-DMA in flight with vmcnt(0), arsb_fused.hip cost 1)
        typedef unsigned u4_t __attribute__((ext_vector_type(4)));
        auto m_row = [&](const float16_t& ac, int mr, int y0, bool inx) {
            const bool in = inx & ((unsigned)(y0 - 1 + mr) < (unsigned)a.H);
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                u4_t hv;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const h2_t pr = {(half_t)ac[8 * gq + 2 * k], (half_t)ac[8 * gq + 2 * k + 1]};
                    unsigned u = __builtin_bit_cast(unsigned, pr), t;
                    asm("v_pk_mul_f16 %0, %1, %2" : "=v"(t) : "v"(u), "v"(slope2));
                    asm("v_pk_max_f16 %0, %1, %2" : "=v"(u) : "v"(u), "v"(t));
                    hv[k] = in ? u : 0u;
                }
                if (!(PC_DBG & 8)) asm volatile("ds_write_b128 %0, %1" ::"v"(lds0 + (unsigned)XBYTES + (unsigned)(mr * (MPITCH * 128)) + (unsigned)mw[gq]), "v"(hv));
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
                PC_STAMP(2 * s)
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                PC_STAMP(2 * s + 1)
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
        // Every vector-memory operation of the loop is a raw-buffer operation: lane part of the address in a VGPR that is computed once per
        // patch, row part in an SGPR, slice as the instruction's immediate -- no address arithmetic beside the MFMAs.  Whatever must not
        // be fetched or stored (columns or rows outside the image, dead periods) gets an offset >= num_records: the fetch returns zeros --
        // which is conv_1's zero padding -- and the store is dropped.
        const unsigned nrec = (unsigned)a.B * a.H * a.W * 128u;       // < 2^31 (launcher): lane part + row part never wraps
        const unsigned rowbytes = (unsigned)a.W * 128u;
        const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc((void*)a.x_hi, 0, nrec, 0x00020000);
        const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc((void*)(LO ? a.x_lo : a.x_hi), 0, nrec, 0x00020000);
        const __amdgpu_buffer_rsrc_t ryh = __builtin_amdgcn_make_buffer_rsrc((void*)a.y_hi, 0, nrec, 0x00020000);
        const __amdgpu_buffer_rsrc_t ryl = __builtin_amdgcn_make_buffer_rsrc((void*)(LO ? a.y_lo : a.y_hi), 0, nrec, 0x00020000);
        typedef unsigned u4_t __attribute__((ext_vector_type(4)));
        // x-ring piece k = 3c + k3 of a row: lane = (column 8k + lane / 8, physical 16-byte slot lane % 8)
        auto lane_x = [&](const Item& it, int k3) {
            const int k = c * 3 + k3, cc = 8 * k + (lane >> 3), x = it.pxi * TW - 2 + cc;
            const int sl = (lane & 7) ^ ((cc >> 1) & 7);                   // logical slot behind this lane's physical slot
            const bool ok = ((unsigned)x < (unsigned)a.W) & (cc < 34) & (k < 5);
            return ok ? (unsigned)(x * 128 + sl * 16) : nrec;
        };
        // residual fetch / store of a row: lane (j, hh) = channels 32c + 16gp + 8hh .. +7 of pixel x0 + j (gp: immediate offset 32 gp)
        auto lane_r = [&](const Item& it) {
            const int x = it.pxi * TW + j;
            return ((j < TW) & (x < a.W)) ? (unsigned)(x * 128 + (32 * c + 8 * hh) * 2) : nrec;
        };
        // SGPR part: start of image row y0 + row of a plane, `base` = byte offset of row y0 (computed once per patch); branch-free select
        auto row_s = [&](unsigned base, int y0, int row, bool live) {
            const unsigned m = 0u - (unsigned)(live & ((unsigned)(y0 + row) < (unsigned)a.H));
            return ((base + (unsigned)row * rowbytes) & m) | (nrec & ~m);
        };
        auto base_of = [&](const Item& it) { return (unsigned)(it.b * a.H + it.pyi * TH) * rowbytes; };
        float16_t acc[8];
        Item itq = decode(g), itprev = itq;   // patch q of the steps s >= 5 (= p) and of the steps s < 5 (= p - 1)
        unsigned vr_q = lane_r(itq), vr_prev = vr_q;
        unsigned bs_q = base_of(itq), bs_prev = bs_q;
        __builtin_amdgcn_s_barrier();         // (pairs with the producers' barrier behind their first fragment reads)
        asm volatile("" ::: "memory");
        for (int p = 0; p <= K; ++p) {
            const bool live_p = p < K, live_prev = p >= 1;
            const Item itn = (p + 1 < K) ? decode(g + (p + 1) * G) : itq;
            const bool live_n = p + 1 < K;
            unsigned vx[3];
#pragma unroll
            for (int k3 = 0; k3 < 3; ++k3) vx[k3] = lane_x(itn, k3);
            const unsigned vr_n = lane_r(itn), bs_n = base_of(itn);
            // one step, `s` a compile-time constant (a generic lambda over integral_constant: as a `#pragma unroll` loop the body was too big
            // to be unrolled reliably, and everything below -- register arrays, wait counts, the scheduling pattern -- needs constant s)
            auto step = [&](auto S_) __attribute__((always_inline)) {
                constexpr int s = decltype(S_)::value;
                constexpr int jr = (s + 7) % 12;                     // m row this step multiplies (10, 11: none)
                const Item& it = s >= 5 ? itq : itprev;
                const bool live = s >= 5 ? live_p : live_prev;
                const unsigned vr = s >= 5 ? vr_q : vr_prev, bs = s >= 5 ? bs_q : bs_prev;
                const int y0 = it.pyi * TH;
                // The step is written as twelve CHUNKS, one per B fragment f of m row jr: its (up to) three MFMAs, the read of fragment f of the
                // next m row (row 0 during the idle step jr = 11; published by the previous barrier), and one PIECE of the step's other work --
                // about twenty instructions for the ~21 issue slots beside three 32-cycle MFMAs:
                //   0..2  DMA piece k3 of x_{p+1} row s (its slot was read into P's registers during the previous step; the sixth piece is a dummy)
                //   3     residual loads of output row jr (drained three steps from now)
                //   4..6, 7..9   output row o = jr - 3 (complete since the end of the previous step), slices gp = 0, 1:
                //                accumulators + low residual | high residual + split | lane pairing + stores
                // A sched_barrier closes every chunk, a sched_group_barrier pattern orders it inside.
                constexpr bool has_epi = jr >= 3 && jr <= 10 && !(PC_DBG & 1);
                constexpr int o = jr - 3;
                float v[2][8];
                unsigned hw[2][4], lw[2][4];
                uint4 rw[2][2];                                      // residual words of row o: slice gp, {hi, lo}
                auto chunk = [&](auto F_) __attribute__((always_inline)) {
                    constexpr int f = decltype(F_)::value;
                    constexpr int dx = f >> 2, ks = f & 3;
                    int nm = 0;
                    constexpr bool rd_chunk = has_epi && f == 3;      // this chunk also fetches the residual words of row o from the LDS ring
                    if constexpr (rd_chunk) {
                        // One asm statement: the ring reads, this chunk's MFMAs, lgkmcnt(0).  As C++ reads the compiler put vmcnt(0) in front
                        // of them (they do alias LDS-DMA writes; that those are long complete is only known to wait_keep below); inside one
                        // statement the destination registers are defined only behind the wait, and the LDS latency hides under the MFMAs.
                        constexpr int so = (o & 3) * 8192;
                        const unsigned ra = lds0 + (unsigned)(XBYTES + MBYTES + 1024) + (unsigned)(c * 4096 + lane * 16);
#define PC_RD_HI "ds_read_b128 %[r0], %[ra] offset:%[k0]\n ds_read_b128 %[r1], %[ra] offset:%[k1]\n"
#define PC_RD_LO "ds_read_b128 %[r2], %[ra] offset:%[k2]\n ds_read_b128 %[r3], %[ra] offset:%[k3]\n"
#define PC_MF(N) "v_mfma_f32_32x32x16_f16 %[c" #N "], %[w" #N "], %[b], %[c" #N "]\n"
#define PC_OUT_HI [r0] "=&v"(rw[0][0]), [r1] "=&v"(rw[1][0])
#define PC_OUT_LO [r2] "=&v"(rw[0][1]), [r3] "=&v"(rw[1][1])
#define PC_IN_HI [ra] "v"(ra), [k0] "n"(so), [k1] "n"(so + 2048), [b] "v"(fr[jr & 1][f])
#define PC_IN_LO [k2] "n"(so + 1024), [k3] "n"(so + 3072)
#define PC_W(DY) "a"(wf[((DY) * 3 + dx) * 4 + ks])
                        if constexpr (jr <= 7) {
                            if (LO) asm volatile(PC_RD_HI PC_RD_LO PC_MF(0) PC_MF(1) PC_MF(2) "s_waitcnt lgkmcnt(0)"
                                                 : PC_OUT_HI, PC_OUT_LO, [c0] "+a"(acc[jr]), [c1] "+a"(acc[jr - 1]), [c2] "+a"(acc[jr - 2])
                                                 : PC_IN_HI, PC_IN_LO, [w0] PC_W(0), [w1] PC_W(1), [w2] PC_W(2));
                            else asm volatile(PC_RD_HI PC_MF(0) PC_MF(1) PC_MF(2) "s_waitcnt lgkmcnt(0)"
                                              : PC_OUT_HI, [c0] "+a"(acc[jr]), [c1] "+a"(acc[jr - 1]), [c2] "+a"(acc[jr - 2])
                                              : PC_IN_HI, [w0] PC_W(0), [w1] PC_W(1), [w2] PC_W(2));
                        } else if constexpr (jr == 8) {
                            if (LO) asm volatile(PC_RD_HI PC_RD_LO PC_MF(1) PC_MF(2) "s_waitcnt lgkmcnt(0)"
                                                 : PC_OUT_HI, PC_OUT_LO, [c1] "+a"(acc[7]), [c2] "+a"(acc[6]) : PC_IN_HI, PC_IN_LO, [w1] PC_W(1), [w2] PC_W(2));
                            else asm volatile(PC_RD_HI PC_MF(1) PC_MF(2) "s_waitcnt lgkmcnt(0)"
                                              : PC_OUT_HI, [c1] "+a"(acc[7]), [c2] "+a"(acc[6]) : PC_IN_HI, [w1] PC_W(1), [w2] PC_W(2));
                        } else if constexpr (jr == 9) {
                            if (LO) asm volatile(PC_RD_HI PC_RD_LO PC_MF(2) "s_waitcnt lgkmcnt(0)" : PC_OUT_HI, PC_OUT_LO, [c2] "+a"(acc[7]) : PC_IN_HI, PC_IN_LO, [w2] PC_W(2));
                            else asm volatile(PC_RD_HI PC_MF(2) "s_waitcnt lgkmcnt(0)" : PC_OUT_HI, [c2] "+a"(acc[7]) : PC_IN_HI, [w2] PC_W(2));
                        } else {
                            if (LO) asm volatile(PC_RD_HI PC_RD_LO "s_waitcnt lgkmcnt(0)" : PC_OUT_HI, PC_OUT_LO : PC_IN_HI, PC_IN_LO);
                            else asm volatile(PC_RD_HI "s_waitcnt lgkmcnt(0)" : PC_OUT_HI : PC_IN_HI);
                        }
#undef PC_RD_HI
#undef PC_RD_LO
#undef PC_MF
#undef PC_OUT_HI
#undef PC_OUT_LO
#undef PC_IN_HI
#undef PC_IN_LO
#undef PC_W
                    } else if (jr <= 9) {
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int oo = jr - dy;
                            if (oo >= 0 && oo < 8) {
                                acc[oo] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[(dy * 3 + dx) * 4 + ks], fr[jr & 1][f], (dy == 0 && f == 0) ? zero16 : acc[oo], 0, 0, 0);
                                ++nm;
                            }
                        }
                    }
                    if (jr <= 8) fr[(jr + 1) & 1][f] = *(lds_h8_t)(unsigned)(fa[f] + (jr + 1) * (MPITCH * 128));
                    if (jr == 11) fr[0][f] = *(lds_h8_t)(unsigned)fa[f];
                    if (f < 3 && !(PC_DBG & 2)) {
                        const int k = c * 3 + f;
                        char* const dst = k < 5 ? xring + (s * XPITCH + 8 * k) * 128 : dummy;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rxh, (__attribute__((address_space(3))) void*)dst, 16, vx[f], row_s(bs_n, itn.pyi * TH, s - 2, live_n), 0, 0);
                    }
                    if (f == 3 && jr <= 7 && !(PC_DBG & 1)) {
                        // The residual words travel through LDS as well (one DMA piece per slice and part, each lane fetching the 16 bytes it will
                        // add): as ordinary loads the compiler waited for them with vmcnt(0) -- it cannot order loads against LDS-DMA in flight --
                        // a full memory latency three times per period; as inline-asm loads their in-flight destination registers are invisible to
                        // the register allocator (it re-used them, and the late data turned stores into wild writes).
                        const unsigned so = row_s(bs, y0, jr, live);
                        char* const dst = rring + ((jr & 3) * 2 + c) * 4096;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rxh, (__attribute__((address_space(3))) void*)dst, 16, vr, so, 0, 0);
                        unsigned vr1 = vr + 32;                           // (slice 1; not as the instruction's immediate: that moves the LDS address too)
                        asm volatile("" : "+v"(vr1));
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rxh, (__attribute__((address_space(3))) void*)(dst + 2048), 16, vr1, so, 0, 0);
                        if (LO) {
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxl, (__attribute__((address_space(3))) void*)(dst + 1024), 16, vr, so, 0, 0);
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rxl, (__attribute__((address_space(3))) void*)(dst + 3072), 16, vr1, so, 0, 0);
                        }
                    }
                    if (has_epi && f >= 4 && f <= 9) {
                        constexpr int gp = (f - 4) / 3, part = (f - 4) % 3;
                        float (&vv)[8] = v[gp];
                        if (part == 0) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) vv[e] = acc[o & 7][gp * 8 + e];     // channels 32c + 16gp + {0..3 | 8..11} + 4hh
                            if (LO) {
                                const uint4 w = rw[gp][1];
                                vv[0] = mix_lo(w.x, kc.lowscale, vv[0]); vv[1] = mix_hi(w.x, kc.lowscale, vv[1]);
                                vv[2] = mix_lo(w.y, kc.lowscale, vv[2]); vv[3] = mix_hi(w.y, kc.lowscale, vv[3]);
                                vv[4] = mix_lo(w.z, kc.lowscale, vv[4]); vv[5] = mix_hi(w.z, kc.lowscale, vv[5]);
                                vv[6] = mix_lo(w.w, kc.lowscale, vv[6]); vv[7] = mix_hi(w.w, kc.lowscale, vv[7]);
                            }
                        } else if (part == 1) {
                            const uint4 w = rw[gp][0];
                            vv[0] = mix_lo(w.x, kc.one, vv[0]); vv[1] = mix_hi(w.x, kc.one, vv[1]);
                            vv[2] = mix_lo(w.y, kc.one, vv[2]); vv[3] = mix_hi(w.y, kc.one, vv[3]);
                            vv[4] = mix_lo(w.z, kc.one, vv[4]); vv[5] = mix_hi(w.z, kc.one, vv[5]);
                            vv[6] = mix_lo(w.w, kc.one, vv[6]); vv[7] = mix_hi(w.w, kc.one, vv[7]);
                            if (LO) {
                                split2(vv[0], vv[1], kc.neg2048, hw[gp][0], lw[gp][0]); split2(vv[2], vv[3], kc.neg2048, hw[gp][1], lw[gp][1]);
                            }
                        } else {
                            if (LO) {
                                split2(vv[4], vv[5], kc.neg2048, hw[gp][2], lw[gp][2]); split2(vv[6], vv[7], kc.neg2048, hw[gp][3], lw[gp][3]);
                            } else {
                                const h2_t p0 = {(half_t)vv[0], (half_t)vv[1]}, p1 = {(half_t)vv[2], (half_t)vv[3]}, p2 = {(half_t)vv[4], (half_t)vv[5]}, p3 = {(half_t)vv[6], (half_t)vv[7]};
                                hw[gp][0] = __builtin_bit_cast(unsigned, p0); hw[gp][1] = __builtin_bit_cast(unsigned, p1);
                                hw[gp][2] = __builtin_bit_cast(unsigned, p2); hw[gp][3] = __builtin_bit_cast(unsigned, p3);
                            }
                            const unsigned so = row_s(bs, y0, o, live);
                            {
                                const u4_t d = {hw[gp][0], hw[gp][1], hw[gp][2], hw[gp][3]};
                                __builtin_amdgcn_raw_buffer_store_b128(d, ryh, vr + 32 * gp, so, 0);
                            }
                            if (LO) {
                                const u4_t d = {lw[gp][0], lw[gp][1], lw[gp][2], lw[gp][3]};
                                __builtin_amdgcn_raw_buffer_store_b128(d, ryl, vr + 32 * gp, so, 0);
                            }
                        }
                    }
#ifndef PC_NOPIN
                    // MFMA, fragment read, a third of the piece; MFMA, a third; MFMA, the rest (VALU | SALU groups)
                    constexpr int nmfma = (jr <= 9 && !rd_chunk) ? ((jr >= 2 ? 1 : 0) + ((jr >= 1 && jr <= 8) ? 1 : 0) + (jr <= 7 ? 1 : 0)) : 0;
#pragma unroll
                    for (int i_ = 0; i_ < 3; ++i_) {
                        if (i_ < nmfma) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i_ == 0 && (jr <= 8 || jr == 11)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        if (i_ < nmfma && i_ < 2) __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
                    }
#endif
                    (void)nm;
                    __builtin_amdgcn_sched_barrier(0);
                    PC_SUB(f)
                };
#define PC_CHUNK(F) chunk(std::integral_constant<int, F>{});
                PC_CHUNK(0) PC_CHUNK(1) PC_CHUNK(2) PC_CHUNK(3) PC_CHUNK(4) PC_CHUNK(5) PC_CHUNK(6) PC_CHUNK(7) PC_CHUNK(8) PC_CHUNK(9) PC_CHUNK(10) PC_CHUNK(11)
#undef PC_CHUNK
                __builtin_amdgcn_sched_barrier(0);
                // everything issued before the previous step has completed: residual words (used from the next step on) and DMA pieces
                // (read by P eleven steps after their issue); what this step and the previous one issued may still be in flight
                wait_keep(c_ops<LO>(s) + c_ops<LO>((s + 11) % 12) + PC_TRACE_OPS);       // (a constant once the step loop is unrolled)
                PC_STAMP(2 * s)
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                PC_STAMP(2 * s + 1)
            };
#define PC_STEP(S) step(std::integral_constant<int, S>{});
            PC_STEP(0) PC_STEP(1) PC_STEP(2) PC_STEP(3) PC_STEP(4) PC_STEP(5) PC_STEP(6) PC_STEP(7) PC_STEP(8) PC_STEP(9) PC_STEP(10) PC_STEP(11)
#undef PC_STEP
            itprev = itq; itq = itn;
            vr_prev = vr_q; vr_q = vr_n;
            bs_prev = bs_q; bs_q = bs_n;
        }
    }
#undef PC_READ_ROW
#undef PC_PIN
#endif
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
    if ((long long)a.B * a.H * a.W * 128 >= (1ll << 31)) return false;   // 32-bit buffer offsets: lane part + row part (either may be num_records = "nowhere")
    if ((a.x_lo == nullptr) != (a.y_lo == nullptr)) return false;
    a.px = (a.W + TW - 1) / TW;
    a.py = (a.H + TH - 1) / TH;
    const long long items = (long long)a.B * a.px * a.py;
    const int G = (int)std::min<long long>(items, max_groups);
    if (a.x_lo) arsb_pc_kernel<true><<<dim3(G), dim3(256), LDS_BYTES, s>>>(a);
    else arsb_pc_kernel<false><<<dim3(G), dim3(256), LDS_BYTES, s>>>(a);
    return true;
}
