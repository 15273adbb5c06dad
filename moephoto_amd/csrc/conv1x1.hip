// conv1x1.hip -- the 1x1 convolutions of MoeNet_lite2 (python/MoeNet_lite2.py:22-54 of the reference: conv_input2 and the upsampler
// stages `ures` / `uim`, nn.Conv2d(48, 48 | 192, 1) [-> PixelShuffle(2) -> PReLU]), optionally with the 48->1 tail conv (also 1x1)
// folded in, in fp16 or with split operands (three products, MOE_PREC_FP16X3).
//
// These layers move 128-512 bytes per pixel for 8-24 MFMAs per 32 pixels: they are HBM-bound, and the generic kernel (conv_mfma.hip)
// ran them at 1.4-2.8 TB/s because its workgroup drains vmcnt to zero and meets at a barrier once per 256-pixel patch.  A 1x1 conv has
// no halo, so here NOTHING is shared between waves but the weights:
//
//   * a wave owns 32-pixel tiles (b, y, 32 x) round-robin and keeps a private ring of RING tiles in LDS, filled by raw-buffer LDS-DMA
//     RING-1 tiles ahead; it waits with a COUNTED vmcnt (every tile issues the same number of loads, live or not: an offset the buffer
//     unit rejects makes a load return zeros and a store vanish), so the loads of several tiles are in flight per wave, store
//     acknowledgements are not waited for, and there is no barrier after the weight load;
//   * all output chunks of a tile are computed from the one copy of the input in LDS (the generic kernel read it once per chunk);
//   * weight rows are permuted when they are copied to LDS so that registers 8g..8g+7 of a lane are eight consecutive channels:
//     16-byte stores without any lane pairing; bias and the tail weights sit in LDS / registers in the same order.
//
// LDS: weights NCH x 8 KiB (x2 with split operands) + bias + 4 waves x RING x (4 | 8) KiB.  Round 6: the split-operand four-chunk stages of the 48-channel nets (lite's
// default arithmetic; template NKS = 3: the all-zero fourth k-slice is not computed) hold their 48 weight fragments in registers (192 AGPRs): no weight bytes in LDS, RING = 4.
#include "common.h"
#include "rowtile.h"

namespace {

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;
typedef const __attribute__((address_space(3))) float4_t* lds_f4_t;

template <bool X3, int NCH, bool TAIL, int NKS = 4>
struct Cfg {
    static constexpr int NP = X3 ? 8 : 4;                                    // DMA pieces per tile
    static constexpr int NS = TAIL ? 2 : NCH * 4 * (X3 ? 2 : 1);            // stores per tile
    static constexpr int TILE = X3 ? 8192 : 4096;
    // Round 6: the split-operand upsampler stages of the 48-channel nets (X3, four chunks, three k-slices: lite's default arithmetic) keep their 48 weight fragments in
    // REGISTERS (192 AGPRs) instead of 64 KiB of LDS: the ring grows from two tiles to four per wave (one tile ahead was 32 KiB in flight per CU against the ~46 KiB that
    // 6 TB/s x 2 us over 256 CUs ask for: these launches ran at 0.6 of the HBM roofline), and 48 KiB of LDS reads per tile disappear
    static constexpr bool WREG = X3 && NCH == 4 && NKS == 3;
    static constexpr int WB = WREG ? 0 : NCH * 8 * 1024 * (X3 ? 2 : 1);
    static constexpr int BIASB = NCH * 256;
    static constexpr int byLds = (163840 - WB - BIASB) / (4 * TILE);
    static constexpr int byCnt = 1 + 63 / NP;
    static constexpr int RING = byLds < byCnt ? (byLds < 8 ? byLds : 8) : (byCnt < 8 ? byCnt : 8);
    static constexpr int LDS = WB + BIASB + 4 * RING * TILE;
    static_assert(RING >= 2, "ring");
};

template <bool X3, int NCH, bool TAIL, int NKS = 4>
__global__ __launch_bounds__(256) void conv1x1_kernel(Conv1x1Args a)
{
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass only needs the launch stub)
    using C = Cfg<X3, NCH, TAIL, NKS>;
    constexpr bool WREG = C::WREG;
    constexpr int R = NCH == 4 ? 2 : 1, RING = C::RING, SEG = X3 ? 2 : 1;
    constexpr unsigned kOOR = 0xFFFF0000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hh = lane >> 5;

    half8_t wr[WREG ? NCH : 1][WREG ? SEG : 1][WREG ? 2 * NKS : 1];      // (WREG: fragment (chunk, part, 2 ks + nb), in AGPRs)
    // ---- weights -> LDS: fragment (chunk c, part s, k-slice ks, n-block nb) at ((c * SEG + s) * 8 + 2 ks + nb) KiB.  MFMA row i = 8q + 4h + e
    // of a 32x32 result lands in register 4q + e of the lanes hh = h; giving row i the channel 16 (q >> 1) + 8 h + 4 (q & 1) + e makes registers
    // 8g .. 8g+7 of lane (j, hh) the consecutive channels 32 nb + 16 g + 8 hh ..
    {
        const int wi = lane & 31, wq = wi >> 3;
        const int src = (lane & 32) | (16 * (wq >> 1) + 8 * ((wi >> 2) & 1) + 4 * (wq & 1) + (wi & 3));
        if constexpr (WREG) {
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int sg = 0; sg < SEG; ++sg)
#pragma unroll
                    for (int k = 0; k < 2 * NKS; ++k) wr[c][sg][k] = *(const half8_t*)((sg == 0 ? a.w_hi : a.w_lo) + ((c * 8 + k) * 64 + src) * 8);
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int sg = 0; sg < SEG; ++sg)
#pragma unroll
                    for (int k = 0; k < 2 * NKS; ++k) asm volatile("" : "+a"(wr[c][sg][k]));
        }
        for (int f = w4; f < (WREG ? 0 : NCH * SEG * 8); f += 4) {
            const int c = f / (SEG * 8), s = (f / 8) % SEG, k = f & 7;
            const half_t* wsrc = (s == 0 ? a.w_hi : a.w_lo) + ((c * 8 + k) * 64 + src) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wsrc, (__attribute__((address_space(3))) void*)(smem + f * 1024), 16, 0, 0);
        }
        if (tid < NCH * 64) ((float*)(smem + C::WB))[tid] = a.bias[tid];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    char* const ring = smem + C::WB + C::BIASB + w4 * (RING * C::TILE);
    const unsigned ring0 = lds0 + (unsigned)(C::WB + C::BIASB) + (unsigned)(w4 * (RING * C::TILE));

    // ---- buffers: lane part of an address in a VGPR (per launch / per tile), tile part in an SGPR ---------------------------------
    const unsigned in_bytes = (unsigned)a.B * a.H * a.W * 128u;
    const unsigned out_px = (unsigned)a.out_cs * 2u;
    const unsigned Wo = (unsigned)a.W * R;
    const unsigned out_bytes = TAIL ? (unsigned)a.B * a.H * R * Wo * 4u : (unsigned)a.B * a.H * R * Wo * out_px;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in_hi, 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rinl = __builtin_amdgcn_make_buffer_rsrc((void*)(X3 ? a.in_lo : a.in_hi), 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(TAIL ? (void*)a.tail_out : (void*)a.out_hi, 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t routl = __builtin_amdgcn_make_buffer_rsrc((X3 && !TAIL) ? (void*)a.out_lo : (void*)a.out_hi, 0, out_bytes, 0x00020000);
    // DMA piece i: pixels 8i .. 8i+7 of the tile, lane = (pixel 8i + lane / 8, physical 16-byte slot lane % 8 = logical slot ^ ((pixel >> 1) & 7)).
    // Pixels beyond the end of the row are the next row's (harmless: their outputs are not stored); beyond the tensor the unit returns zeros.
    unsigned vin[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pi = 8 * i + (lane >> 3);
        vin[i] = (unsigned)(pi * 128 + (((lane & 7) ^ ((pi >> 1) & 7)) << 4));
    }
    // B fragment of k-slice ks: lane (j, hh) reads logical slot 2 ks + hh of pixel j
    unsigned bofs[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) bofs[ks] = (unsigned)(j * 128 + (((2 * ks + hh) ^ ((j >> 1) & 7)) << 4));
    const unsigned wl = lds0 + (unsigned)lane * 16u;                                   // A fragments
    const unsigned bl = lds0 + (unsigned)C::WB + (unsigned)hh * 32u;                   // bias: channel 64 c + 32 nb + 16 g + 8 hh + e
    float tw[2][2][8];                                                                 // tail weights in the same order (TAIL only)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int e = 0; e < 8; ++e) tw[nb][g][e] = TAIL ? a.tail_w[32 * nb + 16 * g + 8 * hh + e] : 0.f;
    const RowConsts kc = {1.0f, 0.00048828125f, -2048.f};
    const bool k4 = a.nks > 3;           // (uniform: one scalar branch per chunk)

    // ---- this wave's tiles: t = gw, gw + S, ... over (row = b H + y, xt) ------------------------------------------------------------
    const int px = (a.W + 31) >> 5;
    const int ntiles = a.B * a.H * px;
    const int gw = blockIdx.x * 4 + w4, S = gridDim.x * 4;
    const int Sx = S % px, Sr = S / px;
    struct Tile { int row, xt, t; };
    auto advance = [&](Tile& q) {
        q.t += S; q.xt += Sx; q.row += Sr;
        if (q.xt >= px) { q.xt -= px; q.row += 1; }
    };
    auto issue = [&](const Tile& q, int slot) {
        const unsigned so = q.t < ntiles ? (unsigned)(q.row * a.W + q.xt * 32) * 128u : kOOR;
        char* const dst = ring + slot * C::TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, vin[i], so, 0, 0);
            if (X3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rinl, (__attribute__((address_space(3))) void*)(dst + 4096 + i * 1024), 16, vin[i], so, 0, 0);
        }
    };
    Tile tq{gw / px, gw % px, gw};       // next tile to fetch
    Tile tc = tq;                        // tile being computed
    if (gw >= ntiles) return;
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) {      // prologue: RING - 1 tiles in flight
        issue(tq, i);
        advance(tq);
    }
    int slot = 0, fslot = RING - 1;
    for (; tc.t < ntiles; advance(tc)) {
        issue(tq, fslot);
        advance(tq);
        fslot = fslot + 1 == RING ? 0 : fslot + 1;
        // This tile's pieces have landed once at most the (RING - 1) NP younger LOADS are outstanding.  Loads return in order among
        // themselves but NOT relative to stores (counting this tile's stores into the allowance let store acknowledgements stand in for
        // loads still in flight: intermittent garbage in the shallowest-ring variant), so the allowance counts loads only; stores still
        // outstanding then merely make the wait longer than necessary.
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 1) * C::NP) : "memory");
        const unsigned tb = ring0 + (unsigned)(slot * C::TILE);
        half8_t bh[4] = {}, blo[4] = {};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (ks >= NKS || (ks == 3 && !k4)) continue;      // (48-channel nets: channels 48..63 are zeros in activations and weights -- the fourth k-slice adds nothing: same bits)
            bh[ks] = *(lds_h8_t)(tb + bofs[ks]);
            if (X3) blo[ks] = *(lds_h8_t)(tb + 4096u + bofs[ks]);
        }
        const int x = tc.xt * 32 + j;
        const bool okx = x < a.W;
        float dots[2] = {0.f, 0.f};          // TAIL: the two horizontal sub-pixels of the current output row
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int si = c >> 1, sj = c & 1;
            float16_t ah[2], al[2];
            // the bias is the accumulators' initial value (round 6: one add per output less in an epilogue that holds the folded-tail form at 13.7 VALU instructions per MFMA)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float4_t b0 = *(lds_f4_t)(bl + (unsigned)((64 * c + 32 * nb + 16 * g) * 4));
                    const float4_t b1 = *(lds_f4_t)(bl + (unsigned)((64 * c + 32 * nb + 16 * g + 4) * 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ah[nb][8 * g + e] = b0[e]; ah[nb][8 * g + 4 + e] = b1[e]; al[nb][8 * g + e] = 0.f; al[nb][8 * g + 4 + e] = 0.f; }
                }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (ks >= NKS || (ks == 3 && !k4)) continue;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    half8_t whi;
                    if constexpr (WREG) whi = wr[c][0][2 * ks + nb];
                    else whi = *(lds_h8_t)(wl + (unsigned)(((c * SEG + 0) * 8 + 2 * ks + nb) * 1024));
                    ah[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, bh[ks], ah[nb], 0, 0, 0);
                    if (X3) {
                        half8_t wlo;
                        if constexpr (WREG) wlo = wr[c][SEG - 1][2 * ks + nb];
                        else wlo = *(lds_h8_t)(wl + (unsigned)(((c * SEG + 1) * 8 + 2 * ks + nb) * 1024));
                        al[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, bh[ks], al[nb], 0, 0, 0);
                        al[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, blo[ks], al[nb], 0, 0, 0);
                    }
                }
            }
            // epilogue of chunk c: + low-order products, PReLU in fp32 (slope <= 1), then the tail dot or fp16 (hi [, lo]) stores
            float dot = 0.f;
            unsigned so = 0, vst = 0;
            if (!TAIL) {
                if (R == 1) {
                    so = (unsigned)(tc.row * a.W + tc.xt * 32) * out_px;
                    vst = okx ? (unsigned)j * out_px + (unsigned)hh * 16u : kOOR;
                } else {
#ifdef C1_ABL_COALESCE      // timing ablation (results WRONG): the same bytes into the same 8-KiB row segment, every store instruction 1 KiB contiguous
                    so = ((unsigned)(2 * tc.row + si) * Wo + (unsigned)(64 * tc.xt)) * out_px + (unsigned)(sj * 4096);
                    vst = (unsigned)lane * 16u;
#else
                    so = ((unsigned)(2 * tc.row + si) * Wo + (unsigned)(64 * tc.xt + sj)) * out_px;
                    vst = okx ? (unsigned)(2 * j) * out_px + (unsigned)hh * 16u : kOOR;
#endif
                }
                if (tc.t >= ntiles) so = kOOR;
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float t = ah[nb][8 * g + e];
                        if (X3) t = __builtin_fmaf(al[nb][8 * g + e], 0.00048828125f, t);
                        const float ts = t * a.slope;
                        asm("v_max_f32 %0, %1, %2" : "=v"(v[e]) : "v"(t), "v"(ts));
                    }
                    if (TAIL) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) dot = __builtin_fmaf(v[e], tw[nb][g][e], dot);
                    } else {
                        unsigned h[4], l[4] = {0, 0, 0, 0};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            if (X3) split2(v[2 * k], v[2 * k + 1], kc.neg2048, h[k], l[k]);
                            else {
                                const half2_t pr = {(half_t)v[2 * k], (half_t)v[2 * k + 1]};
                                h[k] = __builtin_bit_cast(unsigned, pr);
                            }
                        }
                        const u4_t dh = {h[0], h[1], h[2], h[3]};
#ifdef C1_ABL_COALESCE
                        constexpr unsigned kPiece = 1024u;
#else
                        constexpr unsigned kPiece = 32u;
#endif
                        __builtin_amdgcn_raw_buffer_store_b128(dh, rout, vst + (unsigned)(nb * 2 + g) * kPiece, so, 0);
                        if (X3) {
                            const u4_t dl = {l[0], l[1], l[2], l[3]};
                            __builtin_amdgcn_raw_buffer_store_b128(dl, routl, vst + (unsigned)(nb * 2 + g) * kPiece, so, 0);
                        }
                    }
                }
            if (TAIL) {
                // the lanes (j, 0) and (j, 1) hold the two halves of every 16-channel group: their sum is the 64-channel dot product of HR
                // pixel (2y + si, 2x + sj); the two sj of a row leave as one 8-byte store of lane (j, 0)
                dot += __shfl_xor(dot, 32);
                dots[sj] = dot;
                if (sj == 1) {
                    const unsigned so2 = tc.t < ntiles ? ((unsigned)(2 * tc.row + si) * Wo + (unsigned)(64 * tc.xt)) * 4u : kOOR;
                    const unsigned v2 = (okx && hh == 0) ? (unsigned)j * 8u : kOOR;
                    const u2_t d = {__builtin_bit_cast(unsigned, dots[0]), __builtin_bit_cast(unsigned, dots[1])};
                    __builtin_amdgcn_raw_buffer_store_b64(d, rout, v2, so2, 0);
                }
            }
        }
        slot = slot + 1 == RING ? 0 : slot + 1;
    }
#endif
}

template <bool X3, int NCH, bool TAIL, int NKS = 4>
hipError_t set_limit()
{
    return hipFuncSetAttribute((const void*)conv1x1_kernel<X3, NCH, TAIL, NKS>, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg<X3, NCH, TAIL, NKS>::LDS);
}

template <bool X3, int NCH, bool TAIL, int NKS = 4>
void launch_t(const Conv1x1Args& a, int groups, hipStream_t s)
{
    conv1x1_kernel<X3, NCH, TAIL, NKS><<<dim3(groups), dim3(256), Cfg<X3, NCH, TAIL, NKS>::LDS, s>>>(a);
}

}  // namespace

hipError_t conv1x1_init()
{
    hipError_t e;
    if ((e = set_limit<false, 1, false>()) != hipSuccess) return e;
    if ((e = set_limit<true, 1, false>()) != hipSuccess) return e;
    if ((e = set_limit<false, 4, false>()) != hipSuccess) return e;
    if ((e = set_limit<true, 4, false>()) != hipSuccess) return e;
    if ((e = set_limit<false, 4, true>()) != hipSuccess) return e;
    if ((e = set_limit<true, 4, true>()) != hipSuccess) return e;
    if ((e = set_limit<true, 4, false, 3>()) != hipSuccess) return e;
    if ((e = set_limit<true, 4, true, 3>()) != hipSuccess) return e;
    return hipSuccess;
}

// false: the layer is not one of the compiled shapes (caller uses conv_mfma_kernel)
bool launch_conv1x1(const Conv1x1Args& a, int max_groups, hipStream_t s)
{
    const bool x3 = a.in_lo != nullptr, tail = a.tail_out != nullptr;
    if (!((a.r == 1 && a.nchunks == 1) || (a.r == 2 && a.nchunks == 4))) return false;
    if (tail && a.r != 2) return false;
    if (!(a.slope <= 1.f) || !a.bias || !a.w_hi) return false;
    if (x3 && (!a.w_lo || (!tail && !a.out_lo))) return false;
    if (!tail && !a.out_hi) return false;
    const long long in_b = 128ll * a.B * a.H * a.W, out_b = (tail ? 4ll : 2ll * a.out_cs) * a.B * a.H * a.r * a.W * a.r;
    if (in_b >= (1ll << 32) - 65536 || out_b >= (1ll << 32) - 65536) return false;       // 32-bit buffer offsets
    const long long ntiles = (long long)a.B * a.H * ((a.W + 31) / 32);
    const int groups = (int)std::min<long long>(max_groups, (ntiles + 3) / 4);
    if (groups < 1) return false;
    if (a.nchunks == 1) { if (x3) launch_t<true, 1, false>(a, groups, s); else launch_t<false, 1, false>(a, groups, s); }
    else if (!tail) { if (x3 && a.nks == 3) launch_t<true, 4, false, 3>(a, groups, s); else if (x3) launch_t<true, 4, false>(a, groups, s); else launch_t<false, 4, false>(a, groups, s); }
    else { if (x3 && a.nks == 3) launch_t<true, 4, true, 3>(a, groups, s); else if (x3) launch_t<true, 4, true>(a, groups, s); else launch_t<false, 4, true>(a, groups, s); }
    return true;
}
