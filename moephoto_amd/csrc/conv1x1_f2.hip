// conv1x1_f2.hip -- TWO upsampler stages of MoeNet_lite2 in one launch (round 6): `ures` / `uim` stage k and stage k+1 with the 48 -> 1 tail conv folded in
// (python/MoeNet_lite2.py:22-54 of the reference: nn.Conv2d(48, 192, 1) -> PixelShuffle(2) -> PReLU, twice, then nn.Conv2d(48, 1, 1)), split operands (MOE_PREC_FP16X3).
//
// Every layer of the upsampler is POINTWISE (1x1 convs, pixel shuffle, PReLU): one input pixel becomes 4, then 16 output pixels, and nothing else contributes to them.
// conv1x1.hip ran the stages one by one: stage k wrote 4 x (128 + 128) B per input pixel and stage k+1 read them back -- 70 GB per lite8 frame for the storing form, which
// waited on the memory system for 0.64-0.76 of its wave cycles (profiles/r06/z3_, z6_pmc_conv1x1_lite8*.txt).  Here the tensor between the two stages never exists:
//
//   * stage A of a 32-pixel tile is conv1x1.hip's register-weight form (48 fragments in AGPRs, the all-zero fourth k-slice not computed); its epilogue leaves, per
//     sub-pixel chunk (si1, sj1), the activations as packed hi / lo pairs in exactly the registers a B operand of the NEXT 1x1 conv wants: lane (j, hh) holds the channels
//     16 ks + 8 hh .. + 7 of "pixel" j for ks = 0, 1, 2 (the epilogue's channel permutation was chosen for 16-byte stores: it is also the MFMA's k order);
//   * stage B runs on that virtual tile at once: four chunks (si2, sj2), weights from LDS (48 KiB), bias as the accumulators' initial value, PReLU in fp32, the tail dot;
//   * the 16 fp32 results of an input pixel leave as four 16-byte stores (one per output row 4 y + 2 si1 + si2).
//
// Arithmetic: per layer the same products in the same order, the same roundings (hi / lo split between the stages) as conv1x1.hip's kernels: the same bits.
#include "common.h"
#include "rowtile.h"
#include <algorithm>
#include <type_traits>

namespace {

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;
typedef const __attribute__((address_space(3))) float4_t* lds_f4_t;

constexpr int NKS = 3;                          // k-slices of 16 input channels that are not all zeros (48-channel nets)
constexpr int TILE = 8192;                      // a tile of 32 pixels: hi + lo lines
constexpr int NP = 8;                           // DMA pieces per tile
constexpr int WB = 4 * 2 * 2 * NKS * 1024;      // stage B's weights in LDS: [chunk 4][part 2][2 ks + nb][1 KiB] = 48 KiB
constexpr int BIASB = 2 * 1024;                 // both stages' biases
constexpr int RING = 3;
constexpr int LDS = WB + BIASB + 4 * RING * TILE;      // 149,504

__global__ __launch_bounds__(256) void conv1x1_f2_kernel(Conv1x1F2Args a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr unsigned kOOR = 0xFFFF0000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hh = lane >> 5;

    // ---- weights.  MFMA row i = 8q + 4h + e of a 32x32 result lands in register 4q + e of the lanes hh = h; giving row i the channel 16 (q >> 1) + 8 h + 4 (q & 1) + e makes
    // registers 8g .. 8g+7 of lane (j, hh) the consecutive channels 32 nb + 16 g + 8 hh .. (conv1x1.hip).  Stage A: registers; stage B: LDS
    half8_t wr[4][2][2 * NKS];
    {
        const int wi = lane & 31, wq = wi >> 3;
        const int src = (lane & 32) | (16 * (wq >> 1) + 8 * ((wi >> 2) & 1) + 4 * (wq & 1) + (wi & 3));
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int sg = 0; sg < 2; ++sg)
#pragma unroll
                for (int k = 0; k < 2 * NKS; ++k) wr[c][sg][k] = *(const half8_t*)((sg == 0 ? a.wa_hi : a.wa_lo) + ((c * 8 + k) * 64 + src) * 8);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int sg = 0; sg < 2; ++sg)
#pragma unroll
                for (int k = 0; k < 2 * NKS; ++k) asm volatile("" : "+a"(wr[c][sg][k]));
        for (int f = w4; f < 4 * 2 * 2 * NKS; f += 4) {      // fragment ((c 2 + part) 6 + k) of stage B
            const int c = f / (4 * NKS), sg = (f / (2 * NKS)) & 1, k = f % (2 * NKS);
            const half_t* wsrc = (sg == 0 ? a.wb_hi : a.wb_lo) + ((c * 8 + k) * 64 + src) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wsrc, (__attribute__((address_space(3))) void*)(smem + f * 1024), 16, 0, 0);
        }
        ((float*)(smem + WB))[tid] = a.bias_a[tid];
        ((float*)(smem + WB + 1024))[tid] = a.bias_b[tid];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
    }
    char* const ring = smem + WB + BIASB + w4 * (RING * TILE);
    const unsigned ring0 = lds0 + (unsigned)(WB + BIASB) + (unsigned)(w4 * (RING * TILE));

    const unsigned in_bytes = (unsigned)a.B * a.H * a.W * 128u;
    const unsigned Wo = (unsigned)a.W * 4u;
    const unsigned out_bytes = (unsigned)a.B * a.H * 4u * Wo * 4u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in_hi, 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rinl = __builtin_amdgcn_make_buffer_rsrc((void*)a.in_lo, 0, in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)a.tail_out, 0, out_bytes, 0x00020000);
    // DMA piece i: pixels 8i .. 8i+7 of the tile, lane = (pixel 8i + lane / 8, physical 16-byte slot lane % 8 = logical slot ^ ((pixel >> 1) & 7))
    unsigned vin[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pi = 8 * i + (lane >> 3);
        vin[i] = (unsigned)(pi * 128 + (((lane & 7) ^ ((pi >> 1) & 7)) << 4));
    }
    unsigned bofs[NKS];      // B fragment of k-slice ks: lane (j, hh) reads logical slot 2 ks + hh of pixel j
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) bofs[ks] = (unsigned)(j * 128 + (((2 * ks + hh) ^ ((j >> 1) & 7)) << 4));
    const unsigned wl = lds0 + (unsigned)lane * 16u;                                   // stage B's A fragments
    const unsigned bla = lds0 + (unsigned)WB + (unsigned)hh * 32u;                     // bias: channel 64 c + 32 nb + 16 g + 8 hh + e
    const unsigned blb = bla + 1024u;
    float tw[3][8];                                                                    // tail weights of the channels 16 p + 8 hh + e, p = 2 nb + g (p = 3: padding)
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) tw[p][e] = a.tail_w[16 * p + 8 * hh + e];
    const RowConsts kc = {1.0f, 0.00048828125f, -2048.f};

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
        char* const dst = ring + slot * TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, vin[i], so, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rinl, (__attribute__((address_space(3))) void*)(dst + 4096 + i * 1024), 16, vin[i], so, 0, 0);
        }
    };
    Tile tq{gw / px, gw % px, gw};       // next tile to fetch
    Tile tc = tq;                        // tile being computed
    if (gw >= ntiles) return;
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) {
        issue(tq, i);
        advance(tq);
    }
    int slot = 0, fslot = RING - 1;
    for (; tc.t < ntiles; advance(tc)) {
        issue(tq, fslot);
        advance(tq);
        fslot = fslot + 1 == RING ? 0 : fslot + 1;
        // (loads only in the allowance: conv1x1.hip; this kernel's sixteen stores per tile are few)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((RING - 1) * NP) : "memory");
        const unsigned tb = ring0 + (unsigned)(slot * TILE);
        half8_t bh[NKS], blo[NKS];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            bh[ks] = *(lds_h8_t)(tb + bofs[ks]);
            blo[ks] = *(lds_h8_t)(tb + 4096u + bofs[ks]);
        }
        const int x = tc.xt * 32 + j;
        const bool okx = x < a.W;
        float dprev[2][2];                   // [si2][sj2] of the sub-pixel column sj1 = 0 of the current row pair si1
#pragma unroll 1
        for (int c1 = 0; c1 < 4; ++c1) {
            const int si1 = c1 >> 1, sj1 = c1 & 1;
            // ================= stage A, chunk c1: 64 (48) channels of the sub-pixel (si1, sj1) of every pixel of the tile =================
            float16_t ah[2], al[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float4_t b0 = *(lds_f4_t)(bla + (unsigned)((64 * c1 + 32 * nb + 16 * g) * 4));
                    const float4_t b1 = *(lds_f4_t)(bla + (unsigned)((64 * c1 + 32 * nb + 16 * g + 4) * 4));
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ah[nb][8 * g + e] = b0[e]; ah[nb][8 * g + 4 + e] = b1[e]; al[nb][8 * g + e] = 0.f; al[nb][8 * g + 4 + e] = 0.f; }
                }
            // (c1 is a loop variable: the register-resident fragments are selected by a switch over its four values -- wr[] must be indexed by constants)
            auto stage_a = [&](auto C_) __attribute__((always_inline)) {
                constexpr int c = decltype(C_)::value;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        ah[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[c][0][2 * ks + nb], bh[ks], ah[nb], 0, 0, 0);
                        al[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[c][1][2 * ks + nb], bh[ks], al[nb], 0, 0, 0);
                        al[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[c][0][2 * ks + nb], blo[ks], al[nb], 0, 0, 0);
                    }
            };
            if (c1 == 0) stage_a(std::integral_constant<int, 0>{});
            else if (c1 == 1) stage_a(std::integral_constant<int, 1>{});
            else if (c1 == 2) stage_a(std::integral_constant<int, 2>{});
            else stage_a(std::integral_constant<int, 3>{});
            // epilogue A: + low-order products, PReLU in fp32, hi / lo split: the pieces p = 2 nb + g = 0, 1, 2 are stage B's k-slices (p = 3: the padding channels)
            half8_t ph[NKS], pl[NKS];
#pragma unroll
            for (int p = 0; p < NKS; ++p) {
                const int nb = p >> 1, g = p & 1;
                unsigned h[4], l[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int e = 2 * k + u;
                        const float t = __builtin_fmaf(al[nb][8 * g + e], 0.00048828125f, ah[nb][8 * g + e]);
                        const float ts = t * a.slope_a;
                        asm("v_max_f32 %0, %1, %2" : "=v"(v[u]) : "v"(t), "v"(ts));
                    }
                    split2(v[0], v[1], kc.neg2048, h[k], l[k]);
                }
                ph[p] = __builtin_bit_cast(half8_t, u4_t{h[0], h[1], h[2], h[3]});
                pl[p] = __builtin_bit_cast(half8_t, u4_t{l[0], l[1], l[2], l[3]});
            }
            // ================= stage B on the virtual tile: four chunks (si2, sj2), the tail dot =================
            float dcur[2][2];
#pragma unroll
            for (int c2 = 0; c2 < 4; ++c2) {
                const int si2 = c2 >> 1, sj2 = c2 & 1;
                float16_t bhh[2], bll[2];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const float4_t b0 = *(lds_f4_t)(blb + (unsigned)((64 * c2 + 32 * nb + 16 * g) * 4));
                        const float4_t b1 = *(lds_f4_t)(blb + (unsigned)((64 * c2 + 32 * nb + 16 * g + 4) * 4));
#pragma unroll
                        for (int e = 0; e < 4; ++e) { bhh[nb][8 * g + e] = b0[e]; bhh[nb][8 * g + 4 + e] = b1[e]; bll[nb][8 * g + e] = 0.f; bll[nb][8 * g + 4 + e] = 0.f; }
                    }
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const half8_t whi = *(lds_h8_t)(wl + (unsigned)(((c2 * 2 + 0) * 2 * NKS + 2 * ks + nb) * 1024));
                        const half8_t wlo = *(lds_h8_t)(wl + (unsigned)(((c2 * 2 + 1) * 2 * NKS + 2 * ks + nb) * 1024));
                        bhh[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, ph[ks], bhh[nb], 0, 0, 0);
                        bll[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, ph[ks], bll[nb], 0, 0, 0);
                        bll[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, pl[ks], bll[nb], 0, 0, 0);
                    }
                float dot = 0.f;
#pragma unroll
                for (int p = 0; p < NKS; ++p) {      // (the padding channels 48..63 carry zero tail weights: left out)
                    const int nb = p >> 1, g = p & 1;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float t = __builtin_fmaf(bll[nb][8 * g + e], 0.00048828125f, bhh[nb][8 * g + e]);
                        const float ts = t * a.slope_b;
                        float v;
                        asm("v_max_f32 %0, %1, %2" : "=v"(v) : "v"(t), "v"(ts));
                        dot = __builtin_fmaf(v, tw[p][e], dot);
                    }
                }
                // the lanes (j, 0) and (j, 1) hold the two halves of every 16-channel group: their sum is the 48-channel dot product of output pixel
                // (4 y + 2 si1 + si2, 4 x + 2 sj1 + sj2)
                dot += __shfl_xor(dot, 32);
                dcur[si2][sj2] = dot;
            }
            if (sj1 == 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) { dprev[u][0] = dcur[u][0]; dprev[u][1] = dcur[u][1]; }
            } else {             // both sub-pixel columns of the row pair si1 are in: two rows of four consecutive outputs, 16 bytes each, from lane (j, 0)
#pragma unroll
                for (int si2 = 0; si2 < 2; ++si2) {
                    const unsigned so = tc.t < ntiles ? ((unsigned)(4 * tc.row + 2 * si1 + si2) * Wo + (unsigned)(128 * tc.xt)) * 4u : kOOR;
                    const unsigned vo = (okx && hh == 0) ? (unsigned)j * 16u : kOOR;
                    const u4_t d = {__builtin_bit_cast(unsigned, dprev[si2][0]), __builtin_bit_cast(unsigned, dprev[si2][1]),
                                    __builtin_bit_cast(unsigned, dcur[si2][0]), __builtin_bit_cast(unsigned, dcur[si2][1])};
                    __builtin_amdgcn_raw_buffer_store_b128(d, rout, vo, so, 0);
                }
            }
        }
        slot = slot + 1 == RING ? 0 : slot + 1;
    }
#endif
}

}  // namespace

hipError_t conv1x1_f2_init()
{
    return hipFuncSetAttribute((const void*)conv1x1_f2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
}

// false: not this kernel's case (the caller runs the stages one by one on conv1x1.hip)
bool launch_conv1x1_f2(const Conv1x1F2Args& a, int max_groups, hipStream_t s)
{
    if (!a.in_hi || !a.in_lo || !a.wa_hi || !a.wa_lo || !a.wb_hi || !a.wb_lo || !a.bias_a || !a.bias_b || !a.tail_w || !a.tail_out) return false;
    if (!(a.slope_a <= 1.f) || !(a.slope_b <= 1.f)) return false;
    const long long in_b = 128ll * a.B * a.H * a.W, out_b = 64ll * a.B * a.H * a.W;
    if (in_b >= (1ll << 32) - 65536 || out_b >= (1ll << 32) - 65536) return false;       // 32-bit buffer offsets
    const long long ntiles = (long long)a.B * a.H * ((a.W + 31) / 32);
    const int groups = (int)std::min<long long>(max_groups, (ntiles + 3) / 4);
    if (groups < 1) return false;
    conv1x1_f2_kernel<<<dim3(groups), dim3(256), LDS, s>>>(a);
    return true;
}
