// conv3x3_sp.hip -- the hot kernel: 3x3, 64-in-channel implicit-GEMM convolution, one wave per SIMD, with the
// epilogue of patch p-1 and the LDS-DMA of patch p+1 software-pipelined INTO the MFMA stream of patch p.
//
// Measured background (profiles/r01): the matrix loop alone runs at ~69 % of the fp16 MFMA peak, but the epilogue
// (64 outputs per lane per 144 MFMAs: K is only 576) costs as many issue cycles as the MFMAs.  Running it in a second
// wave on the same SIMD (the round-1 "ping-pong" variant, since removed) does not hide it: s_memtime traces show the two
// waves of a SIMD slowing each other down to the SUM of their solo times.  What the hardware does hide is a wave's OWN independent VALU / LDS / VMEM
// instructions issued in the shadow of its MFMAs (about 5 issue slots per 32-cycle v_mfma_f32_32x32x16_f16).  So:
//
//   * 4 waves per workgroup (one per SIMD, all 512 registers each), two accumulator sets per wave;
//   * iteration p multiplies patch p into set A while draining set B (patch p-1): in k-steps 0..7, next to its 12 MFMAs and
//     10 ds_read_b128, the wave finishes one eighth of the previous tile (fp16 conversion, PReLU on packed halves or the
//     fp32 residual add, v_permlane32_swap pair, one 16-byte store) and in k-steps 0..5 issues the DMA pieces of patch p+1;
//   * the bias is the C operand of each accumulator's first MFMA (no reset); the one workgroup barrier per patch sits in
//     k-step 11, behind which the first fragments of patch p+1 are read while that step's MFMAs run;
//   * every k-step is its own scheduling region (sched_barrier) with a sched_group_barrier slot pattern inside: measured
//     74 % MFMA occupancy in cycles (97 % for the bare MFMA/LDS stream); DESIGN.md section 4 has the cycle budget.
//
// GEMM view, LDS image (column-keyed XOR swizzle), weight fragment order and the fused epilogue are those of
// conv_mfma.hip; wave tile = 2 output rows x 32 pixels x 64 output channels.
#include "common.h"
#include "rowtile.h"

#ifndef MOE_ABL
#define MOE_ABL 0
#endif

namespace {

constexpr int PW = kTileW + 2, PH = kTileH + 2, NPIX = PW * PH;   // 34 x 10 halo'd patch
constexpr int NDMA = (NPIX + 7) / 8;                               // 43 one-KiB pieces
constexpr int NDMA_W = (NDMA + 3) / 4;                             // 11 per wave
constexpr int PATCH_BYTES = NDMA_W * 4 * 1024;                      // 45,056: 44 pieces, the last one all padding
constexpr int NFRAG = 72;                                          // 9 taps x 4 k-slices x 2 n-blocks
constexpr int WBYTES = NFRAG * kFragBytes;                         // 73,728
constexpr int LDS_BYTES = WBYTES + 2 * PATCH_BYTES;                // 163,840 = all of the CU's LDS

__device__ __forceinline__ void dma16(const half_t* src, char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct Item { int b, pyi, pxi; };

__device__ __forceinline__ Item decode(int item, int px, int py)
{
    Item it;
    it.pxi = item % px;
    const int t = item / px;
    it.pyi = t % py;
    it.b = t / py;
    return it;
}

struct PatchSrc { unsigned soff; int y0, x0; };      // byte offset of the patch origin behind the (padded) descriptor base

// EPI selects the fused epilogue at compile time (no control flow inside the pipelined iteration):
//   0  plain                  (conv_input2, SEDN rblock.4, lite conv_2)
//   1  PReLU/LeakyReLU        (ARSB conv_1, SEDN rblock.0/2, lite conv_1, upsampler convs)   max(x, slope*x), slope <= 1
//   2  + residual             (ARSB conv_2; its ScaleLayer is folded into the packed weights by the engine)
//   3  PReLU + FUSED TAIL     (last upsampler conv of a branch): the activated fp16 tile is not stored; it is the B operand
//                             of a second implicit GEMM against the 64->1 tail conv's weights, G[tap][pixel] += Wt[tap][c]*act[c][pixel]
//                             (8 extra MFMAs per tile), and only the nine per-tap partial sums per HR pixel are written (fp32,
//                             planar).  A small gather kernel then adds the 9 shifted taps of both branches: the 64-channel
//                             HR tensor (1.6 GB per 12 planes of 1024x1024, written once and read back) never exists.
// The bias (upsampler convs) costs nothing here: the accumulators are initialised with it instead of zero.
// timing trace (MOE_DBG & 64): acc32 doubles as a [wg<8][iter<32][wave<4][slot<16] table of s_memtime stamps
// (slots 0..3: iteration start / body end / after vmcnt wait / after barrier; built with -DMOE_STEP_STAMPS also 4+s: end of k-step s)
#ifdef MOE_STAMP_MIN          /* only the iteration-start stamp: the period without the cost of the other stamps */
#define MOE_STAMP_ON(SLOT) ((SLOT) == 0 || (SLOT) >= 13)
#else
#define MOE_STAMP_ON(SLOT) true
#endif
#define MOE_STAMP(SLOT)                                                                                 \
    if (MOE_STAMP_ON(SLOT) && (a.dbg & 64) && a.acc32 && bid < 8 && p < 32 && lane == 0)                   \
        ((unsigned long long*)a.acc32)[((bid * 32 + p) * 4 + w4) * 16 + (SLOT)] = __builtin_amdgcn_s_memtime();

template <int EPI>
__global__ __launch_bounds__(256) void conv3x3_sp_kernel(ConvArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass only needs the launch stub; it drops an instantiation that uses the gfx950 buffer builtins)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wlds = smem;
    char* const pbuf = smem + WBYTES;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);     // 0..3
    const int j = lane & 31, hh = lane >> 5;

    // EPI 6 (SEDN's fused block tail): the `nchunks` weight sets are PER PLANE, not per output chunk -- workgroups of "chunk" b take
    // only the patches of plane b and all write the same 64 output channels
    constexpr bool PLANEW = (EPI == 6);
    const int bid = blockIdx.x;
    // default: the nchunks workgroups that share an input patch sit on one XCD (blocks go round-robin to the 8 XCDs).  Per-plane
    // weights: nothing is shared between workgroups, so block b simply takes plane b % B and is the (b / B)-th of that plane's
    // G = ceil((blocks - plane) / B) groups -- every CU is used whatever the plane count (12 planes: 21 or 22 groups each)
    const int chunk = PLANEW ? bid % a.nchunks : (bid >> 3) % a.nchunks;
    const int g = PLANEW ? bid / a.nchunks : (bid & 7) + 8 * (bid / (8 * a.nchunks));
    const int G = PLANEW ? ((int)gridDim.x - chunk + a.nchunks - 1) / a.nchunks : a.G;
    if (g >= G) return;
    const int nitems = (PLANEW ? 1 : a.B) * a.py * a.px;
    const int bofs = PLANEW ? chunk : 0;                          // plane offset of this workgroup's items
    const int K = (nitems - g + G - 1) / G;                       // this workgroup's items: g, g+G, ...
    if (K <= 0) return;

    // The 11 DMA pieces of a patch are raw-buffer loads to LDS: lane part of the source offset in a VGPR, patch origin in an SGPR.
    // Per lane and piece: the byte offset of its pixel of the 34 x 10 patch from the patch origin.  A pixel outside the image gets an offset the buffer unit rejects -- it then writes ZEROS to LDS, which is the conv's
    // zero padding (tools/micro/buffer_oor.hip) -- so only patches on the image border need per-lane work at all, and only when the
    // border pattern differs from the previous patch's (`vkey`).  The descriptor starts one row + one pixel BEFORE the tensor, so
    // that the origin (-1, -1) of the first patch is offset 0 (those bytes are never touched: their lanes are rejected).
    constexpr unsigned kOOR = 0xFFFF0000u;                        // >= num_records for every launch (launcher: < 2^32 - 2^16)
    const unsigned in_px = (unsigned)a.in_cs * 2u;                // bytes per input pixel
    const unsigned in_pad = (unsigned)(a.W + 1) * in_px;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in - in_pad), 0,
                                                                         (unsigned)a.B * a.H * a.W * in_px + in_pad, 0x00020000);
    unsigned vofs[NDMA_W];                                        // this lane's source offset in piece i, for the border pattern `vkey`
    int vkey = -2;                                                // (0: interior patch, -1: nothing to fetch, -2: not built yet)
    // work items g, g+G, g+2G, ... are walked with carries instead of divisions
    const int Gx = G % a.px, Gy = (G / a.px) % a.py, Gb = G / (a.px * a.py);
    auto advance = [&](const Item& it) {
        Item n;
        int x = it.pxi + Gx;
        const int cx = x >= a.px;
        x -= cx ? a.px : 0;
        int y = it.pyi + Gy + cx;
        const int cy = y >= a.py;
        y -= cy ? a.py : 0;
        n.pxi = x; n.pyi = y; n.b = it.b + Gb + cy;
        return n;
    };
    auto patch_src = [&](const Item& it) {
        PatchSrc ps;
        ps.y0 = it.pyi * kTileH - 1; ps.x0 = it.pxi * kTileW - 1;
        ps.soff = (unsigned)(((it.b + bofs) * a.H + ps.y0) * a.W + ps.x0 + a.W + 1) * in_px;
        return ps;
    };
    // called once per patch, before its pieces are issued: rebuild `vofs` if the patch cuts the image border differently (the per-lane
    // pixel coordinates are recomputed here rather than kept: 22 registers for a path that runs on border patches only)
    auto prep_patch = [&](const PatchSrc& ps, bool live) {
        const int xlo = ps.x0 < 0, ylo = ps.y0 < 0;
        const int xhi = max(0, ps.x0 + PW - a.W), yhi = max(0, ps.y0 + PH - a.H);       // columns / rows beyond the right / bottom edge
        const int key = live ? ((xhi * 16 + yhi) * 4 + xlo * 2 + ylo) : -1;
        if (key != vkey) {
            vkey = key;
#pragma unroll
            for (int i = 0; i < NDMA_W; ++i) {
                const int q = (i * 4 + w4) * 8 + (lane >> 3);
                const int r = (q * 241) >> 13;                        // q / 34 for q < 352
                const int c = q - r * PW;
                const int sl = (lane & 7) ^ ((c >> 1) & 7);           // logical 16-B slot behind this physical slot
                const bool ok = ((unsigned)(ps.y0 + r) < (unsigned)a.H) & ((unsigned)(ps.x0 + c) < (unsigned)a.W) & (q < NPIX) & live;
                vofs[i] = ok ? (unsigned)(r * a.W + c) * in_px + (unsigned)sl * 16u : kOOR;
            }
        }
    };
    auto issue_piece = [&](const PatchSrc& ps, int i, char* dstbuf) {
        const int n = i * 4 + w4;                                 // 0..43, no branch: piece 43 is pure padding
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(dstbuf + n * 1024), 16, vofs[i], ps.soff, 0, 0);
    };

    {   // prologue: weights + the first patch.  Every workgroup of a chunk wants the same 72 KiB at the same moment: each starts
        // at a different fragment (rotation by workgroup index) so that they do not all queue on the same L2 channel
        constexpr int p = 0;
        MOE_STAMP(14)
        const half_t* wsrc = a.wpk + (long long)chunk * (WBYTES / 2);
        const int rot = (bid >> 3) % (NFRAG / 4);
        for (int k = 0; k < NFRAG / 4; ++k) {
            int fk = k + rot;
            fk -= fk >= NFRAG / 4 ? NFRAG / 4 : 0;
            const int f = fk * 4 + w4;
            dma16(wsrc + f * 512 + lane * 8, wlds + f * 1024);
        }
        const PatchSrc ps = patch_src(decode(g, a.px, a.py));
        prep_patch(ps, true);
#pragma unroll
        for (int i = 0; i < NDMA_W; ++i) issue_piece(ps, i, pbuf);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        MOE_STAMP(15)
    }

    // LDS read addressing: pixel (row, col) at (row*34 + col)*128, 16-B slot s stored at slot s ^ ((col>>1)&7)
    int Ad[3], Zd[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int c = j + dx;
        const int z = (c >> 1) & 7;
        Ad[dx] = (w4 * 2 * PW + c) * 128 + ((hh ^ (z & 1)) << 4);
        Zd[dx] = (z >> 1) << 5;
    }
    const char* const wl = wlds + lane * 16;

    const int r = a.r;
    const int si = (r > 1) ? chunk / r : 0, sj = (r > 1) ? chunk % r : 0;
    const int cout0 = (r > 1 || PLANEW) ? 0 : chunk * kCB;         // first output channel of this chunk in `out`
    const int Wo = a.W * r, Ho = a.H * r;

    // The bias (an all-zero image when the layer has none) is the C operand of the FIRST MFMA of every accumulator in an
    // iteration, so accumulators are never reset: lane (j, hh) register e of tile [o][nb] is channel nb*32 + 8*(e>>2) + 4*hh + (e&3)
    float16_t biasv[2];
    {
        const float* bsrc = a.bias_img + (PLANEW ? 0 : chunk * 256) + hh * 4;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4_t b4 = *(const float4_t*)(bsrc + nb * 32 + g4 * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) biasv[nb][g4 * 4 + e] = b4[e];
            }
    }
    // predicated-off lanes read/write the 1-KiB slack every activation buffer carries behind its last element
    const unsigned trash_off = (unsigned)a.B * Ho * Wo * a.out_cs + lane * 8;

    float16_t accA[2][2], accB[2][2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) { accA[o][nb] = biasv[nb]; accB[o][nb] = biasv[nb]; }   // (iteration 0 drains B with its stores off)

    // One eighth (index s8 = (o, nb, gp)) of the epilogue of a finished tile held in `ac`.  Branch-free: lanes outside the
    // image (or a disabled slice) store to a trash line and read their residual from the zero page.
    //   4  split-precision FINAL pass: + side16 * 2^-11 (the two low-order products), PReLU in fp32, hi AND lo outputs
    //   5  the same with the residual add (its low part was folded into side16 by the engine)
    //   6  LeakyReLU/PReLU in fp32, THEN the residual add; per-plane weights (see PLANEW above)
    //   7  = 3 with the tail conv's ACTIVATION operand split as well: PReLU in fp32, act = hi + lo 2^-11, a second 8-MFMA GEMM multiplies lo by
    //      the fp16 tail weights placed in rows 16..24 (where the low-order sums of the weight split already accumulate): MOE_PREC_MIXED, R branch
    constexpr bool ACT = (EPI == 1) || (EPI == 3), RES = (EPI == 2) || (EPI == 5) || (EPI == 6), TAIL = (EPI == 3) || (EPI == 7), X3 = (EPI == 4) || (EPI == 5);
    constexpr bool TAIL2 = (EPI == 7);
    constexpr int DRAIN0 = 0;      // first of the eight k-steps that carry a slice of the previous tile's epilogue
#ifndef MOE_DRAIN2
#define MOE_DRAIN2 0      // 1: measured +0.9 % on the launch (the early k-steps get too crowded); the wait it shortens is 300 of 6,260 cycles
#endif
    constexpr bool DRAIN2 = TAIL && MOE_DRAIN2;
    // EPI 6 (SEDN's fused block tail: activation + residual): stores FIRST, loads behind them.  The default schedule alternates one store of the previous
    // tile and one residual load of this one in every k-step, with the DMA pieces in between: every store is issued behind loads in flight and stalls
    // the wave at issue until they return (arsb32c.hip, profiles/r03/k_arsb32c_trace.txt).  PHASED: the eight slices of the previous tile are drained in
    // k-steps 0, 1 (nothing is in flight behind the barrier's vmcnt(0)), the DMA pieces go out in k-steps 2..7, the residual words of this tile in
    // k-steps 2..9 (their registers were released in steps 0, 1); the wait in front of the barrier counts: all but the loads issued behind the last DMA
    // piece.  l25 per 1080p frame: 34.2 -> 33.2 ms (parking the drained slices in registers and storing them in one burst behind the barrier, with
    // the old load timing, measured the same: tools/r03_l.sh).
#ifndef MOE_SP_PHASED
#define MOE_SP_PHASED 1
#endif
    constexpr bool PHASED = (EPI == 6) && MOE_SP_PHASED;
    unsigned slope2;               // {slope, slope} as packed halves
    {
        typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
        const half2_t s2 = {(half_t)a.slope, (half_t)a.slope};
        slope2 = __builtin_bit_cast(unsigned, s2);
    }
    // fused tail: A fragments of the 64->1 conv for the four 16-channel k-slices, rows = taps (9 of 32 used)
    half8_t tailw[4], tailw2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        tailw[i] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        tailw2[i] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        if (TAIL) tailw[i] = *(const half8_t*)(a.tail_w + i * 512 + lane * 8);
        if (TAIL2) tailw2[i] = *(const half8_t*)(a.tail_w + (4 + i) * 512 + lane * 8);     // rows 16..24 = fp16 tail weights (engine.cpp)
    }
    float16_t Gacc[2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int e = 0; e < 16; ++e) Gacc[o][e] = 0.f;
    // tap planes: fp32 [tap 9][phase r*r][b][y][x] at the kernel's INPUT resolution (x contiguous): the HR image of each tap is
    // kept as its r*r pixel-shuffle phases, because one workgroup (= one 64-channel chunk) produces exactly one phase
    const unsigned lrplane = (unsigned)a.B * a.H * a.W;                  // elements per (tap, phase) plane
    const unsigned ph = (unsigned)(si * r + sj);
    const unsigned ttrash = 9u * (unsigned)(r * r) * lrplane + lane * 4;   // slack behind the planes (16 B per lane)
    float tap8 = 0.f;                                                    // centre-bottom tap of row 0, parked until row 1 is done
    // Byte offset of this lane's 16 bytes of slice (o, nb, gp) of tile `it` in `out` (or `res`): everything but the lane term is
    // wave-uniform (SALU), so a slice costs one v_add (SGPR operand) and one v_cndmask, and the access uses the SGPR-base +
    // 32-bit-offset form.  Lanes outside the image (or a disabled slice) go to the slack behind the tensor.
    const unsigned lane_ob = ((unsigned)(j * r) * (unsigned)a.out_cs + (unsigned)hh * 8u) * 2u;
    const unsigned trash_ob = trash_off * 2u;
    struct TileOut { unsigned s00; int y0; bool okx; };      // per tile: offset of its first row in `out`, that row's y, lane inside the image in x
    const unsigned rstride = (unsigned)(r * Wo) * (unsigned)a.out_cs;      // elements between the wave's two output rows
    auto tile_out = [&](const Item& it) {
        TileOut t;
        t.y0 = it.pyi * kTileH + w4 * 2;
        t.s00 = ((unsigned)((it.b + bofs) * Ho + t.y0 * r + si) * (unsigned)Wo + (unsigned)(it.pxi * kTileW * r + sj)) * (unsigned)a.out_cs + (unsigned)cout0;
        t.okx = it.pxi * kTileW + j < a.W;
        return t;
    };
    auto out_off = [&](const TileOut& t, int o, int nb, int gp, bool live) {
        const unsigned srow = t.s00 + (unsigned)o * rstride + (unsigned)(nb * 32 + gp * 16);
        const bool oky = (t.y0 + o < a.H) & live;
        return (t.okx & oky) ? lane_ob + srow * 2u : trash_ob;
    };
    auto drain_slice = [&](float16_t (&ac)[2][2], const Item& it, const TileOut& to, int s8, bool live, const uint4* resw = nullptr, const uint4* sidew = nullptr) {
        const int o = s8 >> 2, nb = (s8 >> 1) & 1, gp = s8 & 1;
        const int y = it.pyi * kTileH + w4 * 2 + o, x = it.pxi * kTileW + j;
        const bool ok = (y < a.H) & (x < a.W) & live;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = ac[o][nb][gp * 8 + e];     // channels nb*32 + 16*gp + {0..3 | 8..11} + 4*hh
        if (MOE_ABL & 4) {         // ablation: accumulator reads only
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("" ::"v"(v[e]));
            return;
        }
        if (X3) {                  // low-order products, fetched like a residual (16 bytes per lane in the store layout), in units of 2^-11
            const uint4 w = sidew[s8];
            const auto qx = __builtin_amdgcn_permlane32_swap(w.x, w.z, false, false);
            const auto qy = __builtin_amdgcn_permlane32_swap(w.y, w.w, false, false);
            const half4_t q0 = __builtin_bit_cast(half4_t, make_uint2(qx[0], qy[0]));
            const half4_t q1 = __builtin_bit_cast(half4_t, make_uint2(qx[1], qy[1]));
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += (float)q0[e] * 0.00048828125f; v[4 + e] += (float)q1[e] * 0.00048828125f; }
            if (EPI == 4) {        // PReLU / plain (slope == 1) in fp32: the low output part needs the unrounded value
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = v[e] * a.slope;
                    asm("v_max_f32 %0, %1, %2" : "=v"(v[e]) : "v"(v[e]), "v"(t));
                }
            }
        }
        // fp16 first, PReLU on the packed halves: slope <= 1 (negative slopes included) => PReLU(x) = max(x, slope*x); two
        // v_pk_* per register pair instead of v_mul_f32 + v_max_f32 per value (-64 VALU per tile).  The negative branch is rounded
        // three times instead of once; on the goldens the end-to-end error is unchanged (a2 7.0e-4 -> 6.6e-4, a4 5.3e-4 -> 5.5e-4).
        unsigned hv[4];            // the eight values as four half2 registers (v[2k], v[2k+1])
        unsigned lv[4];            // TAIL2: their rounding remainders (v - hi) * 2^11
        if (TAIL2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = __builtin_fmaxf(v[e], v[e] * a.slope);       // PReLU in fp32 (slope <= 1)
#pragma unroll
            for (int k = 0; k < 4; ++k) split2(v[2 * k], v[2 * k + 1], -2048.f, hv[k], lv[k]);     // (rowtile.h: 5 instructions per pair with v_fma_mixlo/hi_f16; 9 before)
        }
        if (!RES && !X3 && !TAIL2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
                const half2_t pr = {(half_t)v[2 * k], (half_t)v[2 * k + 1]};
                hv[k] = __builtin_bit_cast(unsigned, pr);
                if (ACT) {         // v_pk_mul_f16 + v_pk_max_f16 (the file is built with -fno-honor-nans: no canonicalising max per operand; as
                                   // inline asm each pair cost an s_nop the hazard recogniser puts behind instructions it cannot see into)
                    const half2_t t = pr * __builtin_bit_cast(half2_t, slope2);
                    hv[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(pr, t));
                }
            }
        }
        if (TAIL) {
            const half8_t bf = __builtin_bit_cast(half8_t, make_uint4(hv[0], hv[1], hv[2], hv[3]));   // k = 8*hh + e  <->  channel nb*32 + 16*gp + perm(hh, e)
            Gacc[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tailw[nb * 2 + gp], bf, Gacc[o], 0, 0, 0);
            if (TAIL2) {
                const half8_t bl = __builtin_bit_cast(half8_t, make_uint4(lv[0], lv[1], lv[2], lv[3]));
                Gacc[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(tailw2[nb * 2 + gp], bl, Gacc[o], 0, 0, 0);
            }
            if ((s8 & 3) == 3) {     // row o complete: lane (j, hh) holds taps 4*hh .. 4*hh+3 in regs 0..3 and tap 8 + 4*hh in reg 4
                // rows 16..24 of the A fragments hold the rounding remainders of the tail weights (units of 2^-11, engine.cpp): the
                // same MFMAs formed the low-order sums in regs 8..12 -- fold them in (5 FMAs per row for ~22-bit tail weights)
#pragma unroll
                for (int k = 0; k < 5; ++k) Gacc[o][k] = __builtin_fmaf(Gacc[o][8 + k], 0.00048828125f, Gacc[o][k]);
                // 4-byte-per-lane stores are what this epilogue paid for (~110 cycles each beside the MFMAs; 16-byte ones are nearly
                // free) while VALU work hides: so a 4x4 transpose inside each lane quad (two DPP exchange stages) turns the four
                // one-float-per-lane tap registers into ONE 16-byte store: lane (4m+i, hh) ends up with tap 4hh+i of pixels 4m..4m+3.
                auto xq = [](float v, bool far) {   // value of the quad partner: lane^1 (quad_perm [1,0,3,2]) or lane^2 ([2,3,0,1])
                    const int u = __builtin_bit_cast(int, v);
                    return __builtin_bit_cast(float, far ? __builtin_amdgcn_mov_dpp(u, 0x4E, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(u, 0xB1, 0xF, 0xF, true));
                };
                const bool b0 = j & 1, b1 = j & 2;
                float t[4], f[4];
#pragma unroll
                for (int k = 0; k < 4; k += 2) {
                    const float p0 = xq(Gacc[o][k], false), p1 = xq(Gacc[o][k + 1], false);
                    t[k] = b0 ? p1 : Gacc[o][k];
                    t[k + 1] = b0 ? Gacc[o][k + 1] : p0;
                }
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const float q0 = xq(t[c], true), q2 = xq(t[2 + c], true);
                    f[c] = b1 ? q2 : t[c];
                    f[2 + c] = b1 ? t[2 + c] : q0;
                }
                const unsigned xq0 = (unsigned)(it.pxi * kTileW + (j & ~3));
                const bool okq = (y < a.H) & (xq0 < (unsigned)a.W) & live;        // W is a multiple of 4: a quad is all in or all out
                const unsigned lrow = (unsigned)(it.b * a.H + y) * (unsigned)a.W;
                const unsigned off = ((unsigned)(4 * hh + (j & 3)) * (unsigned)(r * r) + ph) * lrplane + lrow + xq0;
                // tap 8 (register 4 of the hh == 0 lanes): row 0 waits in `tap8`; with row 1 done, one v_permlane32_swap puts row 1
                // into the hh == 1 lanes and a single store covers both rows
                float v8 = 0.f;
                if (o == 0) tap8 = Gacc[0][4];
                else {
                    const float row1 = Gacc[1][4];   // (a bit_cast applied directly to the vector element picks element 0)
                    v8 = __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, tap8), __builtin_bit_cast(unsigned, row1), false, false)[0]);
                }
                const int y8 = it.pyi * kTileH + w4 * 2 + hh;
                const bool ok8 = (y8 < a.H) & (x < a.W) & live;
                const unsigned off8 = (8u * (unsigned)(r * r) + ph) * lrplane + (unsigned)(it.b * a.H + y8) * (unsigned)a.W + (unsigned)x;
                if (MOE_ABL & 8) {
                    asm volatile("" ::"v"(f[0]), "v"(f[1]), "v"(f[2]), "v"(f[3]), "v"(v8));
                } else {
                    *(float4*)(a.tplanes + (okq ? off : ttrash)) = make_float4(f[0], f[1], f[2], f[3]);
                    if (o == 1) a.tplanes[ok8 ? off8 : ttrash] = v8;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) Gacc[o][e] = 0.f;
            }
            return;
        }
        if (EPI == 6) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = v[e] * a.slope;
                asm("v_max_f32 %0, %1, %2" : "=v"(v[e]) : "v"(v[e]), "v"(t));
            }
        }
        if (RES) {
            // the residual was fetched as one 16-byte access per lane in the STORE layout (lane (j,0): channels 16*gp..+7,
            // lane (j,1): +8..+15); the same v_permlane32_swap pair that builds that layout also undoes it
            const uint4 w = resw[s8];
            const auto rx = __builtin_amdgcn_permlane32_swap(w.x, w.z, false, false);
            const auto ry = __builtin_amdgcn_permlane32_swap(w.y, w.w, false, false);
            const half4_t r0 = __builtin_bit_cast(half4_t, make_uint2(rx[0], ry[0]));
            const half4_t r1 = __builtin_bit_cast(half4_t, make_uint2(rx[1], ry[1]));
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += (float)r0[e]; v[4 + e] += (float)r1[e]; }
        }
        // fp16, then one v_permlane32_swap per register: lane (j,0) gets channels 16*gp .. +7, lane (j,1) 16*gp+8 .. +15
        uint2 u0, u1;
        if (RES || X3) {
            half4_t h0, h1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { h0[e] = (half_t)v[e]; h1[e] = (half_t)v[4 + e]; }
            u0 = __builtin_bit_cast(uint2, h0); u1 = __builtin_bit_cast(uint2, h1);
            if (X3) {              // low part of the output: (v - hi) * 2^11, same layout, second store
                half4_t l0, l1;
#pragma unroll
                for (int e = 0; e < 4; ++e) { l0[e] = (half_t)((v[e] - (float)h0[e]) * 2048.f); l1[e] = (half_t)((v[4 + e] - (float)h1[e]) * 2048.f); }
                const uint2 m0 = __builtin_bit_cast(uint2, l0), m1 = __builtin_bit_cast(uint2, l1);
                const auto lx = __builtin_amdgcn_permlane32_swap(m0.x, m1.x, false, false);
                const auto ly = __builtin_amdgcn_permlane32_swap(m0.y, m1.y, false, false);
                *(uint4*)((char*)a.out_lo + out_off(to, o, nb, gp, live)) = make_uint4(lx[0], ly[0], lx[1], ly[1]);
            }
        } else {
            u0 = make_uint2(hv[0], hv[1]); u1 = make_uint2(hv[2], hv[3]);
        }
        const auto sx = __builtin_amdgcn_permlane32_swap(u0.x, u1.x, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(u0.y, u1.y, false, false);
        if (MOE_ABL & 8) {
            asm volatile("" ::"v"(sx[0]), "v"(sy[0]), "v"(sx[1]), "v"(sy[1]));
            return;
        }
        if ((MOE_ABL & 64) && (s8 & 1)) {   // ablation: every other store dropped
            asm volatile("" ::"v"(sx[0]), "v"(sy[0]), "v"(sx[1]), "v"(sy[1]));
            return;
        }
        if (MOE_ABL & 16) {        // ablation: every store goes to the (L2-resident) slack
            *(uint4*)(a.out + trash_off) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            return;
        }
        if (MOE_ABL & 32) {        // ablation: same footprint, but each store instruction writes 1 KiB contiguously
            const unsigned lin = ((unsigned)(it.b * Ho + y * r) * (unsigned)Wo + (unsigned)(it.pxi * kTileW * r)) * (unsigned)a.out_cs + (unsigned)((nb * 2 + gp) * 512 + lane * 8);
            *(uint4*)(a.out + lin) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
            return;
        }
        *(uint4*)((char*)a.out + out_off(to, o, nb, gp, live)) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
    };

    // residual of the tile being MULTIPLIED (drained one iteration later): slice k is fetched in k-step k+1, right after the
    // drain of the previous tile released its registers, so every load has eleven k-steps to land (issued at the start of the
    // draining iteration they stalled its first slice for a full memory latency)
    uint4 resw[8];
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) resw[s8] = make_uint4(0, 0, 0, 0);
    auto fetch_res = [&](const TileOut& to, int s8) {
        unsigned off = out_off(to, s8 >> 2, (s8 >> 1) & 1, s8 & 1, true);
        asm volatile("" : "+v"(off));        // keep the select: the compiler otherwise turns it into two predicated loads behind branches
        resw[s8] = *(const uint4*)((const char*)a.res + off);
    };
    uint4 sidew[8];                // split-precision final pass: the low-order addend of the tile being multiplied, fetched like the residual
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) sidew[s8] = make_uint4(0, 0, 0, 0);
    auto fetch_side = [&](const TileOut& to, int s8) {
        unsigned off = out_off(to, s8 >> 2, (s8 >> 1) & 1, s8 & 1, true);
        asm volatile("" : "+v"(off));
        sidew[s8] = *(const uint4*)((const char*)a.side16 + off);
    };
    Item it_cur = decode(g, a.px, a.py), it_prev = it_cur, it_next = advance(it_cur);
    // iteration p (0 <= p < K): multiply patch p (buffer p&1) into `cur`; drain patch p-1 from `prev` (stores predicated
    // off for p == 0); fetch patch p+1 (source predicated to the zero page for the last one).  No branches inside: the
    // whole iteration is one scheduling region so the VALU/VMEM work lands in the shadow of the MFMAs.
    half8_t wf[2][3][2], af[2][4];     // double-buffered MFMA operand fragments (k-step parity)
    auto iteration = [&](int p, float16_t (&cur)[2][2], float16_t (&prev)[2][2]) {
        const bool drain = (p >= 1) && !(a.dbg & 4);
        const bool fetch = (p + 1 < K) && !(a.dbg & 1);
        const Item itp = it_prev;
        const TileOut top = tile_out(itp), toc = tile_out(it_cur);     // addressing of the drained tile / of the residual being fetched
        const PatchSrc ps = patch_src(it_next);
        const char* abuf = pbuf + (p & 1) * PATCH_BYTES;
        char* nbuf = pbuf + ((p + 1) & 1) * PATCH_BYTES;
        prep_patch(ps, fetch);
        MOE_STAMP(0)
        if (RES) {   // pin the compiler's wait for the residual registers here, before this iteration issues any memory operation
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) asm volatile("" : "+v"(resw[s8].x), "+v"(resw[s8].y), "+v"(resw[s8].z), "+v"(resw[s8].w));
        }
        if (X3) {
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) asm volatile("" : "+v"(sidew[s8].x), "+v"(sidew[s8].y), "+v"(sidew[s8].z), "+v"(sidew[s8].w));
        }
// the ten reads of a k-step in the order its MFMAs consume them (input row pr, then the weights of the tap row it meets first):
// LDS returns in order, so the first MFMAs of the step wait for two reads instead of seven
#define MOE_LOAD_STEP(S, BUF, PB)                                                                            \
    {                                                                                                      \
        constexpr int dx_ = (S) / 4, ks_ = (S) % 4;                                                        \
        const char* ap_ = (PB) + Ad[dx_] + ((ks_ << 5) ^ Zd[dx_]);                                         \
        _Pragma("unroll") for (int pr = 0; pr < 4; ++pr) {                                                 \
            af[BUF][pr] = *(const half8_t*)(ap_ + pr * (PW * 128));                                        \
            if (pr < 3) {                                                                                  \
                _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)                                           \
                    wf[BUF][pr][nb] = *(const half8_t*)(wl + ((((pr * 3 + dx_) * 4 + ks_) * 2 + nb) << 10)); \
            }                                                                                              \
        }                                                                                                  \
    }
        // (the fragments of k-step 0 were read in k-step 11 of the previous iteration, behind the barrier -- see below)
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            const int cb = s & 1;
            // The fragments this step multiplies were read behind the first MFMAs of the previous step: they landed long ago.  Saying so
            // once (lgkmcnt(0), the builtin so that the compiler's wait-count pass sees it) replaces the 7-8 counted waits per step it
            // otherwise puts between the MFMAs -- it has to count the NEXT step's reads, issued in between, against every older one
            // (91 s_waitcnt per 152 MFMAs in the PMC instruction mix, each an issue slot).
#ifndef MOE_NO_STEP_WAIT
            __builtin_amdgcn_s_waitcnt(0xC07F);
#endif
            switch (s + 1) {   // constant after unrolling
#define MOE_CASE(N) case N: MOE_LOAD_STEP(N, ((N) & 1), abuf) break;
                MOE_CASE(1) MOE_CASE(2) MOE_CASE(3) MOE_CASE(4) MOE_CASE(5) MOE_CASE(6)
                MOE_CASE(7) MOE_CASE(8) MOE_CASE(9) MOE_CASE(10) MOE_CASE(11)
#undef MOE_CASE
                default: break;
            }
            if (s == 11) {
                // The workgroup barrier sits HERE, not at the end of the iteration: k-step 11 already holds its operands in
                // registers, so nobody reads the current patch buffer any more (the next iteration may overwrite it), and the
                // DMA pieces of the next patch (issued in k-steps 0..5) have had five steps to land.  Behind the barrier the
                // fragments of the NEXT iteration's k-step 0 are read while this step's twelve MFMAs run: no LDS round trip
                // between the last MFMA of a tile and the first of the next.
                MOE_STAMP(1)
                if (PHASED) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");      // (the three residual words requested behind the last DMA piece may still fly)
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                MOE_STAMP(2)
                __builtin_amdgcn_s_barrier();      // bare: __syncthreads() carries a fence that drains vmcnt/lgkmcnt again
                asm volatile("" ::: "memory");
                MOE_STAMP(3)
                MOE_LOAD_STEP(0, 0, nbuf)
            }
            if (!(MOE_ABL & 1)) {   // compile-time timing ablation (tools/ablate_sp.sh); 0 in the product build
                if (PHASED) {
                    if (s >= 2 && s < 7) { issue_piece(ps, 2 * (s - 2), nbuf); issue_piece(ps, 2 * (s - 2) + 1, nbuf); }
                    if (s == 7) issue_piece(ps, 10, nbuf);
                } else {
                    if (s < 5) { issue_piece(ps, 2 * s, nbuf); issue_piece(ps, 2 * s + 1, nbuf); }
                    if (s == 5) issue_piece(ps, 10, nbuf);
                }
            }
#pragma unroll
            for (int pr = 0; pr < 4; ++pr)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int o = pr - dy;
                    if (o >= 0 && o < 2) {
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)     // (s, dy) == (0, 0) is the first product of accumulator [o][nb]
                            cur[o][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cb][dy][nb], af[cb][pr], (s == 0 && dy == 0) ? biasv[nb] : cur[o][nb], 0, 0, 0);
                    }
                }
            if (PHASED) {
                if (s < 2) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) drain_slice(prev, itp, top, 4 * s + q, drain, resw, sidew);
                }
                if (s >= 2 && s < 10) fetch_res(toc, s - 2);
            } else
            if (DRAIN2) {      // fused tail: two slices per step in k-steps 0..3 -- the tap-plane stores (slices 3 and 7) then have eight k-steps
                               // to be acknowledged before the vmcnt(0) in front of the barrier (traces: that wait was 300 cycles per patch)
                if (!(MOE_ABL & 2) && s < 4) { drain_slice(prev, itp, top, 2 * s, drain, resw, sidew); drain_slice(prev, itp, top, 2 * s + 1, drain, resw, sidew); }
            } else
            if (!(MOE_ABL & 2) && s >= DRAIN0 && s < DRAIN0 + 8) drain_slice(prev, itp, top, s - DRAIN0, drain, resw, sidew);
            if (RES && !PHASED && s >= DRAIN0 + 1 && s <= DRAIN0 + 8) fetch_res(toc, s - DRAIN0 - 1);   // slice s-1's registers were consumed in the previous step
            if (X3 && s >= DRAIN0 + 1 && s <= DRAIN0 + 8) fetch_side(toc, s - DRAIN0 - 1);
#ifdef MOE_STEP_STAMPS
            MOE_STAMP(4 + s)
#endif
            // pin the issue order: the ten LDS reads of the next step go out behind the first five MFMAs (their latency
            // then hides under the other seven), the two DMA pieces and the store sit in the middle, and every MFMA is
            // followed by up to five VALU instructions of the drain / address arithmetic (what fits in a 32-cycle shadow)
#ifndef MOE_NO_SGB
#ifndef SGB_VALU_A
#define SGB_VALU_A 3
#endif
#ifndef SGB_VALU_B
#define SGB_VALU_B 5
#endif
#ifndef SGB_DSR_SLOTS
#define SGB_DSR_SLOTS 5
#endif
#ifndef SGB_DMA_AT
#define SGB_DMA_AT 5
#endif
#ifndef SGB_ST_AT
#define SGB_ST_AT 8
#endif
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < SGB_DSR_SLOTS) __builtin_amdgcn_sched_group_barrier(0x100, 10 / SGB_DSR_SLOTS, 0);
                if (i < SGB_DSR_SLOTS) __builtin_amdgcn_sched_group_barrier(0x002, SGB_VALU_A, 0);
                else __builtin_amdgcn_sched_group_barrier(0x002, SGB_VALU_B, 0);
                if (PHASED) {
                    if ((i == SGB_DMA_AT || i == SGB_DMA_AT + 1) && s >= 2 && s <= 7) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    if (i == SGB_DMA_AT + 2 && s >= 2 && s < 10) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    if ((i == 2 || i == 5 || i == 8 || i == 11) && s < 2) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
                    continue;
                }
                if ((i == SGB_DMA_AT || i == SGB_DMA_AT + 1) && s <= 5) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (RES && i == SGB_DMA_AT + 2 && s >= DRAIN0 + 1 && s <= DRAIN0 + 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (X3 && i == SGB_DMA_AT + 3 && s >= DRAIN0 + 1 && s <= DRAIN0 + 8) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if (X3 && i == SGB_ST_AT + 1 && s >= DRAIN0 && s < DRAIN0 + 8) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
                if (i == SGB_ST_AT && (DRAIN2 ? (s == 1 || s == 3) : (s >= DRAIN0 && s < DRAIN0 + 8))) __builtin_amdgcn_sched_group_barrier(0x040, DRAIN2 ? 2 : 1, 0);
            }
            // the fused tail adds a 13th MFMA to k-steps 0..7: without a slot of its own it takes the next step's first MFMA slot
            // and every later group slides by one, until the LDS reads land right in front of their consumers
            if (TAIL && (DRAIN2 ? s < 4 : (s >= DRAIN0 && s < DRAIN0 + 8))) {
#pragma unroll
                for (int i = 0; i < (DRAIN2 ? 2 : 1) * (TAIL2 ? 2 : 1); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, SGB_VALU_B, 0);
                }
            }
            // Nothing moves across a k-step boundary.  As ONE region the sched_group_barrier slot pattern drifts whenever the amount of
            // filler work per step changes (a 13th MFMA in the fused-tail variant, fewer VALU after an epilogue diet ...) until the
            // LDS reads sit right in front of their consumers: 39-83 of 100 reads with < 6 MFMAs of distance and +15 % cycles were
            // seen; fenced, every step compiles to `wait, 5 x (MFMA, 2 reads), DMA/MFMA mix, 7 MFMAs` (tools/seq_view.py).
#ifndef MOE_STEP_FENCE_NONE
            __builtin_amdgcn_sched_barrier(0);
#endif
#endif
        }
        if (MOE_ABL & 2) {   // keep the MFMAs alive without a drain
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) asm volatile("" ::"a"(prev[o][nb]));
        }
        it_prev = it_cur; it_cur = it_next; it_next = advance(it_next);
    };

    {
        const char* abuf = pbuf;
        MOE_LOAD_STEP(0, 0, abuf)      // k-step 0 of the first tile (every later one is read in k-step 11 of its predecessor)
    }
#undef MOE_LOAD_STEP
    int p = 0;
    for (; p + 1 < K; p += 2) {
        iteration(p, accA, accB);
        iteration(p + 1, accB, accA);
    }
    if (p < K) {
        iteration(p, accA, accB);
        ++p;
    }
    {   // drain the last tile (patch K-1): it sits in A when K is odd, in B when K is even
        const Item itp = it_prev;
        const TileOut top = tile_out(itp);
        const bool live = !(a.dbg & 4);
        // (its residual was fetched in steps 8..11 of the last iteration)
        if (K & 1) {
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) drain_slice(accA, itp, top, s8, live, resw, sidew);
        } else {
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) drain_slice(accB, itp, top, s8, live, resw, sidew);
        }
    }
    {   // trace: end of the workgroup's work (slot 13 of iteration 0; slot 14 there is its start)
        constexpr int p = 0;
        if (a.dbg & 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MOE_STAMP(13)
    }
#endif
}

template <int EPI>
hipError_t set_limit()
{
    return hipFuncSetAttribute((const void*)conv3x3_sp_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}

}  // namespace

hipError_t conv3x3_sp_init()
{
    hipError_t e;
    if ((e = set_limit<0>()) != hipSuccess) return e;
    if ((e = set_limit<1>()) != hipSuccess) return e;
    if ((e = set_limit<2>()) != hipSuccess) return e;
    if ((e = set_limit<3>()) != hipSuccess) return e;
    if ((e = set_limit<4>()) != hipSuccess) return e;
    if ((e = set_limit<5>()) != hipSuccess) return e;
    if ((e = set_limit<6>()) != hipSuccess) return e;
    if ((e = set_limit<7>()) != hipSuccess) return e;
    return hipSuccess;
}

// Returns false when the layer's epilogue is not one of the four compiled variants (caller uses another kernel).
bool launch_conv3x3_sp(const ConvArgs& a, hipStream_t s)
{
    const bool x3 = a.acc_mode == 3 && a.side16 && a.out_lo;          // split-precision final pass (the low-order products are in side16)
    if ((a.acc_mode != 0 && !x3 && !(a.dbg & 64)) || a.slope > 1.f) return false;
    // 32-bit BYTE offsets for stores / residual loads (the fused-tail epilogue stores no tensor: its tap planes are checked below)
    if (!a.tplanes && 2ll * a.B * a.H * a.r * a.W * a.r * a.out_cs >= (1ll << 32) - 8192) return false;
    if (2ll * a.B * a.H * a.W * a.in_cs + 2ll * (a.W + 1) * a.in_cs >= (1ll << 32) - 65536) return false;   // 32-bit buffer offsets of the patch DMA
    if (a.scale != 1.f || !a.bias_img) return false;           // the engine folds ScaleLayer into the weights and always passes a bias vector
    const bool act = a.slope != 1.f, res = a.res != nullptr, tail = a.tplanes != nullptr;
    if ((act || tail) && res && !a.plane_w) return false;
    if (x3 && (tail || a.res_lo)) return false;                        // (the engine folds res_lo into side16)
    if (tail && 36ll * a.B * a.H * a.r * a.W * a.r >= (1ll << 32) - 8192) return false;
    const bool planew = a.plane_w != 0;                                 // per-plane weights: act + residual only (SEDN fused block tail)
    if (planew && (!res || tail || x3 || a.r != 1)) return false;
    const int epi = planew ? 6 : x3 ? (res ? 5 : 4) : tail ? (a.tail_split ? 7 : 3) : (res ? 2 : (act ? 1 : 0));
    const int blocks = planew ? a.G : a.nchunks * ((a.G + 7) / 8) * 8;          // per-plane weights: a.G is the TOTAL number of workgroups
    const dim3 grid(blocks), blk(256);
    switch (epi) {
        case 0: conv3x3_sp_kernel<0><<<grid, blk, LDS_BYTES, s>>>(a); break;
        case 1: conv3x3_sp_kernel<1><<<grid, blk, LDS_BYTES, s>>>(a); break;
        case 3: conv3x3_sp_kernel<3><<<grid, blk, LDS_BYTES, s>>>(a); break;
        case 4: conv3x3_sp_kernel<4><<<grid, blk, LDS_BYTES, s>>>(a); break;
        case 5: conv3x3_sp_kernel<5><<<grid, blk, LDS_BYTES, s>>>(a); break;
        case 6: conv3x3_sp_kernel<6><<<grid, blk, LDS_BYTES, s>>>(a); break;
        case 7: conv3x3_sp_kernel<7><<<grid, blk, LDS_BYTES, s>>>(a); break;
        default: conv3x3_sp_kernel<2><<<grid, blk, LDS_BYTES, s>>>(a); break;
    }
    return true;
}
