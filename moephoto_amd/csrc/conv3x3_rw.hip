// conv3x3_rw.hip -- the upsampler convolutions (3x3, 64 -> 64 r^2, + bias, PixelShuffle(r), PReLU: python/models.py:29-39 of the reference)
// with the weights RESIDENT IN REGISTERS, second form of the hot kernel (conv3x3_sp.hip is the first and stays the fallback):
//
//   EPI 1   PReLU, pixel shuffle folded into the 16-byte stores
//   EPI 4   = 1 + per-plane channel sums of the stored output (one chunk, r = 1: SEDN's rblock.2, whose sums feed the fused block tail)
//   EPI 3   PReLU + the fused 64 -> 1 tail conv: per-tap partial sums to the phase-separated tap planes (see conv3x3_sp.hip)
//
// What the PMC passes of round 2 said about conv3x3_sp (profiles/r02/i_*): 72 % MFMA busy, 4.0 other instructions per MFMA and 120
// ds_read_b128 per 152 MFMAs and wave -- the four waves keep the CU's one LDS pipe 64 % busy with fragment reads, 72 of the 120 for
// weights.  Here a wave owns 32 of the chunk's 64 output channels and half of the patch rows:
//
//   wave (c, h)   output channels 32c .. 32c+31, output rows 4h .. 4h+3 of the 8 x 32 patch; its 36 A fragments (9 taps x 4 k-slices of
//                 v_mfma_f32_32x32x16_f16) are loaded once per launch and parked in accumulator registers (144);
//   rows stream   input row r of the wave's six (one ds_read_b128 per (dx, k-slice): 12 reads) feeds the up to three output rows it
//                 touches: 72 reads for 144 MFMAs per patch and wave, nothing else is read from LDS in the matrix loop;
//   epilogue      output row o is complete after input row o+2; its epilogue rides in the MFMA stream of the following rows (rows 2, 3: in
//                 the first row steps of the next patch), in pieces of a few instructions per B fragment;
//   fused tail    the 64-channel contraction of the tail conv now spans two waves: each forms the partial sums of its 32 channels
//                 (2 MFMAs per row), folds the low-order weight rows in and leaves 5 floats per lane in an LDS exchange area; two patches
//                 later (one barrier per patch publishes them) wave (c, h) adds the two halves of rows 4h+2c, 4h+2c+1 and stores the
//                 tap planes exactly as conv3x3_sp does.  Three exchange buffers rotate (written in patch p / p+1, read in p+2).
//   DMA, barrier  halo'd 10 x 34 patches, double buffered, raw-buffer loads to LDS as in conv3x3_sp; the one barrier per patch sits at
//                 the start of the last row step (its fragments are in registers), behind it the first fragments of the next patch
//                 are read while that step's MFMAs run.
//
// LDS: 2 x 45,056 (patches) + 3 x 20,480 (tail exchange) = 151,552 B.
#include "common.h"
#include "rowtile.h"
#include <type_traits>

namespace {

constexpr int PW = kTileW + 2, PH = kTileH + 2, NPIX = PW * PH;   // 34 x 10 halo'd patch
constexpr int NDMA_W = 11;                                         // 1-KiB pieces per wave (44 >= 340 / 8)
constexpr int PATCH_BYTES = NDMA_W * 4 * 1024;                      // 45,056
constexpr int XROW = 1024 + 256;                                   // exchange record of one (channel half, row): 64 x 4 floats + 64 x 1 float
constexpr int XCH_BYTES = 2 * 8 * XROW;                            // 20,480
constexpr int LDS_BYTES = 2 * PATCH_BYTES + 3 * XCH_BYTES;         // 151,552

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;

struct Item { int b, pyi, pxi; };

template <int EPI>
__global__ __launch_bounds__(256) void conv3x3_rw_kernel(ConvArgs a)
{
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass only needs the launch stub)
    constexpr bool TAIL = EPI == 3, POOL = EPI == 4;
    constexpr bool PERM = !TAIL;         // EPI 1: channel order that makes registers 8g..8g+7 eight consecutive channels (16-byte stores)
    constexpr unsigned kOOR = 0xFFFF0000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = w4 & 1, h = w4 >> 1;
    const int j = lane & 31, hh = lane >> 5;

    // ---- workgroup -> (chunk, group): as conv3x3_sp (the nchunks workgroups that share an input patch sit on one XCD) ----------------
    const int bid = blockIdx.x;
    const int chunk = (bid >> 3) % a.nchunks;
    const int g = (bid & 7) + 8 * (bid / (8 * a.nchunks));
    const int G = a.G;
    if (g >= G) return;
    const int nitems = a.B * a.py * a.px;
    const int K = (nitems - g + G - 1) / G;
    if (K <= 0) return;
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

    // ---- weights: 36 A fragments (tap, k-slice) of this wave's 32 output channels, conv3x3_sp / pack_conv fragment order ----------------
    half8_t wf[36];
    {
        int src = lane;
        if (PERM) {      // MFMA row i = 8q + 4h' + e (register 4q + e of the lanes hh = h') is given channel 16 (q >> 1) + 8 h' + 4 (q & 1) + e
            const int wi = lane & 31, wq = wi >> 3;
            src = (lane & 32) | (16 * (wq >> 1) + 8 * ((wi >> 2) & 1) + 4 * (wq & 1) + (wi & 3));
        }
        const half_t* wsrc = a.wpk + (long long)chunk * (72 * 512);
#pragma unroll
        for (int f = 0; f < 36; ++f) wf[f] = *(const half8_t*)(wsrc + ((f * 2 + c) * 64 + src) * 8);
#pragma unroll
        for (int f = 0; f < 36; ++f) asm volatile("" : "+a"(wf[f]));
    }
    // bias of this lane's 16 channels: the C operand of the first MFMA of every output row
    float16_t biasv;
    {
        const float* bsrc = a.bias_img + chunk * 256 + 32 * c;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) biasv[4 * q + e] = bsrc[PERM ? 16 * (q >> 1) + 8 * hh + 4 * (q & 1) + e : 8 * q + 4 * hh + e];
    }

    // ---- patch DMA (see conv3x3_sp.hip): lane offsets rebuilt only when the border pattern changes, patch origin in an SGPR -------------
    const unsigned in_px = (unsigned)a.in_cs * 2u;
    const unsigned in_pad = (unsigned)(a.W + 1) * in_px;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in - in_pad), 0,
                                                                         (unsigned)a.B * a.H * a.W * in_px + in_pad, 0x00020000);
    unsigned vofs[NDMA_W];
    int vkey = -2;
    auto prep_patch = [&](const Item& it, bool live) {
        const int y0 = it.pyi * kTileH - 1, x0 = it.pxi * kTileW - 1;
        const int xhi = max(0, x0 + PW - a.W), yhi = max(0, y0 + PH - a.H);
        const int key = live ? ((xhi * 16 + yhi) * 4 + (x0 < 0) * 2 + (y0 < 0)) : -1;
        if (key != vkey) {
            vkey = key;
#pragma unroll
            for (int i = 0; i < NDMA_W; ++i) {
                const int q = (i * 4 + w4) * 8 + (lane >> 3);
                const int r = (q * 241) >> 13;                        // q / 34 for q < 352
                const int cc = q - r * PW;
                const int sl = (lane & 7) ^ ((cc >> 1) & 7);          // logical 16-B slot behind this physical slot
                const bool ok = ((unsigned)(y0 + r) < (unsigned)a.H) & ((unsigned)(x0 + cc) < (unsigned)a.W) & (q < NPIX) & live;
                vofs[i] = ok ? (unsigned)(r * a.W + cc) * in_px + (unsigned)sl * 16u : kOOR;
            }
        }
    };
    auto origin = [&](const Item& it) {
        return (unsigned)((it.b * a.H + it.pyi * kTileH - 1) * a.W + it.pxi * kTileW - 1 + a.W + 1) * in_px;
    };

    // ---- LDS read addressing of a B fragment f = (dx, ks): pixel (row, col) at (row * 34 + col) * 128, 16-B slot s at s ^ ((col >> 1) & 7);
    // lane (j, hh) reads slot 2 ks + hh of column j + dx; rows are immediate offsets.  fa[0]: patch buffer 0, fa[1]: buffer 1
    unsigned fa[2][12];
#pragma unroll
    for (int f = 0; f < 12; ++f) {
        const int dx = f >> 2, ks = f & 3;
        const int cc = j + dx, z = (cc >> 1) & 7;
        const unsigned o = (unsigned)((4 * h * PW + cc) * 128 + (((2 * ks + hh) ^ z) << 4));
        fa[0][f] = lds0 + o;
        fa[1][f] = lds0 + (unsigned)PATCH_BYTES + o;
        asm volatile("" : "+v"(fa[0][f]), "+v"(fa[1][f]));
    }
    const unsigned xch0 = lds0 + 2u * PATCH_BYTES;                  // exchange area

    // ---- output addressing -----------------------------------------------------------------------------------------------------------
    const int r = a.r;
    const int si = (r > 1) ? chunk / r : 0, sj = (r > 1) ? chunk % r : 0;
    const int cout0 = (r > 1) ? 0 : chunk * kCB;
    const unsigned Wo = (unsigned)(a.W * r), Ho = (unsigned)(a.H * r);
    const unsigned out_px = (unsigned)a.out_cs * 2u;
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)(TAIL ? (half_t*)a.in : a.out), 0,
                                                                          TAIL ? 4u : (unsigned)a.B * Ho * Wo * out_px, 0x00020000);
    unsigned slope2;
    {
        const half2_t s2 = {(half_t)a.slope, (half_t)a.slope};
        slope2 = __builtin_bit_cast(unsigned, s2);
    }
    // fused tail: A fragments of the 64 -> 1 conv for this wave's two 16-channel k-slices (rows = taps, rows 16..24 their rounding remainders)
    half8_t tailw[2];
#pragma unroll
    for (int gp = 0; gp < 2; ++gp) {
        tailw[gp] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        if (TAIL) tailw[gp] = *(const half8_t*)(a.tail_w + (2 * c + gp) * 512 + lane * 8);
    }
    const unsigned lrplane = (unsigned)a.B * a.H * a.W;
    const unsigned ph = (unsigned)(si * r + sj);
    const unsigned ttrash = 9u * (unsigned)(r * r) * lrplane + lane * 4;

    const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float16_t acc[5];                      // output rows 0..2 of the patch in slots 0..2, row 3 in slot 3 (even patches) / 4 (odd): it is read out during the next patch
    half8_t fr[2][12];
    float v2[16], v3[16];                  // output rows 2 and 3 of the previous patch, read out of their accumulators late in that patch
#pragma unroll
    for (int e = 0; e < 16; ++e) { v2[e] = 0.f; v3[e] = 0.f; }
    float tap8 = 0.f;

    // ===== epilogue pieces ==============================================================================================================
    // PReLU on packed halves (slope <= 1): values v[8 gp .. 8 gp + 7] -> four half2 registers
    auto act8 = [&](const float* v, unsigned (&hv)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const half2_t pr = {(half_t)v[2 * k], (half_t)v[2 * k + 1]};
            const half2_t t = pr * __builtin_bit_cast(half2_t, slope2);
            hv[k] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(pr, t));
        }
    };
    // EPI 4: this lane's sums of the 16 channels it stores (the fp16 values, as a second pass over the tensor would read them), for the plane
    // `pool_b`; when a row of another plane arrives (and at the end) the 32 pixel lanes are added up and lane j = 0 stores the slab
    float psum[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) psum[e] = 0.f;
    int pool_b = -1;
    auto pool_flush = [&]() {
        if (pool_b >= 0) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float t = psum[e];
                t += __shfl_xor(t, 16); t += __shfl_xor(t, 8); t += __shfl_xor(t, 4); t += __shfl_xor(t, 2); t += __shfl_xor(t, 1);
                psum[e] = t;
            }
            if (j == 0) {
                float* dst = a.pool + ((long long)pool_b * a.pool_slabs + 2 * g + h) * 64 + 32 * c + 8 * hh;
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    *(float4_t*)(dst + 16 * gp) = float4_t{psum[8 * gp], psum[8 * gp + 1], psum[8 * gp + 2], psum[8 * gp + 3]};
                    *(float4_t*)(dst + 16 * gp + 4) = float4_t{psum[8 * gp + 4], psum[8 * gp + 5], psum[8 * gp + 6], psum[8 * gp + 7]};
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) psum[e] = 0.f;
    };
    // EPI 1: the 16-byte stores of output row `orow` (0..7 in the patch) of tile `it`: channels cout0 + 32c + 16 gp + 8 hh .. +7 of pixel j
    auto store_row = [&](const Item& it, int orow, bool live, const unsigned (&h0)[4], const unsigned (&h1)[4]) {
        const int y = it.pyi * kTileH + orow, x = it.pxi * kTileW + j;
        if (POOL) {
            if (live && it.b != pool_b) { pool_flush(); pool_b = it.b; }       // (wave-uniform, once per plane)
            const float m = (live & (y < a.H) & (x < a.W)) ? 1.f : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                psum[2 * k] = mix_lo(h0[k], m, psum[2 * k]); psum[2 * k + 1] = mix_hi(h0[k], m, psum[2 * k + 1]);
                psum[8 + 2 * k] = mix_lo(h1[k], m, psum[8 + 2 * k]); psum[8 + 2 * k + 1] = mix_hi(h1[k], m, psum[8 + 2 * k + 1]);
            }
        }
        const unsigned vo = (x < a.W) ? (unsigned)(j * r) * out_px + (unsigned)hh * 16u : kOOR;
        const unsigned so = (live & (y < a.H)) ? ((unsigned)(it.b * (int)Ho + y * r + si) * Wo + (unsigned)(it.pxi * kTileW * r + sj)) * out_px + (unsigned)(cout0 + 32 * c) * 2u : kOOR;
        const u4_t d0 = {h0[0], h0[1], h0[2], h0[3]}, d1 = {h1[0], h1[1], h1[2], h1[3]};
        __builtin_amdgcn_raw_buffer_store_b128(d0, rout, vo, so, 0);
        __builtin_amdgcn_raw_buffer_store_b128(d1, rout, vo + 32u, so, 0);
    };
    // EPI 3: partial tail sums of one row (this wave's 32 channels) -> exchange record (buffer xb, half c, patch row orow)
    auto tail_partial = [&](const unsigned (&h0)[4], const unsigned (&h1)[4], int xb, int orow) {
        const half8_t b0 = __builtin_bit_cast(half8_t, u4_t{h0[0], h0[1], h0[2], h0[3]});
        const half8_t b1 = __builtin_bit_cast(half8_t, u4_t{h1[0], h1[1], h1[2], h1[3]});
        float16_t G = __builtin_amdgcn_mfma_f32_32x32x16_f16(tailw[0], b0, zero16, 0, 0, 0);
        G = __builtin_amdgcn_mfma_f32_32x32x16_f16(tailw[1], b1, G, 0, 0, 0);
        float t[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) t[k] = __builtin_fmaf(G[8 + k], 0.00048828125f, G[k]);     // low-order weight rows (units of 2^-11)
        const unsigned rec = xch0 + (unsigned)(xb * XCH_BYTES + (c * 8 + orow) * XROW);
        const u4_t d = {__builtin_bit_cast(unsigned, t[0]), __builtin_bit_cast(unsigned, t[1]), __builtin_bit_cast(unsigned, t[2]), __builtin_bit_cast(unsigned, t[3])};
        // (inline asm: a C++ store to LDS is ordered behind the LDS-DMA in flight with vmcnt(0), arsb_fused.hip cost 1)
        asm volatile("ds_write_b128 %0, %1" ::"v"(rec + (unsigned)lane * 16u), "v"(d) : "memory");
        asm volatile("ds_write_b32 %0, %1 offset:1024" ::"v"(rec + (unsigned)lane * 4u), "v"(t[4]) : "memory");
    };
    // EPI 3: rows 4h + 2c + i (i = 0, 1) of a patch whose partial sums are complete, in three pieces: (a) fetch the two channel halves from
    // the exchange area, (b) add them and transpose inside each lane quad so that four one-float tap registers become one 16-byte store
    // (conv3x3_sp.hip), (c) the store; tap 8 of both rows shares one store
    float4_t fA0, fA1;
    float fe0, fe1, ff[4], fg8;
    auto fin_fetch = [&](int xb, int i) {
        const int orow = 4 * h + 2 * c + i;
        const unsigned r0 = xch0 + (unsigned)(xb * XCH_BYTES + orow * XROW), r1 = r0 + 8u * XROW;
        fA0 = *(const __attribute__((address_space(3))) float4_t*)(r0 + (unsigned)lane * 16u);
        fA1 = *(const __attribute__((address_space(3))) float4_t*)(r1 + (unsigned)lane * 16u);
        fe0 = *(const __attribute__((address_space(3))) float*)(r0 + 1024u + (unsigned)lane * 4u);
        fe1 = *(const __attribute__((address_space(3))) float*)(r1 + 1024u + (unsigned)lane * 4u);
    };
    auto fin_mix = [&]() {
        float Gv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) Gv[k] = fA0[k] + fA1[k];
        fg8 = fe0 + fe1;
        auto xq = [](float v, bool far) {
            const int u = __builtin_bit_cast(int, v);
            return __builtin_bit_cast(float, far ? __builtin_amdgcn_mov_dpp(u, 0x4E, 0xF, 0xF, true) : __builtin_amdgcn_mov_dpp(u, 0xB1, 0xF, 0xF, true));
        };
        const bool b0 = j & 1, b1 = j & 2;
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; k += 2) {
            const float p0 = xq(Gv[k], false), p1 = xq(Gv[k + 1], false);
            t[k] = b0 ? p1 : Gv[k];
            t[k + 1] = b0 ? Gv[k + 1] : p0;
        }
#pragma unroll
        for (int cq = 0; cq < 2; ++cq) {
            const float q0 = xq(t[cq], true), q2 = xq(t[2 + cq], true);
            ff[cq] = b1 ? q2 : t[cq];
            ff[2 + cq] = b1 ? t[2 + cq] : q0;
        }
    };
    auto fin_store = [&](const Item& it, int i, bool live) {
        const int orow = 4 * h + 2 * c + i;
        const int y = it.pyi * kTileH + orow, x = it.pxi * kTileW + j;
        const unsigned xq0 = (unsigned)(it.pxi * kTileW + (j & ~3));
        const bool okq = (y < a.H) & (xq0 < (unsigned)a.W) & live;
        const unsigned off = ((unsigned)(4 * hh + (j & 3)) * (unsigned)(r * r) + ph) * lrplane + (unsigned)(it.b * a.H + y) * (unsigned)a.W + xq0;
        *(float4*)(a.tplanes + (okq ? off : ttrash)) = make_float4(ff[0], ff[1], ff[2], ff[3]);
        if (i == 0) tap8 = fg8;
        else {
            const float v8 = __builtin_bit_cast(float, __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, tap8), __builtin_bit_cast(unsigned, fg8), false, false)[0]);
            const int y8 = it.pyi * kTileH + 4 * h + 2 * c + hh;
            const bool ok8 = (y8 < a.H) & (x < a.W) & live;
            const unsigned off8 = (8u * (unsigned)(r * r) + ph) * lrplane + (unsigned)(it.b * a.H + y8) * (unsigned)a.W + (unsigned)x;
            a.tplanes[ok8 ? off8 : ttrash] = v8;
        }
    };
    auto finalize_row = [&](const Item& it, int xb, int i, bool live) { fin_fetch(xb, i); fin_mix(); fin_store(it, i, live); };

    // ===== prologue ====================================================================================================================
    Item it_cur{0, 0, 0};
    {
        it_cur.pxi = g % a.px;
        const int t = g / a.px;
        it_cur.pyi = t % a.py;
        it_cur.b = t / a.py;
    }
    Item it_prev = it_cur, it_pp = it_cur, it_next = advance(it_cur);
    prep_patch(it_cur, true);
    {
        const unsigned org = origin(it_cur);
#pragma unroll
        for (int i = 0; i < NDMA_W; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + (i * 4 + w4) * 1024), 16, vofs[i], org, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int f = 0; f < 12; ++f) fr[0][f] = *(lds_h8_t)(fa[0][f]);
    }

    // ===== one patch.  Row step R multiplies input row R (fragments fr[R & 1]) into output rows R, R-1, R-2 and reads the fragments of
    // row R+1; every B fragment f is a CHUNK: its (up to) three MFMAs, one fragment read, one piece of the other work. ===================
    auto iteration = [&](int p, auto BUF_) __attribute__((always_inline)) {
        constexpr int BUF = decltype(BUF_)::value;
        const bool fetch = p + 1 < K;                     // patch p+1 exists
        const bool ep3 = p >= 1, fin = p >= 2;            // rows 2, 3 of patch p-1 / the rows of patch p-2 are pending
        const Item it = it_cur, itp = it_prev, itpp = it_pp, itn = it_next;
        const unsigned orgn = origin(itn);
        prep_patch(itn, fetch);
        const int xb = p % 3, xbp = (p + 2) % 3, xbpp = (p + 1) % 3;     // exchange buffers of patch p, p-1, p-2
        char* const nbuf = smem + (BUF ^ 1) * PATCH_BYTES;
        unsigned hA[4], hB[4];                             // activated halves of the row being finished
        float v[16];

        auto step = [&](auto R_) __attribute__((always_inline)) {
            constexpr int R = decltype(R_)::value;
            constexpr int nm = (R <= 3 ? 1 : 0) + ((R >= 1 && R <= 4) ? 1 : 0) + (R >= 2 ? 1 : 0);     // MFMAs per chunk
            if (R == 5) {
                // Everybody's pieces of patch p+1 have landed, nobody reads patch p's buffer any more (row 5 is in registers), the exchange
                // records written so far are published.  (vmcnt(0) also waits for this patch's store acknowledgements: all issued rows ago.)
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            auto chunk_f = [&](auto F_) __attribute__((always_inline)) {
                constexpr int f = decltype(F_)::value;
                constexpr int dx = f >> 2, ks = f & 3;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int o = R - dy;
                    if (o >= 0 && o < 4) {
                        const int sl = o == 3 ? 3 + BUF : o;
                        acc[sl] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[(dy * 3 + dx) * 4 + ks], fr[R & 1][f], (dy == 0 && f == 0) ? biasv : acc[sl], 0, 0, 0);
                    }
                }
                if (R < 5) fr[(R + 1) & 1][f] = *(lds_h8_t)(fa[BUF][f] + (unsigned)((R + 1) * (PW * 128)));
                else fr[0][f] = *(lds_h8_t)(fa[BUF ^ 1][f]);            // row 0 of patch p+1 (behind the barrier)
                // ---- the piece of other work of this chunk, spread so that every row step carries about three instructions per MFMA.
                // Output row o of this patch is complete after row step o + 2:
                //   row 0   finished in step 3;   row 1   read out and activated in step 4, emitted in step 5 BEHIND the barrier (a global
                //   store issued shortly before the barrier's vmcnt(0) would make it wait for the acknowledgement);   row 2   read out in
                //   step 5, finished in step 1 of the next patch;   row 3   stays in its accumulators (slot 3 + BUF), read out in step 0 of
                //   the next patch, finished in its steps 1, 2;   DMA of patch p+1 in steps 0..2;   the pending rows of patch p-2 in step 2.
                if (R == 0) {
                    if (f == 1 || f == 3 || f == 5 || f == 7)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(nbuf + ((f >> 1) * 4 + w4) * 1024), 16, vofs[f >> 1], orgn, 0, 0);
                    if (f == 9) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v3[e] = acc[3 + (BUF ^ 1)][e];
                    }
                    if (f == 10) {
#pragma unroll
                        for (int e = 8; e < 16; ++e) v3[e] = acc[3 + (BUF ^ 1)][e];
                    }
                }
                if (R == 1) {
                    if (f <= 3)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(nbuf + ((f + 4) * 4 + w4) * 1024), 16, vofs[f + 4], orgn, 0, 0);
                    if (f == 4) act8(v2, hA);
                    if (f == 5) act8(v2 + 8, hB);
                    if (f == 6) { if (TAIL) tail_partial(hA, hB, xbp, 4 * h + 2); else store_row(itp, 4 * h + 2, ep3, hA, hB); }
                    if (f == 9) act8(v3, hA);
                    if (f == 10) act8(v3 + 8, hB);
                }
                if (R == 2) {
                    if (f <= 2)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(nbuf + ((f + 8) * 4 + w4) * 1024), 16, vofs[f + 8], orgn, 0, 0);
                    if (f == 3) { if (TAIL) tail_partial(hA, hB, xbp, 4 * h + 3); else store_row(itp, 4 * h + 3, ep3, hA, hB); }
                    if (TAIL) {
                        if (f == 4) fin_fetch(xbpp, 0);
                        if (f == 5) fin_mix();
                        if (f == 6) fin_store(itpp, 0, fin);
                        if (f == 7) fin_fetch(xbpp, 1);
                        if (f == 8) fin_mix();
                        if (f == 9) fin_store(itpp, 1, fin);
                    }
                }
                if (R == 3) {
                    if (f == 0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = acc[0][e];
                    }
                    if (f == 1) {
#pragma unroll
                        for (int e = 8; e < 16; ++e) v[e] = acc[0][e];
                    }
                    if (f == 2) act8(v, hA);
                    if (f == 3) act8(v + 8, hB);
                    if (f == 5) { if (TAIL) tail_partial(hA, hB, xb, 4 * h + 0); else store_row(it, 4 * h + 0, true, hA, hB); }
                }
                if (R == 4) {
                    if (f == 0) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = acc[1][e];
                    }
                    if (f == 2) {
#pragma unroll
                        for (int e = 8; e < 16; ++e) v[e] = acc[1][e];
                    }
                    if (f == 5) act8(v, hA);
                    if (f == 8) act8(v + 8, hB);
                }
                if (R == 5) {
                    if (f == 1) { if (TAIL) tail_partial(hA, hB, xb, 4 * h + 1); else store_row(it, 4 * h + 1, true, hA, hB); }
                    if (f == 5) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v2[e] = acc[2][e];
                    }
                    if (f == 8) {
#pragma unroll
                        for (int e = 8; e < 16; ++e) v2[e] = acc[2][e];
                    }
                }
#ifndef RW_NOPIN
#pragma unroll
                for (int i_ = 0; i_ < 3; ++i_) {
                    if (i_ < nm) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i_ == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (i_ < nm && i_ + 1 < nm) __builtin_amdgcn_sched_group_barrier(0x006, 5, 0);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            };
#define RW_CHUNK(F) chunk_f(std::integral_constant<int, F>{});
            RW_CHUNK(0) RW_CHUNK(1) RW_CHUNK(2) RW_CHUNK(3) RW_CHUNK(4) RW_CHUNK(5) RW_CHUNK(6) RW_CHUNK(7) RW_CHUNK(8) RW_CHUNK(9) RW_CHUNK(10) RW_CHUNK(11)
#undef RW_CHUNK
            // the fragments of the next row step have landed before it starts (one wait the compiler's count pass can see, see conv3x3_sp.hip)
            __builtin_amdgcn_s_waitcnt(0xC07F);
        };
#define RW_STEP(S) step(std::integral_constant<int, S>{});
        RW_STEP(0) RW_STEP(1) RW_STEP(2) RW_STEP(3) RW_STEP(4) RW_STEP(5)
#undef RW_STEP
        it_pp = it_prev; it_prev = it_cur; it_cur = it_next; it_next = advance(it_next);
    };

    int p = 0;
    for (; p + 1 < K; p += 2) {
        iteration(p, std::integral_constant<int, 0>{});
        iteration(p + 1, std::integral_constant<int, 1>{});
    }
    if (p < K) {
        iteration(p, std::integral_constant<int, 0>{});
        ++p;
    }
    // ===== drain: rows 2 and 3 of the last patch, then (fused tail) the rows of the last two patches =====================================
    {
        unsigned hA[4], hB[4];
        act8(v2, hA);
        act8(v2 + 8, hB);
        if (TAIL) tail_partial(hA, hB, (K + 2) % 3, 4 * h + 2);
        else store_row(it_prev, 4 * h + 2, true, hA, hB);
        if (K & 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) v3[e] = acc[3][e];
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) v3[e] = acc[4][e];
        }
        act8(v3, hA);
        act8(v3 + 8, hB);
        if (TAIL) tail_partial(hA, hB, (K + 2) % 3, 4 * h + 3);
        else store_row(it_prev, 4 * h + 3, true, hA, hB);
        if (POOL) pool_flush();
        if (TAIL) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (K >= 2) {
                finalize_row(it_pp, (K + 1) % 3, 0, true);
                finalize_row(it_pp, (K + 1) % 3, 1, true);
            }
            finalize_row(it_prev, (K + 2) % 3, 0, true);
            finalize_row(it_prev, (K + 2) % 3, 1, true);
        }
    }
#endif
}

template <int EPI>
hipError_t set_limit()
{
    return hipFuncSetAttribute((const void*)conv3x3_rw_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}

}  // namespace

hipError_t conv3x3_rw_init()
{
    hipError_t e;
    if ((e = set_limit<1>()) != hipSuccess) return e;
    if ((e = set_limit<4>()) != hipSuccess) return e;
    return set_limit<3>();
}

// false: not one of the two compiled epilogues / shapes (caller uses conv3x3_sp)
bool launch_conv3x3_rw(const ConvArgs& a, hipStream_t s)
{
    if (a.acc_mode != 0 || a.dbg || a.plane_w || a.res || a.out_lo || a.side16) return false;
    if (!(a.slope < 1.f) || a.scale != 1.f || !a.bias_img || a.in_cs != 64) return false;
    const bool tail = a.tplanes != nullptr;
    if (tail && a.tail_split) return false;
    if (!tail && !a.out) return false;
    if (2ll * a.B * a.H * a.W * a.in_cs + 2ll * (a.W + 1) * a.in_cs >= (1ll << 32) - 65536) return false;
    if (!tail && 2ll * a.B * a.H * a.r * a.W * a.r * a.out_cs >= (1ll << 32) - 65536) return false;
    if (tail && 36ll * a.B * a.H * a.r * a.W * a.r >= (1ll << 32) - 8192) return false;
    const int blocks = a.nchunks * ((a.G + 7) / 8) * 8;
    if (a.pool && (tail || a.nchunks != 1 || a.r != 1 || a.pool_slabs < 2 * a.G)) return false;
    if (tail) conv3x3_rw_kernel<3><<<dim3(blocks), dim3(256), LDS_BYTES, s>>>(a);
    else if (a.pool) conv3x3_rw_kernel<4><<<dim3(blocks), dim3(256), LDS_BYTES, s>>>(a);
    else conv3x3_rw_kernel<1><<<dim3(blocks), dim3(256), LDS_BYTES, s>>>(a);
    return true;
}
