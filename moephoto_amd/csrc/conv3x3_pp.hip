// conv3x3_pp.hip -- the hot kernel: 3x3, 64-in-channel implicit-GEMM convolution with a two-group "ping-pong"
// schedule, so that the matrix pipe of every SIMD always has a wave to run.
//
// Why: rocprofv3 on the single-group kernel (conv_mfma.hip; profiles/r01/v1_summary.txt) showed MFMA busy 23 % of
// wave time with 5.6 VALU + 3.1 SALU instructions per MFMA: the epilogue (accumulator -> bias/PReLU/residual -> fp16
// -> NHWC stores), the DMA address arithmetic and the end-of-patch vmcnt/barrier wait were all SERIAL with the
// 144-MFMA loop, on a CU that holds one workgroup (weights + patches fill the LDS) and therefore one wave per SIMD.
//
// Now: 512 threads = two groups of 4 waves (waves w and w+4 share a SIMD).  The groups work on alternate patches,
// half a period apart, one workgroup-wide barrier per phase:
//
//     phase p :   group p&1      multiplies patch p (LDS -> MFMA, 144 v_mfma_f32_32x32x16_f16 per wave) and, one piece per
//                                k-step between its MFMAs, issues the LDS-DMA of patch p+1 into the OTHER group's buffer
//                 other group    runs the epilogue of patch p-1: bias/scale/PReLU/residual in fp32, fp16, one
//                                v_permlane32_swap per register pair so that every lane owns 16 contiguous bytes of a pixel,
//                                eight 16-byte stores per wave (s_memtime traces: an LDS-staged epilogue cost the same stores
//                                but pushed the 11 DMA issues per wave, ~250 cycles each, onto the critical path)
//
// so the VALU/VMEM work of one wave sits next to the MFMA work of its SIMD partner.  LDS is unchanged:
// 73,728 B weights (shared by both groups) + one 44,032-B patch buffer per group.  Everything else (GEMM view, LDS
// fragment order, fused epilogue, pixel-shuffle store) is as documented in conv_mfma.hip; the LDS swizzle here is keyed
// on the patch COLUMN ((col>>1)&7) instead of the linear pixel index, which is equally conflict-free for the 32-pixel
// row reads and turns the per-row addresses into immediate offsets.
#include "common.h"

namespace {

constexpr int PW = kTileW + 2, PH = kTileH + 2, NPIX = PW * PH;   // 34 x 10 halo'd patch
constexpr int NDMA = (NPIX + 7) / 8;                               // 43 one-KiB pieces
constexpr int NDMA_W = (NDMA + 3) / 4;                             // 11 per wave of the loading group
constexpr int PATCH_BYTES = NDMA * 1024;                           // 44,032
constexpr int NFRAG = 72;                                          // 9 taps x 4 k-slices x 2 n-blocks
constexpr int WBYTES = NFRAG * kFragBytes;                         // 73,728
constexpr int LDS_BYTES = WBYTES + 2 * PATCH_BYTES;                // 161,792

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

// timing trace (MOE_DBG & 64): acc32 doubles as a [wg<8][phase<32][grp<2][slot<8] table of s_memtime stamps
#define MOE_STAMP(SLOT)                                                                                     \
    if (MODE == 0 && (a.dbg & 64) && a.acc32 && bid < 8 && p < 32 && w4 == 0 && lane == 0)                              \
        ((unsigned long long*)a.acc32)[((bid * 32 + p) * 2 + grp) * 8 + (SLOT)] = __builtin_amdgcn_s_memtime();

template <int MODE, bool RES>
__global__ __launch_bounds__(512) void conv3x3_pp_kernel(ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wlds = smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int grp = wave >> 2, w4 = wave & 3;
        char* const mybuf = smem + WBYTES + grp * PATCH_BYTES;

    const int bid = blockIdx.x;
    const int chunk = (bid >> 3) % a.nchunks;
    const int g = (bid & 7) + 8 * (bid / (8 * a.nchunks));
    if (g >= a.G) return;
    const int nitems = a.B * a.py * a.px;
    const int K = (nitems - g + a.G - 1) / a.G;                  // this workgroup's items: g, g+G, ...
    if (K <= 0) return;

    const half_t* const zsrc = a.zero + (lane & 7) * 8;

    // LDS-DMA of the halo'd patch of item k into patch buffer `dstbuf`: piece i (0..10) of this wave
    struct PatchSrc { const half_t* base; int y0, x0; };
    auto patch_src = [&](int k) {
        const Item it = decode(g + k * a.G, a.px, a.py);
        PatchSrc ps;
        ps.y0 = it.pyi * kTileH - 1; ps.x0 = it.pxi * kTileW - 1;
        ps.base = a.in + ((long long)(it.b * a.H + ps.y0) * a.W + ps.x0) * a.in_cs;
        return ps;
    };
    // launch-invariant per-lane source offsets of this wave's 11 pieces (elements, relative to the patch origin)
    int poff[NDMA_W];
#pragma unroll
    for (int i = 0; i < NDMA_W; ++i) {
        const int q = (i * 4 + w4) * 8 + (lane >> 3);
        const int r = (q * 241) >> 13;                            // q / 34 for q < 352
        const int c = q - r * PW;
        const int sl = (lane & 7) ^ ((c >> 1) & 7);               // logical 16-B slot behind this physical slot
        poff[i] = (r * a.W + c) * a.in_cs + sl * 8;
    }
    auto issue_piece = [&](const PatchSrc& ps, int i, char* dstbuf) {
        const int n = i * 4 + w4;
        if (n < NDMA) {                                           // wave-uniform
            // branch-free bounds test (keeps the MFMA stream in one basic block): unsigned compares fold the < 0 cases
            const int q = n * 8 + (lane >> 3);
            const int r = (q * 241) >> 13;
            const int c = q - r * PW;
            const bool ok = ((unsigned)(ps.y0 + r) < (unsigned)a.H) & ((unsigned)(ps.x0 + c) < (unsigned)a.W) & (q < NPIX);
            const half_t* src = ps.base + poff[i];
            src = ok ? src : zsrc;
            dma16(src, dstbuf + n * 1024);
        }
    };

    {   // prologue: weights (all 8 waves) + the first patch of each group
        const half_t* wsrc = a.wpk + (long long)chunk * (WBYTES / 2);
        for (int f = wave; f < NFRAG; f += 8) dma16(wsrc + f * 512 + lane * 8, wlds + f * 1024);
        if (grp == 0) {
            const PatchSrc ps = patch_src(0);
#pragma unroll
            for (int i = 0; i < NDMA_W; ++i) issue_piece(ps, i, mybuf);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const int j = lane & 31, hh = lane >> 5;
    char* const otherbuf = smem + WBYTES + (grp ^ 1) * PATCH_BYTES;
    // wave tile: output rows 2*w4, 2*w4+1 of the patch x 32 pixels x all 64 output channels of the chunk
    float16_t acc[2][2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[o][nb][e] = 0.f;

    for (int p = 0; p <= K; ++p) {
        MOE_STAMP(0)
        if ((p & 1) == grp) {
            // ======================= multiply patch p, fetch patch p+1 for the other group =================
            if (p < K && !(a.dbg & 2)) {
                const bool fetch = (p + 1 < K) && !(a.dbg & 1);
                PatchSrc ps{nullptr, 0, 0};
                if (fetch) ps = patch_src(p + 1);
                const char* abuf = mybuf;
                const char* wl = wlds + lane * 16;
                // LDS image: pixel (row, col) at (row*34 + col)*128, 16-B slot s stored at slot s ^ ((col>>1)&7): the XOR
                // depends on the column only, so the input rows of a step differ by an immediate offset.
                int Ad[3], Zd[3];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int c = j + dx;
                    const int z = (c >> 1) & 7;
                    Ad[dx] = (w4 * 2 * PW + c) * 128 + ((hh ^ (z & 1)) << 4);
                    Zd[dx] = (z >> 1) << 5;
                }
                half8_t wf[2][3][2], af[2][4];
#define MOE_LOAD_STEP(S, BUF)                                                                              \
    {                                                                                                      \
        constexpr int dx_ = (S) / 4, ks_ = (S) % 4;                                                        \
        _Pragma("unroll") for (int dy = 0; dy < 3; ++dy)                                                   \
            _Pragma("unroll") for (int nb = 0; nb < 2; ++nb)                                               \
                wf[BUF][dy][nb] = *(const half8_t*)(wl + ((((dy * 3 + dx_) * 4 + ks_) * 2 + nb) << 10));   \
        const char* ap_ = abuf + Ad[dx_] + ((ks_ << 5) ^ Zd[dx_]);                                         \
        _Pragma("unroll") for (int pr = 0; pr < 4; ++pr)                                                   \
            af[BUF][pr] = *(const half8_t*)(ap_ + pr * (PW * 128));                                        \
    }
                MOE_LOAD_STEP(0, 0)
                __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
#pragma unroll
                for (int s = 0; s < 12; ++s) {
                    const int cb = s & 1;
                    switch (s + 1) {   // constant after unrolling
#define MOE_CASE(N) case N: MOE_LOAD_STEP(N, ((N) & 1)) break;
                        MOE_CASE(1) MOE_CASE(2) MOE_CASE(3) MOE_CASE(4) MOE_CASE(5) MOE_CASE(6)
                        MOE_CASE(7) MOE_CASE(8) MOE_CASE(9) MOE_CASE(10) MOE_CASE(11)
#undef MOE_CASE
                        default: break;
                    }
                    // one DMA piece of the next patch per k-step (11 pieces per wave), issued between the MFMAs
                    if (s < NDMA_W && fetch) issue_piece(ps, s, otherbuf);
#pragma unroll
                    for (int pr = 0; pr < 4; ++pr)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int o = pr - dy;
                            if (o >= 0 && o < 2) {
#pragma unroll
                                for (int nb = 0; nb < 2; ++nb)
                                    acc[o][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cb][dy][nb], af[cb][pr], acc[o][nb], 0, 0, 0);
                            }
                        }
                    if (s + 1 < 12) {
#pragma unroll
                        for (int i = 0; i < 10; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
                    }
                }
#undef MOE_LOAD_STEP
            }
            MOE_STAMP(1)
        } else if (p >= 1 && !(a.dbg & 8)) {
            // ======================= epilogue of patch p-1 ===========================================
            const Item it = decode(g + (p - 1) * a.G, a.px, a.py);
            const int r = a.r;
            const int si = (r > 1) ? chunk / r : 0, sj = (r > 1) ? chunk % r : 0;
            const int cout0 = (r > 1) ? 0 : chunk * kCB;                   // first output channel of this chunk in `out`
            const int Wo = a.W * r, Ho = a.H * r;
            const int yb = it.pyi * kTileH + w4 * 2, x = it.pxi * kTileW + j;
            if (MODE == 3 && a.side16) {
                // the low-order products were formed by two fp16-output launches of conv3x3_sp (sum in `side16`, output layout,
                // scaled by 2^11): fp16 is plenty for a term that is 2^-11 of the result
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const int y = yb + o;
                    if (y < a.H && x < a.W) {
                        const long long spix = ((long long)(it.b * Ho + y * r + si) * Wo + (x * r + sj)) * a.out_cs + cout0 + hh * 4;
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                const half4_t sv = *(const half4_t*)(a.side16 + spix + nb * 32 + g4 * 8);
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[o][nb][g4 * 4 + e] += (float)sv[e] * 0.00048828125f;
                            }
                    }
                }
            } else if (MODE == 1 || MODE == 2 || MODE == 3) {
                // split-precision passes: fp32 partial sums through acc32 (accumulator layout, 16-B accesses)
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const int y = yb + o;
                    if (y < a.H && x < a.W) {
                        const long long apix = ((long long)(it.b * a.H + y) * a.W + x) * (a.nchunks * kCB) + chunk * kCB + hh * 4;
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                float4_t* q = (float4_t*)(a.acc32 + apix + nb * 32 + g4 * 8);
                                float4_t v = {acc[o][nb][g4 * 4 + 0], acc[o][nb][g4 * 4 + 1], acc[o][nb][g4 * 4 + 2], acc[o][nb][g4 * 4 + 3]};
                                if (MODE == 1) *q = v;
                                else if (MODE == 2) *q = *q + v;
                                else {
                                    v = v + *q * 0.00048828125f;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) acc[o][nb][g4 * 4 + e] = v[e];
                                }
                            }
                    }
                }
            }
            if (MODE == 0 || MODE == 3) {
                // ---- fp32 math in the accumulator layout: lane (j, hh) holds channels nb*32 + 8*g4 + 4*hh + e of pixel j
                if (a.bias || a.scale != 1.f) {
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            float4_t b4 = {0.f, 0.f, 0.f, 0.f};
                            if (a.bias) b4 = *(const float4_t*)(a.bias + chunk * kCB + nb * 32 + g4 * 8 + hh * 4);
#pragma unroll
                            for (int o = 0; o < 2; ++o)
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[o][nb][g4 * 4 + e] = (acc[o][nb][g4 * 4 + e] + b4[e]) * a.scale;
                        }
                }
                if (a.slope != 1.f) {
                    if (a.slope <= 1.f) {       // PReLU/LeakyReLU with slope <= 1 (negative slopes included): max(x, slope*x)
#pragma unroll
                        for (int o = 0; o < 2; ++o)
#pragma unroll
                            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                                for (int e = 0; e < 16; ++e) acc[o][nb][e] = fmaxf(acc[o][nb][e], acc[o][nb][e] * a.slope);
                    } else {
#pragma unroll
                        for (int o = 0; o < 2; ++o)
#pragma unroll
                            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                                for (int e = 0; e < 16; ++e) acc[o][nb][e] = acc[o][nb][e] >= 0.f ? acc[o][nb][e] : acc[o][nb][e] * a.slope;
                    }
                }
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const int y = yb + o;
                    const bool ok = (y < a.H) && (x < a.W);
                    const long long opix = ((long long)(it.b * Ho + y * r + si) * Wo + (x * r + sj)) * a.out_cs + cout0;
                    if (RES && ok) {
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                const long long ro = opix + nb * 32 + g4 * 8 + hh * 4;
                                const half4_t rv = *(const half4_t*)(a.res + ro);
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[o][nb][g4 * 4 + e] += (float)rv[e];
                                if (MODE == 3 && a.res_lo) {
                                    const half4_t rl = *(const half4_t*)(a.res_lo + ro);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) acc[o][nb][g4 * 4 + e] += (float)rl[e] * 0.00048828125f;
                                }
                            }
                    }
                    // ---- fp16, then v_permlane32_swap pairs: lane (j, 0) ends up with channels 16*gp .. +7, lane (j, 1)
                    //      with 16*gp + 8 .. +15 of pixel j -> one 16-byte store per lane, 32 bytes per pixel per instruction
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int gp = 0; gp < 2; ++gp) {
                            half4_t h0, h1, l0, l1;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                h0[e] = (half_t)acc[o][nb][(2 * gp) * 4 + e];
                                h1[e] = (half_t)acc[o][nb][(2 * gp + 1) * 4 + e];
                                if (MODE == 3) {
                                    l0[e] = (half_t)((acc[o][nb][(2 * gp) * 4 + e] - (float)h0[e]) * 2048.f);
                                    l1[e] = (half_t)((acc[o][nb][(2 * gp + 1) * 4 + e] - (float)h1[e]) * 2048.f);
                                }
                            }
                            const long long oo = opix + nb * 32 + gp * 16 + hh * 8;
                            {
                                const uint2 u0 = __builtin_bit_cast(uint2, h0), u1 = __builtin_bit_cast(uint2, h1);
                                const auto sx = __builtin_amdgcn_permlane32_swap(u0.x, u1.x, false, false);
                                const auto sy = __builtin_amdgcn_permlane32_swap(u0.y, u1.y, false, false);
                                const uint4 w = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                                if (a.dbg & 4) asm volatile("" ::"v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w));
                                else if (ok) *(uint4*)(a.out + oo) = w;
                            }
                            if (MODE == 3 && a.out_lo) {
                                const uint2 u0 = __builtin_bit_cast(uint2, l0), u1 = __builtin_bit_cast(uint2, l1);
                                const auto sx = __builtin_amdgcn_permlane32_swap(u0.x, u1.x, false, false);
                                const auto sy = __builtin_amdgcn_permlane32_swap(u0.y, u1.y, false, false);
                                if (ok) *(uint4*)(a.out_lo + oo) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                            }
                        }
                }
            }
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[o][nb][e] = 0.f;
            MOE_STAMP(5)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MOE_STAMP(6)
        __syncthreads();
        MOE_STAMP(7)
    }
}

template <int MODE, bool RES>
hipError_t set_limit()
{
    return hipFuncSetAttribute((const void*)conv3x3_pp_kernel<MODE, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}

}  // namespace

hipError_t conv3x3_pp_init()
{
    hipError_t e;
    if ((e = set_limit<0, false>()) != hipSuccess) return e;
    if ((e = set_limit<0, true>()) != hipSuccess) return e;
    if ((e = set_limit<1, false>()) != hipSuccess) return e;
    if ((e = set_limit<2, false>()) != hipSuccess) return e;
    if ((e = set_limit<3, false>()) != hipSuccess) return e;
    if ((e = set_limit<3, true>()) != hipSuccess) return e;
    return hipSuccess;
}

void launch_conv3x3_pp(const ConvArgs& a, hipStream_t s)
{
    const int blocks = a.nchunks * ((a.G + 7) / 8) * 8;
    const dim3 grid(blocks), blk(512);
    const bool res = a.res != nullptr;
    switch (a.acc_mode) {
        case 0:
            if (res) conv3x3_pp_kernel<0, true><<<grid, blk, LDS_BYTES, s>>>(a);
            else conv3x3_pp_kernel<0, false><<<grid, blk, LDS_BYTES, s>>>(a);
            break;
        case 1: conv3x3_pp_kernel<1, false><<<grid, blk, LDS_BYTES, s>>>(a); break;
        case 2: conv3x3_pp_kernel<2, false><<<grid, blk, LDS_BYTES, s>>>(a); break;
        default:
            if (res) conv3x3_pp_kernel<3, true><<<grid, blk, LDS_BYTES, s>>>(a);
            else conv3x3_pp_kernel<3, false><<<grid, blk, LDS_BYTES, s>>>(a);
            break;
    }
}
