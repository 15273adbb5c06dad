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
//     phase p :   group p&1      multiplies patch p              (LDS -> MFMA, 144 v_mfma_f32_32x32x16_f16 per wave)
//                 other group    stores patch p-1 (epilogue) and starts the LDS-DMA of patch p+1 into its own buffer
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

template <int MODE, bool RES>
__global__ __launch_bounds__(512) void conv3x3_pp_kernel(ConvArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wlds = smem;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..7
    const int grp = wave >> 2, w4 = wave & 3;
    const int wr = w4 >> 1, wn = w4 & 1;
    char* const mybuf = smem + WBYTES + grp * PATCH_BYTES;

    const int bid = blockIdx.x;
    const int chunk = (bid >> 3) % a.nchunks;
    const int g = (bid & 7) + 8 * (bid / (8 * a.nchunks));
    if (g >= a.G) return;
    const int nitems = a.B * a.py * a.px;
    const int K = (nitems - g + a.G - 1) / a.G;                  // this workgroup's items: g, g+G, ...
    if (K <= 0) return;

    const half_t* const zsrc = a.zero + (lane & 7) * 8;

    // LDS-DMA of the halo'd patch of item k into this group's buffer (issued by the 4 waves of the group)
    auto issue_patch = [&](int k) {
        const Item it = decode(g + k * a.G, a.px, a.py);
        const int y0 = it.pyi * kTileH - 1, x0 = it.pxi * kTileW - 1;
        const half_t* base = a.in + ((long long)(it.b * a.H + y0) * a.W + x0) * a.in_cs;
        const bool interior = (y0 >= 0) && (y0 + PH <= a.H) && (x0 >= 0) && (x0 + PW <= a.W);
#pragma unroll
        for (int i = 0; i < NDMA_W; ++i) {
            const int n = i * 4 + w4;
            if (n < NDMA) {                                       // wave-uniform
                const int q = n * 8 + (lane >> 3);
                const int r = (q * 241) >> 13;                    // q / 34 for q < 352
                const int c = q - r * PW;
                const int sl = (lane & 7) ^ ((c >> 1) & 7);       // logical 16-B slot behind this physical slot
                const half_t* src = base + (r * a.W + c) * a.in_cs + sl * 8;
                if (!interior || i == NDMA_W - 1) {
                    const int yy = y0 + r, xx = x0 + c;
                    const bool ok = (q < NPIX) && (yy >= 0) && (yy < a.H) && (xx >= 0) && (xx < a.W);
                    src = ok ? src : zsrc;
                }
                dma16(src, mybuf + n * 1024);
            }
        }
    };

    {   // prologue: weights (all 8 waves) + the first patch of each group
        const half_t* wsrc = a.wpk + (long long)chunk * (WBYTES / 2);
        for (int f = wave; f < NFRAG; f += 8) dma16(wsrc + f * 512 + lane * 8, wlds + f * 1024);
        if (grp < K) issue_patch(grp);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const int j = lane & 31, hh = lane >> 5;
    float16_t acc[4];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[o][e] = 0.f;

    for (int p = 0; p <= K; ++p) {
        if ((p & 1) == grp) {
            // ======================= multiply patch p ============================================
            if (p < K) {
                const char* abuf = mybuf;
                const char* wl = wlds + (wn << 10) + lane * 16;
                // LDS image: pixel (row, col) at (row*34 + col)*128, 16-B slot s stored at slot s ^ ((col>>1)&7): the XOR
                // depends on the column only, so the six input rows of a step differ by an immediate offset.
                int Ad[3], Zd[3];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int c = j + dx;
                    const int z = (c >> 1) & 7;
                    Ad[dx] = (wr * 4 * PW + c) * 128 + ((hh ^ (z & 1)) << 4);
                    Zd[dx] = (z >> 1) << 5;
                }
                half8_t wf[2][3], af[2][6];
#define MOE_LOAD_STEP(S, BUF)                                                                              \
    {                                                                                                      \
        constexpr int dx_ = (S) / 4, ks_ = (S) % 4;                                                        \
        _Pragma("unroll") for (int dy = 0; dy < 3; ++dy)                                                   \
            wf[BUF][dy] = *(const half8_t*)(wl + ((((dy * 3 + dx_) * 4 + ks_) * 2) << 10));                \
        const char* ap_ = abuf + Ad[dx_] + ((ks_ << 5) ^ Zd[dx_]);                                         \
        _Pragma("unroll") for (int pr = 0; pr < 6; ++pr)                                                   \
            af[BUF][pr] = *(const half8_t*)(ap_ + pr * (PW * 128));                                        \
    }
                MOE_LOAD_STEP(0, 0)
                __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);
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
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int o = pr - dy;
                            if (o >= 0 && o < 4)
                                acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cb][dy], af[cb][pr], acc[o], 0, 0, 0);
                        }
                    if (s + 1 < 12) {
#pragma unroll
                        for (int i = 0; i < 9; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
                    }
                }
#undef MOE_LOAD_STEP
            }
        } else {
            // ======================= store patch p-1, fetch patch p+1 =============================
            const bool have_prev = (p >= 1);
            Item it{0, 0, 0};
            if (have_prev) it = decode(g + (p - 1) * a.G, a.px, a.py);
            const int x = it.pxi * kTileW + j;
            const int r = a.r;
            const int si = (r > 1) ? chunk / r : 0, sj = (r > 1) ? chunk % r : 0;
            const int cbase = ((r > 1) ? 0 : chunk * kCB) + wn * 32 + hh * 4;
            const int pcb = chunk * kCB + wn * 32 + hh * 4;
            const int Wo = a.W * r, Ho = a.H * r;
            // start the DMA of this group's next patch first: it flies while the epilogue math and stores run
            if (p >= 1 && p + 1 < K) issue_patch(p + 1);
            if (have_prev) {
                float4_t bias4[4];
#pragma unroll
                for (int grp4 = 0; grp4 < 4; ++grp4) {
                    bias4[grp4] = float4_t{0.f, 0.f, 0.f, 0.f};
                    if ((MODE == 0 || MODE == 3) && a.bias) bias4[grp4] = *(const float4_t*)(a.bias + pcb + grp4 * 8);
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const int y = it.pyi * kTileH + wr * 4 + o;
                    if (y < a.H && x < a.W) {
                        const long long orow = ((long long)(it.b * Ho + y * r + si) * Wo + (x * r + sj)) * a.out_cs + cbase;
                        const long long apix = ((long long)(it.b * a.H + y) * a.W + x) * (a.nchunks * kCB) + pcb;
#pragma unroll
                        for (int grp4 = 0; grp4 < 4; ++grp4) {
                            float4_t v = {acc[o][grp4 * 4 + 0], acc[o][grp4 * 4 + 1], acc[o][grp4 * 4 + 2], acc[o][grp4 * 4 + 3]};
                            if (MODE == 1) { *(float4_t*)(a.acc32 + apix + grp4 * 8) = v; continue; }
                            if (MODE == 2) {
                                float4_t* q = (float4_t*)(a.acc32 + apix + grp4 * 8);
                                *q = *q + v;
                                continue;
                            }
                            if (MODE == 3) v = v + *(const float4_t*)(a.acc32 + apix + grp4 * 8) * 0.00048828125f;
                            v = (v + bias4[grp4]) * a.scale;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * a.slope;
                            if (RES) {
                                const half4_t rv = *(const half4_t*)(a.res + orow + grp4 * 8);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += (float)rv[e];
                                if (MODE == 3 && a.res_lo) {
                                    const half4_t rl = *(const half4_t*)(a.res_lo + orow + grp4 * 8);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) v[e] += (float)rl[e] * 0.00048828125f;
                                }
                            }
                            const half4_t hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                            *(half4_t*)(a.out + orow + grp4 * 8) = hv;
                            if (MODE == 3 && a.out_lo) {
                                half4_t lv;
#pragma unroll
                                for (int e = 0; e < 4; ++e) lv[e] = (half_t)((v[e] - (float)hv[e]) * 2048.f);
                                *(half4_t*)(a.out_lo + orow + grp4 * 8) = lv;
                            }
                        }
                    }
                }
#pragma unroll
                for (int o = 0; o < 4; ++o)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[o][e] = 0.f;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
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
