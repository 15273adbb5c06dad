// conv_mfma.hip -- implicit-GEMM 3x3 / 1x1 convolution on the gfx950 matrix cores.
//
// Replaces the nn.Conv2d (+bias) -> [PixelShuffle] -> PReLU/LeakyReLU -> [*scale] -> [+residual] chains of
// the reference nets (python/models.py:29-39 Conv3x3/upsample_block, :76-80 ARSB, :176-196 _Conv_Block;
// python/MoeNet_lite2.py:5-6,8-20).  One kernel, persistent workgroups:
//
//   GEMM view   D[cout][pixel] += W[cout][k] * X[k][pixel],  k = (tap, cin)      (M = cout so that a lane
//               ends up holding 4 consecutive output channels of ONE pixel -> 8-byte NHWC stores)
//   workgroup   256 threads = 4 waves, owns ONE 64-output-channel chunk for the whole launch: its weights
//               (<= 72 KiB as ready-made MFMA A fragments) are loaded into LDS once; then it walks
//               8x32-pixel patches (b, py, px) with stride G.
//   input       the (8+2)x(32+2) halo'd NHWC patch (one 128-B line per pixel) is brought in by LDS-DMA
//               (global_load_lds_dwordx4), double buffered: patch i+1 streams in while patch i is multiplied.
//               Out-of-image taps (the conv's zero padding, ragged edges) are redirected to a zero page.
//   LDS image   pixel-major, 128 B per pixel, 16-B slot index XOR-swizzled with (pixel>>1)&7: the B-fragment
//               reads (32 consecutive pixels, same slot) are then conflict-free for ds_read_b128 while every
//               8-lane DMA group still fetches one whole 128-B line.  The DMA destination is lane-linear, so
//               the swizzle is applied to the per-lane *source* address (and again on the read).
//   wave tile   wave (wr, wn): output rows 4*wr..4*wr+3 of the patch x 32 pixels x 32 output channels.
//               For each (dx, 16-channel k-slice) it loads 3 weight fragments (dy = 0..2) and 6 input-row
//               fragments; each input row feeds the up-to-3 output rows it touches: 9 ds_read_b128 per 12
//               v_mfma_f32_32x32x16_f16.
//   epilogue    straight from the accumulators: +bias, *scale, PReLU, +residual, fp16, 8-byte stores; the
//               pixel shuffle is folded into the store address (chunk -> sub-pixel (i,j), weights are packed
//               in (i, j, c) order so a chunk is exactly the 64 channels of one sub-pixel position).
//
// LDS: 73,728 (weights) + 2 x 44,032 (patches) = 161,792 B of the CU's 163,840.
#include "common.h"

namespace {

template <int TAPS>
struct Geo {
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int KD = (TAPS == 9) ? 3 : 1;
    static constexpr int PW = kTileW + 2 * HALO;
    static constexpr int PH = kTileH + 2 * HALO;
    static constexpr int NPIX = PW * PH;
    static constexpr int NDMA = (NPIX + 7) / 8;          // 1-KiB DMA pieces (8 pixels each)
    static constexpr int NDMA_W = (NDMA + 3) / 4;        // per wave
    static constexpr int PATCH_BYTES = NDMA * 1024;
};

__device__ __forceinline__ void dma16(const half_t* src, char* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int TAPS, int NSEG>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a)
{
    using G = Geo<TAPS>;
    constexpr int NFRAG = NSEG * TAPS * 4 * 2;           // weight fragments per chunk
    constexpr int WBYTES = NFRAG * kFragBytes;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wlds = smem;
    char* const pbuf = smem + WBYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wn = wave & 1;

    // blockIdx -> (group g, chunk): the nchunks workgroups that share an input patch sit on one XCD
    // (the dispatcher places block b on XCD b % 8; a speed hint only, nothing depends on it).
    const int bid = blockIdx.x;
    const int chunk = (bid >> 3) % a.nchunks;
    const int g = (bid & 7) + 8 * (bid / (8 * a.nchunks));
    if (g >= a.G) return;

    const int nitems = a.B * a.py * a.px;
    const int nunits = nitems * NSEG;                    // DMA units: (item, segment)

    // ---- per-thread DMA descriptors (constant for the launch) ------------------------------------
    int d_rc[G::NDMA_W];   // (row << 8) | col of the pixel this lane moves in piece i, or -1
    int d_off[G::NDMA_W];  // element offset of its 16 B inside the (y0-HALO, x0-HALO)-anchored window
#pragma unroll
    for (int i = 0; i < G::NDMA_W; ++i) {
        const int n = i * 4 + wave;
        const int q = n * 8 + (lane >> 3);
        const int sp = lane & 7;                          // physical 16-B slot
        const int sl = sp ^ ((q >> 1) & 7);               // logical slot (8 channels)
        const int r = q / G::PW, c = q - r * G::PW;
        const bool ok = (n < G::NDMA) && (q < G::NPIX);
        d_rc[i] = ok ? ((r << 8) | c) : -1;
        d_off[i] = (r * a.W + c) * a.in_cs + sl * 8;
    }
    const half_t* const zsrc = a.zero + (lane & 7) * 8;

    auto issue_unit = [&](int unit, int buf) {
        const int item = g + (unit / NSEG) * a.G;
        const int seg = unit % NSEG;
        const int pxi = item % a.px;
        const int t = item / a.px;
        const int pyi = t % a.py;
        const int b = t / a.py;
        const int y0 = pyi * kTileH - G::HALO, x0 = pxi * kTileW - G::HALO;
        // acc_mode 4: the three K segments are (w_lo, in), (w_hi, in_lo), (w_hi, in) over the SAME 64 channels
        const half_t* tensor = (a.acc_mode == 4) ? (seg == 1 ? a.in_lo : a.in) : a.in + seg * kCB;
        const half_t* base = tensor + ((long long)(b * a.H + y0) * a.W + x0) * a.in_cs;
        char* dst = pbuf + buf * G::PATCH_BYTES;
#pragma unroll
        for (int i = 0; i < G::NDMA_W; ++i) {
            const int n = i * 4 + wave;
            if (n < G::NDMA) {                             // wave-uniform
                const int rc = d_rc[i];
                const int yy = y0 + (rc >> 8), xx = x0 + (rc & 255);
                const bool ok = (rc >= 0) && (yy >= 0) && (yy < a.H) && (xx >= 0) && (xx < a.W);
                const half_t* src = ok ? (base + d_off[i]) : zsrc;
                dma16(src, dst + n * 1024);
            }
        }
    };

    int wb_loaded = -1;
    auto load_weights = [&](int b) {
        const half_t* wsrc = a.wpk + (long long)b * a.w_batch_stride + (long long)chunk * (WBYTES / 2);
        for (int f = wave; f < NFRAG; f += 4) dma16(wsrc + f * 512 + lane * 8, wlds + f * 1024);
    };

    const int my_items = (nitems - g + a.G - 1) / a.G;   // items g, g+G, ...
    const int my_units = my_items * NSEG;
    if (my_items <= 0) return;

    {   // prologue
        const int b0 = (g / (a.px * a.py));
        load_weights(a.w_batch_stride ? b0 : 0);
        wb_loaded = a.w_batch_stride ? b0 : 0;
        issue_unit(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const int j = lane & 31, hh = lane >> 5;
    float16_t acc[4];
    // the chunk's bias for this lane's 16 channels, once per kernel (loaded inside the epilogue, every tile waited four times for L2)
    float4_t bias4[4];
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
        bias4[grp] = float4_t{0.f, 0.f, 0.f, 0.f};
        if (a.bias) bias4[grp] = *(const float4_t*)(a.bias + chunk * kCB + wn * 32 + hh * 4 + grp * 8);
    }
    float tw[4][4];                                      // fused 1x1 tail: this lane's 16 tail weights (channel wn*32 + hh*4 + 8*grp + e)
#pragma unroll
    for (int grp = 0; grp < 4; ++grp)
#pragma unroll
        for (int e = 0; e < 4; ++e) tw[grp][e] = a.tail1_w ? a.tail1_w[wn * 32 + hh * 4 + 8 * grp + e] : 0.f;

    for (int u = 0; u < my_units; ++u) {
        const int seg = u % NSEG;
        const int cur = u & 1;
        const int item = g + (u / NSEG) * a.G;
        const int pxi = item % a.px;
        const int t = item / a.px;
        const int pyi = t % a.py;
        const int b = t / a.py;

        bool reload_w = false;
        if (u + 1 < my_units) {
            // per-plane weights (SEDN trans): if the next item changes plane, its weights are swapped in
            // after this unit's compute (rare: planes are the slowest-varying index of the item order)
            if (a.w_batch_stride && seg == NSEG - 1) {
                const int nb = (g + (u / NSEG + 1) * a.G) / (a.px * a.py);
                reload_w = (nb != wb_loaded);
            }
            issue_unit(u + 1, cur ^ 1);
        }

        if (seg == 0) {
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[o][e] = 0.f;
        }

        // ---- multiply: 4 output rows x 32 px x 32 couts per wave -----------------------------------
        // Software pipelined over the KD*4 (dx, k-slice) steps: the 3+6 fragments of step s+1 are read from
        // LDS while the 12 MFMAs of step s run (fragment registers double buffered, order pinned below).
        const char* abuf = pbuf + cur * G::PATCH_BYTES;
        const char* wl = wlds + ((seg * TAPS * 4 * 2 + wn) << 10) + lane * 16;
        constexpr int NR = 4 + G::KD - 1;
        constexpr int NSTEP = G::KD * 4;
        int A[G::KD][NR], Z[G::KD][NR];
#pragma unroll
        for (int dx = 0; dx < G::KD; ++dx)
#pragma unroll
            for (int pr = 0; pr < NR; ++pr) {
                const int q = (wr * 4 + pr) * G::PW + j + dx;
                const int z = (q >> 1) & 7;
                A[dx][pr] = q * 128 + ((hh ^ (z & 1)) << 4);
                Z[dx][pr] = (z >> 1) << 5;
            }
        half8_t wf[2][G::KD], af[2][NR];
#define MOE_LOAD_STEP(S, BUF)                                                                              \
    {                                                                                                      \
        constexpr int dx_ = (S) / 4, ks_ = (S) % 4;                                                        \
        _Pragma("unroll") for (int dy = 0; dy < G::KD; ++dy)                                               \
            wf[BUF][dy] = *(const half8_t*)(wl + ((((dy * G::KD + dx_) * 4 + ks_) * 2) << 10));            \
        _Pragma("unroll") for (int pr = 0; pr < NR; ++pr)                                                  \
            af[BUF][pr] = *(const half8_t*)(abuf + A[dx_][pr] + ((ks_ << 5) ^ Z[dx_][pr]));                \
    }
        MOE_LOAD_STEP(0, 0)
        __builtin_amdgcn_sched_group_barrier(0x100, G::KD + NR, 0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int cb = s & 1;
            if (s + 1 < NSTEP) {
                switch (s + 1) {   // constant after unrolling
#define MOE_CASE(N) case N: if (N < NSTEP) MOE_LOAD_STEP((N < NSTEP ? N : 0), ((N) & 1)) break;
                    MOE_CASE(1) MOE_CASE(2) MOE_CASE(3) MOE_CASE(4) MOE_CASE(5) MOE_CASE(6)
                    MOE_CASE(7) MOE_CASE(8) MOE_CASE(9) MOE_CASE(10) MOE_CASE(11)
#undef MOE_CASE
                }
            }
#pragma unroll
            for (int pr = 0; pr < NR; ++pr)
#pragma unroll
                for (int dy = 0; dy < G::KD; ++dy) {
                    const int o = pr - dy;
                    if (o >= 0 && o < 4)
                        acc[o] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cb][dy], af[cb][pr], acc[o], 0, 0, 0);
                }
            // pin the interleave: one LDS read slotted behind each of the first MFMAs of the step
            if (s + 1 < NSTEP) {
#pragma unroll
                for (int i = 0; i < G::KD + NR; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * G::KD - (G::KD + NR) > 0 ? 4 * G::KD - (G::KD + NR) : 0, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * G::KD, 0);
            }
        }
#undef MOE_LOAD_STEP

        if (a.acc_mode == 4 && seg == 1) {   // both low-order products are in: bring them to the scale of the main product
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[o][e] *= 0.00048828125f;
        }
        // ---- epilogue (after the last K segment of the item) ---------------------------------------
        if (seg == NSEG - 1) {
            const int x = pxi * kTileW + j;
            const int r = a.r;
            const int si = (r > 1) ? chunk / r : 0, sj = (r > 1) ? chunk % r : 0;
            const int cbase = ((r > 1) ? 0 : chunk * kCB) + wn * 32 + hh * 4;   // + 8*grp + e
            const int pcb = chunk * kCB + wn * 32 + hh * 4;                      // packed channel (bias, acc32)
            const int Wo = a.W * r, Ho = a.H * r;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int y = pyi * kTileH + wr * 4 + o;
                if (y < a.H && x < a.W) {
                    const long long opix = ((long long)(b * Ho + y * r + si) * Wo + (x * r + sj)) * a.out_cs;
                    const long long apix = ((long long)(b * a.H + y) * a.W + x) * (a.nchunks * kCB);
                    float dot = 0.f;
#pragma unroll
                    for (int grp = 0; grp < 4; ++grp) {
                        float4_t v = {acc[o][grp * 4 + 0], acc[o][grp * 4 + 1], acc[o][grp * 4 + 2], acc[o][grp * 4 + 3]};
                        if (a.acc_mode == 1) { *(float4_t*)(a.acc32 + apix + pcb + grp * 8) = v; continue; }
                        if (a.acc_mode == 2) {
                            float4_t* p = (float4_t*)(a.acc32 + apix + pcb + grp * 8);
                            *p = *p + v; continue;
                        }
                        if (a.acc_mode == 3) v = v + *(const float4_t*)(a.acc32 + apix + pcb + grp * 8) * 0.00048828125f;
                        v = v + bias4[grp];
                        v = v * a.scale;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * a.slope;
                        if (a.tail1_out) {       // fused 1x1 tail: the fp32 activation goes straight into the 48->1 dot product
#pragma unroll
                            for (int e = 0; e < 4; ++e) dot += v[e] * tw[grp][e];
                            continue;
                        }
                        const long long oo = opix + cbase + grp * 8;
                        if (a.res) {
                            const half4_t rv = *(const half4_t*)(a.res + oo);
                            float4_t rf = {(float)rv[0], (float)rv[1], (float)rv[2], (float)rv[3]};
                            if (a.res_lo) {
                                const half4_t rl = *(const half4_t*)(a.res_lo + oo);
#pragma unroll
                                for (int e = 0; e < 4; ++e) rf[e] += (float)rl[e] * 0.00048828125f;
                            }
                            v = v + rf;
                        }
                        half4_t hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                        *(half4_t*)(a.out + oo) = hv;
                        if (a.out_lo) {
                            half4_t lv;
#pragma unroll
                            for (int e = 0; e < 4; ++e) lv[e] = (half_t)((v[e] - (float)hv[e]) * 2048.f);
                            *(half4_t*)(a.out_lo + oo) = lv;
                        }
                    }
                    if (a.tail1_out) {           // lanes (j, 0) and (j, 1) hold the two 16-channel parts of this wave's 32 channels
                        dot += __shfl_xor(dot, 32);
                        if (hh == 0) a.tail1_out[(long long)wn * a.B * Ho * Wo + (long long)(b * Ho + y * r + si) * Wo + (x * r + sj)] = dot;
                    }
                }
            }
        }

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (reload_w) {
            const int nb = (g + (u / NSEG + 1) * a.G) / (a.px * a.py);
            load_weights(nb);
            wb_loaded = nb;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
}

template <int TAPS, int NSEG>
constexpr int lds_bytes() { return NSEG * TAPS * 4 * 2 * kFragBytes + 2 * Geo<TAPS>::PATCH_BYTES; }

template <int TAPS, int NSEG>
void launch_t(const ConvArgs& a, hipStream_t s)
{
    const int blocks = a.nchunks * ((a.G + 7) / 8) * 8;
    constexpr int lds = lds_bytes<TAPS, NSEG>();
    conv_mfma_kernel<TAPS, NSEG><<<dim3(blocks), dim3(256), lds, s>>>(a);
}

// ---------------------------------------------------------------------------------------------------
// Scalar device convolution for kernel debugging (MOE_PREC_DEBUG_DIRECT): independent indexing, plain
// OIHW fp32 weights, pixel shuffle by formula.  One thread per (pixel, output channel).
// ---------------------------------------------------------------------------------------------------
__global__ void conv_direct_kernel(DirectConvArgs a)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)a.B * a.H * a.W * a.cout;
    if (idx >= total) return;
    const int co = (int)(idx % a.cout);
    long long p = idx / a.cout;
    const int x = (int)(p % a.W); p /= a.W;
    const int y = (int)(p % a.H);
    const int b = (int)(p / a.H);
    const int k = a.k, pad = k / 2;
    const float* w = a.w + (long long)b * a.w_batch_stride + (long long)co * a.cin * k * k;
    float acc = a.bias ? a.bias[co] : 0.f;
    for (int ky = 0; ky < k; ++ky)
        for (int kx = 0; kx < k; ++kx) {
            const int yy = y + ky - pad, xx = x + kx - pad;
            if (yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) continue;
            const half_t* ip = a.in + ((long long)(b * a.H + yy) * a.W + xx) * a.in_cs;
            for (int c = 0; c < a.cin; ++c) acc += (float)ip[c] * w[(c * k + ky) * k + kx];
        }
    acc *= a.scale;
    acc = acc >= 0.f ? acc : acc * a.slope;
    const int r = a.r;
    int oc = co, oy = y, ox = x;
    if (r > 1) { oc = co / (r * r); const int rem = co % (r * r); oy = y * r + rem / r; ox = x * r + rem % r; }
    const long long oo = ((long long)(b * a.H * r + oy) * (a.W * r) + ox) * a.out_cs + oc;
    if (a.res) acc += (float)a.res[oo];
    a.out[oo] = (half_t)acc;
}

template <int TAPS, int NSEG>
hipError_t set_lds_limit()
{
    return hipFuncSetAttribute((const void*)conv_mfma_kernel<TAPS, NSEG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               lds_bytes<TAPS, NSEG>());
}

int g_max_groups = 0;

}  // namespace

hipError_t conv_mfma_init()
{
    hipError_t e;
    if ((e = set_lds_limit<9, 1>()) != hipSuccess) return e;
    if ((e = set_lds_limit<1, 1>()) != hipSuccess) return e;
    if ((e = set_lds_limit<1, 4>()) != hipSuccess) return e;
    if ((e = set_lds_limit<1, 3>()) != hipSuccess) return e;
    if ((e = conv3x3_sp_init()) != hipSuccess) return e;
    if ((e = arsb32c_init()) != hipSuccess) return e;
    if ((e = conv64_q8_init()) != hipSuccess) return e;
    if ((e = conv64_sq_init()) != hipSuccess) return e;
    if ((e = arsb_sq_init()) != hipSuccess) return e;
    if ((e = conv64_s_init()) != hipSuccess) return e;
    if ((e = conv64_x3_init()) != hipSuccess) return e;
    if ((e = conv1x1_init()) != hipSuccess) return e;
    if ((e = conv1x1_f2_init()) != hipSuccess) return e;
    if ((e = conv3x3_rw_init()) != hipSuccess) return e;
    if ((e = conv3x3_ps4_init()) != hipSuccess) return e;
    if ((e = conv3x3_ps9_init()) != hipSuccess) return e;
    int dev = 0;
    e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    int cus = 0;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    g_max_groups = cus > 0 ? cus : 256;
    return hipSuccess;
}

int conv_mfma_max_groups() { return g_max_groups > 0 ? g_max_groups : 256; }

void launch_conv_mfma(const ConvArgs& a, int taps, int nseg, hipStream_t s)
{
    if (taps == 9 && nseg == 1) launch_t<9, 1>(a, s);
    else if (taps == 1 && nseg == 1) launch_t<1, 1>(a, s);
    else if (taps == 1 && nseg == 4) launch_t<1, 4>(a, s);
    else if (taps == 1 && nseg == 3) launch_t<1, 3>(a, s);     // acc_mode 4: the three split-precision products of a 64-channel 1x1 conv
}

void launch_conv_direct(const DirectConvArgs& a, hipStream_t s)
{
    const long long total = (long long)a.B * a.H * a.W * a.cout;
    const int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(conv_direct_kernel, dim3(blocks), dim3(256), 0, s, a);
}
