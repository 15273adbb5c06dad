// wino1d_rs_probe.hip -- round 6, the third Winograd probe: tools/micro/wino1d_probe.hip (F(2, 3) along x, 1.5x fewer MFMAs) reached MFMA busy 0.62 with ONE B-fragment read per
// MFMA (an output row at a time: 48 reads for 48 MFMAs).  conv3x3_ps4's design principle was six MFMAs per read; this variant streams INPUT rows the same way: a step = one V row
// r, its 16 fragments (4 positions x 4 k-slices) each feed THREE MFMAs -- output rows r - 1 (kernel row dy = 2), r (dy = 1), r + 1 (dy = 0) -- into three rotating accumulator sets
// (3 x 4 positions x 16 = 192 registers beside the 192 weight registers).  16 reads per 48 MFMAs; a row completes at (k-slice 3, dy = 2) of step r + 1 and its output transform
// rides at the end of that step, its epilogue in the next; six V rows in LDS, ONE barrier per three steps (144 MFMAs), raw input in blocks of three rows a body ahead.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 tools/micro/wino1d_rs_probe.hip -o tools/micro/bin/wino1d_rs_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(2))) _Float16 half2_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
typedef __attribute__((ext_vector_type(16))) float float16_t;
typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;

// a - b on packed halves as ONE v_pk_fma_f16 with a -1 the compiler cannot see through (see wino_probe.hip)
#define hsub8(a_, b_) __builtin_elementwise_fma((b_), neg1_8, (a_))

#ifndef WP_PIN
#define WP_PIN 1
#endif
#ifndef WP_FILL
#define WP_FILL 5          // VALU / SALU instructions pinned behind each MFMA
#endif
#ifndef WP_SLOT
#define WP_SLOT 1          // 1: one unit of <= 16 VALU instructions behind every group of four MFMAs | 0: the first form (32-instruction units)
#endif
#ifndef WP_XCD
#define WP_XCD 1
#endif
#ifndef WP_EVEN
#define WP_EVEN 1          // 1: the micro-ops dealt evenly over a row's twelve groups of four MFMAs | 0: front-loaded (the first v2 run)
#endif
#ifndef WP_AHEAD
#define WP_AHEAD 3         // B fragments read this many MFMA slots ahead
#endif

constexpr int PW = 72, ROWB = PW * 128, BLKB = 3 * ROWB;      // a raw block = three input rows of 66 (+ 6 unused) pixels = 27,648 bytes = 27 pieces
constexpr int NPIECE = 27;
constexpr int OFF_V = 2 * BLKB;                                // 55,296: two raw blocks
constexpr int VROW = 4 * 4096;                                 // one V row: four positions x 32 tiles x 128 B
constexpr int OFF_DUMP = OFF_V + 6 * VROW;                     // 153,600: six V rows
constexpr int OFF_BIAS = OFF_DUMP + 1024;                      // [wave 4][hh 2][16] fp32
constexpr int LDS_BYTES = OFF_BIAS + 512;

struct Args {
    const half_t* in;        // [B][H][W][64]
    const half_t* wpk;       // [cout block 8][dy 3][pos 4][ks 4][lane 64][8]
    const float* bias;       // [256]
    const half_t* tailw;     // [256] (checksum weights)
    half_t* out;             // MODE 1: [B][H][W][256] fp16 (PReLU(conv + bias)), else nullptr
    float* chk;              // MODE 0: [grid][256] per-thread checksums
    float slope;
    int B, H, W;
};

template <int N> using ic = std::integral_constant<int, N>;

template <int MODE>
__global__ __launch_bounds__(256) void wino_kernel(Args a)
{
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, hh = lane >> 5;
    const int H = a.H, W = a.W;
    const int px = W / 64;
#if WP_XCD
    // the two workgroups of a strip (channel halves) on ONE XCD -- workgroups go round-robin over the 8 XCDs, so blocks b and b + 8 share an L2: the partner's read of the
    // same input rows hits it instead of going out to memory a second time
    const int ch = (blockIdx.x >> 3) & 1;
    const int pair = (blockIdx.x >> 4) * 8 + (blockIdx.x & 7);       // 0 .. grid / 2 - 1
#else
    const int ch = blockIdx.x & 1;
    const int pair = blockIdx.x >> 1;
#endif
    const int cb = ch * 4 + w4;                                 // this wave's block of 32 output channels
    constexpr unsigned kOOR = 0xFFFF0000u;

    // ---- weights: 48 A fragments (dy, position, k-slice) in AGPRs ---------------------------------------------------------------------------------------------------------------
    half8_t wf[12][4];
    {
        const half_t* wsrc = a.wpk + (long long)cb * (48 * 512);
#pragma unroll
        for (int p = 0; p < 12; ++p)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wf[p][ks] = *(const half8_t*)(wsrc + ((p * 4 + ks) * 64 + lane) * 8);
#pragma unroll
        for (int p = 0; p < 12; ++p)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) asm volatile("" : "+a"(wf[p][ks]));
    }
    // bias: the initial value of position 1's accumulator (A^T column 1 = (1, 1)); kept in LDS, reloaded where the chain starts
    if (tid < 128) {
        const int bw = tid >> 5, bh = (tid >> 4) & 1, br = tid & 15;
        *(float*)(smem + OFF_BIAS + tid * 4) = a.bias[(ch * 4 + bw) * 32 + 8 * (br >> 2) + 4 * bh + (br & 3)];
    }
    const unsigned bias_ad = lds0 + (unsigned)(OFF_BIAS + (w4 * 2 + hh) * 64);
    half2_t tw2[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { tw2[r][0] = a.tailw[cb * 32 + 8 * (r >> 1) + 4 * hh + 2 * (r & 1)]; tw2[r][1] = a.tailw[cb * 32 + 8 * (r >> 1) + 4 * hh + 2 * (r & 1) + 1]; }
    const half2_t slope2 = {(half_t)a.slope, (half_t)a.slope};
    unsigned neg1_u = 0xBC00BC00u;
    asm volatile("" : "+v"(neg1_u));
    const half2_t neg1_2 = __builtin_bit_cast(half2_t, neg1_u);
    const half8_t neg1_8 = __builtin_shufflevector(neg1_2, neg1_2, 0, 1, 0, 1, 0, 1, 0, 1);

    // ---- B fragment of (V row slot, position p, k-slice ks): tile n at n * 128, 16-byte slot (2 ks + hh) ^ ((n >> 1) & 7): one base per k-slice, the rest immediates ---------
    unsigned fa4[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fa4[ks] = (lds0 + (unsigned)OFF_V + (unsigned)(n * 128 + ((hh ^ ((n >> 1) & 7)) << 4))) ^ (unsigned)(ks << 5);
    // ---- transform thread: tile tx = tid >> 3, channels 8 c8 .. + 7; raw column cc at line cc ^ ((cc >> 1) & 1) (no bank conflicts at a stride of two pixels: wino_probe.hip) ----
    const int tx = tid >> 3, c8 = tid & 7;
    const unsigned va = lds0 + (unsigned)OFF_V + (unsigned)(tx * 128 + ((c8 ^ ((tx >> 1) & 7)) << 4));
    unsigned ra[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { const int cc = 2 * tx + c; ra[c] = lds0 + (unsigned)((cc ^ ((cc >> 1) & 1)) * 128 + (c8 << 4)); }

    // ---- input descriptor (shifted so that block origins are non-negative offsets) -------------------------------------------------------------------------------------------
    const unsigned in_pad = (unsigned)(4 * W + 1) * 128u;      // (block origins: image row 3B - 1 >= -1)
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in - in_pad), 0, (unsigned)a.B * H * W * 128u + in_pad, 0x00020000);
    unsigned d_off[7];                                          // piece p = w4 + 4 m (p < 27): the lane's pixel of the 3 x 72 block | its row in bits 0..1 (the offset is 16-byte aligned)
    auto piece_offsets = [&](int x0) {
#pragma unroll
        for (int m = 0; m < 7; ++m) {
            const unsigned q = (unsigned)((w4 + 4 * m) * 8 + (lane >> 3));
            const unsigned r = q / 72u, lc = q - 72u * r;
            const unsigned cc = lc ^ ((lc >> 1) & 1u);
            const unsigned sl = (unsigned)(lane & 7);
            const bool ok = cc < 66u && (unsigned)(x0 - 1 + (int)cc) < (unsigned)W && (w4 + 4 * m) < NPIECE;
            d_off[m] = ok ? ((((r * (unsigned)W + cc) << 7) | (sl << 4)) | (r & 3u)) : kOOR;
        }
    };
    float chk = 0.f;

    for (int item = pair; item < a.B * px; item += gridDim.x >> 1) {
        const int b = item / px, x0 = (item - b * px) * 64;
        piece_offsets(x0);
        // raw block B = image rows 3B - 1 .. 3B + 1 (columns x0 - 1 .. x0 + 64) into raw slot B & 1; rows / columns outside the image: zeros
        auto dma_piece = [&](int B_, auto M_) __attribute__((always_inline)) {
            constexpr int m = decltype(M_)::value;
            const int y0 = 3 * B_ - 1;
            const unsigned org = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((b * H + y0 + 4) * W + x0) * 128u));
            const bool mine = (w4 + 4 * m) < NPIECE;
            const unsigned dsto = (unsigned)__builtin_amdgcn_readfirstlane((int)(mine ? (B_ & 1) * BLKB + (w4 + 4 * m) * 1024 : OFF_DUMP));
            const unsigned rr = d_off[m] & 3u;
            const bool ok = d_off[m] != kOOR && (unsigned)(y0 + (int)rr) < (unsigned)H;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + dsto), 16, ok ? (d_off[m] & ~3u) : kOOR, org, 0, 0);
        };
        auto dma_block = [&](int B_) { dma_piece(B_, ic<0>{}); dma_piece(B_, ic<1>{}); dma_piece(B_, ic<2>{}); dma_piece(B_, ic<3>{}); dma_piece(B_, ic<4>{}); dma_piece(B_, ic<5>{}); dma_piece(B_, ic<6>{}); };
        half8_t xd[4];
        auto x_load = [&](int rslot, int rrow) __attribute__((always_inline)) {
            const unsigned rb = (unsigned)(rslot * BLKB + rrow * ROWB);
#pragma unroll
            for (int c = 0; c < 4; ++c) xd[c] = *(lds_h8_t)(ra[c] + rb);
        };
        auto x_put = [&](int vslot) __attribute__((always_inline)) {
            const unsigned vb = va + (unsigned)(vslot * VROW);
            *(__attribute__((address_space(3))) half8_t*)(vb) = hsub8(xd[0], xd[2]);
            *(__attribute__((address_space(3))) half8_t*)(vb + 4096u) = xd[1] + xd[2];
            *(__attribute__((address_space(3))) half8_t*)(vb + 8192u) = hsub8(xd[2], xd[1]);
            *(__attribute__((address_space(3))) half8_t*)(vb + 12288u) = hsub8(xd[1], xd[3]);
        };

        // ---- prologue.  Step i = V row i - 1 (slot i % 6); body = three steps; body B reads V rows of raw block B, transforms raw block B + 1 (landed before it starts) into the
        // next body's V slots and fetches raw block B + 2.  Before body 0: raw blocks 0, 1 landed, V rows -1, 0, 1 in slots 0, 1, 2 -------------------------------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // everybody has left the previous strip
        dma_block(0); dma_block(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        x_load(0, 0); x_put(0);
        x_load(0, 1); x_put(1);
        x_load(0, 2); x_put(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();

        const float16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float16_t acc[3][4];                                    // output row o lives in set (o + 1) % 3
#pragma unroll
        for (int sidx = 0; sidx < 3; ++sidx)
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) acc[sidx][pp] = zero16;
        float16_t Y0 = zero16, Y1 = zero16;
        half8_t fr[4];
        unsigned pk[8];
        auto op_e = [&](int oe, auto J_, auto HF_) __attribute__((always_inline)) {
            constexpr int j = decltype(J_)::value, hf = decltype(HF_)::value;
            const float16_t& Yv = j ? Y1 : Y0;
#pragma unroll
            for (int r = 4 * hf; r < 4 * hf + 4; ++r) {
                const half2_t v = {(half_t)Yv[2 * r], (half_t)Yv[2 * r + 1]};
                const half2_t m = __builtin_elementwise_max(v, v * slope2);
                pk[r] = __builtin_bit_cast(unsigned, m);
                if (MODE == 0) chk = __builtin_amdgcn_fdot2(m, tw2[0], chk, false);      // (one weight pair for every channel pair: the stand-in needs the instruction, not 8 registers)
            }
            if (MODE == 1 && hf == 1) {
                const int ox = x0 + 2 * n + j;
                if (oe >= 0 && oe < H) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const u2_t v2 = {pk[2 * qd], pk[2 * qd + 1]};
                        *(u2_t*)(a.out + (((long long)(b * H + oe) * W + ox) * 256 + cb * 32 + 8 * qd + 4 * hh)) = v2;
                    }
                }
            }
        };

        // ---- one step: V row r = i - 1, i = 6 j + V.  Per k-slice: four fragment reads, then 12 MFMAs -- dy = 2 into the set of output row r - 1 (which completes at k-slice 3),
        // dy = 1 into row r's, dy = 0 into row r + 1's (opened here: zero / bias at k-slice 0).  Behind each group of four MFMAs one unit of the other work ------------------------
        auto step = [&](int i, auto V_) __attribute__((always_inline)) {
            constexpr int v = decltype(V_)::value, u = v % 3, bpar = (v / 3) & 1;      // v = i % 6: V slot; u: step of the body; bpar: parity of the body = raw slot of ITS block
            constexpr int sC = (u + 2) % 3, sM = u, sN = (u + 1) % 3;                  // accumulator sets of output rows r - 1 (completing), r, r + 1 (new)
            const int r = i - 1;
            auto grp = [&](auto G_) __attribute__((always_inline)) {                  // G = 0..11 = (ks, d): d = 0 -> dy 2, 1 -> dy 1, 2 -> dy 0
                constexpr int g = decltype(G_)::value, ks = g / 3, d = g % 3;
                if (d == 0) {
#pragma unroll
                    for (int pp = 0; pp < 4; ++pp) fr[pp] = *(lds_h8_t)(fa4[ks] + (unsigned)(v * VROW + pp * 4096));
                }
                if (d == 2 && ks == 0) {                                               // the new row's accumulators: zero, bias for position 1
                    acc[sN][0] = acc[sN][2] = acc[sN][3] = zero16;
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const float4_t bq = *(const __attribute__((address_space(3))) float4_t*)(bias_ad + (unsigned)(q4 * 16));
                        acc[sN][1][4 * q4] = bq[0]; acc[sN][1][4 * q4 + 1] = bq[1]; acc[sN][1][4 * q4 + 2] = bq[2]; acc[sN][1][4 * q4 + 3] = bq[3];
                    }
                }
                constexpr int st = d == 0 ? sC : d == 1 ? sM : sN, dy = 2 - d;
#pragma unroll
                for (int pp = 0; pp < 4; ++pp) acc[st][pp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[dy * 4 + pp][ks], fr[pp], acc[st][pp], 0, 0, 0);
                // ---- what rides behind these four MFMAs: ONE unit of <= 16 VALU instructions per group (an MFMA hides ~5 other instructions when they sit evenly in its gaps) ----
#if WP_SLOT
                // the row completed by the PREVIOUS step (r - 2): the second half of its output transform, then its epilogue; this step's completing row (r - 1) opens its
                // transform behind group 9 (k-slice 3, dy 2); the new row's accumulators are first written in group 2: Y1 is read out before that
                if (g == 0) Y1 = acc[sN][1] - acc[sN][2];                               // (set sN still holds row r - 2 here: it was the completing set of the previous step)
                if (g == 1) Y1 = Y1 - acc[sN][3];
                if (g == 2) op_e(r - 2, ic<0>{}, ic<0>{});
                if (g == 3) op_e(r - 2, ic<0>{}, ic<1>{});
                if (g == 4) op_e(r - 2, ic<1>{}, ic<0>{});
                if (g == 5) op_e(r - 2, ic<1>{}, ic<1>{});
                if (g == 6) x_load(bpar ^ 1, u);
                if (g == 7) x_put((v + 3) % 6);
                if (u == 0 && g >= 8 && g <= 9) {
                    const int B2 = i / 3 + 2;
                    if (g == 8) { dma_piece(B2, ic<0>{}); dma_piece(B2, ic<1>{}); dma_piece(B2, ic<2>{}); dma_piece(B2, ic<3>{}); }
                    if (g == 9) { dma_piece(B2, ic<4>{}); dma_piece(B2, ic<5>{}); dma_piece(B2, ic<6>{}); }
                }
                if (g == 10) Y0 = acc[sC][0] + acc[sC][1];
                if (g == 11) Y0 = Y0 + acc[sC][2];
#else
                if (g == 0) op_e(r - 2, ic<0>{}, ic<0>{});                             // (the row whose Y0 / Y1 closed the previous step)
                if (g == 1) op_e(r - 2, ic<0>{}, ic<1>{});
                if (g == 2) op_e(r - 2, ic<1>{}, ic<0>{});
                if (g == 3) op_e(r - 2, ic<1>{}, ic<1>{});
                if (g == 4) x_load(bpar ^ 1, u);                                        // raw block of the NEXT body, its row u -> V row r + 3
                if (g == 5) x_put((v + 3) % 6);
                if (u == 0 && g >= 2 && g <= 8) {                                       // raw block (body + 2) into this body's own raw slot (transformed in the previous body)
                    const int B2 = i / 3 + 2;
                    if (g == 2) dma_piece(B2, ic<0>{});
                    if (g == 3) dma_piece(B2, ic<1>{});
                    if (g == 4) dma_piece(B2, ic<2>{});
                    if (g == 5) dma_piece(B2, ic<3>{});
                    if (g == 6) dma_piece(B2, ic<4>{});
                    if (g == 7) dma_piece(B2, ic<5>{});
                    if (g == 8) dma_piece(B2, ic<6>{});
                }
                if (g == 10) Y0 = (acc[sC][0] + acc[sC][1]) + acc[sC][2];              // row r - 1 is complete since group 9 (k-slice 3, dy 2)
                if (g == 11) Y1 = (acc[sC][1] - acc[sC][2]) - acc[sC][3];
#endif
#if WP_PIN
#pragma unroll
                for (int i_ = 0; i_ < 4; ++i_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x080, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x006, WP_FILL, 0);
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            };
            grp(ic<0>{}); grp(ic<1>{}); grp(ic<2>{}); grp(ic<3>{}); grp(ic<4>{}); grp(ic<5>{}); grp(ic<6>{}); grp(ic<7>{}); grp(ic<8>{}); grp(ic<9>{}); grp(ic<10>{}); grp(ic<11>{});
            if (u == 2) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        };
        const int nsteps = (H + 2 + 5) / 6 * 6;                 // V rows -1 .. H (and a few zero rows behind), in pairs of bodies
        for (int i = 0; i < nsteps; i += 6) {
            step(i, ic<0>{}); step(i + 1, ic<1>{}); step(i + 2, ic<2>{}); step(i + 3, ic<3>{}); step(i + 4, ic<4>{}); step(i + 5, ic<5>{});
        }
        // the row whose Y closed the last step (row nsteps - 3 >= H - 1 ... its epilogue; rows >= H are discarded by the store's predicate)
#if WP_SLOT
        { constexpr int sL = (5 % 3 + 2) % 3; Y1 = (acc[sL][1] - acc[sL][2]) - acc[sL][3]; }      // (the completing set of the last step, v = 5)
#endif
        op_e(nsteps - 3, ic<0>{}, ic<0>{}); op_e(nsteps - 3, ic<0>{}, ic<1>{}); op_e(nsteps - 3, ic<1>{}, ic<0>{}); op_e(nsteps - 3, ic<1>{}, ic<1>{});
    }
    if (MODE == 0) a.chk[blockIdx.x * 256 + tid] = chk;
}

// ---- host ------------------------------------------------------------------------------------------------------------------------------------------------------------
static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.f * 2.f - 1.f; }

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 96, H = argc > 2 ? atoi(argv[2]) : 512, W = argc > 3 ? atoi(argv[3]) : 512, reps = argc > 4 ? atoi(argv[4]) : 20;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    hipFuncSetAttribute((const void*)wino_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)wino_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    // weights g[cout 256][cin 64][3][3], U = G g G^T in fp32, rounded to fp16, packed as A fragments
    unsigned seed = 12345;
    std::vector<float> g(256 * 64 * 9), bias(256);
    for (auto& v : g) v = frand(seed) * 0.06f;
    for (auto& v : bias) v = frand(seed) * 0.1f;
    std::vector<half_t> wpk((size_t)8 * 48 * 512), tailw(256);
    for (auto& v : tailw) v = (half_t)(frand(seed) * 0.1f);
    static const float G[4][3] = {{1, 0, 0}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0, 0, 1}};
    std::vector<float> U((size_t)256 * 64 * 12);                   // U[o][c][dy][p] = sum_k g[o][c][dy][k] G[p][k]
    for (int o = 0; o < 256; ++o)
        for (int c = 0; c < 64; ++c) {
            const float* gg = &g[((size_t)o * 64 + c) * 9];
            for (int dy = 0; dy < 3; ++dy) for (int p = 0; p < 4; ++p) U[((size_t)o * 64 + c) * 12 + dy * 4 + p] = G[p][0] * gg[dy * 3 + 0] + G[p][1] * gg[dy * 3 + 1] + G[p][2] * gg[dy * 3 + 2];
        }
    for (int cb = 0; cb < 8; ++cb)
        for (int p = 0; p < 12; ++p)
            for (int ks = 0; ks < 4; ++ks)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e)
                        wpk[((((size_t)cb * 12 + p) * 4 + ks) * 64 + l) * 8 + e] = (half_t)U[((size_t)(cb * 32 + (l & 31)) * 64 + ks * 16 + 8 * (l >> 5) + e) * 12 + p];
    half_t *d_w, *d_tw; float* d_bias;
    hipMalloc(&d_w, wpk.size() * 2); hipMalloc(&d_tw, 512); hipMalloc(&d_bias, 1024);
    hipMemcpy(d_w, wpk.data(), wpk.size() * 2, hipMemcpyHostToDevice); hipMemcpy(d_tw, tailw.data(), 512, hipMemcpyHostToDevice); hipMemcpy(d_bias, bias.data(), 1024, hipMemcpyHostToDevice);
    const float slope = 0.2f;

    // ---- validation: 2 planes of 24 x 64 against the direct convolution (fp32 on the fp16 inputs) ---------------------------------------------------------------------------
    {
        const int vB = 2, vH = 24, vW = 128;
        std::vector<half_t> x((size_t)vB * vH * vW * 64);
        for (auto& v : x) v = (half_t)frand(seed);
        half_t *d_x, *d_o;
        hipMalloc(&d_x, x.size() * 2 + 4096); hipMalloc(&d_o, (size_t)vB * vH * vW * 256 * 2);
        hipMemcpy(d_x, x.data(), x.size() * 2, hipMemcpyHostToDevice);
        hipMemset(d_o, 0, (size_t)vB * vH * vW * 256 * 2);
        Args a{d_x, d_w, d_bias, d_tw, d_o, nullptr, slope, vB, vH, vW};
        wino_kernel<1><<<dim3(16), dim3(256), LDS_BYTES>>>(a);
        if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "validation launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
        std::vector<half_t> o((size_t)vB * vH * vW * 256);
        hipMemcpy(o.data(), d_o, o.size() * 2, hipMemcpyDeviceToHost);
        double worst = 0, big = 0;
        long long bad = 0;
        for (int b = 0; b < vB; ++b)
            for (int y = 0; y < vH; ++y)
                for (int xx = 0; xx < vW; ++xx)
                    for (int oc = 0; oc < 256; oc += 3) {
                        double s = bias[oc];
                        for (int dy = 0; dy < 3; ++dy)
                            for (int dx = 0; dx < 3; ++dx) {
                                const int iy = y + dy - 1, ix = xx + dx - 1;
                                if (iy < 0 || iy >= vH || ix < 0 || ix >= vW) continue;
                                const half_t* xp = &x[(((size_t)b * vH + iy) * vW + ix) * 64];
                                for (int c = 0; c < 64; ++c) s += (double)(float)xp[c] * g[((size_t)oc * 64 + c) * 9 + dy * 3 + dx];
                            }
                        const double want = s > 0 ? s : s * slope;
                        const double got = (float)o[(((size_t)b * vH + y) * vW + xx) * 256 + oc];
                        const double e = std::fabs(got - want);
                        if (e > worst) worst = e;
                        if (std::fabs(want) > big) big = std::fabs(want);
                        if (e > 0.02) ++bad;
                    }
        printf("validation (2 x 24 x 128, every third channel): max-abs error %.3e against the direct conv (|y| <= %.2f), %lld values off by more than 0.02 -> %s\n", worst, big, bad, bad == 0 ? "OK" : "WRONG");
        hipFree(d_x); hipFree(d_o);
        if (bad) return 2;
    }

    // ---- timing at the U-up1 shape --------------------------------------------------------------------------------------------------------------------------------------
    std::vector<half_t> x((size_t)B * H * W * 64);
    {
        unsigned s2 = 777;
        for (size_t i = 0; i < x.size(); ++i) x[i] = (half_t)(frand(s2) * 0.7f);
    }
    half_t* d_x; float* d_chk;
    hipMalloc(&d_x, x.size() * 2 + 4096); hipMalloc(&d_chk, (size_t)cus * 256 * 4);
    hipMemcpy(d_x, x.data(), x.size() * 2, hipMemcpyHostToDevice);
    Args a{d_x, d_w, d_bias, d_tw, nullptr, d_chk, slope, B, H, W};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) wino_kernel<0><<<dim3(cus), dim3(256), LDS_BYTES>>>(a);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) wino_kernel<0><<<dim3(cus), dim3(256), LDS_BYTES>>>(a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flop = 2.0 * B * H * W * 256.0 * 64 * 9;           // the conv's algorithmic FLOPs
    const double mfma = (double)B * (W / 64) * 2 * ((H + 2 + 5) / 6 * 6) * 4 * 48;      // MFMAs executed (per wave: 48 a V row)
    const double peak = cus * 4.0 * 1024 * prop.clockRate * 1e3 / 1e12;
    std::vector<float> hc((size_t)cus * 256);
    hipMemcpy(hc.data(), d_chk, hc.size() * 4, hipMemcpyDeviceToHost);
    double cs = 0; for (float v : hc) cs += v;
    printf("wino1d_rs_probe: %d planes of %d x %d: %.3f ms per launch (%d reps) = %.0f TFLOP/s algorithmic = %.3f of the nominal fp16 peak %.0f; MFMA time at 2.4 GHz %.3f ms -> busy x clock/2.4 = %.3f; checksum %.6e\n",
           B, H, W, ms, reps, flop / ms / 1e9, flop / ms / 1e9 / peak, peak, mfma * 32 / (cus * 4.0) / 2.4e9 * 1e3 * (1.0), mfma * 32 / (cus * 4.0) / 2.4e9 * 1e3 / ms, cs);
    printf("reference: conv3x3_ps4<1> takes 5.30 ms for the same 96-plane launch (profiles/r05), the gate of VERDICT r05 item 1 is 0.8 x 5.30 = 4.24 ms\n");
    return 0;
}
