// conv3x3_ps9b.hip -- conv3x3_ps9's layer (x3 nets: 3x3 64 -> 576 + bias, PixelShuffle(3), PReLU, fused 64 -> 1 tail; python/models.py:33-36,125-143 of the reference)
// with ALL FOUR SIMDs of a CU at work.
//
// conv3x3_ps9 gives a wave one phase (64 output channels = 288 weight registers): a phase row is three waves and the fourth SIMD idles; the package power cap returns part of
// its share as clock (1.84 against 1.62 GHz, profiles/r06/t_pmc_ps9_vs_ps4.txt) and 12 % are lost.  Here the 192 output channels of a phase row are dealt as TWELVE 16-channel
// tiles of v_mfma_f32_16x16x32_f16, three per wave (216 weight registers, all in AGPRs):
//
//   wave w        tiles (phase column pj, channel quarter): w0 (0,0) (0,1) (0,2) | w1 (1,0) (1,1) (0,3) | w2 (1,2) (1,3) (2,0) | w3 (2,1) (2,2) (2,3); both 16-pixel halves of the
//                 32-pixel column: per input row 3 dx x 2 k-halves x 2 pixel halves = 12 fragment reads (ds_read_b128), each feeding 3 tiles x 3 output rows = 9 MFMAs: 108
//                 MFMAs of 16 cycles per row step and wave (conv3x3_ps9: 72 of 32 cycles on three waves);
//   LDS image     conv64_x3.hip's: pixel at col * 128, logical 16-byte slot s at s ^ ((col >> 1) & 3); lane (n, kq) of k-half kh reads slot (2 kh + (kq >> 1)) ^ 4 (kq & 1) -- the
//                 weights' k order follows (engine.cpp pack_ps9b);
//   epilogue      PReLU [+ hi / lo split], then the tail GEMM over the wave's own channels: tiles 0, 1 (one phase: K = 32) and tile 2 (K = 16 + zeros) on 16x16x32 MFMAs with
//                 the tail's weights (rows = taps) and their remainders (a second MFMA) in registers; where the wave's three tiles belong to ONE phase (w0, w3) the second
//                 product chains onto the first.  Every phase receives exactly two partial tap images: the tap image in LDS is [row 8][phase 3][part 2][tap 9][32 px];
//   finishing, planes, aprons, tailadd3, ranges, zero-block skip, block -> (range, phase row) map: conv3x3_ps9's (the finishing adds the two parts).
//
// Arithmetic: fp16 operands, fp32 accumulation; per output the products are summed tap row by tap row in time, inside a tap row by (dx, k-half): another order than the
// 32x32x16 forms' (results agree to fp32 summation order and the odd flipped fp16 rounding of an activation).
#include "common.h"
#include "rowtile.h"
#include "../../include/moephoto_amd.h"
#include <algorithm>
#include <type_traits>

#ifndef PS9B_FILL
#define PS9B_FILL 2       // VALU / SALU slots pinned behind each 16-cycle MFMA
#endif

namespace {

constexpr int RB = 4;                           // rows per block: one DMA fill, one barrier
constexpr int PW = kTileW + 2;                  // 34
constexpr int ROWB = PW * 128;
constexpr int BLKB = RB * ROWB;                 // 17,408
constexpr int NPIECE = BLKB / 1024;             // 17
constexpr int TEXP = 54 * 128;                  // one row of the tap image: [phase 3][part 2][tap 9][32 px] fp32, then the exports [side 2][dy 3][part 2] and pad words
constexpr int TREC = TEXP + 64;
constexpr int TROWS = 8;
constexpr int OFF_T = 2 * BLKB;                 // 34,816
constexpr int OFF_BIAS = OFF_T + TROWS * TREC;  // + 55,808: bias [wave 4][tile 3][16] fp32
constexpr int OFF_DUMP = OFF_BIAS + 1024;
constexpr int LDS_BYTES = OFF_DUMP + 1024;      // 92,672

typedef unsigned u4_t __attribute__((ext_vector_type(4)));
typedef unsigned u3_t __attribute__((ext_vector_type(3)));
typedef const __attribute__((address_space(3))) half8_t* lds_h8_t;

template <bool SPLIT, bool MASK>
__global__ __launch_bounds__(256) void conv3x3_ps9b_kernel(Ps9Args a)
{
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr unsigned kOOR = 0xFFFF0000u;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w4 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, q4 = lane >> 4;                 // MFMA coordinates: pixel of a 16-pixel half, k quarter / row quad
    const int j = lane & 31, hh = lane >> 5;                   // finishing coordinates: pixel of the column, task half
    const int H = a.H, W = a.W;

    // ---- block -> (range g of G, phase row pi): conv3x3_ps9.hip ---------------------------------------------------------------------------------------------------
    const int px = (W + kTileW - 1) / kTileW, nyb = H / RB;
    int pi, g, G;
    if (a.xcd_map) {
        const int S = (int)gridDim.x >> 3, q = S / 3, r = S - 3 * q, x = (int)blockIdx.x & 7, i = (int)blockIdx.x >> 3;
        G = 8 * q + (8 * r) / 3;
        if (i < 3 * q) { pi = i % 3; g = x * q + i / 3; }
        else {
            const int l = (i - 3 * q) * 8 + x;
            if (l >= 3 * ((8 * r) / 3)) return;
            pi = l % 3; g = 8 * q + l / 3;
        }
    } else { G = (int)gridDim.x / 3; pi = (int)blockIdx.x % 3; g = (int)blockIdx.x / 3; }
    const long long nitems = (long long)a.B * px * nyb;
    int item = (int)(nitems * g / G);
    const int item_end = (int)(nitems * (g + 1) / G);
    if (item >= item_end) return;

    // ---- weights: 3 tiles x 18 A fragments (tap, k-half) of v_mfma_f32_16x16x32_f16, pack_ps9b order [pi][wave][tile][tap 2 + kh][lane][8]: 216 AGPRs ----------------------
    half8_t wf[3][18];
    {
        const half_t* wsrc = a.wpk + (long long)((pi * 4 + w4) * 3) * (18 * 512);
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int f = 0; f < 18; ++f) wf[mt][f] = *(const half8_t*)(wsrc + ((mt * 18 + f) * 64 + lane) * 8);
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int f = 0; f < 18; ++f) asm volatile("" : "+a"(wf[mt][f]));
    }
    // the tail conv's fragments of this wave (engine.cpp: "<key>.frag9b" [wave][A hi, A lo, B hi, B lo][lane][8]): rows = taps, k = the channels of tiles 0, 1 (A) / tile 2 (B)
    half8_t tw[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) tw[f] = *(const half8_t*)(a.tail_w + ((w4 * 4 + f) * 64 + lane) * 8);
    // ---- LDS tables: bias ([wave][tile][16], pack_ps9b order), the tap-image ring as zeros (the slots no lane ever writes stay zero) ---------------------------------------
    if (tid < 192) *(float*)(smem + OFF_BIAS + tid * 4) = a.bias[pi * 192 + tid];
    {
        const u4_t z = {0u, 0u, 0u, 0u};
        for (int o = tid * 16; o < TROWS * TREC; o += 256 * 16) *(u4_t*)(smem + OFF_T + o) = z;
    }
    const unsigned bias_ad = lds0 + (unsigned)(OFF_BIAS + (w4 * 48 + 4 * q4) * 4);      // (tile mt: + 64)

    // ---- buffers ---------------------------------------------------------------------------------------------------------------------------------------------------
    const unsigned in_pad = (unsigned)(RB * W + 1) * 128u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.in - in_pad), 0, (unsigned)a.B * H * W * 128u + in_pad, 0x00020000);
    const unsigned plane_b = (unsigned)a.B * H * W * 36u;      // bytes of one S plane [B][3H][3W] fp32
    const __amdgpu_buffer_rsrc_t rpl = __builtin_amdgcn_make_buffer_rsrc((void*)a.plane, 0, 3u * plane_b, 0x00020000);
    const unsigned apron_b = (unsigned)a.B * px * H * 12u;     // bytes of one (side, dy) apron array [B][px][3H] fp32
    const __amdgpu_buffer_rsrc_t rap = __builtin_amdgcn_make_buffer_rsrc((void*)a.apron, 0, 6u * apron_b, 0x00020000);
    unsigned d_off = 0, d_r = 0, d_cc = 0;
    auto piece_addr = [&](int m) {                           // piece i = w4 + 4 m (m < 4) / 16 (m = 4): the lane's pixel of the 4 x 34 block, its logical 16-byte slot
        unsigned q = (unsigned)((m < 4 ? w4 + 4 * m : 16) * 8 + (lane >> 3));
        asm volatile("" : "+v"(q));
        d_r = __umul24(q, 241u) >> 13;                        // q / 34 (q < 352)
        d_cc = (unsigned)(__mul24((int)d_r, -PW) + (int)q);
        const unsigned sl = (unsigned)(lane & 7) ^ ((d_cc >> 1) & 3u);
        d_off = ((__umul24(d_r, (unsigned)W) + d_cc) << 7) | (sl << 4);
    };
    auto piece_off = [&](int ya, int xa, bool live) {
        const bool ok = ((unsigned)(ya + (int)d_r) < (unsigned)H) & ((unsigned)(xa + (int)d_cc) < (unsigned)W) & live;
        return ok ? d_off : kOOR;
    };

    // ---- B fragment (dx, kh, nt) of an input row: lane (n, kq) reads logical slot (2 kh + (kq >> 1)) ^ 4 (kq & 1) of column 16 nt + n + dx: one address per dx; kh toggles
    // bit 1 of the slot (address ^ 32), nt adds 16 columns (the swizzle term (col >> 1) & 3 is the same: + 2048)
    unsigned fa[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int cc = n16 + dx, z = (cc >> 1) & 3;
        const int slot = (q4 >> 1) ^ (4 * (q4 & 1));
        fa[dx] = lds0 + (unsigned)(cc * 128 + ((slot ^ z) << 4));
        asm volatile("" : "+v"(fa[dx]));
    }
    // ---- tap image: where the lane's T values go.  The wave's two tail products: out 0 = tiles 0, 1 (phase pa), out 1 = tile 2 [+ out 0 where both are one phase] (phase pb):
    //   w0: (-, 0/part 0)   w1: (1/0, 0/1)   w2: (1/1, 2/0)   w3: (-, 2/1)
    // Lane (n, q) holds the rows 4q + e = taps (q < 2: four, q = 2: tap 8) of pixel 16 nt + n.  Tap (dy, dx) of phase column pj at conv pixel x is a term of the HR column
    // 3 x + pj - dx + 1: stored at the consumer's column; columns 32 / -1 are the export slots [side][dy][part]
    const bool chain = (w4 == 0) | (w4 == 3);
    unsigned wa[2][2][4];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int ph = o == 0 ? 1 : (w4 <= 1 ? 0 : 2);
        const int part = o == 0 ? (w4 == 1 ? 0 : 1) : (w4 == 0 || w4 == 2 ? 0 : 1);
        const bool dead = o == 0 && chain;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = 4 * q4 + e;
                const int dy = t / 3, dx = t - 3 * dy;
                const int x = 16 * nt + n16;
                int col = x;
                if (ph == 2 && dx == 0) col = x + 1;
                if (ph == 0 && dx == 2) col = x - 1;
                unsigned off = (unsigned)(((ph * 2 + part) * 9 + t) * 128 + col * 4);
                if (col == kTileW) off = (unsigned)(TEXP + ((0 * 3 + dy) * 2 + part) * 4);
                if (col < 0) off = (unsigned)(TEXP + ((1 * 3 + dy) * 2 + part) * 4);
                if (t > 8 || dead) off = (unsigned)(TEXP + 48 + (lane & 3) * 4);
                wa[o][nt][e] = lds0 + (unsigned)OFF_T + off;
                asm volatile("" : "+v"(wa[o][nt][e]));
            }
    }
    unsigned slope2;
    {
        const half2_t s2 = {(half_t)a.slope, (half_t)a.slope};
        slope2 = __builtin_bit_cast(unsigned, s2);
    }
    const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};

    float4_t acc[4][3][2];    // out row o lives in slot o & 3: [tile][pixel half]
    half8_t fr[3];            // fragment g of a row in fr[g % 3], read two fragments ahead

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    while (item < item_end) {
        // ===== one strip: plane b, column pxi, blocks [s0, s1) = conv rows [4 s0, 4 s1) =====================================================================
        const int s0 = item % nyb;
        const int t_ = item / nyb;
        const int pxi = t_ % px, b = t_ / px;
        const int s1 = min(nyb, s0 + (item_end - item));
        item += s1 - s0;
        const int x0 = pxi * kTileW;
        const int nblk = s1 - s0 + 3;                         // input blocks s0 - 1 .. s1, then one more iteration for the last finishing rounds
        const bool okx = x0 + j < W;
        const bool okn0 = x0 + n16 < W, okn1 = x0 + 16 + n16 < W;
        const int ylo = RB * s0, yhi = RB * s1;
        const int kfirst = s0 == 0 ? 1 : 0;                   // (zero blocks above / below the image: conv3x3_ps9.hip)
        const int kz = s1 == nyb ? nblk - 2 : nblk - 1;       // blocks [kfirst, kz) run MFMAs

        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        {
            const int ya = RB * (s0 - 1 + kfirst), xa = x0 - 1;
            const unsigned org = (unsigned)((b * H + ya + RB) * W + xa + 1) * 128u;
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                piece_addr(m);
                const bool mine = m < 4 || w4 == 0;
                char* dst = smem + (m < 4 ? (w4 + 4 * m) * 1024 : (mine ? 16 * 1024 : OFF_DUMP));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)dst, 16, piece_off(ya, xa, mine), org, 0, 0);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int mt = 0; mt < 3; ++mt) {
                    const float4_t bv = *(const __attribute__((address_space(3))) float4_t*)(bias_ad + (unsigned)(mt * 64));
                    acc[s][mt][0] = bv; acc[s][mt][1] = bv;
                }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            fr[0] = *(lds_h8_t)(fa[0]);
            fr[1] = *(lds_h8_t)(fa[0] + 2048u);
        }

        auto block = [&](int k, auto BUF_, auto KIND_) __attribute__((always_inline)) {
            constexpr int BUF = decltype(BUF_)::value;
            constexpr bool LAST = decltype(KIND_)::value == 2;     // the iteration behind the last input block: only its finishing rounds are wanted
            constexpr bool ZERO = decltype(KIND_)::value == 1;     // an input block of zeros (below the image): epilogues, finishing and barrier, no MFMAs
            const int Rk = RB * (s0 - 1 + k);                 // first input row of this block
            const bool live = k + 1 < kz;                     // (the next block is one that reads its input)
            const int yan = Rk + RB, xan = x0 - 1;
            const unsigned orgn = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((b * H + yan + RB) * W + xan + 1) * 128u));

            // ---- finishing round t (steps 0, 1): this lane's task (row, dy) = divmod(8 t + 2 w4 + hh, 3) of the 12 of the four rows Rk - 6 .. Rk - 3 (t = 1: waves 0, 1) ----
            auto finish = [&](int t) __attribute__((always_inline)) {
                int hv = hh;
                asm volatile("" : "+v"(hv));
                const int id = 8 * t + 2 * w4 + hv;
                const int rho = (id * 11) >> 5;                                       // id / 3 (id < 16)
                const int dy = id - 3 * rho;
                const int yf = Rk - 6 + rho;
                const bool rok = (yf >= ylo) & (yf < yhi) & (id < 12);
                const unsigned rec = lds0 + (unsigned)OFF_T + (unsigned)((yf + 64) & (TROWS - 1)) * (unsigned)TREC;
                const unsigned fb = rec + (unsigned)(dy * 384 + j * 4);
                float fsum[3];
#pragma unroll
                for (int jp = 0; jp < 3; ++jp) {
                    float v[3];
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int ps = (jp + dx + 2) % 3;
                        const float p0 = *(const __attribute__((address_space(3))) float*)(fb + (unsigned)(((ps * 2 + 0) * 9 + dx) * 128));
                        const float p1 = *(const __attribute__((address_space(3))) float*)(fb + (unsigned)(((ps * 2 + 1) * 9 + dx) * 128));
                        v[dx] = p0 + p1;
                    }
                    fsum[jp] = (v[0] + v[1]) + v[2];
                }
                unsigned fo = (unsigned)dy * plane_b + (unsigned)(((b * 3 * H + 3 * yf + pi) * 3 * W) + 3 * (x0 + j)) * 4u;
                asm volatile("" : "+v"(fo));
                fo = (rok & okx) ? fo : kOOR;
                const u3_t o3 = {__builtin_bit_cast(unsigned, fsum[0]), __builtin_bit_cast(unsigned, fsum[1]), __builtin_bit_cast(unsigned, fsum[2])};
                __builtin_amdgcn_raw_buffer_store_b96(o3, rpl, fo, 0, 0);
                // apron[side = j][dy][b][pxi][3 yf + pi]  (lanes j < 2): the export slots' two parts
                const unsigned fe = rec + (unsigned)(TEXP + (((j & 1) * 3 + dy) * 2) * 4);
                const float av = *(const __attribute__((address_space(3))) float*)fe + *(const __attribute__((address_space(3))) float*)(fe + 4u);
                unsigned ao = (unsigned)((j & 1) * 3 + dy) * apron_b + (unsigned)((b * px + pxi) * 3 * H + 3 * yf + pi) * 4u;
                asm volatile("" : "+v"(ao));
                ao = (rok & (j < 2)) ? ao : kOOR;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, av), rap, ao, 0, 0);
            };

            auto step = [&](auto E_) __attribute__((always_inline)) {
                constexpr int e = decltype(E_)::value;
                const int orow = Rk + e - 2;                  // the conv row whose epilogue rides in this step
                const unsigned trow = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((orow + 64) & (TROWS - 1)) * (unsigned)TREC));
                const bool rowok = (unsigned)orow < (unsigned)H;
                constexpr int SL = (e + 2) & 3;               // its accumulator slot

                // ---- the row epilogue of pixel half nt: PReLU [+ split], the tail products, the tap-image writes --------------------------------------------------------
                auto epilogue = [&](auto NT_) __attribute__((always_inline)) {
                    constexpr int nt = decltype(NT_)::value;
                    unsigned hP[3][2], lP[3][2];
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const float s0v = acc[SL][mt][nt][2 * k], s1v = acc[SL][mt][nt][2 * k + 1];
                            unsigned hv, lv = 0;
                            if (SPLIT) {
                                const float t0 = __builtin_fmaxf(s0v, s0v * a.slope), t1 = __builtin_fmaxf(s1v, s1v * a.slope);
                                split2(t0, t1, -2048.f, hv, lv);
                            } else {
                                const half2_t pr = {(half_t)s0v, (half_t)s1v};
                                const half2_t t = pr * __builtin_bit_cast(half2_t, slope2);
                                hv = __builtin_bit_cast(unsigned, __builtin_elementwise_max(pr, t));
                            }
                            hP[mt][k] = hv; lP[mt][k] = lv;
                        }
                    const half8_t bA = __builtin_bit_cast(half8_t, u4_t{hP[0][0], hP[0][1], hP[1][0], hP[1][1]});
                    const half8_t bB = __builtin_bit_cast(half8_t, u4_t{hP[2][0], hP[2][1], 0u, 0u});
                    float4_t g1h = __builtin_amdgcn_mfma_f32_16x16x32_f16(tw[0], bA, zero4, 0, 0, 0);
                    float4_t g1l = __builtin_amdgcn_mfma_f32_16x16x32_f16(tw[1], bA, zero4, 0, 0, 0);
                    if (SPLIT) {
                        const half8_t lA = __builtin_bit_cast(half8_t, u4_t{lP[0][0], lP[0][1], lP[1][0], lP[1][1]});
                        g1l = __builtin_amdgcn_mfma_f32_16x16x32_f16(tw[0], lA, g1l, 0, 0, 0);
                    }
                    float4_t c2h, c2l;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { c2h[i] = chain ? g1h[i] : 0.f; c2l[i] = chain ? g1l[i] : 0.f; }
                    float4_t g2h = __builtin_amdgcn_mfma_f32_16x16x32_f16(tw[2], bB, c2h, 0, 0, 0);
                    float4_t g2l = __builtin_amdgcn_mfma_f32_16x16x32_f16(tw[3], bB, c2l, 0, 0, 0);
                    if (SPLIT) {
                        const half8_t lB = __builtin_bit_cast(half8_t, u4_t{lP[2][0], lP[2][1], 0u, 0u});
                        g2l = __builtin_amdgcn_mfma_f32_16x16x32_f16(tw[2], lB, g2l, 0, 0, 0);
                    }
                    const bool ok = MASK ? (rowok & (nt == 0 ? okn0 : okn1)) : rowok;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float t1 = __builtin_fmaf(g1l[i], 0.00048828125f, g1h[i]);      // remainders: units of 2^-11
                        float t2 = __builtin_fmaf(g2l[i], 0.00048828125f, g2h[i]);
                        t1 = ok ? t1 : 0.f; t2 = ok ? t2 : 0.f;
                        const unsigned a1 = wa[0][nt][i] + trow, a2 = wa[1][nt][i] + trow;
                        asm volatile("ds_write_b32 %0, %1" ::"v"(a1), "v"(t1) : "memory");
                        asm volatile("ds_write_b32 %0, %1" ::"v"(a2), "v"(t2) : "memory");
                    }
                };
                auto bias_in = [&]() __attribute__((always_inline)) {                 // the drained slot becomes the accumulator of conv row Rk + e + 2
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) {
                        const float4_t bv = *(const __attribute__((address_space(3))) float4_t*)(bias_ad + (unsigned)(mt * 64));
                        acc[SL][mt][0] = bv; acc[SL][mt][1] = bv;
                    }
                };
                auto dma = [&](auto M_) __attribute__((always_inline)) {
                    constexpr int m = decltype(M_)::value;
                    piece_addr(m);
                    const bool mine = m < 4 || w4 == 0;
                    const unsigned dsto = (unsigned)__builtin_amdgcn_readfirstlane((int)(m < 4 ? (BUF ^ 1) * BLKB + (w4 + 4 * m) * 1024 : (mine ? (BUF ^ 1) * BLKB + 16 * 1024 : OFF_DUMP)));
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem + dsto), 16, piece_off(yan, xan, live & mine), orgn, 0, 0);
                };

                if constexpr (LAST) {
                    if constexpr (e < 2) finish(e);
                    return;
                }
                // ---- twelve fragment groups g = (dx, kh, nt): nine MFMAs each; the step's other work is dealt to them -------------------------------------------------
                auto group = [&](auto G_) __attribute__((always_inline)) {
                    constexpr int gi = decltype(G_)::value;
                    constexpr int dx = gi >> 2, kh = (gi >> 1) & 1, nt = gi & 1;
                    if (e == 3 && gi == 10) {
                        // this block's T rows are written, the next block's pieces have landed, nobody reads this block's input rows any more
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                    if constexpr (!ZERO) {
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int sl = (e + 1 - dy + 4) & 3;
#pragma unroll
                            for (int mt = 0; mt < 3; ++mt)
                                acc[sl][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[mt][(dy * 3 + dx) * 2 + kh], fr[gi % 3], acc[sl][mt][nt], 0, 0, 0);
                        }
                        {      // the fragment of group gi + 2
                            constexpr int g2 = (gi + 2) % 12;
                            constexpr int rowsel = gi + 2 < 12 ? BUF * RB + e : (e < 3 ? BUF * RB + e + 1 : (BUF ^ 1) * RB);
                            constexpr int dx2 = g2 >> 2, kh2 = (g2 >> 1) & 1, nt2 = g2 & 1;
                            fr[(gi + 2) % 3] = *(lds_h8_t)((fa[dx2] ^ (unsigned)(kh2 * 32)) + (unsigned)(rowsel * ROWB + nt2 * 2048));
                        }
                    }
                    if constexpr (gi == 0) epilogue(std::integral_constant<int, 0>{});
                    if constexpr (gi == 3) epilogue(std::integral_constant<int, 1>{});
                    if constexpr (gi == 6 && e < 2) finish(e);
                    if constexpr (gi == 8 && e == 0) { dma(std::integral_constant<int, 0>{}); dma(std::integral_constant<int, 1>{}); }
                    if constexpr (gi == 8 && e == 1) { dma(std::integral_constant<int, 2>{}); dma(std::integral_constant<int, 3>{}); dma(std::integral_constant<int, 4>{}); }
                    if constexpr (gi == 11) bias_in();
#ifndef PS9B_NOPIN
                    if constexpr (!ZERO) {
#pragma unroll
                        for (int i_ = 0; i_ < 9; ++i_) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            if (i_ == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x006, PS9B_FILL, 0);
                        }
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                };
#define PS9B_G(F) group(std::integral_constant<int, F>{});
                PS9B_G(0) PS9B_G(1) PS9B_G(2) PS9B_G(3) PS9B_G(4) PS9B_G(5) PS9B_G(6) PS9B_G(7) PS9B_G(8) PS9B_G(9) PS9B_G(10) PS9B_G(11)
#undef PS9B_G
            };
            step(std::integral_constant<int, 0>{});
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
            step(std::integral_constant<int, 3>{});
        };

        typedef std::integral_constant<int, 0> Run;
        typedef std::integral_constant<int, 1> Zero;
        typedef std::integral_constant<int, 2> Last;
        int k = kfirst;
        for (; k + 1 < kz; k += 2) {
            block(k, std::integral_constant<int, 0>{}, Run{});
            block(k + 1, std::integral_constant<int, 1>{}, Run{});
        }
        if (k < kz) { block(k, std::integral_constant<int, 0>{}, Run{}); ++k; }
        if (k < nblk - 1) { block(k, std::integral_constant<int, 0>{}, Zero{}); ++k; }
        block(k, std::integral_constant<int, 0>{}, Last{});
    }
#endif
}

template <bool SPLIT, bool MASK>
hipError_t set_limit()
{
    return hipFuncSetAttribute((const void*)conv3x3_ps9b_kernel<SPLIT, MASK>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}

}  // namespace

hipError_t conv3x3_ps9b_init()
{
    hipError_t e;
    if ((e = set_limit<false, false>()) != hipSuccess) return e;
    if ((e = set_limit<false, true>()) != hipSuccess) return e;
    if ((e = set_limit<true, false>()) != hipSuccess) return e;
    return set_limit<true, true>();
}

// the same buffers, predicate and block map as conv3x3_ps9 (a.wpk / a.bias / a.tail_w in pack_ps9b's orders); false: not applicable
bool launch_conv3x3_ps9b(const Ps9Args& a, int max_groups, hipStream_t s)
{
    if (!ps9_tail_applicable(a.B, a.H, a.W, a.slope) || max_groups < 3) return false;
    const int px = (a.W + kTileW - 1) / kTileW;
    const long long items = (long long)a.B * px * (a.H / RB);
    Ps9Args q = a;
    int grid;
    {
        const int S = max_groups / 8, nq = S / 3, r = S - 3 * nq, G = 8 * nq + (8 * r) / 3;
        if (nq >= 1 && items >= G) { q.xcd_map = 1; grid = 8 * S; }
        else { q.xcd_map = 0; grid = 3 * (int)std::min<long long>(items, max_groups / 3); }
    }
    const bool ragged = a.W % kTileW != 0;
    if (a.split) {
        if (ragged) conv3x3_ps9b_kernel<true, true><<<dim3(grid), dim3(256), LDS_BYTES, s>>>(q);
        else conv3x3_ps9b_kernel<true, false><<<dim3(grid), dim3(256), LDS_BYTES, s>>>(q);
    } else {
        if (ragged) conv3x3_ps9b_kernel<false, true><<<dim3(grid), dim3(256), LDS_BYTES, s>>>(q);
        else conv3x3_ps9b_kernel<false, false><<<dim3(grid), dim3(256), LDS_BYTES, s>>>(q);
    }
    return true;
}
