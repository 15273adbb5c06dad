// misc_kernels.hip -- the HBM-bound kernels around the MFMA convolution: 1->C stem, C->1 tail (+branch sum),
// squeeze-excite pooling / gates, tile stitch, and the uint8/uint16 <-> float image edges.
// All of them stream NHWC fp16 activations with 16-byte accesses (8 lanes = one 128-B pixel line).
#include "common.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include "../../include/moephoto_amd.h"

namespace {

__device__ __forceinline__ float prelu(float v, float slope) { return v >= 0.f ? v : v * slope; }

template <typename T>
__device__ __forceinline__ float ld1(const void* p, long long i) { return (float)((const T*)p)[i]; }

// ---------------------------------------------------------------------------------------------------
// Stem: PReLU(conv 1->64) (models.py:112,117 conv_input+relu; SEDN :219,223; lite 1x1 MoeNet_lite2.py:28,40).
// fp32 weights and arithmetic (the stem's weights are the single most sensitive ones to fp16 rounding).
// One thread = 8 channels x PX consecutive pixels of a row: its 72 weights sit in registers, the 3 x (PX + 2) input window is loaded
// once, and it issues PX 16-byte stores; 8 consecutive threads (the 8 channel groups of the same pixels) write whole 128-B lines.
// The kernel is VALU-bound otherwise (one pixel per thread: ~200 instructions for 72 FMAs).  PX = 8 since round 5 (4 before: 18 LDS
// reads of weights + 18 input loads per 4 pixels -- the launch ran at 2.6 TB/s of stores); per pixel the same taps in the same order.
// ---------------------------------------------------------------------------------------------------
template <typename TIN, int TAPS>
__global__ __launch_bounds__(256) void stem_kernel(StemArgs a)
{
    __shared__ float w[TAPS * 64];
    for (int i = threadIdx.x; i < TAPS * 64; i += 256) w[i] = a.w[i];
    __syncthreads();
    // fp8 low parts: MODE.FP16_OVFL makes the conversion saturate (+-448) instead of producing NaN (conv64_q8.hip runs the same way)
    if (a.out_lo8) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1\n\ts_nop 3" ::: "memory");
    constexpr int PX = 8;
    const int nq = (a.W + PX - 1) / PX;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long q = idx >> 3;                        // (b, y, x-quad)
    if (q >= (long long)a.B * a.H * nq) return;
    const int cg = (int)(idx & 7) * 8;
    const int x0 = (int)(q % nq) * PX;
    const long long t = q / nq;
    const int y = (int)(t % a.H);
    const int b = (int)(t / a.H);
    const TIN* xp = (const TIN*)a.x + (a.x_off ? a.x_off[b] : (long long)b * a.sB);
    constexpr int KD = (TAPS == 9) ? 3 : 1, PAD = (TAPS == 9) ? 1 : 0;
    float wr[TAPS][8];
#pragma unroll
    for (int k = 0; k < TAPS; ++k) {
        const float4 w0 = *(const float4*)(w + k * 64 + cg), w1 = *(const float4*)(w + k * 64 + cg + 4);
        wr[k][0] = w0.x; wr[k][1] = w0.y; wr[k][2] = w0.z; wr[k][3] = w0.w;
        wr[k][4] = w1.x; wr[k][5] = w1.y; wr[k][6] = w1.z; wr[k][7] = w1.w;
    }
    float v[KD][PX + 2 * PAD];
#pragma unroll
    for (int dy = 0; dy < KD; ++dy) {
        const int yy = y + dy - PAD;
        const bool rok = yy >= 0 && yy < a.H;
        const TIN* rp = xp + (long long)yy * a.sH;
#pragma unroll
        for (int c = 0; c < PX + 2 * PAD; ++c) {
            const int xx = x0 + c - PAD;
            v[dy][c] = (rok && xx >= 0 && xx < a.W) ? (float)rp[(long long)xx * a.sW] : 0.f;
        }
    }
#pragma unroll
    for (int e4 = 0; e4 < PX; ++e4) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int dy = 0; dy < KD; ++dy)       // same accumulation order as one pixel per thread: taps ascending
#pragma unroll
            for (int dx = 0; dx < KD; ++dx)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += v[dy][e4 + dx] * wr[dy * KD + dx][e];
        if (x0 + e4 >= a.W) continue;
        const long long p = ((long long)b * a.H + y) * a.W + x0 + e4;
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)prelu(acc[e], a.slope);
        *(half8_t*)(a.out + p * 64 + cg) = o;
        if (a.out_lo) {
            half8_t l;
#pragma unroll
            for (int e = 0; e < 8; ++e) l[e] = (half_t)((prelu(acc[e], a.slope) - (float)o[e]) * 2048.f);
            if (!a.out_lo8) *(half8_t*)(a.out_lo + p * 64 + cg) = l;
            else {
                // the eight values / 4 as e4m3: the word conv64_q8's own conversion would make of them (its cvt4: one asm block, the two words' conversions
                // alternating, a wait state behind the half-register writes)
                typedef unsigned u4v __attribute__((ext_vector_type(4)));
                const u4v lw = __builtin_bit_cast(u4v, l);
                unsigned p0, p1;
                const float quarter = 4.0f;
                asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %2, %6\n\t"
                             "v_cvt_scalef32_pk_fp8_f16 %1, %4, %6\n\t"
                             "v_cvt_scalef32_pk_fp8_f16 %0, %3, %6 op_sel:[0,0,1]\n\t"
                             "v_cvt_scalef32_pk_fp8_f16 %1, %5, %6 op_sel:[0,0,1]\n\t"
                             "s_nop 0"
                             : "=&v"(p0), "=&v"(p1) : "v"(lw[0]), "v"(lw[1]), "v"(lw[2]), "v"(lw[3]), "v"(quarter));
                *(uint2*)((unsigned char*)a.out_lo + p * 64 + cg) = make_uint2(p0, p1);
            }
        }
        if (TAPS == 1 && a.w2) {      // conv_input2 of this pixel in closed form: x * P (x >= 0) or x * Q (StemArgs::w2)
            const float xv = v[0][e4];
            const float* pq = a.w2 + (xv < 0.f ? 64 : 0) + cg;
            const float4 q0 = *(const float4*)pq, q1 = *(const float4*)(pq + 4);
            const float t[8] = {xv * q0.x, xv * q0.y, xv * q0.z, xv * q0.w, xv * q1.x, xv * q1.y, xv * q1.z, xv * q1.w};
            half8_t o2, l2;
#pragma unroll
            for (int e = 0; e < 8; ++e) { o2[e] = (half_t)t[e]; l2[e] = (half_t)((t[e] - (float)o2[e]) * 2048.f); }
            *(half8_t*)(a.out2 + p * 64 + cg) = o2;
            if (a.out2_lo) *(half8_t*)(a.out2_lo + p * 64 + cg) = l2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Tail: y = conv C->1 (in0, w0) [+ conv C->1 (in1, w1)] [+ skip]   (models.py:129-131,149-153 last Conv3x3(64,1)
// of u / convt_R1 and their sum in multiConvt :41-43; NetDN :161; SEDN :220,224; lite MoeNet_lite2.py:33-34,49-51).
// Block = 32-px x 8-row strip; thread = (pixel column, 8-channel group).  Each thread walks the 10 input rows of
// its column once (3 dx each), feeding every loaded 16 B into the up-to-3 output rows it contributes to
// (v_dot2_f32_f16, fp32 accumulate); the 8 channel groups are summed with 3 xor-shuffles.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float dot8(half8_t v, half8_t w, float acc)
{
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        half2_t a = {v[2 * e], v[2 * e + 1]}, b = {w[2 * e], w[2 * e + 1]};
        acc = __builtin_amdgcn_fdot2(a, b, acc, false);
    }
    return acc;
}

template <int TAPS>
__global__ __launch_bounds__(256) void tail_kernel(TailArgs a)
{
    constexpr int KD = (TAPS == 9) ? 3 : 1, PAD = (TAPS == 9) ? 1 : 0;
    const int cgi = threadIdx.x & 7, pc = threadIdx.x >> 3;
    const int nbx = (a.W + 31) / 32, nby = (a.H + 7) / 8;
    int blk = blockIdx.x;
    const int bx = blk % nbx; blk /= nbx;
    const int by = blk % nby;
    const int b = blk / nby;
    const int x = bx * 32 + pc, y0 = by * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
        const half_t* in = which ? a.in1 : a.in0;
        if (!in) continue;
        const half_t* wsrc = (which ? a.w1 : a.w0) + cgi * 8;
        half8_t w[TAPS];
#pragma unroll
        for (int t = 0; t < TAPS; ++t) w[t] = *(const half8_t*)(wsrc + t * 64);
        const half_t* wlo_src = (which ? a.w1_lo : a.w0_lo);
        const half_t* in_lo = which ? a.in1_lo : a.in0_lo;
#pragma unroll
        for (int r = 0; r < 8 + 2 * PAD; ++r) {
            const int yy = y0 + r - PAD;
            const bool rok = (yy >= 0) && (yy < a.H);
#pragma unroll
            for (int dx = 0; dx < KD; ++dx) {
                const int xx = x + dx - PAD;
                const bool ok = rok && xx >= 0 && xx < a.W;
                half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
                const long long off = ((long long)(b * a.H + yy) * a.W + xx) * 64 + cgi * 8;
                if (ok) v = *(const half8_t*)(in + off);
#pragma unroll
                for (int dy = 0; dy < KD; ++dy) {
                    const int o = r - dy;
                    if (o >= 0 && o < 8) acc[o] = dot8(v, w[dy * KD + dx], acc[o]);
                }
                if (in_lo) {   // FP16X3: (a_hi + a_lo/2048) * (w_hi + w_lo/2048), cross terms kept
                    half8_t vl = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (ok) vl = *(const half8_t*)(in_lo + off);
#pragma unroll
                    for (int dy = 0; dy < KD; ++dy) {
                        const int o = r - dy;
                        if (o >= 0 && o < 8) {
                            const half8_t wl = *(const half8_t*)(wlo_src + cgi * 8 + (dy * KD + dx) * 64);
                            float c = dot8(vl, w[dy * KD + dx], 0.f);
                            c = dot8(v, wl, c);
                            acc[o] += c * 0.00048828125f;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        float v = acc[o];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        acc[o] = v;
    }
    if (cgi == 0 && x < a.W) {
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            const int y = y0 + o;
            if (y >= a.H) break;
            float v = acc[o];
            if (a.skip) {
                const long long so = (a.skip_off ? a.skip_off[b] : (long long)b * a.skip_sB) + y * a.skip_sH + x * a.skip_sW;
                v += (a.skip_dtype == MOE_F16) ? ld1<half_t>(a.skip, so) : ld1<float>(a.skip, so);
            }
            const long long yo = (a.y_off ? a.y_off[b] : (long long)b * a.H * a.W) + (long long)y * a.W + x;
            if (a.y_dtype == MOE_F16) ((half_t*)a.y)[yo] = (half_t)v;
            else ((float*)a.y)[yo] = v;
        }
    }
}

// 3x3 tail, second form: every 16 bytes of the input are loaded ONCE.  Block = 32 columns (30 outputs + 1 halo column per side) x 8
// output rows; thread (column c, channel group cg) walks the 10 rows of its own column and accumulates, per output row, the three
// per-dx partial sums S[dx] of its column (they belong to the outputs at columns c+1, c, c-1 for dx = 0, 1, 2).  The partials go through LDS once:
// thread (c, o = cg) then adds S0[c-1] + S1[c] + S2[c+1] over the eight channel groups and stores one pixel.
// (The first form loads every element three times -- once per dx -- and ran at 1.3-2 TB/s.)
// LO (the hi + lo form: FP16X3, MIXED on NetDN): the operands are put together in fp32 -- x = hi + lo 2^-11 and w = w_hi + w_lo 2^-11 are exact there (22 significant bits) --
// and every tap is four packed fp32 FMAs on two accumulators (even / odd channel pairs), instead of three v_dot2 products (hi hi, lo hi, hi lo) of four instructions each: the
// kernel was held by its VALU stream (profiles/r06/zf_pmc_tail3_sedn.txt: 130 VALU instructions per loaded pixel-group pair), and the lo lo term comes with it.
// LO: 0 fp16 operands, 1 hi + lo pairs, 2 hi + lo pairs with in1's low part as fp8 words (TailArgs::in1_lo8)
template <int LO>
__global__ __launch_bounds__(256) void tail3_kernel(TailArgs a)
{
    typedef float f2v __attribute__((ext_vector_type(2)));
    __shared__ float part[3][8][32][9];          // [dx][output row][column][channel group], padded: 27.6 KB
    const int cgi = threadIdx.x & 7, pc = threadIdx.x >> 3;
    const int nbx = (a.W + 29) / 30, nby = (a.H + 7) / 8;
    int blk = blockIdx.x;
    const int bx = blk % nbx; blk /= nbx;
    const int by = blk % nby;
    const int b = blk / nby;
    const int x = bx * 30 - 1 + pc, y0 = by * 8;
    const bool xin = x >= 0 && x < a.W;
    float S[3][8];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int o = 0; o < 8; ++o) S[dx][o] = 0.f;
    if constexpr (LO != 0) {
        f2v S2[3][8];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int o = 0; o < 8; ++o) S2[dx][o] = f2v{0.f, 0.f};
        const int xc = min(max(x, 0), a.W - 1);
        auto branch = [&](auto which_c) {
            constexpr int which = decltype(which_c)::value;
            constexpr bool lo8 = which == 1 && LO == 2;
            const half_t* in = which ? a.in1 : a.in0;
            if (!in) return;
            const half_t* wsrc = (which ? a.w1 : a.w0) + cgi * 8;
            const half_t* wlo_src = (which ? a.w1_lo : a.w0_lo) + cgi * 8;
            const half_t* in_lo = which ? a.in1_lo : a.in0_lo;
            f2v w[9][4];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const half8_t wh = *(const half8_t*)(wsrc + t * 64), wl = *(const half8_t*)(wlo_src + t * 64);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    w[t][e] = f2v{(float)wh[2 * e] + (float)wl[2 * e] * 0.00048828125f, (float)wh[2 * e + 1] + (float)wl[2 * e + 1] * 0.00048828125f};
            }
            // rows are fetched two ahead of their use into a ring of three, from clamped addresses (no branch around a load: the values of a pixel outside the image are
            // zeroed at their use); the scheduler left to itself hoists all twenty loads: 256 VGPRs, one wave per SIMD
            half8_t vh[3];
            uint4 vq[3];
            auto fetch = [&](int r, half8_t& h, uint4& q) {
                const int yy = min(max(y0 + r - 1, 0), a.H - 1);
                const long long off = ((long long)(b * a.H + yy) * a.W + xc) * 64 + cgi * 8;
                h = *(const half8_t*)(in + off);
                if constexpr (lo8) { const uint2 t = *(const uint2*)((const unsigned char*)in_lo + off); q = make_uint4(t.x, t.y, 0, 0); }
                else q = *(const uint4*)(in_lo + off);
            };
            fetch(0, vh[0], vq[0]);
            fetch(1, vh[1], vq[1]);
#pragma unroll
            for (int r = 0; r < 10; ++r) {
                if (r + 2 < 10) fetch(r + 2, vh[(r + 2) % 3], vq[(r + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);
                const int yy = y0 + r - 1;
                const float keep = (xin && yy >= 0 && yy < a.H) ? 1.f : 0.f;
                const half8_t v = vh[r % 3];
                const uint4 q = vq[r % 3];
                f2v xv[4];
                if constexpr (lo8) {      // the stem's low part as fp8 words of lo / 4 (bytes in channel order)
                    const f2v l0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)q.x, false), l1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)q.x, true);
                    const f2v l2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)q.y, false), l3 = __builtin_amdgcn_cvt_pk_f32_fp8((int)q.y, true);
                    const f2v ll[4] = {l0, l1, l2, l3};
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[e] = f2v{(float)v[2 * e] + ll[e][0] * 0.001953125f, (float)v[2 * e + 1] + ll[e][1] * 0.001953125f} * keep;
                } else {
                    const half8_t vl = __builtin_bit_cast(half8_t, q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[e] = f2v{(float)v[2 * e] + (float)vl[2 * e] * 0.00048828125f, (float)v[2 * e + 1] + (float)vl[2 * e + 1] * 0.00048828125f} * keep;
                }
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int o = r - dy;
                    if (o < 0 || o >= 8) continue;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                        for (int e = 0; e < 4; ++e) S2[dx][o] = __builtin_elementwise_fma(xv[e], w[dy * 3 + dx][e], S2[dx][o]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        branch(std::integral_constant<int, 0>{});
        branch(std::integral_constant<int, 1>{});
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int o = 0; o < 8; ++o) S[dx][o] = S2[dx][o][0] + S2[dx][o][1];
    } else {
#pragma unroll 1
        for (int which = 0; which < 2; ++which) {
            const half_t* in = which ? a.in1 : a.in0;
            if (!in) continue;
            const half_t* wsrc = (which ? a.w1 : a.w0) + cgi * 8;
            half8_t w[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) w[t] = *(const half8_t*)(wsrc + t * 64);
#pragma unroll
            for (int r = 0; r < 10; ++r) {
                const int yy = y0 + r - 1;
                const bool ok = xin && yy >= 0 && yy < a.H;
                const long long off = ((long long)(b * a.H + yy) * a.W + x) * 64 + cgi * 8;
                half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) v = *(const half8_t*)(in + off);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int o = r - dy;
                    if (o < 0 || o >= 8) continue;
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) S[dx][o] = dot8(v, w[dy * 3 + dx], S[dx][o]);
                }
            }
        }
    }
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int o = 0; o < 8; ++o) part[dx][o][pc][cgi] = S[dx][o];
    __syncthreads();
    const int o = cgi, y = y0 + o;               // second role of the thread: output pixel (column pc, row o)
    if (pc < 1 || pc > 30 || x >= a.W || y >= a.H) return;
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) v += part[0][o][pc - 1][g] + part[1][o][pc][g] + part[2][o][pc + 1][g];   // input column = output column + dx - 1
    if (a.skip) {
        const long long so = (a.skip_off ? a.skip_off[b] : (long long)b * a.skip_sB) + y * a.skip_sH + x * a.skip_sW;
        v += (a.skip_dtype == MOE_F16) ? ld1<half_t>(a.skip, so) : ld1<float>(a.skip, so);
    }
    const long long yo = (a.y_off ? a.y_off[b] : (long long)b * a.H * a.W) + (long long)y * a.W + x;
    if (a.y_dtype == MOE_F16) ((half_t*)a.y)[yo] = (half_t)v;
    else ((float*)a.y)[yo] = v;
}

// ---------------------------------------------------------------------------------------------------
// Fused-tail gather: the last upsampler conv wrote, per HR pixel q and tap t, G[t][q] = sum_c Wt[t][c] * act[c][q]; the 3x3 tail
// conv (zero padded) is then  y[p] = sum_t G[t][p + off_t]  over the in-image neighbours, summed over both branches.
// G is stored as [tap][phase (Y%r)*r + X%r][b][Y/r][X/r] (conv3x3_sp.hip: one workgroup produces one pixel-shuffle phase).
// ---------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(256) void tapsum_kernel(TapSumArgs a)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y, b = blockIdx.z;
    if (x >= a.W) return;
    const int h = a.H / R, w = a.W / R;
    const long long lrplane = (long long)a.B * h * w;
    float acc = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int yy = y + dy - 1, xx = x + dx - 1;
            if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
                const int ph = (yy % R) * R + (xx % R);
                const long long o = ((long long)(dy * 3 + dx) * (R * R) + ph) * lrplane + ((long long)b * h + yy / R) * w + xx / R;
                acc += a.t0[o];
                if (a.t1) acc += a.t1[o];
            }
        }
    const long long yo = (a.y_off ? a.y_off[b] : (long long)b * a.H * a.W) + (long long)y * a.W + x;
    if (a.y_dtype == MOE_F16) ((half_t*)a.y)[yo] = (half_t)acc;
    else ((float*)a.y)[yo] = acc;
}

// R = 2 specialisation: one thread makes 8 consecutive outputs of one HR row from 16-byte loads.  With x0 = 4*xq, the outputs
// X = 2*x0 + e (e = 0..7) of tap column dx read HR columns 2*x0 + e + dx - 1: for each column phase sj that is the aligned
// quad [x0, x0+4) of the phase plane, shifted by one element for (dx = 0, sj = 1) and (dx = 2, sj = 0) -- those two take one
// extra scalar load.  Same summation order per output as the generic kernel (taps ascending, branch 0 then 1).
__global__ __launch_bounds__(256) void tapsum2_kernel(TapSumArgs a)
{
    const int h = a.H >> 1, w = a.W >> 1, nq = w >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.B * a.H * nq) return;
    const int xq = (int)(idx % nq);
    const int y = (int)((idx / nq) % a.H), b = (int)(idx / ((long long)nq * a.H));
    const int x0 = xq * 4;
    const long long lrplane = (long long)a.B * h * w;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + dy - 1;
        if (yy < 0 || yy >= a.H) continue;
        const int si = yy & 1;
        const long long row = ((long long)b * h + (yy >> 1)) * w + x0;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const long long p0 = ((long long)(dy * 3 + dx) * 4 + si * 2) * lrplane + row, p1 = p0 + lrplane;   // sj = 0 / 1 planes
#pragma unroll
            for (int br = 0; br < 2; ++br) {
                const float* t = br ? a.t1 : a.t0;
                if (!t) continue;
                const float4 q0 = *(const float4*)(t + p0), q1 = *(const float4*)(t + p1);
                if (dx == 1) {
                    acc[0] += q0.x; acc[1] += q1.x; acc[2] += q0.y; acc[3] += q1.y;
                    acc[4] += q0.z; acc[5] += q1.z; acc[6] += q0.w; acc[7] += q1.w;
                } else if (dx == 0) {      // even outputs read plane sj=1 one element to the left
                    const float l = x0 > 0 ? t[p1 - 1] : 0.f;
                    acc[0] += l;    acc[1] += q0.x; acc[2] += q1.x; acc[3] += q0.y;
                    acc[4] += q1.y; acc[5] += q0.z; acc[6] += q1.z; acc[7] += q0.w;
                } else {                   // odd outputs read plane sj=0 one element to the right
                    const float rr = x0 + 4 < w ? t[p0 + 4] : 0.f;
                    acc[0] += q1.x; acc[1] += q0.y; acc[2] += q1.y; acc[3] += q0.z;
                    acc[4] += q1.z; acc[5] += q0.w; acc[6] += q1.w; acc[7] += rr;
                }
            }
        }
    }
    const long long yo = (a.y_off ? a.y_off[b] : (long long)b * a.H * a.W) + (long long)y * a.W + 2 * x0;
    if (a.y_dtype == MOE_F16) {
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)acc[e];
        *(half8_t*)((half_t*)a.y + yo) = o;
    } else {
        float* yp = (float*)a.y + yo;
        *(float4*)yp = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *(float4*)(yp + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

// Gather of the "phase-class sums" form (conv3x3_rw.hip EPI 3 / 7; the arithmetic is stated in tests/tailsum_model.py): an output pixel
// (2y' + i', 2x' + j') is the sum over the four phases (i, j) and both branches of S_(i,j)[i' ^ i][j' ^ j][y'][x'] -- 32 bytes read per output
// pixel instead of 72 -- plus, on patch edges, the aprons of the neighbouring patch in the fixed order S + RA + CA + CO.  One thread makes
// the 8 consecutive outputs over 4 conv-input columns: per phase and branch two 16-byte loads (the class of the even and of the odd outputs).
template <bool VEC>
__global__ __launch_bounds__(256) void tapsum4_kernel(TapSumArgs a)
{
    const int h = a.H >> 1, w = a.W >> 1, nq = w >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)a.B * a.H * nq) return;
    const int xq = (int)(idx % nq);
    const int Y = (int)((idx / nq) % a.H), b = (int)(idx / ((long long)nq * a.H));
    const int ip = Y & 1, yl = Y >> 1, x0 = xq * 4;
    const TailSumLayout L = tailsum_layout(a.B, h, w);
    const unsigned plane = (unsigned)(a.B * h * w);
    const unsigned row = (unsigned)((b * h + yl) * w + x0);
    const int pyi = yl / kTileH, rho = yl - pyi * kTileH, pxi = x0 / kTileW, chi = x0 - pxi * kTileW;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int br = 0; br < 2; ++br) {
        const float* t = br ? a.t1 : a.t0;
        if (!t) continue;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int i = ph >> 1, j = ph & 1, ci = ip ^ i;
            const int sv = i == 0 ? 1 : -1, sh = j == 0 ? 1 : -1;
            const float* S = t + L.S + (unsigned)(ph * 4 + 2 * ci) * plane + row;
            float4 e = *(const float4*)(S + (unsigned)j * plane);            // class (ci, j' ^ j) of the even outputs (j' = 0)
            float4 o = *(const float4*)(S + (unsigned)(1 ^ j) * plane);      // ... of the odd outputs
            const int pyn = pyi + sv, pxn = pxi + sh;
            const bool rowfix = ci == 1 && rho == (sv == 1 ? kTileH - 1 : 0) && pyn >= 0 && pyn < L.py;
            const bool colfix = chi == (sh == 1 ? kTileW - 4 : 0) && pxn >= 0 && pxn < L.px;      // the quad that holds the patch's edge column
            if (rowfix) {
                const float* RA = t + L.RA + (unsigned)(ph * 2) * (unsigned)(a.B * L.py * w) + (unsigned)((b * L.py + pyn) * w + x0);
                const float4 r0 = *(const float4*)(RA + (unsigned)j * (unsigned)(a.B * L.py * w));
                const float4 r1 = *(const float4*)(RA + (unsigned)(1 ^ j) * (unsigned)(a.B * L.py * w));
                e.x += r0.x; e.y += r0.y; e.z += r0.z; e.w += r0.w;
                o.x += r1.x; o.y += r1.y; o.z += r1.z; o.w += r1.w;
            }
            if (colfix) {      // the class with cj = 1 is the odd outputs' for j = 0, the even outputs' for j = 1; its edge element is the quad's last (sh = +1) / first
                float fix = t[L.CA + (unsigned)(ph * 2 + ci) * (unsigned)(a.B * h * L.px) + (unsigned)((b * h + yl) * L.px + pxn)];
                float4& q = j == 0 ? o : e;
                float& el = sh == 1 ? q.w : q.x;
                el += fix;
                if (rowfix) el += t[L.CO + (unsigned)ph * (unsigned)(a.B * L.py * L.px) + (unsigned)((b * L.py + pyn) * L.px + pxn)];
            }
            acc[0] += e.x; acc[1] += o.x; acc[2] += e.y; acc[3] += o.y;
            acc[4] += e.z; acc[5] += o.z; acc[6] += e.w; acc[7] += o.w;
        }
    }
    const long long yo = (a.y_off ? a.y_off[b] : (long long)b * a.H * a.W) + (long long)Y * a.W + 2 * x0;
    if (a.y_dtype == MOE_F16) {
        if (VEC) {
            half8_t v;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (half_t)acc[k];
            *(half8_t*)((half_t*)a.y + yo) = v;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) ((half_t*)a.y)[yo + k] = (half_t)acc[k];
        }
    } else {
        float* yp = (float*)a.y + yo;
        if (VEC) {
            *(float4*)yp = make_float4(acc[0], acc[1], acc[2], acc[3]);
            *(float4*)(yp + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) yp[k] = acc[k];
        }
    }
}

// Fused 1x1 tail (lite): y = sum of the four partial planes (two branches x two 32-channel halves), fixed order.
__global__ __launch_bounds__(256) void tail1sum_kernel(Tail1SumArgs a)
{
    const long long hw = (long long)a.H * a.W, plane = (long long)a.B * hw;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= plane) return;
    const int b = (int)(idx / hw);
    const long long p = idx - (long long)b * hw;
    float v = a.p0[idx];
    if (a.nparts == 2) v += a.p0[plane + idx];
    if (a.p1) {
        v += a.p1[idx];
        if (a.nparts == 2) v += a.p1[plane + idx];
    } else if (a.lut) {      // the U branch of a pointwise net is a function of the input pixel's value and the HR pixel's phase
        const int Y = (int)(p / a.W), X = (int)(p - (long long)Y * a.W);
        const int ly = Y / a.r, lx = X / a.r;
        const unsigned bits = ((const unsigned short*)a.x)[(a.x_off ? a.x_off[b] : (long long)b * a.sB) + (long long)ly * a.sH + (long long)lx * a.sW];
        v += a.lut[((long long)(bits >> 8) * a.r + (Y - ly * a.r)) * (256 * a.r) + (bits & 255) * a.r + (X - lx * a.r)];
    }
    const long long yo = (a.y_off ? a.y_off[b] : (long long)b * hw) + p;
    if (a.y_dtype == MOE_F16) ((half_t*)a.y)[yo] = (half_t)v;
    else ((float*)a.y)[yo] = v;
}

// the table form (Tail1SumArgs::lut), four consecutive outputs of a row per thread (W % 4 == 0, one part per branch, output rows 8- / 16-byte aligned): r = 4, 8 -- the four
// lie in ONE input pixel, their table entries are 16 contiguous bytes; r = 2 -- two input pixels, 8 bytes each.  The same additions as tail1sum_kernel: the same bits.
__global__ __launch_bounds__(256) void tail1sum_lut4_kernel(Tail1SumArgs a)
{
    const int wq = a.W >> 2;
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= (long long)a.B * a.H * wq) return;
    const int xq = (int)(q % wq);
    const long long t = q / wq;
    const int Y = (int)(t % a.H), b = (int)(t / a.H);
    const int X0 = xq * 4, ly = Y / a.r, py = Y - ly * a.r;
    const long long p = (long long)Y * a.W + X0, idx = (long long)b * a.H * a.W + p;
    const float4 pv = *(const float4*)(a.p0 + idx);
    const unsigned short* xrow = (const unsigned short*)a.x + (a.x_off ? a.x_off[b] : (long long)b * a.sB) + (long long)ly * a.sH;
    float v[4] = {pv.x, pv.y, pv.z, pv.w};
    if (a.r >= 4) {
        const int lx = X0 / a.r, px = X0 - lx * a.r;
        const unsigned bits = xrow[(long long)lx * a.sW];
        const float4 u = *(const float4*)(a.lut + ((long long)(bits >> 8) * a.r + py) * (256 * a.r) + (bits & 255) * a.r + px);
        v[0] += u.x; v[1] += u.y; v[2] += u.z; v[3] += u.w;
    } else {
        const int lx = X0 >> 1;
        const unsigned b0 = xrow[(long long)lx * a.sW], b1 = xrow[(long long)(lx + 1) * a.sW];
        const float2 u0 = *(const float2*)(a.lut + ((long long)(b0 >> 8) * 2 + py) * 512 + (b0 & 255) * 2);
        const float2 u1 = *(const float2*)(a.lut + ((long long)(b1 >> 8) * 2 + py) * 512 + (b1 & 255) * 2);
        v[0] += u0.x; v[1] += u0.y; v[2] += u1.x; v[3] += u1.y;
    }
    const long long yo = (a.y_off ? a.y_off[b] : (long long)b * a.H * a.W) + p;
    if (a.y_dtype == MOE_F16) {
        half4_t h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (half_t)v[e];
        *(half4_t*)((half_t*)a.y + yo) = h;
    } else *(float4*)((float*)a.y + yo) = make_float4(v[0], v[1], v[2], v[3]);
}

// ---------------------------------------------------------------------------------------------------
// Global average pool partial sums (AdaptiveAvgPool2d(1): models.py:190,274): in [B][HW][C] -> [B][nslab][C].
// Deterministic two-stage reduction (the second stage lives in the gate kernels).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_partial_kernel(const half_t* in, const half_t* in_lo, float* partial, long long HW, int C, int nslab)
{
    __shared__ float red[256 * 8];
    const int b = blockIdx.y, slab = blockIdx.x;
    const int ngrp = C / 8;                    // 8 or 32 channel groups
    const int cg = threadIdx.x % ngrp, pl = threadIdx.x / ngrp, npl = 256 / ngrp;
    const long long per = (HW + nslab - 1) / nslab;
    const long long p0 = slab * per, p1 = (p0 + per < HW) ? p0 + per : HW;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (long long p = p0 + pl; p < p1; p += npl) {
        const long long off = ((long long)b * HW + p) * C + cg * 8;
        const half8_t v = *(const half8_t*)(in + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
        if (in_lo) {
            const half8_t l = *(const half8_t*)(in_lo + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += (float)l[e] * 0.00048828125f;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = acc[e];
    __syncthreads();
    if (pl == 0) {
        for (int k = 1; k < npl; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += red[(k * ngrp + cg) * 8 + e];
#pragma unroll
        for (int e = 0; e < 8; ++e) partial[((long long)b * nslab + slab) * C + cg * 8 + e] = acc[e];
    }
}

// ---------------------------------------------------------------------------------------------------
// SEDN squeeze-excite gate (models.py:198-208): g = sigmoid(W_up * lrelu(W_down * mean)); the gate multiplies the
// 256 input channels of the following 1x1 `trans` conv, so it is folded into a per-plane copy of trans's packed
// weights: W'[b][o][c] = W[o][c] * g[b][c]  (fp32 product, then fp16 -- the activations themselves stay untouched).
// kSeSplit blocks per plane: each recomputes the (tiny) gate and rescales its share of the weights.  The 256->16 squeeze runs
// on all 256 threads (16 per hidden unit, fixed-order tree sum) -- as 16 serial 256-term dot products it was 45 us per launch.
// ---------------------------------------------------------------------------------------------------
constexpr int kSeSplit = 8;
__global__ __launch_bounds__(256) void sedn_se_kernel(SednSeArgs a)
{
    __shared__ float mean[256];
    __shared__ float part[256];
    __shared__ float hid[16];
    __shared__ float gate[256];
    const int b = blockIdx.x, t = threadIdx.x;
    float s = 0.f;
    for (int k = 0; k < a.nslab; ++k) s += a.partial[((long long)b * a.nslab + k) * 256 + t];
    mean[t] = s / (float)a.HW;
    __syncthreads();
    {
        const int k = t >> 4, q = t & 15;            // hidden unit k, channels 16q .. 16q+15
        float h = 0.f;
#pragma unroll
        for (int c = 0; c < 16; ++c) h += a.w_down[k * 256 + q * 16 + c] * mean[q * 16 + c];
        part[t] = h;
    }
    __syncthreads();
    if (t < 16) {
        float h = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) h += part[t * 16 + q];
        hid[t] = prelu(h, 0.2f);
    }
    __syncthreads();
    {
        float u = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) u += a.w_up[t * 16 + k] * hid[k];
        gate[t] = 1.f / (1.f + __expf(-u));
    }
    __syncthreads();
    const int part_i = blockIdx.y;
    if (a.nfrag < 0) {   // MOE_PREC_DEBUG_DIRECT: plain fp32 [cout][256] weights, -nfrag elements
        const int total = -a.nfrag;
        float* o = (float*)a.trans_out;
        for (int i = part_i * 256 + t; i < total; i += 256 * kSeSplit) o[(long long)b * total + i] = a.trans_pk32[i] * gate[i & 255];
        return;
    }
    // packed fragment element idx -> input channel:  frag f = (seg*4 + ks)*2 + nblk, lane l, e
    const int total = a.nfrag * 512;
    for (int i = part_i * 256 + t; i < total; i += 256 * kSeSplit) {
        const int e = i & 7, l = (i >> 3) & 63, f = i >> 9;
        const int ks = (f >> 1) & 3, seg = f >> 3;
        const int cin = seg * 64 + ks * 16 + 8 * (l >> 5) + e;
        const float v = a.trans_pk32[i] * gate[cin];
        const half_t hv = (half_t)v;
        a.trans_out[(long long)b * total + i] = hv;
        if (a.trans_out_lo) a.trans_out_lo[(long long)b * total + i] = (half_t)((v - (float)hv) * 2048.f);
    }
}

// ---------------------------------------------------------------------------------------------------
// SEDN fused block tail.  Between rblock.4 (3x3 64->256, no activation) and the gated 1x1 `trans` (256->64) everything is linear:
//     trans(g * conv256(x)) = conv(W_eff, x),   W_eff[b] = W_t diag(g_b) W_256      (64 x 64 x 9, one set per plane)
// and the pooled mean the gate needs is W_256 applied to the nine shifted-window sums of x (total minus border rows/columns,
// zero padding).  The 256-channel tensor, the pooling pass over it and 4.4x of these two convs' FLOPs disappear.
//   1. sedn_xsum:  per plane and channel the total and the first/last row/column sums of x (two-stage, fixed order)
//   2. sedn_fmean: shifted sums -> channel means of the (never formed) 256-channel tensor
//   3. sedn_weff:  squeeze-excite gate g from the means, then W_eff = (W_t * g) W_256 in fp32, stored as fp16 MFMA A fragments for conv3x3_sp (EPI 6, per-plane weights)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sedn_xsum_kernel(SednFuseArgs a)
{
    __shared__ float red[5][32][64];
    const int b = blockIdx.y, slab = blockIdx.x;
    const int cg = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const long long HW = (long long)a.H * a.W;
    const long long per = (HW + a.nslab - 1) / a.nslab;
    const long long p0 = slab * per, p1 = (p0 + per < HW) ? p0 + per : HW;
    float acc[5][8];
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[c][e] = 0.f;
    // with the totals already formed by the producing conv only the border pixels are visited: index i in [0, 2W + 2H) = first row,
    // last row, first column, last column (corner pixels appear in a row list and in a column list, as they must)
    const long long nb = a.pooled ? 2ll * a.W + 2ll * a.H : HW;
    const long long bper = (nb + a.nslab - 1) / a.nslab;
    const long long q0 = a.pooled ? slab * bper : p0, q1 = a.pooled ? ((q0 + bper < nb) ? q0 + bper : nb) : p1;
    for (long long i = q0 + pl; i < q1; i += 32) {
        int y, x;
        float f0 = 1.f, fr0, frl, fc0, fcl;
        if (a.pooled) {
            const long long w2 = 2ll * a.W;
            if (i < w2) { y = i < a.W ? 0 : a.H - 1; x = (int)(i < a.W ? i : i - a.W); fr0 = i < a.W ? 1.f : 0.f; frl = 1.f - fr0; fc0 = fcl = 0.f; }
            else { const long long k = i - w2; x = k < a.H ? 0 : a.W - 1; y = (int)(k < a.H ? k : k - a.H); fc0 = k < a.H ? 1.f : 0.f; fcl = 1.f - fc0; fr0 = frl = 0.f; }
            f0 = 0.f;
        } else {
            y = (int)(i / a.W); x = (int)(i - (long long)y * a.W);
            fr0 = y == 0 ? 1.f : 0.f; frl = y == a.H - 1 ? 1.f : 0.f; fc0 = x == 0 ? 1.f : 0.f; fcl = x == a.W - 1 ? 1.f : 0.f;
        }
        const long long p = (long long)y * a.W + x;
        const half8_t v = *(const half8_t*)(a.x + ((long long)b * HW + p) * 64 + cg * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float t = (float)v[e];
            acc[0][e] += t * f0; acc[1][e] += t * fr0; acc[2][e] += t * frl; acc[3][e] += t * fc0; acc[4][e] += t * fcl;
        }
    }
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[c][pl][cg * 8 + e] = acc[c][e];
    __syncthreads();
    for (int i = threadIdx.x; i < 5 * 64; i += 256) {
        const int c = i >> 6, ch = i & 63;
        float t = 0.f;
        for (int k = 0; k < 32; ++k) t += red[c][k][ch];
        a.partial[(((long long)b * a.nslab + slab) * 5 + c) * 64 + ch] = t;
    }
}

constexpr int kMeanSplit = 4;          // blocks per plane in sedn_fmean: 64 of the 256 channel means each
__global__ __launch_bounds__(256) void sedn_fmean_kernel(SednFuseArgs a)
{
    __shared__ float sums[5][64];
    __shared__ float corner[4][64];      // x[0][0], x[0][W-1], x[H-1][0], x[H-1][W-1]
    __shared__ float sh[576];            // shifted-window sums, k = tap*64 + ci
    __shared__ float red[4][64];
    const int b = blockIdx.x, t = threadIdx.x;
    const long long HW = (long long)a.H * a.W;
    if (a.pooled) {
        // totals from the producing conv: [pooled_slabs][64] per plane (hundreds of slabs): thread = (slab class t / 16, channels 4 (t % 16) ..),
        // 16-byte loads, eight in flight; then a fixed-order sum of the 16 classes
        __shared__ float4_t pred[16][16];
        const int cls = t >> 4, c4 = t & 15;
        float4_t v = {0.f, 0.f, 0.f, 0.f};
        const float4_t* src = (const float4_t*)(a.pooled + (long long)b * a.pooled_slabs * 64) + c4;
#pragma unroll 8
        for (int k = cls; k < a.pooled_count; k += 16) v = v + src[(long long)k * 16];
        pred[cls][c4] = v;
        __syncthreads();
        if (t < 64) {
            float u = 0.f;
            for (int k = 0; k < 16; ++k) u += pred[k][t >> 2][t & 3];
            sums[0][t] = u;
        }
    }
    if (a.pooled) {
        // round 6: with the totals formed by the producing conv only the border is left of sedn_xsum's work -- 2 W + 2 H pixels, visited HERE (by each of the plane's four
        // blocks) instead of by a launch of its own: index i in [0, 2W + 2H) = first row, last row, first column, last column (corner pixels appear in a row list and in a
        // column list, as they must); thread = (pixel lane t / 8 of 32, channel group t % 8), then a fixed-order sum over the 32 lanes
        __shared__ float bred[4][32][64];
        const int cg = t & 7, pl = t >> 3;
        float acc[4][8];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[c][e] = 0.f;
        const int nb = 2 * a.W + 2 * a.H;
#pragma unroll 8
        for (int i = pl; i < nb; i += 32) {
            int y, x, c;
            if (i < 2 * a.W) { c = i < a.W ? 0 : 1; y = c ? a.H - 1 : 0; x = c ? i - a.W : i; }
            else { const int k = i - 2 * a.W; c = k < a.H ? 2 : 3; x = c == 3 ? a.W - 1 : 0; y = c == 3 ? k - a.H : k; }
            const half8_t v = *(const half8_t*)(a.x + ((long long)b * HW + (long long)y * a.W + x) * 64 + cg * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float u = (float)v[e];
                acc[0][e] += c == 0 ? u : 0.f; acc[1][e] += c == 1 ? u : 0.f; acc[2][e] += c == 2 ? u : 0.f; acc[3][e] += c == 3 ? u : 0.f;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) bred[c][pl][cg * 8 + e] = acc[c][e];
        __syncthreads();
        {
            const int c = t >> 6, ch = t & 63;
            float v = 0.f;
#pragma unroll 8
            for (int k = 0; k < 32; ++k) v += bred[c][k][ch];
            sums[1 + c][ch] = v;
        }
    } else {
        for (int i = t; i < 5 * 64; i += 256) {
            const int c = i >> 6, ch = i & 63;
            float v = 0.f;
#pragma unroll 8
            for (int k = 0; k < a.nslab; ++k) v += a.partial[(((long long)b * a.nslab + k) * 5 + c) * 64 + ch];
            sums[c][ch] = v;
        }
    }
    {
        const int q = t >> 6, ch = t & 63;
        const long long p = (q & 2 ? (long long)(a.H - 1) * a.W : 0) + (q & 1 ? a.W - 1 : 0);
        corner[q][ch] = (float)a.x[((long long)b * HW + p) * 64 + ch];
    }
    __syncthreads();
    for (int k = t; k < 576; k += 256) {
        const int tap = k >> 6, ci = k & 63, dy = tap / 3, dx = tap % 3;
        float v = sums[0][ci];
        if (dy == 0) v -= sums[2][ci];           // rows y-1: the last row never contributes
        if (dy == 2) v -= sums[1][ci];           // rows y+1: the first row never contributes
        if (dx == 0) v -= sums[4][ci];
        if (dx == 2) v -= sums[3][ci];
        if (dy == 0 && dx == 0) v += corner[3][ci];
        if (dy == 0 && dx == 2) v += corner[2][ci];
        if (dy == 2 && dx == 0) v += corner[1][ci];
        if (dy == 2 && dx == 2) v += corner[0][ci];
        sh[k] = v;
    }
    __syncthreads();
    {   // thread (part q of 4, channel m): 144 of the 576 terms, then a fixed-order sum of the four parts
        const int q = t >> 6, m = blockIdx.y * 64 + (t & 63);
        float m0 = 0.f, m1 = 0.f;
#pragma unroll 8
        for (int k = q * 144; k < q * 144 + 144; k += 2) {
            m0 += a.w256t[(k + 0) * 256 + m] * sh[k + 0];
            m1 += a.w256t[(k + 1) * 256 + m] * sh[k + 1];
        }
        red[q][t & 63] = m0 + m1;
    }
    __syncthreads();
    if (t < 64) a.gate[b * 256 + blockIdx.y * 64 + t] = ((red[0][t] + red[1][t]) + (red[2][t] + red[3][t])) / (float)HW;   // (the MEAN; the gate is formed in sedn_weff)
}

// W_eff[b][co][k] = sum_m (W_t[co][m] g[b][m]) W_256[m][k]: 64 x 576 x 256 per plane.  Block = all 64 co x 64 k (one tap); grid 9 x B.  (Round 4: 32 co x 64 k, 2 x 4 per
// thread on VALU FMAs, LDS-bound at 36 us a launch; round 5: 4 x 4 per thread, 27 us.)
__global__ __launch_bounds__(256) void sedn_weff_kernel(SednFuseArgs a)
{
    __shared__ __attribute__((aligned(16))) half_t Cs[64][72];        // the block's result [co][ci] (+8: rows stay 16-byte aligned)
    __shared__ __attribute__((aligned(16))) float gate[256];
    __shared__ float mean[256], part[256], hid[16];
    const int tap = blockIdx.x, b = blockIdx.y;
    {   // squeeze-excite gate of this plane from its channel means (16 KFLOP: every block redoes it rather than wait for a launch)
        const int t = threadIdx.x;
        mean[t] = a.gate[b * 256 + t];
        __syncthreads();
        {
            const int k = t >> 4, q = t & 15;
            float h = 0.f;
#pragma unroll
            for (int c = 0; c < 16; ++c) h += a.w_down[k * 256 + q * 16 + c] * mean[q * 16 + c];
            part[t] = h;
        }
        __syncthreads();
        if (t < 16) {
            float h = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) h += part[t * 16 + q];
            hid[t] = prelu(h, 0.2f);
        }
        __syncthreads();
        float u = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) u += a.w_up[t * 16 + k] * hid[k];
        gate[t] = 1.f / (1.f + __expf(-u));
        __syncthreads();
    }
    // round 6: the 64 x 64 x 256 product of a block on fp32 MFMAs (v_mfma_f32_32x32x2_f32: wave (cob, cib) owns a 32 x 32 quadrant, 128 MFMAs), operands straight from
    // global memory -- no staging, no barrier in the loop.  The k order of an MFMA is free as long as A and B agree: lane (ln, lk) takes the 16 bytes W_t[co][8 q + 4 lk ..]
    // and the four W_256 rows 8 q + 4 lk + e; MFMA (q, e) multiplies the k pair (8 q + e, 8 q + 4 + e).  (The VALU form -- 4 x 4 outputs per thread, 16 FMAs per two 16-byte
    // LDS reads, a barrier pair per 64 m -- ran 27 us a launch: profiles/r06/zf_pmc_tail3_sedn.txt.)
    typedef float f16v __attribute__((ext_vector_type(16)));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cob = wave >> 1, cib = wave & 1, lk = lane >> 5, ln = lane & 31;
    f16v acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* wrow = a.wt + (cob * 32 + ln) * 256 + 4 * lk;
    const float* brow = a.w256 + (long long)(4 * lk) * 576 + tap * 64 + cib * 32 + ln;
#pragma unroll 8
    for (int q = 0; q < 32; ++q) {
        const float4 av = *(const float4*)(wrow + 8 * q);
        const float4 gv = *(const float4*)&gate[8 * q + 4 * lk];
        const float b0 = brow[(8 * q + 0) * 576], b1 = brow[(8 * q + 1) * 576], b2 = brow[(8 * q + 2) * 576], b3 = brow[(8 * q + 3) * 576];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x * gv.x, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y * gv.y, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z * gv.z, b2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w * gv.w, b3, acc, 0, 0, 0);
    }
    // through LDS as fp16 [co][ci], then 16-byte stores: the tap's eight fragments are 8 KB in a row.  fragment f = (tap*4 + ks)*2 + nblk, lane l = 32*(ci%16/8) + co%32,
    // element e = ci%8 (pack_conv's order)
#pragma unroll
    for (int r = 0; r < 16; ++r) Cs[cob * 32 + (r >> 2) * 8 + lk * 4 + (r & 3)][cib * 32 + ln] = (half_t)acc[r];      // (D: register r of lane -> row 8 (r / 4) + 4 lk + r % 4, column ln)
    __syncthreads();
#pragma unroll
    for (int q = threadIdx.x; q < 512; q += 256) {
        const int fl = q >> 6, l = q & 63;
        const int ci8 = (fl >> 1) * 2 + (l >> 5), co = (fl & 1) * 32 + (l & 31);
        *(half8_t*)(a.weff + ((long long)b * 72 + tap * 8) * 512 + q * 8) = *(const half8_t*)&Cs[co][ci8 * 8];
    }
}

// ---------------------------------------------------------------------------------------------------
// lite FRM gate (models.py:270-287) + LB residual (MoeNet_lite2.py:16-20): out = t * gate + x
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frm_gate_kernel(FrmArgs a)
{
    __shared__ float red[4][64];
    __shared__ float mean[64];
    __shared__ float hid[3];
    const int b = blockIdx.x, t = threadIdx.x & 63, part = threadIdx.x >> 6;
    float s = 0.f;
    for (int k = part; k < a.nslab; k += 4) s += a.partial[((long long)b * a.nslab + k) * 64 + t];     // (fixed order: deterministic)
    red[part][t] = s;
    __syncthreads();
    if (part == 0) mean[t] = ((red[0][t] + red[1][t]) + (red[2][t] + red[3][t])) / (float)a.HW;
    __syncthreads();
    if (threadIdx.x < 3) {
        float h = a.b0[threadIdx.x];
        for (int c = 0; c < 64; ++c) h += a.w0[threadIdx.x * 64 + c] * mean[c];
        hid[threadIdx.x] = h > 0.f ? h : 0.f;
    }
    __syncthreads();
    if (part == 0) {
        float u = a.b2[t];
#pragma unroll
        for (int k = 0; k < 3; ++k) u += a.w2[t * 3 + k] * hid[k];
        a.gate[b * 64 + t] = 1.f / (1.f + __expf(-u));
    }
}

// the gate from conv_2's INPUT (FrmPreArgs in common.h): one block per plane
__global__ __launch_bounds__(256) void frm_pre_kernel(FrmPreArgs a)
{
    __shared__ float sums[5][64];        // total, first row, last row, first column, last column
    __shared__ float corner[4][64];      // m[0][0], m[0][W-1], m[H-1][0], m[H-1][W-1]
    __shared__ float sh[576];            // shifted-window sums, k = tap*64 + ci
    __shared__ float red[4][64];
    __shared__ float bred[4][32][64];
    __shared__ float mean[64];
    __shared__ float hid[3];
    const int b = blockIdx.x, t = threadIdx.x;
    const long long HW = (long long)a.H * a.W;
    {   // totals: the slabs of conv_1's epilogue, four parts in a fixed order
        const int ch = t & 63, part = t >> 6;
        float s = 0.f;
#pragma unroll 16
        for (int k = part; k < a.nslab; k += 4) s += a.partial[((long long)b * a.nslab + k) * 64 + ch];      // (unrolled: sixteen loads in flight -- one load per wait made this loop 2/3 of the kernel)
        red[part][ch] = s;
    }
    {   // border: index i in [0, 2W + 2H) = first row, last row, first column, last column; thread = (pixel lane t / 8 of 32, channel group t % 8)
        const int cg = t & 7, pl = t >> 3;
        float acc[4][8];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[c][e] = 0.f;
        const int nb = 2 * a.W + 2 * a.H;
#pragma unroll 8
        for (int i = pl; i < nb; i += 32) {
            int y, x, c;
            if (i < 2 * a.W) { c = i < a.W ? 0 : 1; y = c ? a.H - 1 : 0; x = c ? i - a.W : i; }
            else { const int k = i - 2 * a.W; c = k < a.H ? 2 : 3; x = c == 3 ? a.W - 1 : 0; y = c == 3 ? k - a.H : k; }
            const long long off = ((long long)b * HW + (long long)y * a.W + x) * 64 + cg * 8;
            const half8_t v = *(const half8_t*)(a.m + off);
            half8_t vl = {0, 0, 0, 0, 0, 0, 0, 0};
            if (a.m_lo) vl = *(const half8_t*)(a.m_lo + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float u = (float)v[e] + (float)vl[e] * 0.00048828125f;
                acc[0][e] += c == 0 ? u : 0.f; acc[1][e] += c == 1 ? u : 0.f; acc[2][e] += c == 2 ? u : 0.f; acc[3][e] += c == 3 ? u : 0.f;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) bred[c][pl][cg * 8 + e] = acc[c][e];
    }
    {
        const int q = t >> 6, ch = t & 63;
        const long long p = (q & 2 ? (long long)(a.H - 1) * a.W : 0) + (q & 1 ? a.W - 1 : 0);
        const long long off = ((long long)b * HW + p) * 64 + ch;
        corner[q][ch] = (float)a.m[off] + (a.m_lo ? (float)a.m_lo[off] * 0.00048828125f : 0.f);
    }
    __syncthreads();
    {
        const int c = t >> 6, ch = t & 63;
        float v = 0.f;
#pragma unroll 8
        for (int k = 0; k < 32; ++k) v += bred[c][k][ch];
        sums[1 + c][ch] = v;
        if (c == 0) sums[0][ch] = (red[0][ch] + red[1][ch]) + (red[2][ch] + red[3][ch]);
    }
    __syncthreads();
    for (int k = t; k < 576; k += 256) {
        const int tap = k >> 6, ci = k & 63, dy = tap / 3, dx = tap % 3;
        float v = sums[0][ci];
        if (dy == 0) v -= sums[2][ci];           // rows y-1: the last row never contributes
        if (dy == 2) v -= sums[1][ci];           // rows y+1: the first row never contributes
        if (dx == 0) v -= sums[4][ci];
        if (dx == 2) v -= sums[3][ci];
        if (dy == 0 && dx == 0) v += corner[3][ci];
        if (dy == 0 && dx == 2) v += corner[2][ci];
        if (dy == 2 && dx == 0) v += corner[1][ci];
        if (dy == 2 && dx == 2) v += corner[0][ci];
        sh[k] = v;
    }
    __syncthreads();
    {   // thread (part q of 4, channel co): 144 of the 576 terms, then a fixed-order sum of the four parts
        const int q = t >> 6, co = t & 63;
        float m0 = 0.f, m1 = 0.f;
#pragma unroll 8
        for (int k = q * 144; k < q * 144 + 144; k += 2) {
            m0 += a.c2t[(k + 0) * 64 + co] * sh[k + 0];
            m1 += a.c2t[(k + 1) * 64 + co] * sh[k + 1];
        }
        red[q][co] = m0 + m1;
    }
    __syncthreads();
    if (t < 64) mean[t] = ((red[0][t] + red[1][t]) + (red[2][t] + red[3][t])) / (float)HW;
    __syncthreads();
    if (t < 3) {
        float h = a.b0[t];
        for (int c = 0; c < 64; ++c) h += a.w0[t * 64 + c] * mean[c];
        hid[t] = h > 0.f ? h : 0.f;
    }
    __syncthreads();
    if (t < 64) {
        float u = a.b2[t];
#pragma unroll
        for (int k = 0; k < 3; ++k) u += a.w2[t * 3 + k] * hid[k];
        const float g = fmaxf(1.f / (1.f + __expf(-u)), 1e-12f);      // (a gate that small contributes nothing; its reciprocal must stay finite: the residual is added as x / g)
        a.gate[b * 64 + t] = g;
        a.gate[((long long)a.B + b) * 64 + t] = 1.f / g;
    }
}

__global__ __launch_bounds__(256) void frm_apply_kernel(FrmArgs a)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // one 16-B group each
    const long long total = (long long)a.B * a.HW * 8;
    if (idx >= total) return;
    const int cg = (int)(idx & 7) * 8;
    const int b = (int)((idx >> 3) / a.HW);
    const float* g = a.gate + b * 64 + cg;
    const half8_t tv = *(const half8_t*)(a.t + idx * 8);
    const half8_t xv = *(const half8_t*)(a.x + idx * 8);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)tv[e];
    if (a.t_lo) {
        const half8_t tl = *(const half8_t*)(a.t_lo + idx * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)tl[e] * 0.00048828125f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * g[e] + (float)xv[e];
    if (a.x_lo) {
        const half8_t xl = *(const half8_t*)(a.x_lo + idx * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += (float)xl[e] * 0.00048828125f;
    }
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
    *(half8_t*)(a.out + idx * 8) = o;
    if (a.out_lo) {
        half8_t l;
#pragma unroll
        for (int e = 0; e < 8; ++e) l[e] = (half_t)((v[e] - (float)o[e]) * 2048.f);
        *(half8_t*)(a.out_lo + idx * 8) = l;
    }
}

// ---------------------------------------------------------------------------------------------------
// Stitch: per-pixel fold of doCrop's sequential blend (imageProcess.py:120-131,167-170).  Every HR pixel visits,
// in raster tile order, the tiles whose written region covers it and applies
//     v1 = ex + wH*(r-ex)  (row inside the tile's blend band, else r);   v = ex + wW*(v1-ex)  (column likewise)
// with the same fp32 operation order as the reference, so the result is bit-identical to the sequential loop.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float stitch_pixel(const StitchArgs& a, int X, int Y, int c)
{
    const int i0 = a.row_first[Y], ni = a.row_cnt[Y];
    const int j0 = a.col_first[X], nj = a.col_cnt[X];
    float cur = 0.f;
    for (int i = max(i0, a.row_lo); i < i0 + ni; ++i) {      // (row_lo: a band starts at the solid part of its first tile row, which overwrites whatever the rows above wrote)
        const int fy = a.row_tab[i * 4 + 0], sy = a.row_tab[i * 4 + 1], oy = a.row_tab[i * 4 + 2], eh = a.row_tab[i * 4 + 3];
        for (int j = j0; j < j0 + nj; ++j) {
            const int fx = a.col_tab[j * 4 + 0], sx = a.col_tab[j * 4 + 1], ox = a.col_tab[j * 4 + 2], ew = a.col_tab[j * 4 + 3];
            const float r = a.tiles[a.tile_off[i * a.step_w + j] + ((long long)c * eh + (Y - oy)) * ew + (X - ox)];
            float v1 = r;
            if (Y < sy) v1 = cur + a.ramp[Y - fy] * (r - cur);
            float v = v1;
            if (X < sx) v = cur + a.ramp[X - fx] * (v1 - cur);
            cur = v;
        }
    }
    return cur;
}

__global__ __launch_bounds__(256) void stitch_kernel(StitchArgs a)
{
    const int X = blockIdx.x * 256 + threadIdx.x;
    const int Y = blockIdx.y + a.y0, c = blockIdx.z;
    if (X >= a.out_w) return;
    const float cur = stitch_pixel(a, X, Y, c);
    const long long o = ((long long)c * a.rows + (Y - a.y0)) * a.out_w + X;
    if (a.out_dtype == MOE_F16) ((half_t*)a.out)[o] = (half_t)cur;
    else ((float*)a.out)[o] = cur;
}

// Four consecutive pixels per thread (out_w % 4 == 0).  Almost every quad lies inside one tile's solid region: then the fold is
// the identity on that tile's value and the quad is one 16-byte load and one 8/16-byte store; the others take the exact
// per-pixel fold above.  (One pixel per thread: 465 us per 8K frame; this: 350; eight rows per thread was slower again.)
__global__ __launch_bounds__(256) void stitch4_kernel(StitchArgs a)
{
    const int X0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int Y = blockIdx.y + a.y0, c = blockIdx.z;
    if (X0 >= a.out_w) return;
    float v[4];
    const int i0 = a.row_first[Y], j0 = a.col_first[X0];
    bool fast = a.row_cnt[Y] == 1 && a.col_cnt[X0] == 1 && a.col_cnt[X0 + 3] == 1 && a.col_first[X0 + 3] == j0;
    if (fast) {
        const int sy = a.row_tab[i0 * 4 + 1], oy = a.row_tab[i0 * 4 + 2], eh = a.row_tab[i0 * 4 + 3];
        const int sx = a.col_tab[j0 * 4 + 1], ox = a.col_tab[j0 * 4 + 2], ew = a.col_tab[j0 * 4 + 3];
        const long long o = a.tile_off[i0 * a.step_w + j0] + ((long long)c * eh + (Y - oy)) * ew + (X0 - ox);
        fast = Y >= sy && X0 >= sx && (o & 3) == 0;
        if (fast) {
            const float4 q = *(const float4*)(a.tiles + o);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        }
    }
    if (!fast) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = stitch_pixel(a, X0 + e, Y, c);
    }
    const long long o = ((long long)c * a.rows + (Y - a.y0)) * a.out_w + X0;
    if (a.out_dtype == MOE_F16) {
        half4_t h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (half_t)v[e];
        *(half4_t*)((half_t*)a.out + o) = h;
    } else {
        *(float4*)((float*)a.out + o) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// R rows x eight consecutive pixels per thread (out_w % 8 == 0, at most 64 tile columns).  History of this kernel on the 8K / 32K canvas:
// one pixel per thread 465 us; four pixels (stitch4 above) 350 us -- a chain of four dependent table loads per thread (col_first -> col_tab
// -> tile_off -> tile data) for 16 bytes of payload; eight pixels with the column table staged in LDS and a block-uniform row side
// 388 us / 4.46 ms = 1.6 / 2.1 TB/s; this form 1.88 ms on the 32K canvas = 5.1 TB/s.  What the last step removed: (1) a block lived for one
// chain of dependent loads and moved 12 KB with it -- now a block takes R rows, so one chain carries R x 2 independent 16-byte loads per
// thread; (2) seam ROWS (two covering tile rows: 4 % of the rows) sent every thread through the per-pixel fold -- now a thread whose eight
// columns lie in the solid part of ONE tile column folds the covering tile rows as vectors (the reference's order per pixel: cur = r, or
// cur + ramp (r - cur) inside the tile's blend band, python/imageProcess.py:120-131, so the result stays bit-identical to the sequential
// loop); (3) a thread on a COLUMN seam walked its 8 pixels one after the other, ~5 dependent loads each, and held its wave meanwhile --
// every fourth wave of a 2048-px tile column; now such threads only enlist their group and the whole block folds the seam pixels one
// pixel per thread.
template <int R>
__global__ __launch_bounds__(256) void stitch8r_kernel(StitchArgs a)
{
    __shared__ int s_col[64 * 4];
    __shared__ int s_seam[256];
    __shared__ int s_nseam;
    const int Y0 = blockIdx.y * R + a.y0, c = blockIdx.z;
    const int yend = a.y0 + a.rows;                       // (a band of the canvas: moe_stitch_band; the whole canvas: y0 = 0, rows = out_h)
    const int nsw = a.step_w;
    for (int t = threadIdx.x; t < nsw * 4; t += 256) s_col[t] = a.col_tab[t];
    if (threadIdx.x == 0) s_nseam = 0;
    __syncthreads();
    const int X0 = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (X0 < a.out_w) {
        const int j0 = a.col_first[X0], j7 = a.col_first[X0 + 7];
        bool colfast = a.col_cnt[X0] == 1 && a.col_cnt[X0 + 7] == 1 && j0 == j7;
        const int sx = s_col[j0 * 4 + 1], ox = s_col[j0 * 4 + 2], ew = s_col[j0 * 4 + 3];
        colfast = colfast && X0 >= sx;
        if (!colfast) s_seam[atomicAdd(&s_nseam, 1)] = threadIdx.x;      // eight columns on a seam: handed to the whole block below
        else {
            float v[R][8];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int Y = min(Y0 + r, yend - 1);               // (rows past the band repeat the last one and are not stored)
                const int i0 = a.row_first[Y], ni = a.row_cnt[Y];  // block-uniform
                float cur[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int i = max(i0, a.row_lo); i < i0 + ni; ++i) {
                    const int fy = a.row_tab[i * 4 + 0], sy = a.row_tab[i * 4 + 1], oy = a.row_tab[i * 4 + 2], eh = a.row_tab[i * 4 + 3];
                    const long long o = a.tile_off[i * nsw + j0] + ((long long)c * eh + (Y - oy)) * ew + (X0 - ox);
                    float q[8];
                    if ((o & 3) == 0) {
                        const float4 q0 = *(const float4*)(a.tiles + o), q1 = *(const float4*)(a.tiles + o + 4);
                        q[0] = q0.x; q[1] = q0.y; q[2] = q0.z; q[3] = q0.w; q[4] = q1.x; q[5] = q1.y; q[6] = q1.z; q[7] = q1.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) q[e] = a.tiles[o + e];
                    }
                    if (Y < sy) {
                        const float wgt = a.ramp[Y - fy];
#pragma unroll
                        for (int e = 0; e < 8; ++e) cur[e] = cur[e] + wgt * (q[e] - cur[e]);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) cur[e] = q[e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[r][e] = cur[e];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (Y0 + r >= yend) break;
                const long long o = ((long long)c * a.rows + (Y0 + r - a.y0)) * a.out_w + X0;
                if (a.out_dtype == MOE_F16) {
                    half8_t h;
#pragma unroll
                    for (int e = 0; e < 8; ++e) h[e] = (half_t)v[r][e];
                    *(half8_t*)((half_t*)a.out + o) = h;
                } else {
                    *(float4*)((float*)a.out + o) = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
                    *(float4*)((float*)a.out + o + 4) = make_float4(v[r][4], v[r][5], v[r][6], v[r][7]);
                }
            }
        }
    }
    __syncthreads();
    // column seams: one pixel per thread and pass, the exact per-pixel fold (a thread walking its own 8 x R seam pixels one after the other
    // held its wave for 32 chains of dependent loads: every fourth wave of a 2048-px tile column)
    const int total = s_nseam * 8 * R;
    for (int t = threadIdx.x; t < total; t += 256) {
        const int gidx = t / (8 * R), rem = t - gidx * 8 * R;
        const int r = rem >> 3, e = rem & 7;
        const int X = (blockIdx.x * 256 + s_seam[gidx]) * 8 + e, Y = Y0 + r;
        if (Y >= yend) continue;
        const float cur = stitch_pixel(a, X, Y, c);
        const long long o = ((long long)c * a.rows + (Y - a.y0)) * a.out_w + X;
        if (a.out_dtype == MOE_F16) ((half_t*)a.out)[o] = (half_t)cur;
        else ((float*)a.out)[o] = cur;
    }
}

// ---------------------------------------------------------------------------------------------------
// Image edges: toTorch (imageProcess.py:259-263) and toOutput (:245-257)
// ---------------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ void to_float_kernel(const TS* src, TD* dst, int H, int W, int C, float d, bool divide)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)H * W * C;
    if (idx >= n) return;
    const int c = (int)(idx % C);
    const long long p = idx / C;
    const float v = (float)src[idx];
    dst[(long long)c * H * W + p] = (TD)(divide ? v / d : v * d);
}

template <typename TS, typename TD>
__global__ void to_output_kernel(const TS* src, TD* dst, int H, int W, int C, float quant)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)H * W * C;
    if (idx >= n) return;
    const int c = (int)(idx % C);
    const long long p = idx / C;
    // reference: image * quant, clamp_(0, quant-1), truncate
    // (toFloat casts to fp32 first: imageProcess.py:238-243)
    float v = (float)src[(long long)c * H * W + p] * quant;
    v = fminf(fmaxf(v, 0.f), quant - 1.f);
    if (!(v == v)) v = 0.f;
    dst[idx] = (TD)(int)v;
}

// Three-channel images, eight pixels per thread (H * W % 8 == 0, aligned bases): the per-element kernels above move one 1..4-byte element per thread with a
// stride of C elements on the interleaved side (1.3 TB/s on the 8K canvas); here a thread moves 8 x 3 interleaved elements as one contiguous run and 8
// consecutive elements of each plane as one vector.  Same arithmetic per element.
template <typename T, int N> struct alignas(sizeof(T) * 8 >= 16 ? 16 : 8) VecN { T e[N]; };      // (a run of 8 or 24 elements starting at a multiple of 8 elements)

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void to_float3_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long long HW, float d, bool divide)
{
    const long long p0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (p0 >= HW) return;
    const VecN<TS, 24> in = *(const VecN<TS, 24>*)(src + p0 * 3);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        VecN<TD, 8> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = (float)in.e[e * 3 + c];
            o.e[e] = (TD)(divide ? v / d : v * d);
        }
        *(VecN<TD, 8>*)(dst + c * HW + p0) = o;
    }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void to_output3_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long long HW, float quant)
{
    const long long p0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
    if (p0 >= HW) return;
    VecN<TD, 24> o;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const VecN<TS, 8> in = *(const VecN<TS, 8>*)(src + c * HW + p0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = (float)in.e[e] * quant;
            v = fminf(fmaxf(v, 0.f), quant - 1.f);
            if (!(v == v)) v = 0.f;
            o.e[e * 3 + c] = (TD)(int)v;
        }
    }
    *(VecN<TD, 24>*)(dst + p0 * 3) = o;
}

__global__ void nhwc_to_nchw_kernel(const half_t* in, const half_t* in_lo, float* out, int B, int H, int W, int cs, int C)
{
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)B * C * H * W;
    if (idx >= n) return;
    const int x = (int)(idx % W);
    long long t = idx / W;
    const int y = (int)(t % H); t /= H;
    const int c = (int)(t % C);
    const int b = (int)(t / C);
    const long long o = ((long long)(b * H + y) * W + x) * cs + c;
    float v = (float)in[o];
    if (in_lo) v += (float)in_lo[o] * 0.00048828125f;
    out[idx] = v;
}

}  // namespace

void launch_stem(const StemArgs& a, hipStream_t s)
{
    const long long n = (long long)a.B * a.H * ((a.W + 7) / 8) * 8;      // (b, y, x-octet, channel group): stem_kernel's PX
    const int blocks = (int)((n + 255) / 256);
    if (a.taps == 9) {
        if (a.x_dtype == MOE_F16) hipLaunchKernelGGL((stem_kernel<half_t, 9>), dim3(blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((stem_kernel<float, 9>), dim3(blocks), dim3(256), 0, s, a);
    } else {
        if (a.x_dtype == MOE_F16) hipLaunchKernelGGL((stem_kernel<half_t, 1>), dim3(blocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((stem_kernel<float, 1>), dim3(blocks), dim3(256), 0, s, a);
    }
}

void launch_tail(const TailArgs& a, hipStream_t s)
{
    static const bool old9 = [] { const char* e = getenv("MOE_TAIL_V1"); return e && !strcmp(e, "1"); }();
    const int blocks = ((a.W + 31) / 32) * ((a.H + 7) / 8) * a.B;
    if (a.taps == 9 && (!old9 || a.in1_lo8)) {
        const dim3 g(((a.W + 29) / 30) * ((a.H + 7) / 8) * a.B);
        if (a.in0_lo && a.in1_lo8) hipLaunchKernelGGL(tail3_kernel<2>, g, dim3(256), 0, s, a);
        else if (a.in0_lo) hipLaunchKernelGGL(tail3_kernel<1>, g, dim3(256), 0, s, a);
        else hipLaunchKernelGGL(tail3_kernel<0>, g, dim3(256), 0, s, a);
    }
    else if (a.taps == 9) hipLaunchKernelGGL((tail_kernel<9>), dim3(blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((tail_kernel<1>), dim3(blocks), dim3(256), 0, s, a);
}

void launch_tapsum(const TapSumArgs& a, hipStream_t s)
{
    if (a.form == 1) {          // phase-class sums (r == 2, W % 8 == 0 by construction: the producing conv requires input widths that are multiples of 4)
        const long long n = (long long)a.B * a.H * (a.W / 8);
        if (a.vec_ok) tapsum4_kernel<true><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(a);
        else tapsum4_kernel<false><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(a);
        return;
    }
    const dim3 grid((a.W + 255) / 256, a.H, a.B);
    if (a.r == 3) tapsum_kernel<3><<<grid, dim3(256), 0, s>>>(a);
    else if (a.vec_ok) {
        const long long n = (long long)a.B * a.H * (a.W / 8);
        tapsum2_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(a);
    } else tapsum_kernel<2><<<grid, dim3(256), 0, s>>>(a);
}

void launch_tail1sum(const Tail1SumArgs& a, hipStream_t s)
{
    if (a.lut && !a.p1 && a.nparts == 1 && a.vec_ok && a.W % 4 == 0 && (a.r == 2 || a.r == 4 || a.r == 8) && ((uintptr_t)a.p0 & 15) == 0) {
        const long long nq = (long long)a.B * a.H * (a.W / 4);
        hipLaunchKernelGGL(tail1sum_lut4_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, a);
        return;
    }
    const long long n = (long long)a.B * a.H * a.W;
    hipLaunchKernelGGL(tail1sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
}

void launch_pool_partial(const half_t* in, const half_t* in_lo, float* partial, int B, long long HW, int C, int nslab, hipStream_t s)
{
    hipLaunchKernelGGL(pool_partial_kernel, dim3(nslab, B), dim3(256), 0, s, in, in_lo, partial, HW, C, nslab);
}

void launch_sedn_se(const SednSeArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(sedn_se_kernel, dim3(a.B, kSeSplit), dim3(256), 0, s, a);
}

void launch_sedn_fuse(const SednFuseArgs& a, hipStream_t s)
{
    if (!a.pooled) hipLaunchKernelGGL(sedn_xsum_kernel, dim3(a.nslab, a.B), dim3(256), 0, s, a);      // (with the producing conv's totals sedn_fmean visits the border itself)
    hipLaunchKernelGGL(sedn_fmean_kernel, dim3(a.B, kMeanSplit), dim3(256), 0, s, a);
    hipLaunchKernelGGL(sedn_weff_kernel, dim3(9, a.B), dim3(256), 0, s, a);
}

void launch_frm(const FrmArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(frm_gate_kernel, dim3(a.B), dim3(256), 0, s, a);
    const long long total = (long long)a.B * a.HW * 8;
    hipLaunchKernelGGL(frm_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
}

void launch_frm_pre(const FrmPreArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(frm_pre_kernel, dim3(a.B), dim3(256), 0, s, a);
}

void launch_stitch(const StitchArgs& a, hipStream_t s)
{
    if (a.rows <= 0) return;
    if (a.out_w % 8 == 0 && a.step_w <= 64 && ((uintptr_t)a.out & 15) == 0) hipLaunchKernelGGL(stitch8r_kernel<4>, dim3((a.out_w / 8 + 255) / 256, (a.rows + 3) / 4, a.C), dim3(256), 0, s, a);
    else if (a.out_w % 4 == 0 && ((uintptr_t)a.out & 15) == 0) hipLaunchKernelGGL(stitch4_kernel, dim3((a.out_w / 4 + 255) / 256, a.rows, a.C), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(stitch_kernel, dim3((a.out_w + 255) / 256, a.rows, a.C), dim3(256), 0, s, a);
}

// ---------------------------------------------------------------------------------------------------
// Wire format of tile results between ranks (moephoto_amd/dist.py, wire = 'f16s').  The stitch reads a tile's value at full precision only where
// a blend happens (the tile's own blend band, and the rows / columns a LATER tile blends over: python/imageProcess.py:120-131); everywhere else the
// value either is the canvas pixel (then it is rounded to the canvas dtype) or is overwritten.  So a tile travels as the fp16 image of all its
// values plus the fp32 values of its seam rows and seam columns, and unpacking gives back fp32 tiles whose seams are exact and whose interior is
// float(half(v)): an fp16 canvas folded from them is bit-identical to the one folded from the fp32 tiles, at 0.5 + 0.5 x (seam share) of the bytes.
// A record whose seam rows cover the tile (a band-mode strip) is just its fp32 values.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int wire_slot(int v, int a0, int a1, int b0, int b1)
{
    return (v >= a0 && v < a1) ? v - a0 : (v >= b0 && v < b1) ? (a1 - a0) + v - b0 : -1;
}

template <bool PACK>
__global__ __launch_bounds__(256) void wire_kernel(float* __restrict__ tiles, unsigned* __restrict__ wire, const WireRec* __restrict__ recs)
{
    const WireRec r = recs[blockIdx.y];
    const int nR = (r.ra1 - r.ra0) + (r.rb1 - r.rb0), nC = (r.ca1 - r.ca0) + (r.cb1 - r.cb0);
    const bool raw = nR >= r.th;
    const long long plane = (long long)r.th * r.tw, n = plane * r.C;
    float* t = tiles + r.tile_off;
    unsigned* w = wire + r.wire_off;
    const long long hw = raw ? 0 : (n + 1) / 2;                      // words of the fp16 image
    half_t* h16 = (half_t*)w;
    float* rows = (float*)(w + hw);
    float* cols = rows + (long long)r.C * nR * r.tw;
    // two consecutive values per thread and step (one 4-byte word of the fp16 image)
    for (long long e = ((long long)blockIdx.x * 256 + threadIdx.x) * 2; e < n; e += (long long)gridDim.x * 512) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long q = e + u;
            if (q >= n) break;
            const int c = (int)(q / plane);
            const int rem = (int)(q - c * plane);
            const int y = rem / r.tw, x = rem - y * r.tw;
            const int ry = wire_slot(y, r.ra0, r.ra1, r.rb0, r.rb1);
            const int cx = ry >= 0 ? -1 : wire_slot(x, r.ca0, r.ca1, r.cb0, r.cb1);
            if (PACK) {
                const float v = t[q];
                if (!raw) h16[q] = (half_t)v;
                if (ry >= 0) rows[((long long)c * nR + ry) * r.tw + x] = v;
                else if (cx >= 0) cols[((long long)c * r.th + y) * nC + cx] = v;
            } else {
                t[q] = ry >= 0 ? rows[((long long)c * nR + ry) * r.tw + x] : cx >= 0 ? cols[((long long)c * r.th + y) * nC + cx] : (float)h16[q];
            }
        }
    }
}

long long wire_rec_words(const WireRec& r)
{
    const long long nR = (r.ra1 - r.ra0) + (r.rb1 - r.rb0), nC = (r.ca1 - r.ca0) + (r.cb1 - r.cb0);
    const long long n = (long long)r.C * r.th * r.tw;
    if (nR >= r.th) return n;
    return (n + 1) / 2 + (long long)r.C * nR * r.tw + (long long)r.C * r.th * nC;
}

void launch_wire(bool pack, float* tiles, unsigned* wire, const WireRec* recs, int n, long long max_elems, hipStream_t s)
{
    if (n <= 0) return;
    const unsigned gx = (unsigned)std::min<long long>(std::max<long long>((max_elems + 2047) / 2048, 1), 4096);     // four steps per thread at most on the largest record
    if (pack) hipLaunchKernelGGL(wire_kernel<true>, dim3(gx, n), dim3(256), 0, s, tiles, wire, recs);
    else hipLaunchKernelGGL(wire_kernel<false>, dim3(gx, n), dim3(256), 0, s, tiles, wire, recs);
}

void launch_to_float(const void* src, int src_dtype, float d, bool divide, int H, int W, int C, void* dst, int dst_dtype, hipStream_t s)
{
    const long long n = (long long)H * W * C, HW = (long long)H * W;
    const dim3 blk(256);
    if (C == 3 && HW % 8 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const dim3 g3((unsigned)((HW / 8 + 255) / 256));
        if (src_dtype == MOE_U8) {
            if (dst_dtype == MOE_F16) hipLaunchKernelGGL((to_float3_kernel<uint8_t, half_t>), g3, blk, 0, s, (const uint8_t*)src, (half_t*)dst, HW, d, divide);
            else hipLaunchKernelGGL((to_float3_kernel<uint8_t, float>), g3, blk, 0, s, (const uint8_t*)src, (float*)dst, HW, d, divide);
        } else {
            if (dst_dtype == MOE_F16) hipLaunchKernelGGL((to_float3_kernel<uint16_t, half_t>), g3, blk, 0, s, (const uint16_t*)src, (half_t*)dst, HW, d, divide);
            else hipLaunchKernelGGL((to_float3_kernel<uint16_t, float>), g3, blk, 0, s, (const uint16_t*)src, (float*)dst, HW, d, divide);
        }
        return;
    }
    const dim3 grid((unsigned)((n + 255) / 256));
    if (src_dtype == MOE_U8) {
        if (dst_dtype == MOE_F16) hipLaunchKernelGGL((to_float_kernel<uint8_t, half_t>), grid, blk, 0, s, (const uint8_t*)src, (half_t*)dst, H, W, C, d, divide);
        else hipLaunchKernelGGL((to_float_kernel<uint8_t, float>), grid, blk, 0, s, (const uint8_t*)src, (float*)dst, H, W, C, d, divide);
    } else {
        if (dst_dtype == MOE_F16) hipLaunchKernelGGL((to_float_kernel<uint16_t, half_t>), grid, blk, 0, s, (const uint16_t*)src, (half_t*)dst, H, W, C, d, divide);
        else hipLaunchKernelGGL((to_float_kernel<uint16_t, float>), grid, blk, 0, s, (const uint16_t*)src, (float*)dst, H, W, C, d, divide);
    }
}

void launch_to_output(const void* src, int src_dtype, int H, int W, int C, float quant, void* dst, int dst_dtype, hipStream_t s)
{
    const long long n = (long long)H * W * C, HW = (long long)H * W;
    const dim3 blk(256);
    if (C == 3 && HW % 8 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const dim3 g3((unsigned)((HW / 8 + 255) / 256));
        if (src_dtype == MOE_F16) {
            if (dst_dtype == MOE_U8) hipLaunchKernelGGL((to_output3_kernel<half_t, uint8_t>), g3, blk, 0, s, (const half_t*)src, (uint8_t*)dst, HW, quant);
            else hipLaunchKernelGGL((to_output3_kernel<half_t, uint16_t>), g3, blk, 0, s, (const half_t*)src, (uint16_t*)dst, HW, quant);
        } else {
            if (dst_dtype == MOE_U8) hipLaunchKernelGGL((to_output3_kernel<float, uint8_t>), g3, blk, 0, s, (const float*)src, (uint8_t*)dst, HW, quant);
            else hipLaunchKernelGGL((to_output3_kernel<float, uint16_t>), g3, blk, 0, s, (const float*)src, (uint16_t*)dst, HW, quant);
        }
        return;
    }
    const dim3 grid((unsigned)((n + 255) / 256));
    if (src_dtype == MOE_F16) {
        if (dst_dtype == MOE_U8) hipLaunchKernelGGL((to_output_kernel<half_t, uint8_t>), grid, blk, 0, s, (const half_t*)src, (uint8_t*)dst, H, W, C, quant);
        else hipLaunchKernelGGL((to_output_kernel<half_t, uint16_t>), grid, blk, 0, s, (const half_t*)src, (uint16_t*)dst, H, W, C, quant);
    } else {
        if (dst_dtype == MOE_U8) hipLaunchKernelGGL((to_output_kernel<float, uint8_t>), grid, blk, 0, s, (const float*)src, (uint8_t*)dst, H, W, C, quant);
        else hipLaunchKernelGGL((to_output_kernel<float, uint16_t>), grid, blk, 0, s, (const float*)src, (uint16_t*)dst, H, W, C, quant);
    }
}

// ---------------------------------------------------------------------------------------------------
// resize: F.interpolate(x[None], size=(h, w), mode=..., align_corners=False)[0]  (python/imageProcess.py:555-556, the `resize`
// step of the pipeline builder, python/procedure.py:104-107).  Planar (C, H, W) in, (C, h, w) out, arithmetic in fp32 exactly as
// torch's upsample kernels order it: scale = in / out (float), src = scale * (dst + 0.5) - 0.5; bilinear clamps src at 0, bicubic
// (A = -0.75) clamps the four tap indices; nearest takes floor(dst * scale).  __f*_rn keeps the compiler from contracting the index
// arithmetic into FMAs (a contracted src can land on the other side of an integer).
// ---------------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

template <typename T, int MODE>
__global__ __launch_bounds__(256) void resize_kernel(const T* src, T* dst, int C, int H, int W, int h, int w, float sy, float sx)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)C * h * w;
    if (idx >= total) return;
    const int x = (int)(idx % w);
    const long long t = idx / w;
    const int y = (int)(t % h), c = (int)(t / h);
    const T* p = src + (long long)c * H * W;
    float out;
    if (MODE == 0) {            // nearest: min(floor(dst * scale), in - 1)
        const int iy = min((int)floorf(__fmul_rn((float)y, sy)), H - 1), ix = min((int)floorf(__fmul_rn((float)x, sx)), W - 1);
        out = (float)p[(long long)iy * W + ix];
    } else if (MODE == 1) {     // bilinear
        float fy = __fsub_rn(__fmul_rn(sy, __fadd_rn((float)y, 0.5f)), 0.5f), fx = __fsub_rn(__fmul_rn(sx, __fadd_rn((float)x, 0.5f)), 0.5f);
        fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly = __fsub_rn(fy, (float)y0), lx = __fsub_rn(fx, (float)x0);
        const float hy = __fsub_rn(1.f, ly), hx = __fsub_rn(1.f, lx);
        const float v00 = (float)p[(long long)y0 * W + x0], v01 = (float)p[(long long)y0 * W + x1];
        const float v10 = (float)p[(long long)y1 * W + x0], v11 = (float)p[(long long)y1 * W + x1];
        // h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11), left to right, no contraction
        const float r0 = __fadd_rn(__fmul_rn(hx, v00), __fmul_rn(lx, v01)), r1 = __fadd_rn(__fmul_rn(hx, v10), __fmul_rn(lx, v11));
        out = __fadd_rn(__fmul_rn(hy, r0), __fmul_rn(ly, r1));
    } else {                    // bicubic, A = -0.75
        const float A = -0.75f;
        const float fy = __fsub_rn(__fmul_rn(sy, __fadd_rn((float)y, 0.5f)), 0.5f), fx = __fsub_rn(__fmul_rn(sx, __fadd_rn((float)x, 0.5f)), 0.5f);
        const float gy = floorf(fy), gx = floorf(fx);
        const int iy = (int)gy, ix = (int)gx;
        const float ty = __fsub_rn(fy, gy), tx = __fsub_rn(fx, gx);
        float cy[4], cx[4];
        cy[0] = cubic2(ty + 1.f, A); cy[1] = cubic1(ty, A); cy[2] = cubic1(1.f - ty, A); cy[3] = cubic2(2.f - ty, A);
        cx[0] = cubic2(tx + 1.f, A); cx[1] = cubic1(tx, A); cx[2] = cubic1(1.f - tx, A); cx[3] = cubic2(2.f - tx, A);
        out = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = min(max(iy - 1 + i, 0), H - 1);
            float r = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = min(max(ix - 1 + j, 0), W - 1);
                r += (float)p[(long long)yy * W + xx] * cx[j];
            }
            out += r * cy[i];
        }
    }
    dst[idx] = (T)out;
}

}  // namespace

void launch_resize(const void* src, void* dst, int dtype, int C, int H, int W, int h, int w, int mode, hipStream_t s)
{
    const long long n = (long long)C * h * w;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    const float sy = (float)H / (float)h, sx = (float)W / (float)w;
#define MOE_RZ(T, M) hipLaunchKernelGGL((resize_kernel<T, M>), grid, blk, 0, s, (const T*)src, (T*)dst, C, H, W, h, w, sy, sx)
    if (dtype == MOE_F16) { if (mode == 0) MOE_RZ(half_t, 0); else if (mode == 1) MOE_RZ(half_t, 1); else MOE_RZ(half_t, 2); }
    else { if (mode == 0) MOE_RZ(float, 0); else if (mode == 1) MOE_RZ(float, 1); else MOE_RZ(float, 2); }
#undef MOE_RZ
}

void launch_nhwc_to_nchw_f32(const half_t* in, const half_t* in_lo, float* out, int B, int H, int W, int cs, int C, hipStream_t s)
{
    const long long n = (long long)B * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, in_lo, out, B, H, W, cs, C);
}

// max |a - b| over n floats into *out as the bit pattern of a non-negative float (order-preserving as unsigned; a NaN difference counts as +inf): the comparison of
// moe_net_calibrate's noise tiles stays on the device -- 150 MB of results per arithmetic would otherwise cross PCIe for one number
__global__ __launch_bounds__(256) void maxabsdiff_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, unsigned* __restrict__ out)
{
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = __builtin_fabsf(a[i] - b[i]);
        m = (d <= m) ? m : (d == d ? d : __builtin_inff());
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

void launch_maxabsdiff(const float* a, const float* b, long long n, unsigned* out, hipStream_t s)
{
    const unsigned g = (unsigned)std::min<long long>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(maxabsdiff_kernel, dim3(g ? g : 1), dim3(256), 0, s, a, b, n, out);
}
