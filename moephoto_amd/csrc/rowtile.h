// rowtile.h -- epilogue helpers shared by the register-weight kernels (arsb_fused.hip, conv64_x3.hip): a finished output row is two
// 16x16 MFMA tiles per wave (lane (n, q): 4 channels 16w + 4q.. of pixel columns n and 16 + n).
//
// The epilogues ride in the MFMA stream of a single wave per SIMD; beside 16-cycle MFMAs only ~2 other instructions per MFMA are
// hidden (profiles/r02: a row's 36 MFMAs take 640 cycles bare, +6.4 cycles for every filler beyond ~70), so these helpers are
// written instruction by instruction: v_fma_mix_* reads an fp16 half of a packed register as an fp32 operand (no v_cvt), and the
// low part of a value is one multiply + one mixed FMA that writes its fp16 result straight into a packed half.
#pragma once
#include "common.h"

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

// v + c * (float)lo16(h2)   /   v + c * (float)hi16(h2)
__device__ __forceinline__ float mix_lo(unsigned h2, float c, float v)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(c), "v"(v));
    return r;
}
__device__ __forceinline__ float mix_hi(unsigned h2, float c, float v)
{
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(c), "v"(v));
    return r;
}

// (v0, v1) -> packed fp16 parts {hi(v0), hi(v1)} and packed remainders {(v0 - hi) 2^11, (v1 - hi) 2^11}; neg2048 holds -2048.f
__device__ __forceinline__ void split2(float v0, float v1, float neg2048, unsigned& hi2, unsigned& lo2)
{
    const h2_t pr = {(half_t)v0, (half_t)v1};             // one v_cvt_pk_f16_f32 (round to nearest even)
    hi2 = __builtin_bit_cast(unsigned, pr);
    const float t0 = v0 * 2048.f, t1 = v1 * 2048.f;
    unsigned l = 0;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi2), "v"(neg2048), "v"(t0));     // v0 2^11 - hi 2^11: exact in fp32
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(hi2), "v"(neg2048), "v"(t1));
    lo2 = l;
}

// Finish one output row held as two tiles v[cb][4] (fp32): [+ side * 2^-11] [+ residual], split into hi / lo parts and stored with 16 bytes
// per lane.  v_permlane16_swap(X, Y) exchanges the odd 16-lane rows of X with the even rows of Y (an involution): applied to a loaded
// 16-byte word's two channel quads it yields this lane's own quad for tile 0 (X) and tile 1 (Y); applied to the two tiles' packed
// results it yields 8 consecutive channels of one pixel.
struct RowConsts { float one, lowscale, neg2048; };

template <bool SIDE, bool RES, bool LO>
__device__ __forceinline__ void finish_row(float (&v)[2][4], const uint4& sidew, const uint4& resw, const RowConsts& k, char* out_hi, char* out_lo, unsigned off)
{
    if (SIDE) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(sidew.x, sidew.z, false, false);     // halves 0,1 | 4,5
        const auto s1 = __builtin_amdgcn_permlane16_swap(sidew.y, sidew.w, false, false);     // halves 2,3 | 6,7
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            v[cb][0] = mix_lo(s0[cb], k.lowscale, v[cb][0]); v[cb][1] = mix_hi(s0[cb], k.lowscale, v[cb][1]);
            v[cb][2] = mix_lo(s1[cb], k.lowscale, v[cb][2]); v[cb][3] = mix_hi(s1[cb], k.lowscale, v[cb][3]);
        }
    }
    if (RES) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(resw.x, resw.z, false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(resw.y, resw.w, false, false);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            v[cb][0] = mix_lo(s0[cb], k.one, v[cb][0]); v[cb][1] = mix_hi(s0[cb], k.one, v[cb][1]);
            v[cb][2] = mix_lo(s1[cb], k.one, v[cb][2]); v[cb][3] = mix_hi(s1[cb], k.one, v[cb][3]);
        }
    }
    unsigned hi[2][2], lo[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        if (LO) {
            split2(v[cb][0], v[cb][1], k.neg2048, hi[cb][0], lo[cb][0]);
            split2(v[cb][2], v[cb][3], k.neg2048, hi[cb][1], lo[cb][1]);
        } else {
            const h2_t p0 = {(half_t)v[cb][0], (half_t)v[cb][1]}, p1 = {(half_t)v[cb][2], (half_t)v[cb][3]};
            hi[cb][0] = __builtin_bit_cast(unsigned, p0); hi[cb][1] = __builtin_bit_cast(unsigned, p1);
        }
    }
    {
        const auto s0 = __builtin_amdgcn_permlane16_swap(hi[0][0], hi[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(hi[0][1], hi[1][1], false, false);
        *(uint4*)(out_hi + off) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
    }
    if (LO) {
        const auto s0 = __builtin_amdgcn_permlane16_swap(lo[0][0], lo[1][0], false, false);
        const auto s1 = __builtin_amdgcn_permlane16_swap(lo[0][1], lo[1][1], false, false);
        *(uint4*)(out_lo + off) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
    }
}
