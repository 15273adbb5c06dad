// blend.hip -- the body of doCrop's tile loop behind the net call, python/imageProcess.py:167-170 of the reference, as ONE kernel:
//
//     t = tmp_image[..., top*sc:bsc, left*sc:rsc]
//     q, _ = blend(*blend(opt.unpad(r), t, topT, padSc, -2, bl.t()), leftT, padSc, -1, bl)      # blend(): python/imageProcess.py:120-131
//     tmp_image[..., bsc-h:bsc, rsc-w:rsc] = q
//
// For a maintainer who keeps MoePhoto's own loop and swaps only the model class: the two `blend` calls and the slice-assign are
// ~14 torch launches per tile that move the tile four times (split views, b - bx, blend * (..), bx + .., cat, again for the columns, the assign); here every
// element of the assigned region is read once and written once, and the rows / columns in front of the blend bands are never touched.
//
// Arithmetic: element for element what the reference's expression does IN THE CANVAS DTYPE -- b = bx + blend * (b - bx) as three separately rounded operations
// (torch rounds every elementwise fp16 op to fp16; no FMA contraction in fp32), rows first, then columns on the row-blended value against the same old canvas
// value (the second blend's `x` is the narrowed window of the untouched canvas).  Bit-identical to the torch expression (tests/test_gpu_parity.py).
#include "common.h"

namespace {

template <typename T> struct Arith;
template <> struct Arith<float> {
    static __device__ __forceinline__ float mix(float b, float bx, float w) { return __fadd_rn(bx, __fmul_rn(w, __fsub_rn(b, bx))); }
};
template <> struct Arith<half_t> {
    static __device__ __forceinline__ half_t mix(half_t b, half_t bx, half_t w)
    {
#pragma clang fp contract(off)      // (the compiler narrows these to v_sub_f16 / v_mul_f16 / v_add_f16 -- same roundings -- and would then fuse the last two into ONE v_fma_f16: one rounding less than torch)
        const half_t d = (half_t)((float)b - (float)bx);          // (fp32 difference of two halves is exact: one rounding, as torch's fp16 sub)
        const half_t p = (half_t)((float)w * (float)d);
        return (half_t)((float)bx + (float)p);
    }
};

// one thread = VEC consecutive columns of one row of one plane of the ASSIGNED region (rows [r0, rh), columns [c0, rw) of the tile result)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void blend_tile_kernel(BlendTileArgs a)
{
    const int ncol = a.rw - a.c0, nrow = a.rh - a.r0;
    const int nv = (ncol + VEC - 1) / VEC;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)a.C * nrow * nv) return;
    const int v = (int)(i % nv);
    const long long t = i / nv;
    const int y = a.r0 + (int)(t % nrow), c = (int)(t / nrow);
    const int x = a.c0 + v * VEC;
    const T* rp = (const T*)a.r + c * a.r_sC + (long long)y * a.r_sH + x;
    T* cp = (T*)a.canvas + c * a.c_sC + (long long)(a.top_sc + y) * a.c_sH + a.left_sc + x;
    const T* ramp = (const T*)a.ramp;
    const bool hb = y < a.lt_h;                       // inside the row band [r0, lt_h) (r0 = lt_h - pad_sc when there is one, else lt_h = r0 = 0)
    const T wh = hb ? ramp[y - a.r0] : (T)0;
    T val[VEC], old[VEC];
    const bool full = x + VEC <= a.rw;
    const bool need_old = hb || x < a.lt_w;           // the canvas is only read where a band covers the element
    if (full) {
        typedef T vec_t __attribute__((ext_vector_type(VEC)));
        const vec_t rv = *(const vec_t*)rp;
#pragma unroll
        for (int k = 0; k < VEC; ++k) val[k] = rv[k];
        if (need_old) {
            const vec_t ov = *(const vec_t*)cp;
#pragma unroll
            for (int k = 0; k < VEC; ++k) old[k] = ov[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            val[k] = x + k < a.rw ? rp[k] : (T)0;
            old[k] = (need_old && x + k < a.rw) ? cp[k] : (T)0;
        }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        if (hb) val[k] = Arith<T>::mix(val[k], old[k], wh);
        if (x + k < a.lt_w) val[k] = Arith<T>::mix(val[k], old[k], ramp[x + k - a.c0]);
    }
    if (full) {
        typedef T vec_t __attribute__((ext_vector_type(VEC)));
        vec_t ov;
#pragma unroll
        for (int k = 0; k < VEC; ++k) ov[k] = val[k];
        *(vec_t*)cp = ov;
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k)
            if (x + k < a.rw) cp[k] = val[k];
    }
}

}  // namespace

void launch_blend_tile(const BlendTileArgs& a, bool f16, hipStream_t s)
{
    const int ncol = a.rw - a.c0, nrow = a.rh - a.r0;
    if (ncol <= 0 || nrow <= 0 || a.C <= 0) return;
    // vector accesses need every row start of both tensors aligned to VEC elements
    const size_t es = f16 ? 2 : 4;
    const bool al4 = ((uintptr_t)a.r / es) % 4 == 0 && ((uintptr_t)a.canvas / es) % 4 == 0 && a.r_sC % 4 == 0 && a.r_sH % 4 == 0 && a.c_sC % 4 == 0 && a.c_sH % 4 == 0 &&
                     a.c0 % 4 == 0 && a.left_sc % 4 == 0;
    const int vec = al4 ? 4 : 1;
    const long long n = (long long)a.C * nrow * ((ncol + vec - 1) / vec);
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    if (f16) { if (al4) blend_tile_kernel<half_t, 4><<<g, b, 0, s>>>(a); else blend_tile_kernel<half_t, 1><<<g, b, 0, s>>>(a); }
    else { if (al4) blend_tile_kernel<float, 4><<<g, b, 0, s>>>(a); else blend_tile_kernel<float, 1><<<g, b, 0, s>>>(a); }
}
