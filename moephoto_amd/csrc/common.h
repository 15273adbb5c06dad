// common.h -- shared types between the gfx950 kernels and the host engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(2))) _Float16 half2_t;
typedef __attribute__((ext_vector_type(4))) _Float16 half4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;
typedef __attribute__((ext_vector_type(16))) float float16_t;

// Activation tensors live in HBM as NHWC fp16 with a channel stride that is a multiple of 64
// (48-channel nets are zero-padded to 64): one pixel of a 64-channel tensor is exactly one 128-byte line.
constexpr int kCB = 64;          // channel block
constexpr int kTileW = 32;       // output pixels per patch row  (= MFMA N)
constexpr int kTileH = 8;        // output rows per patch
constexpr int kFragBytes = 1024; // one MFMA 32x32x16 f16 operand fragment: 64 lanes x 16 B

// ---------------------------------------------------------------------------------------------------
// Implicit-GEMM convolution (conv_mfma.hip)
// ---------------------------------------------------------------------------------------------------
struct ConvArgs {
    const half_t* in;     // [B][H][W][in_cs]
    half_t* out;          // [B][H*r][W*r][out_cs]
    const half_t* res;    // optional residual, same indexing as out (may alias out)
    const half_t* wpk;    // packed weight fragments [wbatch][chunk][frag][lane][8]
    const float* bias;    // [nchunks*64] in packed output-channel order, or nullptr
    const float* bias_img; // [nchunks][256] fp32: the chunk's 64 biases then zeros (1-KiB LDS-DMA piece of conv3x3_sp)
    const half_t* zero;   // >= 256 B of zeros (out-of-image taps are redirected here)
    half_t* trash;        // >= 1 KiB write-only scratch (predicated-off stores of the branch-free epilogue land here)
    // FP16X3 (hi/lo split operands): partial products are combined through an fp32 side buffer
    half_t* out_lo;       // low part of the output activation ((v - hi) * 2^11), or nullptr
    const half_t* in_lo;  // acc_mode 4 only: low part of the input activation (same layout as `in`)
    int plane_w;          // conv3x3_sp only: wpk holds one weight set per PLANE ([B][72 fragments], nchunks = B), 64 output channels
    const half_t* side16; // acc_mode 3 (conv3x3_sp): the two low-order products, already summed, as fp16 in the OUTPUT layout (replaces acc32)
    // conv_mfma_kernel only: fused 1x1 tail (lite's last upsampler stage + its 48->1 conv).  The activated tile is not stored: each
    // lane dots its 16 channels with tail1_w, the two 32-channel halves of the chunk go to two fp32 partial planes [2][B][Ho][Wo]
    const float* tail1_w;  // [64] fp32 in output-channel order of the chunk, or nullptr
    float* tail1_out;
    const half_t* res_lo; // low part of the residual, or nullptr
    float* acc32;         // [B][H][W][nchunks*64] fp32 partial sums (pre-shuffle coordinates)
    int acc_mode;         // 0: none, 1: store acc, 2: acc32 += acc, 3: acc += acc32 * 2^-11 then epilogue,
                          // 4: all three split-precision products in ONE launch (1x1 convs, conv_mfma_kernel<1,3>): K segments
                          //    (w_lo, in), (w_hi, in_lo), (w_hi, in); the accumulator is scaled by 2^-11 after the second
    long long w_batch_stride;  // halfs between per-plane weight sets (0: shared)
    int B, H, W;
    int in_cs, out_cs;    // channel strides of in / out, in halfs
    int r;                // pixel-shuffle factor folded into the store (1: none)
    int nchunks;          // 64-output-channel chunks
    int G;                // persistent workgroups per chunk
    int px, py;           // patches along x / y
    float slope;          // PReLU / LeakyReLU slope (1: identity)
    float scale;          // multiplier applied before the activation (ARSB ScaleLayer); 1: none
    // fused tail (conv3x3_sp EPI 3): A fragments of the 64->1 tail conv and the planar fp32 per-tap sums [9][B][H*r][W*r]
    const half_t* tail_w;
    float* tplanes;
    int tail_form;        // fused tail output: 0 nine tap planes per phase (conv3x3_sp EPI 3/7, tapsum2 / tapsum<R>), 1 phase-class sums + aprons
                          // (conv3x3_rw EPI 3/7, r == 2, layout tailsum_layout(B, H, W) behind `tplanes`, gathered by tapsum4)
    int tail_split;       // fused tail: also split the activation operand (EPI 7; tail_w then holds eight fragments, the second four = fp16 weights in rows 16..24)
    int dbg;              // timing ablations (MOE_DBG env; results are wrong when set): 1 no patch DMA, 2 no MFMA, 4 no stores, 8 no epilogue
    // conv3x3_rw, PReLU epilogue, one chunk, r = 1 only: per-plane channel sums of the STORED (fp16) output, pool[b][slab][64] with
    // slab = 2 * workgroup + row half, zeroed by the caller, pool_slabs >= 2 * G (SEDN's fused block tail needs them, sedn_fuse)
    float* pool; int pool_slabs;
};

// Fused tail, "phase-class sums" form (tests/tailsum_model.py states the arithmetic).  One branch's buffer, fp32 ELEMENT offsets:
//   S  [4 phases][4 classes][B][H][W]      sums of the per-tap products that land on output pixels of one parity class, formed inside a patch
//   RA [4][2][B][py][W]                    row aprons: what patch row pyi exports to the vertically adjacent patch's edge row (classes (1, cj))
//   CA [4][2][B][H][px]                    column aprons (classes (ci, 1)),   CO [4][B][py][px]  the corner term of class (1, 1)
// H, W: the conv's INPUT size (HR = 2H x 2W), py / px: patches of kTileH x kTileW
struct TailSumLayout { unsigned S, RA, CA, CO, total; int py, px; };
__host__ __device__ inline TailSumLayout tailsum_layout(int B, int H, int W)
{
    TailSumLayout l;
    l.py = (H + kTileH - 1) / kTileH; l.px = (W + kTileW - 1) / kTileW;
    auto up = [](unsigned long long n) { return (unsigned)((n + 63) / 64 * 64); };
    l.S = 0;
    l.RA = l.S + up(16ull * B * H * W);
    l.CA = l.RA + up(8ull * B * l.py * W);
    l.CO = l.CA + up(8ull * B * H * l.px);
    l.total = l.CO + up(4ull * B * l.py * l.px);
    return l;
}
inline bool tailsum_fits(int B, int H, int W)      // 32-bit byte offsets into the whole buffer
{
    const unsigned long long el = 16ull * B * H * W + 8ull * B * ((H + kTileH - 1) / kTileH) * W + 8ull * B * H * ((W + kTileW - 1) / kTileW) + 4ull * B * ((H + kTileH - 1) / kTileH) * ((W + kTileW - 1) / kTileW) + 4 * 64;
    return el * 4 < (1ull << 32) - (1ull << 17);
}

// 1x1 convs of lite (conv1x1.hip): 64 (48 real) input channels, one chunk (r = 1) or four (r = 2, pixel shuffle folded into the store),
// fp16 or split operands, optionally the 48->1 tail conv folded in (one fp32 plane [B][2H][2W] of complete dot products)
struct Conv1x1Args {
    const half_t* in_hi; const half_t* in_lo;     // [B][H][W][64]; in_lo != nullptr selects the three-product form
    half_t* out_hi; half_t* out_lo;               // [B][H r][W r][out_cs] (not used with tail_out)
    const half_t* w_hi; const half_t* w_lo;       // packed A fragments [chunk][8] (pack_conv order)
    const float* bias;                            // [nchunks * 64] in packed output-channel order (zeros when the layer has none)
    const float* tail_w; float* tail_out;         // fused tail: [64] fp32 weights in the chunk's channel order, fp32 plane out
    float slope;                                  // PReLU slope (<= 1; 1: none)
    int B, H, W, r, nchunks, out_cs;
    int nks;                                      // k-slices of 16 input channels that are not all zeros: 3 for the 48-channel nets (option k48), else 4
};
bool launch_conv1x1(const Conv1x1Args& a, int max_groups, hipStream_t s);   // false: shape not compiled (caller uses conv_mfma_kernel)
hipError_t conv1x1_init();

// Two upsampler stages of MoeNet_lite2 (1x1 48 -> 192, PixelShuffle(2), PReLU, twice) + the folded 48 -> 1 tail in one launch, split operands (conv1x1_f2.hip):
// the tensor between the stages never exists; same bits as the two launches of conv1x1.hip
struct Conv1x1F2Args {
    const half_t* in_hi; const half_t* in_lo;     // [B][H][W][64]
    const half_t* wa_hi; const half_t* wa_lo;     // stage A: packed A fragments [chunk 4][8] (pack_conv order)
    const half_t* wb_hi; const half_t* wb_lo;     // stage B
    const float* bias_a; const float* bias_b;     // [256] each, packed output-channel order
    const float* tail_w; float* tail_out;         // [64] fp32 tail weights; fp32 plane [B][4H][4W]
    float slope_a, slope_b;                       // PReLU slopes (<= 1)
    int B, H, W;
};
bool launch_conv1x1_f2(const Conv1x1F2Args& a, int max_groups, hipStream_t s);   // false: not applicable
hipError_t conv1x1_f2_init();

void launch_conv_mfma(const ConvArgs& a, int taps, int nseg, hipStream_t s);
int conv_mfma_max_groups();  // persistent workgroups the device holds (1 per CU)
hipError_t conv_mfma_init(); // raise dynamic-LDS limits once per process
// 3x3 / 64-channel specialisation: one wave per SIMD with the epilogue software-pipelined into the MFMA stream (conv3x3_sp.hip)
bool launch_conv3x3_sp(const ConvArgs& a, hipStream_t s);   // false: epilogue variant not compiled, use another kernel
hipError_t conv3x3_sp_init();
// the upsampler convs with register-resident weights (conv3x3_rw.hip): PReLU (+ fused tail) epilogues only
bool launch_conv3x3_rw(const ConvArgs& a, hipStream_t s);   // false: not applicable, use conv3x3_sp
hipError_t conv3x3_rw_init();

// The last upsampler stage (3x3, 64 -> 256, + bias, PixelShuffle(2), PReLU) with the 64 -> 1 tail conv, all four phases in one workgroup (conv3x3_ps4.hip)
struct Ps4Args {
    const half_t* in;        // [B][H][W][64]
    const half_t* wpk;       // pack_conv fragments [phase 4][72][lane 64][8]
    const float* bias;       // [256] fp32 in packed output-channel order (ConvLayer::bias)
    const half_t* tail_w;    // eight A fragments of the tail conv (engine.cpp tail(): "<key>.frag")
    half_t* out;             // store form (not the last stage): [B][2H][2W][64] fp16, PReLU'd and pixel-shuffled; plane / apron / tail_w unused
    float* plane;            // out: [B][2H][2W] fp32, this branch's tail-conv sums without the terms that cross a 32-pixel column
    float* apron;            // out: [side 2][B][px][2H] fp32, those terms (side 0: for the column to the right, 1: to the left)
    float slope;             // PReLU slope (< 1)
    int B, H, W;             // the conv's input size
    int split;               // the tail conv's activation operand as hi + lo 2^-11 (MOE_PREC_MIXED, R branch)
};
bool launch_conv3x3_ps4(const Ps4Args& a, int max_groups, hipStream_t s);   // false: not applicable (caller keeps conv3x3_rw + tapsum4)
bool ps4_applicable(int B, int H, int W);                                    // the shape conditions of the launcher (planning)
// the ONE predicate of the fused-tail form: the engine plans its buffers with it and launch_conv3x3_ps4 refuses with it -- a condition added to one side only would
// turn a performance choice into a failed forward
inline bool ps4_tail_applicable(int B, int H, int W, float slope) { return slope < 1.f && ps4_applicable(B, H, W); }
size_t ps4_plane_bytes(int B, int H, int W);
size_t ps4_apron_bytes(int B, int H, int W);
hipError_t conv3x3_ps4_init();
// y = plane_0 + plane_1 + the column aprons of both (conv3x3_ps4.hip), cast to the caller's type
struct TailAddArgs {
    const float* p0; const float* p1;      // [B][H][W] fp32 (p1 may be nullptr)
    const float* a0; const float* a1;      // [2][B][px][H]
    void* y; int y_dtype; const long long* y_off;
    int B, H, W, px;                       // HR size; px: 32-pixel columns of the conv input (64 HR columns each)
    int vec_ok;                            // every output row start is 16-byte aligned
};
void launch_tailadd(const TailAddArgs& a, hipStream_t s);

// The upsampler stage of a x3 net (3x3, 64 -> 576, + bias, PixelShuffle(3), PReLU) with the 64 -> 1 tail conv: a workgroup = the three phases of one phase row
// (conv3x3_ps9.hip), three workgroups per range of rows
struct Ps9Args {
    const half_t* in;        // [B][H][W][64]
    const half_t* wpk;       // pack_conv fragments [phase 9][72][lane 64][8]
    const float* bias;       // [576] fp32 in packed output-channel order (ConvLayer::bias)
    const half_t* tail_w;    // eight A fragments of the tail conv (engine.cpp tail(): "<key>.frag")
    float* plane;            // out: [dy 3][B][3H][3W] fp32, S[dy][HR row][HR column] = the tail conv's sums over dx for tap row dy, terms crossing a 32-pixel column left out
    float* apron;            // out: [side 2][dy 3][B][px][3H] fp32, those terms (side 0: for the column to the right, 1: to the left)
    float slope;             // PReLU slope (< 1)
    int B, H, W;             // the conv's input size
    int split;               // the tail conv's activation operand as hi + lo 2^-11 (MOE_PREC_MIXED, R branch)
    int xcd_map;             // set by the launcher: the block -> (range, phase row) map
};
bool launch_conv3x3_ps9(const Ps9Args& a, int max_groups, hipStream_t s);   // false: not applicable (caller keeps conv3x3_sp + nine tap planes + tapsum<3>)
bool ps9_applicable(int B, int H, int W);
inline bool ps9_tail_applicable(int B, int H, int W, float slope) { return slope < 1.f && ps9_applicable(B, H, W); }      // the ONE predicate (see ps4_tail_applicable)
size_t ps9_plane_bytes(int B, int H, int W);
size_t ps9_apron_bytes(int B, int H, int W);
hipError_t conv3x3_ps9_init();
// y[Y][X] = sum over both branches of S0[Y - 1][X] + S1[Y][X] + S2[Y + 1][X] + their column aprons (conv3x3_ps9.hip), cast to the caller's type
struct TailAdd3Args {
    const float* p0; const float* p1;      // [3][B][H][W] fp32 (p1 may be nullptr)
    const float* a0; const float* a1;      // [2][3][B][px][H]
    void* y; int y_dtype; const long long* y_off;
    int B, H, W, px;                       // HR size; px: 32-pixel columns of the conv input (96 HR columns each)
    int vec_ok;                            // every output row start is 16-byte aligned
};
void launch_tailadd3(const TailAdd3Args& a, hipStream_t s);

// One fused ARSB  y = x + conv_2(PReLU(conv_1(x)))  (arsb32c.hip; conv_2's weights carry the ScaleLayer factor)
struct ArsbArgs {
    const half_t* x_hi; const half_t* x_lo;   // stream in  [B][H][W][64] (x_lo: low part in units of 2^-11, or nullptr)
    half_t* y_hi; half_t* y_lo;               // stream out, NOT aliasing x (neighbouring patches read x's halo)
    const half_t* w1; const half_t* w2;       // packed A fragments of conv_1 / conv_2 (one 64-channel chunk each, 72 fragments)
    const half_t* zero;                       // >= 256 B of zeros
    float slope;                              // PReLU slope of conv_1 (<= 1)
    int B, H, W;
    int px, py;                               // set by the launcher
    int cin;                                  // channels that carry data (0 = 64): arsb32c leaves the fourth k-slice out for the 48-channel nets
    int drop_lo;                              // y_lo is never read: its stores are issued against an empty range (the last ARSB of an SR net)
    unsigned long long* trace;                // -DARSB_TRACE builds only: s_memtime stamps [workgroup < 8][patch < 16][wave 4][slot 40]
};
// arsb32c.hip: v_mfma_f32_32x32x16_f16, four waves in lock-step, both convs' weights resident (w1 / w2 in the pack_conv fragment order, ConvLayer::w_hi), ten output rows
// per patch, the two last m rows of a patch stay in LDS for the patch below.  false: not applicable (the caller runs the two convs)
bool launch_arsb32c(ArsbArgs a, int max_groups, hipStream_t s);
hipError_t arsb32c_init();

// One 3x3 64->64 conv with split operands, three products in one launch (conv64_x3.hip); weights in the fused-ARSB order
struct ConvX3Args {
    const half_t* in_hi; const half_t* in_lo;     // [B][H][W][64], low part in units of 2^-11
    half_t* out_hi; half_t* out_lo;               // may alias res (never in)
    const half_t* res_hi; const half_t* res_lo;   // residual (both or none)
    const half_t* w_hi; const half_t* w_lo;       // [wave 4][fragment 18][lane 64][8]
    const half_t* zero;
    float slope;                                  // PReLU slope (1: none; <= 1)
    int B, H, W;
    int px, py;                                   // set by the launcher
    // optional (plain epilogue only): per-plane channel sums of the output, the global average pool of lite's FRM / LB (MoeNet_lite2.py:16-20,
    // models.py:274) without a second pass over the tensor: pool[b][workgroup][64], zeroed by the caller, pool_slabs >= workgroups
    float* pool; int pool_slabs;
    // optional (with a residual, conv64_x3 only): out = gate[b][c] * conv + residual; gate = [2][B][64] fp32: g, then 1 / g (frm_pre_kernel)
    const float* gate;
    // conv64_q8.hip only: w_hi as A fragments of v_mfma_f32_32x32x16_f16 (ConvLayer::w_hi, pack_conv order), w_hi 2^8 and w_lo 2^8 as fp8 e4m3
    // A fragments of v_mfma_scale_f32_32x32x64_f8f6f4: [tap 9][channel half 2][lane 64][32 bytes]
    const half_t* wq_hi16; const unsigned char* wq_hi8; const unsigned char* wq_lo8;
    // conv64_q8.hip only: in_lo (and res_lo) / out_lo hold fp8 e4m3 words -- (v - fp16(v)) 2^11 / 4, one byte a channel, [B][H][W][64] -- instead of fp16
    int in8, out8;
};
bool launch_conv64_x3(ConvX3Args a, int max_groups, hipStream_t s);   // false: not applicable (caller uses the three-launch form)
hipError_t conv64_x3_init();
// the same conv with its two correction products on fp8 operands (conv64_q8.hip); false: not applicable (caller uses conv64_x3)
bool launch_conv64_q8(ConvX3Args a, int max_groups, hipStream_t s);
hipError_t conv64_q8_init();
// the chain form of that layer (fp8 low parts in) streamed down a column by one fp16 wave + one fp8 wave per workgroup (conv64_sq.hip); false: not applicable (caller uses conv64_q8)
bool launch_conv64_sq(ConvX3Args a, int max_groups, hipStream_t s);
hipError_t conv64_sq_init();
// conv64_s.hip: SEDN's fused block tail (per-plane weights, LeakyReLU, residual) streamed down 32-pixel columns with the plane's weights in registers; false: not applicable
bool launch_conv64_s(const ConvArgs& a, int max_groups, hipStream_t s);
hipError_t conv64_s_init();
// arsb_sq.hip: ONE exact ARSB of such a chain in one launch -- conv_1's rows stay in LDS (producer / consumer wave pairs); false: not applicable (caller runs the two convs)
struct ArsbSqArgs {
    const half_t* x_hi; const unsigned char* x_lo8;      // stream in: [B][H][W][64] fp16 + the fp8 low words
    half_t* y_hi; void* y_lo;                            // stream out, NOT aliasing x; y_lo: fp8 words (out8) or fp16
    const half_t* w16[2]; const unsigned char* wh8[2]; const unsigned char* wl8[2];      // conv_1, conv_2: ConvLayer::w_hi, wq_hi8, wq_lo8
    float slope;                                         // PReLU slope of conv_1 (<= 1)
    int B, H, W, out8;
};
bool launch_arsb_sq(const ArsbSqArgs& a, int max_groups, hipStream_t s);
hipError_t arsb_sq_init();

// Workgroup count of a launch whose epilogue pools per plane into per-workgroup slabs (conv3x3_rw EPI 4, conv64_x3 EPI 3): a multiple or a divisor
// of the patches per plane P, so that the patch -> workgroup map (item % G with item = plane * P + k) -- and with it every slab's content and
// summation order -- does not depend on how many planes share the launch.  (The SE / FRM gates feed fp16-rounded weights and multipliers: with
// batch-dependent sums the last bit of a gate moved and SEDN's output with it by up to 5e-4; now a tile's result is independent of its launch set.)
// P above max_groups with awkward factors (17 x 31 patches: largest divisor 31 of 256 CUs; a prime: 1) would collapse the launch onto a few CUs: pooled_groups_ok
// tells the caller to leave the pooling to the separate pass (launch_pool_partial / sedn_xsum), which costs one read of the tensor instead of up to 10x the conv.
inline int pooled_groups(long long P, long long items, int max_groups)
{
    long long G = 1;
    if (P <= max_groups) G = P * (max_groups / P);
    else for (long long d = max_groups; d >= 1; --d) if (P % d == 0) { G = d; break; }
    return (int)(G < items ? G : items);
}
inline bool pooled_groups_ok(long long P, long long items, int max_groups)
{
    const long long full = items < max_groups ? items : max_groups;
    return 2ll * pooled_groups(P, items, max_groups) >= full;      // at least half of the workgroups a plain launch would use
}

struct DirectConvArgs {
    const half_t* in; half_t* out; const half_t* res;
    const float* w;       // plain fp32 OIHW weights [cout][cin][k][k] (original channel counts)
    const float* bias;    // [cout] original order or nullptr
    long long w_batch_stride;
    int B, H, W, in_cs, out_cs, cin, cout, k, r;
    float slope, scale;
};
void launch_conv_direct(const DirectConvArgs& a, hipStream_t s);

// ---------------------------------------------------------------------------------------------------
// HBM-bound kernels (misc_kernels.hip)
// ---------------------------------------------------------------------------------------------------
struct StemArgs {
    const void* x; int x_dtype;      // MOE_F32 / MOE_F16
    const long long* x_off;          // device [B] element offsets, or nullptr: b * sB
    long long sB, sH, sW;
    const float* w;                  // [taps][64] fp32 (padded channels zero)
    float slope;
    half_t* out;                     // [B][H][W][64]
    half_t* out_lo;                  // FP16X3 low part or nullptr
    int B, H, W, taps;
    int out_lo8;                     // the low part as fp8 e4m3 words of lo / 4, one byte a channel (what conv64_q8 with in8 reads: ConvX3Args)
    // lite (taps == 1; MoeNet_lite2.py:40-41): conv_input2(PReLU(conv_input(x))) is x times a fixed vector -- P for x >= 0, Q for x < 0 (engine.cpp: "stem.p2") -- so the stem
    // writes that tensor as well and the 48 -> 48 1x1 conv is never launched.  w2 = [2][64] fp32 (P, Q) or nullptr; out2 / out2_lo as out / out_lo (out2_lo may be nullptr)
    const float* w2; half_t* out2; half_t* out2_lo;
};
void launch_stem(const StemArgs& a, hipStream_t s);

struct TailArgs {
    const half_t* in0; const half_t* in1;  // [B][H][W][64]; in1 may be nullptr
    const half_t* w0; const half_t* w1;    // [taps][64] fp16
    const half_t* in0_lo; const half_t* in1_lo;  // FP16X3 low parts (all four or none)
    const half_t* w0_lo; const half_t* w1_lo;
    int in1_lo8;                           // in1_lo holds fp8 e4m3 words of lo / 4, one byte a channel (what the stem wrote for conv_input2: NetDN on the fp8-correction chain); 3x3 taps only
    const void* skip; int skip_dtype;      // optional 1-channel skip (SEDN: + x), strided like the stem input
    const long long* skip_off; long long skip_sB, skip_sH, skip_sW;   // skip_off nullptr: b * skip_sB
    void* y; int y_dtype;                  // MOE_F32 / MOE_F16
    const long long* y_off;                // device [B] element offsets of each output plane, or nullptr: b*H*W
    int B, H, W, taps;
};
void launch_tail(const TailArgs& a, hipStream_t s);

// y = sum over the nine taps of both branches' planar partial sums (fused-tail path):  t0/t1 [9][B][H][W] fp32
struct TapSumArgs {
    const float* t0; const float* t1;
    void* y; int y_dtype; const long long* y_off;
    int B, H, W;   // HR size
    int r;         // pixel-shuffle factor of the producing conv (2 or 3)
    int form;      // 0: nine tap planes per phase, 1: phase-class sums + aprons (r == 2; t0 / t1 are tailsum_layout buffers)
    int vec_ok;    // r == 2 and every output row start is 16-byte aligned: the 8-outputs-per-thread kernel may be used
};
void launch_tapsum(const TapSumArgs& a, hipStream_t s);

// y = sum of the four partial planes of the fused 1x1 tail (two branches x two channel halves), each [B][H][W] fp32
struct Tail1SumArgs {
    const float* p0; const float* p1;      // [nparts][B][H][W] per branch
    int nparts;                            // 2: two 32-channel halves per branch (conv_mfma_kernel), 1: complete dot products (conv1x1.hip)
    void* y; int y_dtype; const long long* y_off;
    int B, H, W;
    // lite, fp16 input: the U branch's value from the table of the net (engine.cpp, moe_net::lut: [256 r][256 r] fp32, the entry of input pattern v and phase (py, px) at
    // ((v >> 8) r + py) 256 r + (v & 255) r + px) instead of p1 (nullptr then); x = the forward's fp16 input (strides in elements, x_off = device [B] offsets or nullptr: b sB)
    const float* lut; int r; const void* x; const long long* x_off; long long sB, sH, sW;
    int vec_ok;                            // every output plane starts 16-byte aligned (the four-outputs-per-thread table kernel)
};
void launch_tail1sum(const Tail1SumArgs& a, hipStream_t s);

// per-plane channel sums: in [B][HW][C] fp16 -> partial [B][nslab][C] fp32
void launch_pool_partial(const half_t* in, const half_t* in_lo, float* partial, int B, long long HW, int C, int nslab, hipStream_t s);

struct SednSeArgs {   // _Conv_Block squeeze-excite (models.py:198-213) + per-plane scaling of the 1x1 `trans` weights
    const float* partial; int nslab; long long HW;
    const float* w_down;  // [16][256]
    const float* w_up;    // [256][16]
    const float* trans_pk32;  // trans weights in packed fragment order, fp32 [nfrag*512]
    half_t* trans_out;        // [B][nfrag*512] fp16: trans * sigmoid-gate[cin]
    half_t* trans_out_lo;     // low parts or nullptr
    int B, nfrag;
};
void launch_sedn_se(const SednSeArgs& a, hipStream_t s);

// SEDN fused block tail: trans(g * conv256(x)) == conv(W_t diag(g) W_256, x) -- one 3x3 64->64 conv with per-plane weights.
struct SednFuseArgs {
    const half_t* x;          // [B][H][W][64] input of rblock.4
    float* partial;           // [B][nslab][5][64]: total, first row, last row, first column, last column sums
    const float* pooled; int pooled_slabs;   // optional: the totals, already formed by the producing conv ([B][pooled_slabs][64]); sedn_xsum then only visits the border pixels
    int pooled_count;                        // ... of which the first pooled_count slabs of every plane are written by that conv (all of them, every launch: no memset)
    int nslab, B, H, W;
    const float* w256t;       // [576][256] fp32, k = tap*64 + ci   (rblock.4 weights, transposed)
    const float* w256;        // [256][576]
    const float* wt;          // [64][256]  trans weights
    const float* w_down;      // [16][256]
    const float* w_up;        // [256][16]
    float* gate;              // [B][256]: the channel means (sedn_fmean -> sedn_weff)
    half_t* weff;             // [B][72 fragments][64 lanes][8] packed A fragments of the fused conv
};
void launch_sedn_fuse(const SednFuseArgs& a, hipStream_t s);

struct FrmArgs {     // FRM gate (models.py:270-287) then out = t*gate + x   (MoeNet_lite2.py:16-20)
    const float* partial; int nslab; long long HW;
    const float* w0; const float* b0;   // [3][64], [3]
    const float* w2; const float* b2;   // [64][3], [64]
    const half_t* t; const half_t* x; half_t* out;  // [B][HW][64]
    const half_t* t_lo; const half_t* x_lo; half_t* out_lo;  // FP16X3 low parts or nullptr
    float* gate;                         // [B][64] scratch
    int B;
};
void launch_frm(const FrmArgs& a, hipStream_t s);
// The FRM gate of an LB BEFORE its conv_2 runs (round 6).  conv_2 has no bias and no activation, so the pooled mean the gate needs is linear in conv_2's INPUT m:
//   mean_t[co] = 1/HW  sum_tap sum_ci W2[co][ci][tap] S_tap[ci],   S_tap = the sum of m over the plane shifted by the tap (total minus border rows / columns, zero padding)
// (what sedn_fmean does for SEDN).  conv_1's epilogue forms the totals of m (conv64_x3 EPI 4), this kernel visits the border, applies W2 and the gate's two 1x1 layers
// (models.py:270-287) and writes g and 1 / g; conv_2's epilogue then stores g * conv + x (conv64_x3 EPI 5) -- frm_apply's pass over six tensors is gone.
struct FrmPreArgs {
    const half_t* m; const half_t* m_lo;         // [B][H][W][64] conv_2's input (low part or nullptr)
    const float* partial; int nslab;             // the totals of m: [B][nslab][64]
    const float* c2t;                            // conv_2's weights, fp32, [tap*64 + ci][co 64]
    const float* w0; const float* b0;            // [3][64], [3]
    const float* w2; const float* b2;            // [64][3], [64]
    float* gate;                                 // out: [2][B][64]: g, then 1 / g
    int B, H, W;
};
void launch_frm_pre(const FrmPreArgs& a, hipStream_t s);

struct StitchArgs {
    const float* tiles; const long long* tile_off;   // device
    const int* row_first; const int* row_cnt;        // device [out_h]: first covering tile row, count
    const int* col_first; const int* col_cnt;        // device [out_w]
    const int* row_tab; const int* col_tab;          // device [step][4] = (first, solid, origin, extent)
    const float* ramp;                               // device [pad_sc]
    void* out; int out_dtype;
    int C, out_h, out_w, step_w;
    int row_lo;                                      // first tile row the fold visits (0; a band: its first tile row)
    int y0, rows;                                    // the canvas rows [y0, y0 + rows) are folded into `out` = (C, rows, out_w): the whole canvas (0, out_h) or a band (moe_stitch_band)
};
void launch_stitch(const StitchArgs& a, hipStream_t s);

// blend.hip: the two blend() calls + slice-assign of doCrop's loop body for ONE tile, in the canvas dtype (moe_blend_tile)
struct BlendTileArgs {
    const void* r; void* canvas; const void* ramp;   // tile result (C planes, unit column stride), canvas (C planes), ramp [pad_sc]: all of the canvas dtype, device
    long long r_sC, r_sH, c_sC, c_sH;                // strides in elements
    int C, rh, rw;                                   // window extent = rows / columns of the tile result that are used (opt.unpad)
    int top_sc, left_sc;                             // window origin in the canvas
    int r0, c0;                                      // first assigned row / column of the window (lt - pad_sc, or 0 without a band)
    int lt_h, lt_w;                                  // first un-blended row / column (band = [r0, lt_h) / [c0, lt_w); equal to r0 / c0 without a band)
};
void launch_blend_tile(const BlendTileArgs& a, bool f16, hipStream_t s);

// One record of the inter-rank wire format (moe_wire_pack / moe_wire_unpack; the public moe_wire_rec has the same layout): a tile (or strip) of C planes of
// th x tw fp32 values <-> [fp16 image of all values | fp32 seam rows | fp32 seam columns], offsets in 4-byte words.
struct WireRec {
    long long tile_off, wire_off;
    int C, th, tw;
    int ra0, ra1, rb0, rb1;          // seam rows [ra0, ra1) and [rb0, rb1), tile-local, ra1 <= rb0
    int ca0, ca1, cb0, cb1;          // seam columns likewise
    int pad_;
};
long long wire_rec_words(const WireRec& r);
void launch_wire(bool pack, float* tiles, unsigned* wire, const WireRec* recs, int n, long long max_elems, hipStream_t s);

void launch_to_float(const void* src, int src_dtype, float inv_or_div, bool divide, int H, int W, int C, void* dst, int dst_dtype, hipStream_t s);
void launch_to_output(const void* src, int src_dtype, int H, int W, int C, float quant, void* dst, int dst_dtype, hipStream_t s);
// (C, H, W) -> (C, h, w); mode 0 nearest, 1 bilinear, 2 bicubic (torch F.interpolate semantics, align_corners = False)
void launch_resize(const void* src, void* dst, int dtype, int C, int H, int W, int h, int w, int mode, hipStream_t s);
// max |a - b| over n floats, atomically folded into *out as the bits of a non-negative float (zero it first)
void launch_maxabsdiff(const float* a, const float* b, long long n, unsigned* out, hipStream_t s);
void launch_nhwc_to_nchw_f32(const half_t* in, const half_t* in_lo, float* out, int B, int H, int W, int cs, int C, hipStream_t s);
