/* moephoto_amd.h -- C ABI of libmoephoto_amd.so: the MI355X (gfx950) engine behind MoePhoto's
 * tiled SR / denoise hot path.  Plain pointers and sizes only; no torch types.
 *
 * Every entry point replaces one piece of the reference's Python/PyTorch path (paths relative to
 * the MoePhoto repo; see INTEGRATION.md for the ctypes stub a maintainer adds):
 *
 *   moe_net_create        the model constructors of the plugin tables' "constructor slot":
 *                         Net2x/Net3x/Net4x/NetDN/SEDN (python/models.py:125-223) and
 *                         MoeNet_lite2.Net(upscale) (python/MoeNet_lite2.py:22-37), as selected by
 *                         runSR.mode_switch (python/runSR.py:10-24) / runDN.mode_switch (runDN.py:10-21)
 *   moe_net_set_param     nn.Module.load_state_dict, one call per state-dict entry
 *                         (python/imageProcess.py:319-328: model.load_state_dict(weights))
 *   moe_net_finalize      castModel / model.to(dtype, device) (python/imageProcess.py:309-317)
 *   moe_net_forward       modelCached(x): the torch.nn forward invoked from Option.__call__
 *                         (python/imageProcess.py:391-395) inside doCrop's tile loop (:164-166)
 *   moe_plan_*            the tile planner prepare/getAnchors (python/imageProcess.py:19-35,73-118)
 *   moe_stitch            the blend + slice-assign part of doCrop (python/imageProcess.py:120-131,167-170)
 *   moe_run_plan          doCrop as a whole (python/imageProcess.py:157-172): tile gather -> net -> stitch,
 *                         device resident, batched over same-shaped tiles
 *   moe_to_float/_output  toTorch / toOutput (python/imageProcess.py:245-263)
 *
 * All functions return 0 on success or a negative MOE_E* code; moe_last_error() returns a
 * thread-local description (the Python wrapper raises RuntimeError / MemoryError from it, matching
 * worker.enhance's error envelope, python/worker.py:52-74).  Device work is enqueued on the caller's
 * hipStream_t (passed as void*) and never synchronised here: the reference runs on torch's current
 * stream in one thread (python/worker.py:89-93), so must we.
 */
#ifndef MOEPHOTO_AMD_H
#define MOEPHOTO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOE_ABI_VERSION 4      /* 4: moe_net_forward_ex / MOE_FWD_INPUT_SINCE_PREV (round 6); 3: moe_net_calibrate / moe_net_exact_blocks, MOE_PREC_AUTO measures the checkpoint at finalize, moe_blend_tile (round 5); 2: MOE_PREC_AUTO, moe_net_resolved_precision, moe_plan_rows / moe_stitch_band, moe_plan_seams / moe_wire_* (round 4); 1 also lacked a bump for moe_net_set_option / moe_device_info / moe_stitch_dev */

/* error codes */
#define MOE_OK 0
#define MOE_EINVAL (-1)   /* bad argument / unknown parameter name / shape mismatch */
#define MOE_ENOMEM (-2)   /* host or device allocation failed, or a tile does not fit (-> MemoryError) */
#define MOE_EHIP (-3)     /* a HIP runtime call failed (no device, launch failure, ...) */
#define MOE_ESTATE (-4)   /* call order violated (forward before finalize, missing parameters) */

/* architectures: the constructor slot of the plugin tables */
#define MOE_ARCH_NET2X 0  /* models.Net2x  : a2/p2 */
#define MOE_ARCH_NET3X 1  /* models.Net3x  : a3/p3 */
#define MOE_ARCH_NET4X 2  /* models.Net4x  : a4/p4 */
#define MOE_ARCH_NETDN 3  /* models.NetDN  : dn_lite5/10/15 */
#define MOE_ARCH_SEDN 4   /* models.SEDN   : l15/l25/l50 */
#define MOE_ARCH_LITE 5   /* MoeNet_lite2.Net(upscale = `scale` in {2,4,8}) */

/* element types of caller-visible buffers */
#define MOE_F32 0
#define MOE_F16 1
#define MOE_U8 2
#define MOE_U16 3

/* arithmetic of the MFMA convolutions (weights/activations as operands; accumulate is always fp32,
 * the 1-channel stem and the final branch sum are always fp32) */
#define MOE_PREC_FP16 0        /* fp16 operands, one MFMA pass: the reference's GPU fp16 mode (config.fp16) */
#define MOE_PREC_FP16X3 1      /* hi/lo split operands, three MFMA passes: ~fp32 products */
#define MOE_PREC_DEBUG_DIRECT 2 /* slow scalar device convolution, fp32 accumulate; kernel debugging only */
#define MOE_PREC_MIXED 3       /* Net2x/3x/4x, NetDN: fp16 operands; the trunk stream x + s*conv2(..) of the six ARSBs
                                * (python/models.py:76-80) and the stem output are carried as hi + lo pairs (~fp32),
                                * conv_input2 and the first moe_net_set_exact_blocks() ARSBs use split operands, the 64->1
                                * tail convs see their weights to ~22 bits: <= 1e-3 vs the fp32 reference on every input
                                * class at ~1.1-1.2x the FP16 time */

#define MOE_PREC_AUTO 4        /* the family's default: MIXED for Net2x/3x/4x and NetDN, FP16 for SEDN, FP16X3 for lite -- the arithmetic that
                                * holds 1e-3 max-abs against the reference's fp32 CPU path on every input class.  What a drop-in caller passes
                                * (INTEGRATION.md section 1): the reference's own dtype policy is one line too (python/imageProcess.py:309-317) */

typedef struct moe_net moe_net;
typedef struct moe_plan moe_plan;

const char* moe_last_error(void);
int moe_abi_version(void);
/* number of visible HIP devices (0 when none); never fails */
int moe_device_count(void);

/* info[0..7] = compute units, max engine clock (kHz), memory clock (kHz), memory bus width (bits), L2 bytes, total memory bytes,
 * wall-clock rate (kHz), LDS bytes per CU -- what bench.py derives the MFMA / HBM peaks of the roofline from
 * (the reference reads its device through NVML, python/readgpu.py; config.py:61-71) */
int moe_device_info(int device, int64_t info[8]);

/* ---- model ------------------------------------------------------------------------------------ */
int moe_net_create(int arch, int scale, moe_net** out);
void moe_net_destroy(moe_net* net);
/* scale factor per side (1 for denoisers) */
int moe_net_scale(const moe_net* net);
/* number of state-dict entries this architecture expects, and the i-th name/shape
 * (shape receives up to 4 dims, *ndim their count) */
int moe_net_num_params(const moe_net* net);
int moe_net_param_info(const moe_net* net, int index, const char** name, int64_t shape[4], int* ndim);
/* one load_state_dict entry: fp32, C-contiguous, host memory; copied */
int moe_net_set_param(moe_net* net, const char* name, const float* data, const int64_t* shape, int ndim);
/* pack + upload weights to HIP device `device`; strict like load_state_dict (every parameter set).
 * May be called again to move / change precision.  precision: MOE_PREC_AUTO for the family's default (what a drop-in caller
 * passes), or a specific arithmetic (MIXED is refused for SEDN / lite, which have no such recipe). */
int moe_net_finalize(moe_net* net, int device, int precision);
/* what `precision` resolves to for this net's family (MOE_PREC_AUTO -> FP16 / FP16X3 / MIXED; anything else -> itself).  After a finalize with MOE_PREC_AUTO:
 * what that finalize settled on for the loaded weights (see moe_net_calibrate). */
int moe_net_resolved_precision(const moe_net* net, int precision);
/* The precision policy for checkpoints the build has never seen (the reference's contract is "load any state dict, get the fp32 answer":
 * python/imageProcess.py:319-334; its own dtype policy is castModel, :309-317).  Net2x/3x/4x and NetDN under MOE_PREC_MIXED run their first `blocks` ARSBs with
 * split operands; how many a checkpoint needs depends on how wide its trunk swings.  moe_net_calibrate measures it on the device: uniform uint8-noise tiles (12 seeds
 * x 3 planes of 256 x 256, the tile size that ships; compared on the device) through the exact arithmetic (FP16X3) and through MIXED with blocks = the architecture's default .. 6.  The worst
 * max-abs difference of that sample times 1.10 -- the inflation observed between such a sample and the worst tile of full frames -- is the PREDICTED worst tile of a
 * full frame; *blocks = the smallest count whose prediction is <= target (target <= 0: 8.5e-4; the architecture's default count is kept up to 5 % above it, so that a
 * zoo key next to the target does not flip with the device or driver), *err = that prediction; *blocks = -1 when six blocks do not reach it (*err = what they reach).
 * The count is kept (moe_net_exact_blocks) until a parameter changes or moe_net_set_exact_blocks overrides it.  moe_net_finalize(MOE_PREC_AUTO) runs this by itself,
 * once per checkpoint, and finalizes in MOE_PREC_FP16X3 when no count reaches the target: a drop-in caller needs no extra line.  SEDN / lite: *blocks = 0, nothing
 * is measured (their AUTO arithmetic has no such knob).  Synchronises `stream`; ~0.2-0.4 s.  (moe_net_finalize runs its measurement on a private non-blocking stream.) */
int moe_net_calibrate(moe_net* net, double target, int* blocks, double* err, void* stream);
/* the count of split-operand ARSBs the next forward runs with (0 when the net is not in MOE_PREC_MIXED) */
int moe_net_exact_blocks(const moe_net* net);
/* device bytes of scratch a forward of B planes of h x w needs (allocated lazily, grow-only, owned by the net) */
int64_t moe_net_workspace_bytes(const moe_net* net, int B, int h, int w);
/* largest tile (pixels of one input plane) a forward accepts: the convolution kernels address their tensors with 32-bit byte
 * offsets.  Batches of planes beyond that range are split into several launch sets internally; a single larger plane is
 * refused with MOE_ENOMEM (the planner's "tile does not fit" convention, python/imageProcess.py:58-59). */
int64_t moe_net_max_tile_pixels(const moe_net* net);
/* y[b] = Net(x[b]),  x: B planes of h x w (h, w >= 1), element (b,i,j) at x + x_off[b] + i*sH + j*sW
 * (strides in elements; x_off == NULL means b*sB), dtype MOE_F32/MOE_F16.  y: B contiguous planes of
 * (scale*h) x (scale*w), plane b at y + (y_off ? y_off[b] : b*scale*h*scale*w), dtype MOE_F32/MOE_F16.
 * x_off / y_off are HOST arrays of B element offsets.  Asynchronous on `stream`. */
int moe_net_forward(moe_net* net, const void* x, int x_dtype, int B, int h, int w,
                    int64_t sB, int64_t sH, int64_t sW, const int64_t* x_off,
                    void* y, int y_dtype, const int64_t* y_off, void* stream);
/* moe_net_forward with flags.  MOE_FWD_INPUT_SINCE_PREV: the caller states that x was COMPLETE on `stream` when the previous forward of this net was enqueued on it --
 * true of the reference's tile loop, whose inputs are slices of one padded image that exists before the loop (python/imageProcess.py:157-170: s = x[..., top:bottom,
 * left:right]; r = opt(s); blend; assign).  Small forwards of the SR nets (up to four planes of 256 x 256, no offset tables, option overlap_calls) then alternate between
 * two internal (stream, workspace) sets: forward k+1 starts beside forward k instead of behind it and behind the caller's blend of tile k.  What the caller sees is
 * unchanged: y is written by the forward's LAST kernel, which waits for everything enqueued on `stream` before this call, and `stream` waits for the forward before the
 * call returns, so the next thing the caller enqueues (its blend) finds y complete.  Without the flag, for other shapes / nets, for the first call of a burst (or after
 * a call on another stream) and under stream capture this IS moe_net_forward.  The torch wrapper (moephoto_amd/models.py) sets the flag when x is a view of the same
 * live storage, at the same version counter, as the previous call's input. */
#define MOE_FWD_INPUT_SINCE_PREV 1u
int moe_net_forward_ex(moe_net* net, const void* x, int x_dtype, int B, int h, int w,
                       int64_t sB, int64_t sH, int64_t sW, const int64_t* x_off,
                       void* y, int y_dtype, const int64_t* y_off, void* stream, unsigned flags);
/* Live kernel timing for the roofline report: bracket the MFMA-conv launches of every layer whose key contains one of the
 * comma-separated `layer_substrings` (e.g. "up1,c2_": the 64->256 upsampler convs at 2x resolution, and conv_2 of the ARSBs)
 * with hipEvents on the launch stream.  NULL / "" disables.  moe_net_get_profile_at waits for the recorded events of the
 * index-th substring and returns their summed kernel time, the number of launches and their algorithmic FLOPs
 * (2*pixels*Cout*Cin*taps, real channel counts); moe_net_get_profile = index 0, and resets the recording. */
int moe_net_set_profile(moe_net* net, const char* layer_substrings);
int moe_net_get_profile_at(moe_net* net, int index, double* total_ms, int64_t* launches, double* flops);
int moe_net_get_profile(moe_net* net, double* total_ms, int64_t* launches, double* flops);
/* MOE_PREC_MIXED only: how many leading ARSBs run with split operands (0..6; -1 = the calibrated count of these weights if there is one, else the
 * architecture's default: Net2x 4, Net3x 2, Net4x 1, NetDN 1).  Takes effect at the next forward. */
int moe_net_set_exact_blocks(moe_net* net, int blocks);
/* Kernel-form switches of one net, for A/B measurements and the parity tests that compare forms of one layer in-process
 * ("sp_impl" = "auto" | "rw" | "sp", "arsb_fuse" / "x3_fuse" / "conv1x1" / "fuse_tail" / "sedn_fuse" / "pool_fuse" = "0" | "1",
 * "tail_split" = "0" | "r" | "ru", "tail_form" = "sums" | "planes", "conv_impl" = "sp" | "v1", "tiles_per_batch", "max_groups",
 * "k48" = "0" | "1", "repeat" = "<layer key>:<n>" (measurement: the matching bracketed launches are issued n times -- tools/kernel_power.py), "x3_impl" = "auto" | "x3" | "q8" (the split-operand layers' kernel: three fp16 products, or the two
 * corrections on fp8 operands; auto = q8 for the SR nets), "lo8" = "on" | "off" (fp8 low parts between conv64_q8 layers)).
 * Defaults come from the MOE_* environment variables of the same names ONCE, at moe_net_create; the forward path itself reads no
 * environment.  The reference has no such switches: its forward is torch.nn (python/imageProcess.py:391-395). */
int moe_net_set_option(moe_net* net, const char* key, const char* value);
/* keep fp32 copies of named intermediates during forwards (slow; debugging / layer-by-layer parity only) */
int moe_net_set_debug(moe_net* net, int enable);
/* copy a named intermediate of the LAST forward to host as fp32 NCHW (debug / layer-by-layer parity);
 * synchronises the stream.  Returns the element count written (<= capacity) or a negative error. */
int64_t moe_net_debug_tap(moe_net* net, const char* tap, float* host, int64_t capacity, int64_t shape[4], void* stream);

/* ---- tile planner (host only, no device needed) ------------------------------------------------ */
/* shape = (C, H, W) of the image as handed to doCrop (before planes-as-batch); ram / ram_coef as in
 * Option.ramCoef & config.calcFreeMem(); cropsize 0 = unlimited */
int moe_plan_create(const int64_t shape[3], double ram, double ram_coef, int pad, int scale, int align,
                    int cropsize, moe_plan** out);
void moe_plan_destroy(moe_plan* plan);
/* info[0..11] = n_tiles, step_h, step_w, out_h, out_w, pad_h_to, pad_w_to, pad_sc, tile_h, tile_w, clip_h, clip_w */
int moe_plan_info(const moe_plan* plan, int64_t info[12]);
/* tiles[k*8 .. k*8+7] = (top, bottom, left, right, topT, leftT, bsc, rsc) in raster order */
int moe_plan_tiles(const moe_plan* plan, int32_t* tiles);
/* blend ramp b[k] = sigmoid(9*(k/padSc - .5)), k < padSc, fp32 */
int moe_plan_ramp(const moe_plan* plan, float* ramp);
/* per tile row i (moe_plan_info: step_h of them) four ints: first output row the tile row writes, first un-blended row S(i) ("solid"), output row of the
 * tiles' own row 0, height of the tiles' output extent -- what the stitch kernel folds with (python/imageProcess.py:120-131,167-170) */
int moe_plan_rows(const moe_plan* plan, int32_t* rows);

/* ---- device-side stitch / whole-image run ------------------------------------------------------ */
/* Fold the per-tile results into the canvas exactly as doCrop's sequential blend does.
 * tiles_dev: device buffer holding every tile's result as C contiguous fp32 planes of
 * ((bottom-top)*scale) x ((right-left)*scale); tile k starts at element tile_off[k] (HOST array; NULL = the
 * default layout of moe_plan_tile_offsets).
 * out: (C, out_h, out_w) contiguous, dtype MOE_F32/MOE_F16. */
int moe_stitch(const moe_plan* plan, int device, const float* tiles_dev, const int64_t* tile_off, int C,
               void* out, int out_dtype, void* stream);
/* The same with the tile offsets already on the device (n_tiles int64 elements): no table upload, no cache lookup --
 * moephoto_amd/dist.py keeps one device table per frame it stitches. */
int moe_stitch_dev(const moe_plan* plan, int device, const float* tiles_dev, const int64_t* tile_off_dev, int C,
                   void* out, int out_dtype, void* stream);
/* One ROW BAND of the canvas: the output rows [S(row0), S(row1)) (S(i) = first un-blended row of tile row i, moe_plan_rows; S(step_h) = out_h) folded
 * from tile rows row0 .. row1 into out = (C, S(row1) - S(row0), out_w).  The band ends with the blend band of tile row row1 (when row1 < step_h);
 * strip != 0: the entries of tile_off_dev for that tile row point at STRIPS -- the pad_sc rows [first, solid) of each plane, C planes of pad_sc x width --
 * instead of whole tiles.  Lets the ranks of a single-frame job each fold their own band of the canvas (moephoto_amd/dist.py: band-sharded stitch;
 * only those strips cross between neighbouring bands), bit-identical to the rows of moe_stitch's canvas. */
int moe_stitch_band(const moe_plan* plan, int device, const float* tiles_dev, const int64_t* tile_off_dev, int C,
                    void* out, int out_dtype, int row0, int row1, int strip, void* stream);
/* The body of the reference's tile loop behind the net call, for a caller that KEEPS that loop -- python/imageProcess.py:167-170:
 *     t = tmp_image[..., top*sc:bsc, left*sc:rsc];  q, _ = blend(*blend(unpad(r), t, topT, padSc, -2, bl.t()), leftT, padSc, -1, bl);  tmp_image[..., bsc-h:bsc, rsc-w:rsc] = q
 * as one kernel, in place on the canvas.  r: the tile result, C planes, element (c,i,j) at r + c*r_sC + i*r_sH + j (its first bsc-top_sc rows and rsc-left_sc
 * columns are used: opt.unpad); canvas: element (c,i,j) at canvas + c*c_sC + i*c_sH + j; (top_sc, left_sc, bsc, rsc, topT, leftT) = the tile tuple of iterClip
 * with top, left already multiplied by the scale; ramp: opt.blend, pad_sc values.  r, canvas and ramp share `dtype` (MOE_F16 on the reference's GPU path, MOE_F32)
 * and live on the device; the arithmetic is the reference's expression b = bx + blend * (b - bx) evaluated in that dtype, operation by operation: the canvas gets
 * the very bits the two torch calls produce.  Asynchronous on `stream`. */
int moe_blend_tile(const void* r, int64_t r_sC, int64_t r_sH, void* canvas, int64_t c_sC, int64_t c_sH, int dtype, int C,
                   int top_sc, int left_sc, int bsc, int rsc, int topT, int leftT, int pad_sc, const void* ramp, void* stream);

/* ---- wire format of tile results between ranks (moephoto_amd/dist.py, wire = 'f16s') ------------------------------------------------
 * The reference has no multi-device code; this belongs to the tile-parallel layer around doCrop (python/imageProcess.py:120-172).  A tile's fp32
 * value is needed exactly only where a blend reads it: in the tile's own blend band and under the blend bands of later tiles (imageProcess.py:
 * 120-131); elsewhere it is rounded to the canvas dtype or overwritten.  moe_plan_seams gives those rows / columns per tile (8 ints: two row ranges,
 * two column ranges, tile-local, half-open); a record then travels as [fp16 image of all values | fp32 seam rows | fp32 seam columns]
 * (moe_wire_words 4-byte words; a record whose seam rows cover it -- a band-mode strip -- is its fp32 values alone).  Unpacked tiles hold the exact
 * fp32 value in the seams and float(half(v)) elsewhere: an fp16 canvas folded from them is bit-identical to the fp32-wire one. */
typedef struct moe_wire_rec {
    int64_t tile_off;                /* fp32 elements into the tile buffer */
    int64_t wire_off;                /* 4-byte words into the wire buffer */
    int32_t C, th, tw;               /* planes, rows, columns of the tile (or strip) */
    int32_t ra0, ra1, rb0, rb1;      /* seam rows [ra0, ra1) and [rb0, rb1), ra1 <= rb0 */
    int32_t ca0, ca1, cb0, cb1;      /* seam columns likewise */
    int32_t reserved;
} moe_wire_rec;
int moe_plan_seams(const moe_plan* plan, int32_t* seams /* n_tiles x 8 */);
int64_t moe_wire_words(const moe_wire_rec* rec);
/* recs_dev: n records on the device; max_elems: the largest C*th*tw among them (sizes the launch). */
int moe_wire_pack(const float* tiles_dev, void* wire_dev, const moe_wire_rec* recs_dev, int n, int64_t max_elems, void* stream);
int moe_wire_unpack(float* tiles_dev, const void* wire_dev, const moe_wire_rec* recs_dev, int n, int64_t max_elems, void* stream);

/* doCrop on device: img = (C, Hp, Wp) planes (already padded per moe_plan_info's pad_*_to, see
 * python wrapper), element (c,i,j) at img + c*sC + i*sH + j*sW; out as in moe_stitch.
 * max_tiles_per_batch <= 0 picks a default. */
int moe_run_plan(moe_net* net, const moe_plan* plan, const void* img, int img_dtype, int64_t sC, int64_t sH, int64_t sW,
                 void* out, int out_dtype, int max_tiles_per_batch, void* stream);

/* General form: `pool` (device, fp32, >= moe_plan_pool_elems) receives the raw per-tile results (NULL: internal);
 * only tiles with raster index k % shard_count == shard_index are computed (tile-parallel sharding across GPUs:
 * the ranks then exchange pool slices, see moephoto_amd/dist.py); do_stitch = 0 skips the final fold. */
int moe_run_plan_ex(moe_net* net, const moe_plan* plan, const void* img, int img_dtype, int64_t sC, int64_t sH, int64_t sW,
                    void* out, int out_dtype, int max_tiles_per_batch, float* pool, int shard_index, int shard_count,
                    int do_stitch, void* stream);

/* Multi-frame, owner-sharded form of the tile loop (the multi-GPU step of moephoto_amd/dist.py; the reference runs frames one
 * after the other through doCrop, python/video.py:349-360 -> python/imageProcess.py:157-172).  `imgs` holds n_frames equally
 * shaped padded images, frame f at imgs + f*frame_stride elements; frame f's raw tile results go to pools + f*pool_stride
 * (fp32 elements, >= moe_plan_pool_elems each).  Only pairs with (f * n_tiles + k) % owner_count == owner_index are
 * computed; same-shaped tiles of different frames share launches.  No stitch: exchange, then moe_stitch per frame. */
int moe_run_plan_frames(moe_net* net, const moe_plan* plan, const void* imgs, int img_dtype, int64_t frame_stride,
                        int64_t sC, int64_t sH, int64_t sW, int n_frames, float* pools, int64_t pool_stride,
                        int owner_index, int owner_count, int max_tiles_per_batch, void* stream);
/* The same tile loop with an explicit destination per (frame, tile): tile k of frame f is computed iff
 * tile_dst[f * n_tiles + k] >= 0 and its C result planes go to dst + tile_dst[f * n_tiles + k] (fp32 elements; HOST table).
 * dist.py points the entries straight into its all-to-all send buffer (tiles another rank stitches) and into the stitch
 * buffer (tiles this rank stitches itself), so nothing is copied between the net, the collective and moe_stitch. */
int moe_run_plan_tiles(moe_net* net, const moe_plan* plan, const void* imgs, int img_dtype, int64_t frame_stride,
                       int64_t sC, int64_t sH, int64_t sW, int n_frames, float* dst, const int64_t* tile_dst,
                       int max_tiles_per_batch, void* stream);
/* elements of the fp32 tile pool for C planes, and the element offset of every tile inside it (default layout:
 * tile k = C contiguous planes of its HR extent, tiles in raster order) */
int64_t moe_plan_pool_elems(const moe_plan* plan, int C);
int moe_plan_tile_offsets(const moe_plan* plan, int C, int64_t* off);

/* ---- image I/O edges ---------------------------------------------------------------------------- */
/* src: H x W x C interleaved MOE_U8 (v/255) or MOE_U16 (v/2^bits); dst: C planes H x W, MOE_F32/MOE_F16 */
int moe_to_float(const void* src, int src_dtype, int bits, int H, int W, int C, void* dst, int dst_dtype, int device, void* stream);
/* src: C planes H x W (MOE_F32/MOE_F16); dst: H x W x C interleaved, v*2^bits clamped to [0, 2^bits-1], truncated */
int moe_to_output(const void* src, int src_dtype, int H, int W, int C, int bits, void* dst, int dst_dtype, int device, void* stream);

/* the pipeline's `resize` step: resizeByTorch = F.interpolate(x[None], size=(h, w), mode, align_corners=False)
 * (python/imageProcess.py:174-195, 555-556; python/procedure.py:104-107).  src: C planes H x W, dst: C planes h x w, same dtype
 * (MOE_F32 / MOE_F16); arithmetic in fp32 in torch's operation order. */
#define MOE_RESIZE_NEAREST 0
#define MOE_RESIZE_BILINEAR 1
#define MOE_RESIZE_BICUBIC 2
int moe_resize(const void* src, void* dst, int dtype, int C, int H, int W, int h, int w, int mode, int device, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOEPHOTO_AMD_H */
