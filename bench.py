#!/usr/bin/env python
"""Headline benchmark: megapixels/s of tiled 4x super-resolution of 1080p frames (model a4 = Net4x,
256-px tiles with 5-px overlap: BASELINE.json configs[1]) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N ...                       (started plainly: re-executes itself under torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W          (the driver's form)

A step = one pass of the hot path (doCrop: tile gather -> Net4x on the MFMA kernels -> stitch) over one batch
of N frames that are already resident in HBM.  N = 1: one frame.  N > 1: the N frames' tiles are sharded
round-robin over the ranks, exchanged with one RCCL all-to-all, and rank f % N stitches frame f (weak scaling:
every rank computes 40 tiles and stitches one frame per step); --strong: ONE frame per step, its 40 tiles over the
N ranks, every rank folding its row band of the canvas ("scaling": "strong").  `value` = input megapixels of all frames / s,
from EXACTLY --steps steps bracketed by barrier + synchronize, max over ranks.

Arithmetic: the product default ('auto' -> 'mixed' for Net4x: fp16 MFMA operands, fp32 accumulate, hi+lo trunk stream,
split operands on conv_input2 + the first ARSB, ~22-bit tail weights), the mode every parity test asserts at 1e-3.

Weights: `a4` is absent from the reference mount (.MISSING_LARGE_BLOBS); a synthetic Net4x in the zoo's exact
schema/format is used (tests/golden_defs.py: a2's real trunk + its upsampler duplicated), written and re-read
through the legacy-format file path.  Inputs: the headline is a seeded natural-image-like synthetic frame; the
SURVEY 8(d) input (seed-0 uniform uint8 noise / 255) is timed and parity-checked as a second input (`inputs`).

Extra objects on the JSON line:
  roofline          the kernel group with the largest share of the frame among the ones bracketed by hipEvents on the launch stream
                    (the two 3x3 64->256 implicit-GEMM convs at 2x resolution -- R and U branch are timed SEPARATELY, they run different
                    epilogues -- and the one-launch ARSBs).  `achieved` / `frac` use ALGORITHMIC FLOPs (the frame's pixels,
                    no tile overlap); `achieved_executed` / `frac_executed` count the overlapping tile pixels the kernels really compute.
                    `peak` = compute units x 4 SIMDs x 1024 FLOP/clk x max engine clock, all read from the device (moe_device_info).
                    `traffic` = HBM bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE) from rocprofv3 --pmc passes over a 2-frame child of THIS run on
                    THIS box (`traffic_source: "this run"`: three counter-only passes, ~45 s) when rocprofv3 is present, else from the committed,
                    digest-gated profiles/pmc_bench.json (`"committed"`); --no-pmc skips the child.
  roofline_kernels  every bracketed group, dominant first
  roofline_split_operand   the split-operand layers (conv_input2 + ARSB 1: conv64_sq / arsb_sq), bound "mfma": fp16-equivalent EXECUTED product-times over the
                    fp16 MFMA peak (round 4 labelled them "hbm": they stopped being held by bytes when conv_1's rows stayed in LDS); `hbm_side` keeps the bytes view
  configs           BASELINE's configs 3, 4, 5 -- one timed step each of `python bench.py --config N`, run as children of this command so that the
                    driver's default run times them: ms_per_step, value, parity_max_abs_vs_oracle, roofline (--no-configs skips them)
  clock             shader clock / package power sampled with rocm-smi during the sustained leg (the part is power-capped), and
                    `frac_at_clock` = roofline.frac rescaled to the peak at the sampled clock
  sustained         a >= --sustain second leg after the timed steps
  dropin_loop       the reference's own per-tile loop (python/imageProcess.py:157-172: slice view -> model(x) -> torch blends ->
                    slice assign) around the drop-in module, i.e. what a maintainer gets by swapping the class in runSR.mode_switch only;
                    `breakdown` (the 40 forwards alone / the torch blends alone) and `with_moe_blend_tile` (the two blend calls + the assign as one kernel)
  cpu_baseline      the oracle (a port of the reference's PyTorch-CPU fp32 path, proven equal to it on the goldens) timed on
                    this host with one socket's physical cores on a full tile row (8 tiles) + the ragged corner of the same
                    frame, and on BASELINE config 1 (256x256, a2) in full; rank 0 only.
Parity gate (same oracle tiles): the worst max-abs error of the engine's fp32 tile results vs the oracle over BOTH inputs must be
<= 1e-3, and the delivered fp16 canvas must equal those tiles to fp16 rounding (<= 1e-3 + half an fp16 ulp of the value); otherwise every
rank exits non-zero.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

FRAME = (3, 1080, 1920)
CROP, PAD, SCALE = 256, 5, 4
MFLOP_PER_PX_PLANE = 3.9456          # Net4x conv FLOPs per LR pixel per plane
PARITY_TOL = 1e-3
# bracketed layer groups: (profile substring, label, algorithmic FLOPs per LR pixel and plane)
GROUPS = [
    ('convt_R1.up1', 'R-branch 3x3 64->256 @2x res (+bias +PixelShuffle(2) +PReLU, fused 64->1 tail with split activations; conv3x3_ps4_kernel<2>: four phases per workgroup, rows streamed down a column, one fp32 plane + column aprons out)', 4 * 2 * 256 * 64 * 9),
    ('u.up1', 'U-branch 3x3 64->256 @2x res (+bias +PixelShuffle(2) +PReLU, fused 64->1 tail; conv3x3_ps4_kernel<1>)', 4 * 2 * 256 * 64 * 9),
    ('arsb', 'arsb32c_kernel (one ARSB per launch: two 3x3 64->64 convs @1x res + PReLU + hi/lo residual stream, 32x32x16 MFMAs, conv_1 rows kept in LDS down a patch column; 5 of 6 ARSBs)', 5 * 2 * 2 * 64 * 64 * 9),
]


# The split-operand layers (conv_input2, ARSB 1 of Net4x: conv64_sq.hip) are held by their bytes, not by the matrix pipe (DESIGN.md sections 4.4, 4.7): their roofline object is
# an HBM one.  Algorithmic bytes per LR pixel and plane, hi + low part in and out (+ the residual): with the fp8 low parts of a conv64_q8 chain 192 in + 192 out |
# 192 + 192 | 192 + 192 (residual) + 256 (fp16 low part again for the fused ARSB kernels); with fp16 low parts 512 | 512 | 768
HBM_KEYS = ['input2', 'c1_', 'c2_', 'xpair']      # (xpair: the exact ARSB as one launch, arsb_sq.hip -- then c1_ / c2_ record nothing)


def _exact_bytes_px(fused=False):
    chain = os.environ.get('MOE_X3_IMPL', 'auto') in ('auto', 'q8') and os.environ.get('MOE_LO8', 'on') not in ('0', 'off')
    if fused:
        return 384 + (192 + 256)      # conv_input2 | the exact ARSB in one launch: x in (fp16 + fp8 low word), y out (fp16 + fp16 low part for the single-pass ARSBs); conv_1's rows stay in LDS
    return (384 + 384 + 640) if chain else (512 + 512 + 768)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--precision', default='auto', choices=['auto', 'mixed', 'fp16', 'fp16x3'])
    ap.add_argument('--tiles-per-batch', type=int, default=0)
    ap.add_argument('--sustain', type=float, default=10.0, help='seconds of the sustained leg after the timed steps (0: skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true', help='skip the CPU oracle legs (baseline + parity gate)')
    ap.add_argument('--cpu-tiles', type=int, default=9, help='tiles of the headline frame run through the CPU oracle (parity + baseline)')
    ap.add_argument('--no-noise-input', action='store_true', help='skip the second (uniform uint8 noise) input')
    ap.add_argument('--no-dropin-loop', action='store_true', help='skip the per-tile drop-in loop leg')
    ap.add_argument('--config', type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json's configs[N-1]: 2 = the headline (default), 3 = 4K l25 -> a2 chain, "
                    '4 = batch of 64 1080p frames, 5 = one 8K frame -> 32K with 512-px tiles (bench_extra.py)')
    ap.add_argument('--wire', default='f16s', choices=['f32', 'f16s'], help='--gpus N: tile results between ranks as fp16 + fp32 seams (default: the canvases are fp16 and come out bit-identical, '
                    '0.54-0.57 of the bytes on the links; dist.py), or as fp32')
    ap.add_argument('--strong', action='store_true', help='--gpus N: ONE frame per step, its 40 tiles dealt over the N ranks and the canvas folded in row bands (strong scaling: '
                    'the north star\'s "tiles of a frame across the GPUs") instead of N frames per step (weak scaling, the default)')
    ap.add_argument('--no-extras', action='store_true', help='config 2: skip the roofline objects of the HBM-bound members and the I/O edges')
    ap.add_argument('--no-configs', action='store_true', help="config 2: skip the short legs of BASELINE's configs 3, 4, 5 (object `configs`: one timed step each, in child processes)")
    ap.add_argument('--no-floor', action='store_true', help='config 2: skip the value_floor legs (the same frame with six split-operand blocks / in fp16x3)')
    ap.add_argument('--no-pmc', action='store_true', help='config 2: do not try the rocprofv3 --pmc passes over a short child of this command (`traffic` then comes from the committed, '
                    'digest-gated profiles/pmc_bench.json)')
    args = ap.parse_args()
    args.steps_given = any(a == '--steps' or a.startswith('--steps=') for a in sys.argv[1:])

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # started plainly: become the launcher (one rank per GPU, rendezvous on the loopback address); the ranks re-enter main()
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
               '--master-port', str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)

    if args.config != 2:
        import bench_extra
        return bench_extra.run_config(args.config, args)

    import numpy as np
    import torch
    import torch.distributed as dist
    import golden_defs as gd
    from moephoto_amd import _lib, imageProcess as ip, runSR
    from moephoto_amd.config import config
    from moephoto_amd.weights import load_state_dict_file, save_state_dict_file

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('MOE_FORCE_DEVICE', os.environ.get('LOCAL_RANK', '0')))   # MOE_FORCE_DEVICE: test mode, ranks share a GPU
    backend = os.environ.get('MOE_DIST_BACKEND', 'nccl')                                     # 'gloo' only for that test mode
    args.gpus = world
    _lib.require_device()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)

    # ---- model through the plugin table, from a zoo-format file -------------------------------------
    config.deviceId, config.fp16, config.crop_sr, config.tilesPerBatch = local, True, CROP, args.tiles_per_batch
    wpath = '/tmp/moe_bench_a4_rank{}.pth'.format(rank)
    sd_np = gd.synth_state_dict('a4', load_state_dict_file) if (rank == 0 or world == 1) else None
    if world > 1:       # rank 0 owns the weights: one broadcast, every rank then writes its zoo-format file
        from moephoto_amd.dist import broadcast_state_dict
        sd_t = broadcast_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()} if rank == 0 else None, src=0, device=dev)
        from collections import OrderedDict
        sd_np = OrderedDict((k, v.numpy()) for k, v in sd_t.items())
    save_state_dict_file(sd_np, wpath)
    runSR.mode_switch['a4'] = (wpath, runSR.mode_switch['a4'][1])
    opt = runSR.getOpt({'op': 'SR', 'model': 'a', 'scale': SCALE, 'ensemble': 0})
    model = opt.modelCached.set_precision(args.precision)
    precision = model.resolved_precision()
    dinfo = _lib.device_info(local)
    # dense fp16 MFMA peak: every CU has 4 SIMDs, each retires 1024 FLOP per clock (v_mfma_f32_32x32x16_f16: 32768 FLOP in 32 cycles)
    peak_tflops = dinfo['compute_units'] * 4 * 1024 * dinfo['clock_khz'] * 1e3 / 1e12

    # ---- input frames, resident in HBM ------------------------------------------------------------------
    strong = bool(args.strong and world > 1)
    nframes = 1 if strong else world

    def make_frames(kind):
        if kind == 'natural':
            arr = [gd.natural_image(1000 + f, FRAME) for f in range(nframes)]
        else:           # seed-0 uniform uint8 -> /255
            arr = [(gd.noise_u8(f, FRAME).astype(np.float32) / np.float32(255)) for f in range(nframes)]
        return list(torch.stack([torch.from_numpy(a) for a in arr]).to(dev).half().unbind(0))   # slices of one tensor
    frames = make_frames('natural')
    plan = ip._plan_for(opt, frames[0].shape)
    assert plan.n_tiles == 40, plan.n_tiles
    tile_px_total = sum((t[1] - t[0]) * (t[3] - t[2]) for t in plan.tiles)
    overlap = tile_px_total / float(FRAME[1] * FRAME[2])             # executed / algorithmic pixels (1.064 for this grid)

    def step(fr):
        if world == 1:
            return ip.doCrop(opt, fr[0])
        from moephoto_amd.dist import run_frame_bands, run_frames
        if strong:      # one frame: tiles round-robin over the ranks, every rank folds its row band; the canvas stays sharded
            return run_frame_bands(opt, fr[:1], out_dtype=torch.float16, max_tiles_per_batch=args.tiles_per_batch, wire=args.wire)
        return run_frames(opt, fr, out_dtype=torch.float16, max_tiles_per_batch=args.tiles_per_batch, wire=args.wire)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def all_max(v):
        if world > 1:
            t = torch.tensor([v], dtype=torch.float64, device=dev if backend == 'nccl' else None)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            v = float(t.item())
        return v

    def timed(fr, steps, fn=None):
        fn = fn or step
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn(fr)
        fence()
        return all_max(time.perf_counter() - t0)

    in_mp = nframes * FRAME[1] * FRAME[2] / 1e6
    # ---- first contact of the multi-rank path with real links: before anything is timed, one step whose result is held, bit for bit, against a
    # single-rank doCrop of the same frame on every rank that holds a piece of it; how many ranks took part comes out of an all-reduce, not out of WORLD_SIZE
    first_contact = None
    if world > 1:
        from moephoto_amd import dist as mdist
        got = step(frames)
        worst = 0.0
        if strong:      # every rank holds its row band of frame 0's canvas
            y0, band = got[0]
            if band.shape[-2]:
                ref = ip.doCrop(opt, frames[0])
                worst = float((band.float() - ref[:, y0:y0 + band.shape[-2]].float()).abs().max())
        else:           # rank r holds the canvases of the frames f = r mod N
            for f, y in got.items():
                worst = max(worst, float((y.float() - ip.doCrop(opt, frames[f]).float()).abs().max()))
        t = torch.tensor([1.0, worst], dtype=torch.float64, device=dev if backend == 'nccl' else None)
        cnt = t.clone()
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        first_contact = {'ranks_seen': int(cnt[0].item()), 'parity_vs_single_gpu_max_abs': float(t[1].item()), 'bit_identical': bool(float(t[1].item()) == 0.0),
                         'exchange_mode': mdist.EXCHANGE_MODE, 'exchange_fallback': mdist.EXCHANGE_FALLBACK,
                         'what': 'one untimed step before the warm-up: every rank compares what it holds of the result (its frames / its row band) with its OWN single-rank doCrop '
                                 'of the same frame; max over ranks'}
        del got
    for _ in range(args.warmup):
        step(frames)
    model.set_profile(','.join([g[0] for g in GROUPS] + HBM_KEYS + ['tailadd']))      # hipEvent pairs on the launch stream around the launches of each group
    dt = timed(frames, args.steps)
    profs = model.get_profile(all_keys=True)
    model.set_profile(None)
    ms_per_step = dt / args.steps * 1e3
    value = in_mp / (ms_per_step / 1e3)

    res = {
        'metric': 'megapixels/sec (input), 1080p 4x SR (Net4x a4), 256-px tiles with overlap',
        'value': round(value, 3), 'unit': 'MP/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'strong' if strong else 'weak', 'vs_baseline': None,
        'dtype': 'fp16' if precision != 'fp16x3' else 'fp16x3 (split fp16 operands, three MFMA passes)', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: {} frame(s) 1920x1080 RGB -> 7680x4320, model a4 (Net4x, synthetic weights in zoo format), '
                               'crop 256 pad 5 align 8 -> 40 tiles/frame, fp16 I/O, fp16 MFMA operands + fp32 accumulate'.format(nframes),
                   'frames_per_step': nframes, 'tiles_per_frame': plan.n_tiles, 'output_mp_per_s': round(value * SCALE * SCALE, 2),
                   'tflops_algorithmic': round(nframes * 3 * FRAME[1] * FRAME[2] * MFLOP_PER_PX_PLANE * 1e6 / (ms_per_step / 1e3) / 1e12, 2),
                   'tile_overlap_factor': round(overlap, 4),
                   'parallelism': ('tile-parallel x{}: ONE frame, tiles round-robin over the ranks, all-to-all of tile results + blend strips, every rank folds its row band (canvas stays sharded)'.format(world) if strong else
                                   'tile-parallel x{} (round-robin tiles, all-to-all of tile results, stitch on rank f%N)'.format(world)) if world > 1 else 'single GPU',
                   'precision': precision, 'exact_blocks': model.exact_blocks(),
                   **({'wire': args.wire} if world > 1 else {}),
                   'input': 'natural-image-like synthetic frame (tests/golden_defs.natural_image)'},
        'device': {'compute_units': dinfo['compute_units'], 'max_clock_ghz': round(dinfo['clock_khz'] / 1e6, 3),
                   'peak_fp16_mfma_tflops': round(peak_tflops, 1), 'peak_formula': 'CUs x 4 SIMD x 1024 FLOP/clk x max clock (hipDeviceProp)'},
    }

    if first_contact:
        res['config']['ranks_seen'] = first_contact['ranks_seen']
        res['config']['parity_vs_single_gpu'] = first_contact['parity_vs_single_gpu_max_abs']
        res['config']['exchange_mode'] = first_contact['exchange_mode']
        res['first_contact'] = first_contact

    # ---- what the headline would cost on weights that need more split-operand blocks ------------------------------------------------
    # `value` is the speed of a4-synth with the block count the calibration gives ITS weights (config.exact_blocks); the zoo's real a4 is absent from the mount
    # (.MISSING_LARGE_BLOBS).  moe_net_calibrate moves a checkpoint whose trunk swings wider to more blocks (a 15 % wider trunk: six) and to fp16x3 when six do
    # not reach the target: the same frame timed in those two arithmetics is the floor of what such weights get.
    if precision == 'mixed' and args.precision == 'auto' and not args.no_floor and world == 1:      # (single-GPU disclosure; the exact mode's workspace for a 96-plane launch
        floor = {}                                                                                     # set is 94 GB: not something to triple on a GPU that ranks share in tests)
        for tag, setup in (('exact_blocks_6', lambda: model.set_exact_blocks(6)), ('fp16x3', lambda: model.set_precision('fp16x3'))):
            try:
                setup()
                step(frames)
                dtf = timed(frames, 3)
                floor[tag] = {'ms_per_step': round(dtf / 3 * 1e3, 3), 'value': round(in_mp / (dtf / 3), 3), 'steps': 3}
            except MemoryError as e:
                floor[tag] = {'error': str(e)[:200]}
        model.set_exact_blocks(-1)
        model.set_precision(args.precision)
        step(frames)
        floor['note'] = 'the same frame with six split-operand ARSBs / in the exact arithmetic: what a checkpoint gets that the calibration (moe_net_calibrate) moves there'
        res['config']['value_floor'] = floor

    # ---- roofline objects: one per bracketed group, dominant (by time) first ----------------------------------------
    # HBM bytes per launch (`traffic`) and MFMA busy from rocprofv3 --pmc passes over a 2-frame child of THIS command on THIS box, when the tool is here
    # (`traffic_source: "this run"`); otherwise from the committed, digest-gated profiles/pmc_bench.json (`"committed"`)
    pmc, pmc_source = None, None
    if rank == 0 and world == 1 and not args.no_pmc:
        pmc = _pmc_live()
        pmc_source = 'this run' if pmc else None
    if not pmc:
        pmc = _pmc_table()
        pmc_source = 'committed' if pmc else None
    kernels = []
    frames_timed = args.steps / float(world) if strong else args.steps        # every rank computes one frame's worth of tiles per step (--strong: 1/N of one)
    for (key, label, flop_px), prof in zip(GROUPS, profs):
        if prof['launches'] <= 0 or prof['total_ms'] <= 0:
            continue
        secs = prof['total_ms'] / 1e3
        alg = 3.0 * FRAME[1] * FRAME[2] * flop_px * frames_timed                      # algorithmic FLOPs of the group in the timed steps
        exe = prof['flops']                                                             # what the launches computed (tile pixels)
        k = {'bound': 'mfma', 'kernel': label, 'layer_key': key,
             'achieved': round(alg / secs / 1e12, 1), 'peak': round(peak_tflops, 1), 'unit': 'TFLOP/s', 'frac': round(alg / secs / 1e12 / peak_tflops, 4),
             'achieved_executed': round(exe / secs / 1e12, 1), 'frac_executed': round(exe / secs / 1e12 / peak_tflops, 4),
             'launches': prof['launches'], 'avg_launch_ms': round(prof['total_ms'] / prof['launches'], 4),
             'ms_per_frame': round(prof['total_ms'] / frames_timed, 3), 'share_of_step': round(prof['total_ms'] / frames_timed / ms_per_step, 4),
             'gflop_per_launch_algorithmic': round(alg / prof['launches'] / 1e9, 2)}
        t = pmc.get(key)
        if t:       # PMC passes of this same command: bytes per frame / launches per frame
            k['traffic'] = int(t['hbm_bytes_per_frame'] / max(1, t['launches_per_frame']))
            k['traffic_source'] = pmc_source
            k['traffic_note'] = t.get('note', '')
            if t.get('mfma_busy') is not None:
                k['mfma_busy_pmc'] = t['mfma_busy']
        else:
            k['traffic'] = None
            if getattr(_pmc_table, 'stale', None):
                k['traffic_note'] = _pmc_table.stale
        kernels.append(k)
    kernels.sort(key=lambda k: -k['ms_per_frame'])
    # what the package power cap lets the matrix pipe deliver on operands whose bits toggle: a loop of nothing but v_mfma_f32_32x32x16_f16 on uniform random
    # fp16 data (tools/micro/mfma_power.hip), measured here on this box -- the practical ceiling beside the nominal peak every `frac` is quoted against
    power = _power_roofline() if (rank == 0 and world == 1) else None      # (single-GPU runs only: a second process on the device has no place in a scaling run)
    if power:
        res['power_roofline'] = power
        for k in kernels:
            k['frac_of_power_roofline'] = round(k['achieved'] / power['uniform_random_tflops'], 4)
            k['frac_executed_of_power_roofline'] = round(k['achieved_executed'] / power['uniform_random_tflops'], 4)
    if kernels:
        res['roofline'] = dict(kernels[0])
        res['roofline_kernels'] = kernels
        trunk = [k for k in kernels if k['layer_key'] == 'arsb']
        if trunk:
            res['roofline_trunk'] = trunk[0]
    hb = profs[len(GROUPS):len(GROUPS) + len(HBM_KEYS)]
    fused = len(hb) == len(HBM_KEYS) and hb[3]['launches'] > 0
    hb = [p for p in hb if p['launches'] > 0 and p['total_ms'] > 0]
    if len(hb) >= 2:
        secs = sum(p['total_ms'] for p in hb) / 1e3
        launches = sum(p['launches'] for p in hb)
        alg = 3.0 * FRAME[1] * FRAME[2] * _exact_bytes_px(fused) * frames_timed            # algorithmic bytes of the three layers in the timed steps
        peak_gbs = 8000.0                                                                   # MI355X_MICROARCH.md: HBM3E ~8 TB/s
        flops = 3.0 * FRAME[1] * FRAME[2] * frames_timed * 3 * 2 * 64 * 64 * 9            # three 3x3 64->64 convs; executed: two product-times each (fp16 + two fp8 products at twice the rate)
        exe_tflops = 2 * flops / secs / 1e12      # fp16-equivalent executed: per conv one fp16 product-time + two fp8 products at twice the rate = two product-times
        k = {'bound': 'mfma', 'kernel': ('conv64_sq_kernel (conv_input2) + arsb_sq_kernel (ARSB 1 as ONE launch: conv_1 on producer waves, conv_2 on consumer waves, its rows in LDS)' if fused else
                                        'conv64_sq_kernel (conv_input2 + the two convs of ARSB 1)') +
                                       ': split operands -- fp16 product + two fp8 correction products, fp8 low parts between the layers; rows streamed down columns by '
                                       'register-weight waves (small or odd shapes: conv64_q8_kernel; MOE_X3_IMPL=x3: conv64_x3_kernel)', 'layer_key': 'exact',
             'achieved': round(exe_tflops, 1), 'peak': round(peak_tflops, 1), 'unit': 'TFLOP/s', 'frac': round(exe_tflops / peak_tflops, 4),
             'frac_note': 'fp16-equivalent EXECUTED product-times over the fp16 MFMA peak (VERDICT r04 item 5c): these layers are held by the matrix pipe under the power cap since '
                          'conv_1\'s rows stay in LDS (arsb_sq); algorithmic (one fp32-grade conv = 2 x 64 x 64 x 9 FLOP a pixel) is half of it',
             'achieved_algorithmic': round(flops / secs / 1e12, 1), 'frac_algorithmic': round(flops / secs / 1e12 / peak_tflops, 4),
             'hbm_side': {'achieved': round(alg / secs / 1e9, 1), 'peak': peak_gbs, 'unit': 'GB/s', 'frac': round(alg / secs / 1e9 / peak_gbs, 4), 'bytes_per_pixel_algorithmic': _exact_bytes_px(fused)},
             'launches': launches, 'avg_launch_ms': round(secs * 1e3 / launches, 4),
             'ms_per_frame': round(secs * 1e3 / frames_timed, 3), 'share_of_step': round(secs * 1e3 / frames_timed / ms_per_step, 4),
             }
        t = pmc.get('exact')
        k['traffic'] = int(t['hbm_bytes_per_frame'] / max(1, t['launches_per_frame'])) if t else None
        if t:
            k['traffic_source'] = pmc_source
            k['traffic_note'] = t.get('note', '')
            k['achieved_measured_bytes'] = round(t['hbm_bytes_per_frame'] * frames_timed / secs / 1e9, 1)      # the PMC passes' bytes over this run's time
        res['roofline_split_operand'] = k

    # ---- the HBM-bound members of the path and the I/O edges the headline excludes (bench_extra.py) --------------------------------------
    if world == 1 and not args.no_extras:
        import bench_extra
        tp = profs[len(GROUPS) + len(HBM_KEYS)] if len(profs) > len(GROUPS) + len(HBM_KEYS) else None
        res['roofline_hbm_kernels'] = bench_extra.hbm_members(torch, _lib, ip, gd, dev, opt, model, plan, frames[0], tp, frames_timed, args.tiles_per_batch, load_state_dict_file)
        res['io_edges'] = bench_extra.io_edges(torch, _lib, dev)
        config.crop_sr = CROP

    # ---- sustained leg (power-capped part: a 0.5 s burst flatters the clock), with the clock / power sampled beside it ------------
    if args.sustain > 0:
        n_s = max(args.steps, int(args.sustain / (ms_per_step / 1e3)) + 1)
        sampler = _ClockSampler(local) if rank == 0 else None
        if sampler:
            sampler.start()
        dts = timed(frames, n_s)
        clock = sampler.stop() if sampler else None
        res['sustained'] = {'seconds': round(dts, 2), 'steps': n_s, 'ms_per_step': round(dts / n_s * 1e3, 3), 'value': round(in_mp / (dts / n_s), 3), 'unit': 'MP/s'}
        if clock:
            res['clock'] = clock
            if kernels and clock.get('sclk_ghz_mean'):
                res['clock']['frac_at_clock'] = round(kernels[0]['frac'] * (dinfo['clock_khz'] / 1e6) / clock['sclk_ghz_mean'], 4)
                res['clock']['note'] = 'frame-average shader clock under the package power cap; frac_at_clock = roofline.frac against the MFMA peak at that clock'

    # ---- second input: uniform uint8 noise ------------------------------------------------------
    inputs = {'natural': {'value': res['value'], 'ms_per_step': res['ms_per_step']}}
    noise_frames = None
    if not args.no_noise_input:
        noise_frames = make_frames('noise_u8')
        step(noise_frames)
        dtn = timed(noise_frames, args.steps)
        inputs['noise_u8'] = {'value': round(in_mp / (dtn / args.steps), 3), 'ms_per_step': round(dtn / args.steps * 1e3, 3)}
    res['inputs'] = inputs

    # ---- the reference's own tile loop around the drop-in module (rank 0, one GPU) ---------------------------------------
    if world == 1 and not args.no_dropin_loop:
        ramp = torch.from_numpy(plan.ramp.copy()).to(dev).half()
        xin = frames[0]

        def loop(_):
            return _reference_style_loop(opt, xin, plan, ramp, torch)

        def loop_fused(_):
            return _reference_style_loop(opt, xin, plan, ramp, torch, blend_tile=ip.blendTile)
        y_loop = loop(None)
        y_fused = loop_fused(None)
        torch.cuda.synchronize()
        y_dev = ip.doCrop(opt, xin)
        n_l = max(3, args.steps // 2)
        dtl = timed(None, n_l, loop)
        ms_l = dtl / n_l * 1e3
        ms_f = timed(None, n_l, loop_fused) / n_l * 1e3
        # round 6: the wrapper marks forwards on slices of one image (moe_net_forward_ex, MOE_FWD_INPUT_SINCE_PREV) and the engine overlaps them; the same loop with the
        # option off is what rounds 4-5 measured
        model.set_option('overlap_calls', 0)
        y_plain = loop(None)
        ms_l0 = timed(None, n_l, loop) / n_l * 1e3
        ms_f0 = timed(None, n_l, loop_fused) / n_l * 1e3
        model.set_option('overlap_calls', 1)
        # where the loop's time goes: the 40 forwards alone (results dropped), and the blends + assigns alone (on kept tile results)
        xb_ = plan.padImage(xin).unsqueeze(1)

        def engine_only(_):
            for t in plan.tiles:
                opt(xb_[..., t[0]:t[1], t[2]:t[3]])
        kept = [opt(xb_[..., t[0]:t[1], t[2]:t[3]]).squeeze(1).clone() for t in plan.tiles]

        class _Kept(object):
            def __init__(self):
                self.k = 0

            def __call__(self, _):
                self.k += 1
                return kept[self.k - 1].unsqueeze(1)

        def blends_only(_):
            return _reference_style_loop(_Kept(), xin, plan, ramp, torch)

        def blends_fused_only(_):
            return _reference_style_loop(_Kept(), xin, plan, ramp, torch, blend_tile=ip.blendTile)
        ms_e = timed(None, n_l, engine_only) / n_l * 1e3
        ms_b = timed(None, n_l, blends_only) / n_l * 1e3
        ms_bf = timed(None, n_l, blends_fused_only) / n_l * 1e3
        del kept
        res['dropin_loop'] = {'value': round(FRAME[1] * FRAME[2] / 1e6 / (ms_l / 1e3), 3), 'unit': 'MP/s', 'ms_per_step': round(ms_l, 3), 'steps': n_l,
                              'ratio_to_value': round(ms_per_step / ms_l, 3),
                              'max_abs_vs_device_docrop': float('{:.3e}'.format(float((y_loop.float() - y_dev.float()).abs().max()))),
                              'breakdown': {'engine_forwards_only_ms': round(ms_e, 3), 'torch_blends_and_assigns_only_ms': round(ms_b, 3),
                                            'sum_ms': round(ms_e + ms_b, 3), 'loop_ms': round(ms_l, 3),
                                            'device_resident_doCrop_ms': round(ms_per_step, 3),
                                            'note': 'host-inclusive wall per frame, synchronised at both ends; engine_forwards_only = the 40 per-tile calls of 3 planes each with '
                                                    'the results dropped: what separates it from the headline is launch geometry (3 planes per launch set instead of up to 96); '
                                                    'rocprofv3 kernel trace of the loop: profiles/r05/'},
                              'without_overlap_calls': {'ms_per_step': round(ms_l0, 3), 'ratio_to_value': round(ms_per_step / ms_l0, 3), 'with_moe_blend_tile_ms': round(ms_f0, 3),
                                                        'bit_identical_to_overlapped': bool(torch.equal(y_plain, y_loop)),
                                                        'what': 'option overlap_calls = 0: every forward on the caller\'s stream, behind the previous tile\'s blend (rounds 4-5); default since round 6: '
                                                                'consecutive forwards on slices of one image run on two internal stream + workspace sets (moe_net_forward_ex)'},
                              'with_moe_blend_tile': {'value': round(FRAME[1] * FRAME[2] / 1e6 / (ms_f / 1e3), 3), 'ms_per_step': round(ms_f, 3), 'ratio_to_value': round(ms_per_step / ms_f, 3),
                                                      'blend_tile_only_ms': round(ms_bf, 3),
                                                      'max_abs_vs_device_docrop': float('{:.3e}'.format(float((y_fused.float() - y_dev.float()).abs().max()))),
                                                      'what': 'the same loop with its two blend() calls + slice-assign replaced by ONE kernel per tile (moe_blend_tile = imageProcess.blendTile, '
                                                              "INTEGRATION.md section 2: a two-line change in the reference's doCrop); arithmetic = the reference's own fp16 expression, operation by operation"},
                              'what': "the reference's per-tile loop (python/imageProcess.py:157-172) with torch blends, calling models.Net4x.__call__ = moe_net_forward on "
                                      '3 planes of <= 256x256 per call (40 calls per frame, fp16 canvas AND fp16 blends as in the reference GPU path); '
                                      'moe_run_plan (the headline) batches up to 32 same-shaped tiles per launch set and stitches from fp32 tiles'}

    # ---- CPU baseline + parity gate (rank 0) -----------------------------------------------------------------
    parity_ok = True
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import nets as onets        # test infrastructure: the checker / baseline, never the product path
        from oracle import planner as oplanner, stitch as ostitch
        import ctypes
        ncores, nthreads = _one_socket_cores()
        torch.set_num_threads(nthreads)
        # tile row 1 of the 5 x 8 grid (8 tiles incl. the ragged right column), then the ragged bottom-right corner
        ks = ([8, 9, 10, 11, 12, 13, 14, 15, 39] + [35, 0, 20])[:max(1, args.cpu_tiles)]
        if world > 1:
            ks = ks[1:2] + ks[8:9]            # multi-GPU runs: a bounded check (the other ranks wait)
        off = plan.tile_offsets(3)
        opl = oplanner.prepare(FRAME, 1 << 40, 1e-3, PAD, SCALE, 8, CROP)
        rows = ostitch.axis_cover(opl.anchors_h, SCALE, opl.pad_sc, plan.outH)
        cols = ostitch.axis_cover(opl.anchors_w, SCALE, opl.pad_sc, plan.outW)

        def check(fr, tiles):
            pool = torch.empty(plan.pool_elems(3), dtype=torch.float32, device=dev)
            y = torch.empty((3, plan.outH, plan.outW), dtype=torch.float16, device=dev)
            sC, sH, sW = fr.stride()
            _lib.check(_lib.lib().moe_run_plan_ex(model._h, plan._h, fr.data_ptr(), _lib.F16, sC, sH, sW, y.data_ptr(), _lib.F16, args.tiles_per_batch,
                                                  ctypes.c_void_p(pool.data_ptr()), 0, 1, 1, torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            x16 = fr.float().cpu().numpy()          # the same fp16-quantised input the engine saw
            px, cpu_s, errs, cerrs, cbounds = 0, 0.0, [], [], []
            for k in tiles:
                top, bottom, left, right = plan.tiles[k][:4]
                xt = np.ascontiguousarray(x16[:, None, top:bottom, left:right])
                c0 = time.perf_counter()
                want = onets.forward('net4x', sd_np, xt).numpy()[:, 0]
                cpu_s += time.perf_counter() - c0
                px += (bottom - top) * (right - left)
                got = pool[off[k]:off[k] + want.size].reshape(want.shape).cpu().numpy()
                errs.append(float(np.abs(got - want).max()))
                # the DELIVERED fp16 canvas where this tile is final and un-blended: rows / columns from the tile's solid start up to the
                # next tile's first written row / column (python/imageProcess.py:120-131,167-170)
                i, j = divmod(k, opl.step_w)
                y0, y1 = rows[i][1], (rows[i + 1][0] if i + 1 < len(rows) else plan.outH)
                x0, x1 = cols[j][1], (cols[j + 1][0] if j + 1 < len(cols) else plan.outW)
                oy, ox = rows[i][2], cols[j][2]
                if y1 > y0 and x1 > x0:
                    w_reg = want[:, y0 - oy:y1 - oy, x0 - ox:x1 - ox]
                    c_reg = y[:, y0:y1, x0:x1].float().cpu().numpy()
                    d = np.abs(c_reg - w_reg)
                    cerrs.append(float(d.max()))
                    # fp16 rounding of the delivered value: half an ulp of |v| (2^-11 relative, rounded up to the binade)
                    cbounds.append(float((d - (PARITY_TOL + np.exp2(np.ceil(np.log2(np.maximum(np.abs(w_reg), 1e-3))) - 11))).max()))
            return px, cpu_s, errs, cerrs, cbounds
        px, cpu_s, errs, cerrs, cb = check(frames[0], ks)
        frame_s = cpu_s / px * tile_px_total              # all 40 tiles at the sampled per-pixel rate

        def pobj(tiles, errs, cerrs, cb):
            return {'tiles': tiles, 'worst_max_abs': float('{:.3e}'.format(max(errs))), 'per_tile': [float('{:.2e}'.format(e)) for e in errs],
                    'canvas_fp16_worst_max_abs': float('{:.3e}'.format(max(cerrs))) if cerrs else None,
                    'canvas_within_tol_plus_half_ulp': bool(max(cb) <= 0) if cb else None,
                    'ok': bool(max(errs) <= PARITY_TOL and (not cb or max(cb) <= 0))}
        parity = {'natural': pobj(ks, errs, cerrs, cb)}
        parity_ok = parity['natural']['ok']
        if noise_frames is not None:
            kn = ks if world == 1 else ks[:1]
            _, cpu_n, errs_n, cerrs_n, cb_n = check(noise_frames[0], kn)
            parity['noise_u8'] = pobj(kn, errs_n, cerrs_n, cb_n)
            parity_ok = parity_ok and parity['noise_u8']['ok']
        res['cpu_baseline'] = {'value': round(FRAME[1] * FRAME[2] / 1e6 / frame_s, 5), 'unit': 'MP/s', 'cores': ncores, 'threads': nthreads,
                               'kind': 'port', 'sample': '{} of 40 tiles (tile row 1 + the ragged corner: {} tile pixels x 3 planes) of the headline frame through the fp32 '
                               'oracle (torch/oneDNN conv backend), {:.1f} s; extrapolated to the frame by tile pixels'.format(len(ks), px, cpu_s),
                               'cpu_model': _cpu_model(), 'host_logical_cpus': os.cpu_count(),
                               'note': 'the GPU idles during this leg: device-busy averages over the whole process include it'}
        if world == 1:      # BASELINE config 1 in full: 256x256 RGB, a2 (real weights), one tile, on the CPU oracle
            sd_a2 = gd.state_dict_for('a2', load_state_dict_file)
            x1 = gd.noise_u8(0, (3, 256, 256)).astype(np.float32)[:, None] / np.float32(255)
            onets.forward('net2x', sd_a2, x1[:, :, :64, :64])
            c0 = time.perf_counter()
            onets.forward('net2x', sd_a2, x1)
            c1 = time.perf_counter() - c0
            res['cpu_baseline']['config1'] = {'workload': 'BASELINE configs[0]: 256x256 RGB -> 512x512, a2, one tile, fp32 oracle', 'seconds': round(c1, 3),
                                              'value': round(256 * 256 / 1e6 / c1, 5), 'unit': 'MP/s'}
        res['config']['parity'] = parity
        res['config']['parity_max_abs_vs_oracle'] = max(p['worst_max_abs'] for p in parity.values())
        res['config']['parity_tolerance'] = PARITY_TOL
        res['config']['parity_ok'] = bool(parity_ok)
    # ---- the other BASELINE configs, one timed step each, in child processes of this same script (so that the driver's default command times them too) -------
    if rank == 0 and world == 1 and not args.no_configs:
        res['configs'] = _other_configs(args)
        parity_ok = parity_ok and all(c.get('parity_ok', True) is not False for c in res['configs'].values() if isinstance(c, dict))
    if first_contact and not first_contact['bit_identical']:
        parity_ok = False
        res['config']['parity_ok'] = False
    # ---- the short view LAST: the driver keeps the standard keys, `config`, `roofline`, `cpu_baseline` and the last 2,000 characters of the line ----
    if rank == 0:
        rk = {k['layer_key']: k for k in res.get('roofline_kernels', [])}
        dl = res.get('dropin_loop') or {}
        res['summary'] = {
            'value': res['value'], 'ms_per_step': res['ms_per_step'], 'exact_blocks': res['config'].get('exact_blocks'),
            'value_floor_ms': {k: v.get('ms_per_step') for k, v in (res['config'].get('value_floor') or {}).items() if isinstance(v, dict)},
            'kernels_ms_per_frame_and_frac': {k: [v['ms_per_frame'], v['frac']] for k, v in rk.items()},
            'split_operand_ms': (res.get('roofline_split_operand') or {}).get('ms_per_frame'),
            'dropin_ratio': dl.get('ratio_to_value'), 'dropin_ratio_without_overlap_calls': (dl.get('without_overlap_calls') or {}).get('ratio_to_value'),
            'dropin_blend_tile_ratio': (dl.get('with_moe_blend_tile') or {}).get('ratio_to_value'),
            'configs_ms': {k: v.get('ms_per_step') for k, v in (res.get('configs') or {}).items() if isinstance(v, dict)},
            'parity_max_abs_vs_oracle': res['config'].get('parity_max_abs_vs_oracle'), 'sustained_ms': (res.get('sustained') or {}).get('ms_per_step'),
            'clock_ghz': (res.get('clock') or {}).get('sclk_ghz_mean'), 'power_w': (res.get('clock') or {}).get('package_power_w_mean')}
        print(json.dumps(res))
        sys.stdout.flush()
    if world > 1:       # every rank learns the verdict of rank 0's gate before anybody exits
        flag = torch.tensor([1 if parity_ok else 0], dtype=torch.int32, device=dev if backend == 'nccl' else None)
        dist.broadcast(flag, src=0)
        parity_ok = bool(int(flag.item()))
        dist.barrier()
        dist.destroy_process_group()
    if not parity_ok:
        raise SystemExit('parity gate failed: the engine differs from the oracle by more than {} (tiles) or the fp16 canvas by more than that plus half an ulp'.format(PARITY_TOL))


def _reference_style_loop(opt, x, plan, ramp, torch, blend_tile=None):
    """The loop a MoePhoto maintainer keeps when only the class in runSR.mode_switch is swapped (python/imageProcess.py:157-172): for every
    tile, hand the net a slice VIEW of the planes-as-batch image, cross-fade the fresh result into what the canvas window already
    holds (rows first, then columns, each over `padSc` entries in front of the tile's first new entry), and assign it aligned to the
    window's bottom-right corner.  Written against the plan's tile tuples; dtype of canvas and blends = dtype of x (fp16 on the GPU path)."""
    C = x.shape[0]
    sc, psc = plan.sc, plan.padSc
    xb = plan.padImage(x).unsqueeze(1)
    canvas = x.new_empty((C, plan.outH, plan.outW))

    def fade(fresh, held, first_new, dim, weights):
        n = fresh.shape[dim]
        if first_new < 0:
            first_new += n
        if first_new < 1:
            return fresh, held
        a = first_new - psc
        band = torch.lerp(held.narrow(dim, a, psc), fresh.narrow(dim, a, psc), weights)
        return torch.cat([band, fresh.narrow(dim, first_new, n - first_new)], dim), held.narrow(dim, a, n - a)
    wr, wc = ramp.view(-1, 1), ramp.view(1, -1)
    for (top, bottom, left, right, tt, lt, bsc, rsc) in plan.tiles:
        if blend_tile is not None:       # moe_blend_tile: the two fades + the assign below as one kernel (in place on the canvas)
            blend_tile(opt(xb[..., top:bottom, left:right]), canvas, (top, bottom, left, right, tt, lt, bsc, rsc), sc, psc, ramp)
            continue
        r = opt(xb[..., top:bottom, left:right]).squeeze(1)[..., :plan.outH - top * sc, :plan.outW - left * sc]
        held = canvas[..., top * sc:bsc, left * sc:rsc]
        r, held = fade(r, held, tt, -2, wr)
        r, _ = fade(r, held, lt, -1, wc)
        h, w = r.shape[-2:]
        canvas[..., bsc - h:bsc, rsc - w:rsc] = r
    return canvas


class _ClockSampler(object):
    """rocm-smi's shader clock and package power, sampled twice a second in a side thread while the sustained leg runs."""

    def __init__(self, device):
        self.device, self.samples, self._stop, self._t = device, [], threading.Event(), None

    def _run(self):
        while not self._stop.is_set():
            try:
                out = subprocess.run(['rocm-smi', '-d', str(self.device), '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=5).stdout
                d = json.loads(out)
                card = next(iter(d.values()))
                sclk = next((v for k, v in card.items() if 'sclk' in k.lower() and 'mhz' in str(v).lower()), None)
                pw = next((v for k, v in card.items() if 'power' in k.lower() and 'graphics' in k.lower()), None)
                mhz = float(str(sclk).strip('()').lower().replace('mhz', '')) if sclk else None
                if mhz and mhz > 300:       # (idle samples at the start / end of the leg are dropped)
                    self.samples.append((mhz, float(pw) if pw else None))
            except Exception:
                pass
            self._stop.wait(0.5)

    def start(self):
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def stop(self):
        self._stop.set()
        if self._t:
            self._t.join(timeout=10)
        if not self.samples:
            return None
        mhz = [s[0] for s in self.samples]
        pw = [s[1] for s in self.samples if s[1] is not None]
        return {'sclk_ghz_mean': round(sum(mhz) / len(mhz) / 1e3, 3), 'sclk_ghz_min': round(min(mhz) / 1e3, 3), 'sclk_ghz_max': round(max(mhz) / 1e3, 3),
                'package_power_w_mean': round(sum(pw) / len(pw), 1) if pw else None, 'samples': len(mhz), 'source': 'rocm-smi --showclocks --showpower during the sustained leg'}


def _cpu_model():
    try:
        for l in open('/proc/cpuinfo'):
            if l.startswith('model name'):
                return l.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def _one_socket_cores():
    """(physical cores of socket 0, threads to use): one thread per physical core of one socket -- oversubscribing every
    logical CPU of a 2-socket box made the round-1 baseline 4x slower than the same oracle on 8 threads."""
    cores, phys, core = set(), None, None
    try:
        for l in open('/proc/cpuinfo'):
            if l.startswith('physical id'):
                phys = l.split(':')[1].strip()
            elif l.startswith('core id'):
                core = l.split(':')[1].strip()
            elif not l.strip():
                if phys == '0' and core is not None:
                    cores.add(core)
                phys = core = None
    except Exception:
        pass
    n = len(cores) or max(1, (os.cpu_count() or 2) // 2)
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return n, n


def _power_roofline():
    exe = os.path.join(ROOT, 'tools', 'micro', 'bin', 'mfma_power')
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe, '60000'], capture_output=True, text=True, timeout=120).stdout
        import re
        v = {}
        for kind, key in (('zeros', 'zeros_tflops'), ('uniform random', 'uniform_random_tflops'), ('random signs', 'random_signs_tflops'), ('random, B per 8 MFMAs', 'uniform_random_shared_b_tflops')):
            m = re.search(re.escape(kind) + r'\s+idle states 0:\s+([0-9.]+) TFLOP/s', out)
            if m:
                v[key] = float(m.group(1))
        if 'uniform_random_tflops' not in v:
            return None
        v['what'] = ('back-to-back v_mfma_f32_32x32x16_f16 from registers (no LDS, no memory), one 4-wave workgroup per CU, ~25 ms per variant: the rate the package power cap '
                     'sustains; zeros = operands that do not toggle')
        return v
    except Exception:
        return None


def _other_configs(args):
    """{'config3' | 'config4' | 'config5': {...}}: `python bench.py --config N --steps 1` run as children (their own process: config 5 alone holds a 3.19-GB canvas and
    a 6.4-GB tile pool), condensed to ms_per_step / value / parity / roofline.  A child that fails or times out leaves {'error': ...}: the headline line must not die with it."""
    out = {}
    for cfg in (3, 4, 5):
        t0 = time.perf_counter()
        try:
            cmd = [sys.executable, os.path.abspath(__file__), '--config', str(cfg), '--steps', '1', '--warmup', '1']
            if args.no_cpu_baseline:
                cmd.append('--no-cpu-baseline')
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=150)      # (12-14 s when healthy)
            line = next((l for l in reversed(p.stdout.strip().splitlines()) if l.startswith('{')), None)
            if p.returncode != 0 or line is None:
                out['config%d' % cfg] = {'error': 'rc {}: {}'.format(p.returncode, (p.stderr or p.stdout)[-300:]), 'parity_ok': False if 'parity gate failed' in (p.stderr or '') else None}
                continue
            r = json.loads(line)
            rf = r.get('roofline') or {}
            out['config%d' % cfg] = {'workload': r['config']['workload'], 'metric': r['metric'], 'value': r['value'], 'unit': r['unit'], 'ms_per_step': r['ms_per_step'], 'steps': r['steps'],
                                     'scaling': r['scaling'], 'dtype': r['dtype'], 'tflops_algorithmic': r['config'].get('tflops_algorithmic'),
                                     'parity_max_abs_vs_oracle': r['config'].get('parity_max_abs_vs_oracle'), 'parity_ok': r['config'].get('parity_ok'),
                                     'roofline': {k: rf.get(k) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac')} if rf else None,
                                     'roofline_hbm': ({k: r['roofline_hbm'].get(k) for k in ('kernel', 'achieved', 'peak', 'unit', 'frac', 'ms')} if r.get('roofline_hbm') else None),
                                     'cpu_baseline': ({k: r['cpu_baseline'].get(k) for k in ('value', 'unit', 'cores', 'kind', 'sample')} if r.get('cpu_baseline') else None),
                                     'child_wall_s': round(time.perf_counter() - t0, 1)}
        except Exception as e:
            out['config%d' % cfg] = {'error': repr(e)[:300]}
    out['note'] = 'one timed step each (after one warm-up step) of `python bench.py --config 3 | 4 | 5`, run as children of this command; full lines: profiles/r06/bench_c{3,4,5}.json'
    return out


def _pmc_live():
    """rocprofv3 --pmc over a short child of this command (counters only: no tracing beside them), three passes -- FETCH_SIZE | WRITE_SIZE + GRBM_GUI_ACTIVE | eight SQ
    counters (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not share a pass) -- condensed by tools/pmc_collect.py into the table _pmc_table() reads.  None when
    rocprofv3 is not on the box, a pass fails or takes too long: the caller then falls back to the committed table."""
    import shutil
    import tempfile
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    collect = os.path.join(ROOT, 'tools', 'pmc_collect.py')
    if not exe or not os.path.exists(collect):
        return None
    if os.environ.get('HSA_TOOLS_LIB') or any(k.startswith(('ROCPROF', 'ROCP_', 'ROCTRACER')) for k in os.environ):
        return None      # this very run is being profiled: a profiler inside a profiled process tree is asking for trouble -- the committed table serves
    out = tempfile.mkdtemp(prefix='moe_pmc_', dir='/tmp')
    steps, warm = 1, 1
    child = [sys.executable, os.path.abspath(__file__), '--steps', str(steps), '--warmup', str(warm), '--no-cpu-baseline', '--sustain', '0', '--no-noise-input', '--no-dropin-loop',
             '--no-extras', '--no-configs', '--no-pmc', '--no-floor']
    regex = 'conv3x3_ps4|arsb32c|arsb_sq|conv64_sq|conv64_q8|conv64_x3'
    env = dict(os.environ, TMPDIR='/tmp')
    t0 = time.perf_counter()
    try:
        for name, counters in (('fetch', ['FETCH_SIZE']), ('grbm', ['WRITE_SIZE', 'GRBM_GUI_ACTIVE']),
                               ('sq', ['SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_MFMA', 'SQ_LDS_BANK_CONFLICT'])):
            cmd = [exe, '--pmc'] + counters + ['--kernel-include-regex', regex, '-d', os.path.join(out, 'pmc_' + name), '-o', 'pmc', '-f', 'csv', '--'] + child
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=90, env=env, cwd='/tmp')      # (~15 s a pass when healthy)
            if p.returncode != 0:
                return None
        p = subprocess.run([sys.executable, collect, out, str(steps + warm)], capture_output=True, text=True, timeout=60)
        d = json.load(open(os.path.join(out, 'pmc_bench.json')))
        g = d.get('groups', {})
        if not g:
            return None
        for v in g.values():
            v['note'] = (v.get('note', '') + ' | rocprofv3 --pmc passes of a {}-frame child of this run, {:.0f} s'.format(steps + warm, time.perf_counter() - t0)).strip(' |')
        return g
    except Exception:
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def _pmc_table():
    """{layer key: {'hbm_bytes_per_frame', 'launches_per_frame', ...}} from the committed rocprofv3 PMC passes of THIS command
    (tools/pmc_bench.sh -> tools/pmc_collect.py -> profiles/pmc_bench.json), or {} when absent."""
    try:
        d = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_bench.json')))
        from moephoto_amd.build import source_digest
        if d.get('source_sha256') != source_digest():      # counters of ANOTHER tree: no traffic figure rather than a stale one
            _pmc_table.stale = 'profiles/pmc_bench.json was collected on other sources (digest {} != {}): rerun tools/pmc_bench.sh'.format(str(d.get('source_sha256'))[:12], source_digest()[:12])
            return {}
        return d.get('groups', {})
    except Exception:
        return {}


if __name__ == '__main__':
    main()
