#!/usr/bin/env python
"""Headline benchmark: megapixels/s of tiled 4x super-resolution of 1080p frames (model a4 = Net4x,
256-px tiles with 5-px overlap: BASELINE.json configs[1]) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (doCrop: tile gather -> Net4x on the MFMA kernels -> stitch) over one batch
of N frames that are already resident in HBM.  N = 1: one frame.  N > 1: the N frames' tiles are sharded
round-robin over the ranks, exchanged with one RCCL all-to-all, and rank f % N stitches frame f (weak scaling:
every rank computes 40 tiles and stitches one frame per step).  `value` = input megapixels of all frames / s,
from EXACTLY --steps steps bracketed by barrier + synchronize, max over ranks.

Arithmetic: the product default ('auto' -> 'mixed' for Net4x: fp16 MFMA operands, fp32 accumulate, hi+lo trunk stream,
split operands on conv_input2 + the first ARSB, ~22-bit tail weights), the mode every parity test asserts at 1e-3.

Weights: `a4` is absent from the reference mount (.MISSING_LARGE_BLOBS); a synthetic Net4x in the zoo's exact
schema/format is used (tests/golden_defs.py: a2's real trunk + its upsampler duplicated), written and re-read
through the legacy-format file path.  Inputs: the headline is a seeded natural-image-like synthetic frame; the
SURVEY 8(d) input (seed-0 uniform uint8 noise / 255) is timed and parity-checked as a second input (`inputs`).

Extra objects on the JSON line:
  roofline        the dominant kernel (3x3 64->256 implicit-GEMM conv at 2x resolution, 59.8 % of the FLOPs):
                  algorithmic FLOPs / launch time from hipEvents recorded on the launch stream inside the timed steps
  roofline_trunk  the same for the one-launch ARSBs with fp16 operands (two 64->64 implicit GEMMs, the shape the north star names)
  sustained       a >= --sustain second leg after the timed steps (the part is power-capped: short bursts run faster)
  cpu_baseline    the oracle (a port of the reference's PyTorch-CPU fp32 path, proven equal to it on the goldens) timed on
                  this host with one socket's physical cores on a full tile row (8 tiles) + the ragged corner of the same
                  frame, and on BASELINE config 1 (256x256, a2) in full; rank 0 only.  The same tiles are the parity gate:
                  the worst max-abs error of the engine's tiles vs the oracle must be <= 1e-3 or the run exits non-zero.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'tests')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

FRAME = (3, 1080, 1920)
CROP, PAD, SCALE = 256, 5, 4
MFLOP_PER_PX_PLANE = 3.9456          # Net4x conv FLOPs per LR pixel per plane
PEAK_FP16_TFLOPS = 2500.0            # MI355X dense fp16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense)
PARITY_TOL = 1e-3


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
    args = ap.parse_args()

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
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node {}'.format(args.gpus))
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

    # ---- input frames, resident in HBM ------------------------------------------------------------------
    nframes = world

    def make_frames(kind):
        if kind == 'natural':
            arr = [gd.natural_image(1000 + f, FRAME) for f in range(nframes)]
        else:           # seed-0 uniform uint8 -> /255
            arr = [(gd.noise_u8(f, FRAME).astype(np.float32) / np.float32(255)) for f in range(nframes)]
        return list(torch.stack([torch.from_numpy(a) for a in arr]).to(dev).half().unbind(0))   # slices of one tensor
    frames = make_frames('natural')
    plan = ip._plan_for(opt, frames[0].shape)
    assert plan.n_tiles == 40, plan.n_tiles

    def step(fr):
        if world == 1:
            return ip.doCrop(opt, fr[0])
        from moephoto_amd.dist import run_frames
        return run_frames(opt, fr, out_dtype=torch.float16, max_tiles_per_batch=args.tiles_per_batch)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed(fr, steps):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(fr)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == 'nccl' else None)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    in_mp = nframes * FRAME[1] * FRAME[2] / 1e6
    for _ in range(args.warmup):
        step(frames)
    model.set_profile('up1,arsb')      # hipEvent pairs around the 64->256 @2x convs (both branches) and the one-launch ARSBs, on the launch stream
    dt = timed(frames, args.steps)
    prof_up1, prof_c2 = model.get_profile(all_keys=True)
    model.set_profile(None)
    ms_per_step = dt / args.steps * 1e3
    value = in_mp / (ms_per_step / 1e3)

    res = {
        'metric': 'megapixels/sec (input), 1080p 4x SR (Net4x a4), 256-px tiles with overlap',
        'value': round(value, 3), 'unit': 'MP/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'fp16', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: {} frame(s) 1920x1080 RGB -> 7680x4320, model a4 (Net4x, synthetic weights in zoo format), '
                               'crop 256 pad 5 align 8 -> 40 tiles/frame, fp16 I/O, fp16 MFMA operands + fp32 accumulate'.format(nframes),
                   'frames_per_step': nframes, 'tiles_per_frame': plan.n_tiles, 'output_mp_per_s': round(value * SCALE * SCALE, 2),
                   'tflops_algorithmic': round(nframes * 3 * FRAME[1] * FRAME[2] * MFLOP_PER_PX_PLANE * 1e6 / (ms_per_step / 1e3) / 1e12, 2),
                   'parallelism': 'tile-parallel x{} (round-robin tiles, all-to-all of tile results, stitch on rank f%N)'.format(world) if world > 1 else 'single GPU',
                   'precision': precision, 'input': 'natural-image-like synthetic frame (tests/golden_defs.natural_image)'},
    }

    def roof(prof, kernel):
        ach = prof['flops'] / (prof['total_ms'] / 1e3) / 1e12
        return {'bound': 'mfma', 'kernel': kernel, 'achieved': round(ach, 1), 'peak': PEAK_FP16_TFLOPS, 'unit': 'TFLOP/s',
                'frac': round(ach / PEAK_FP16_TFLOPS, 4), 'launches': prof['launches'], 'avg_launch_ms': round(prof['total_ms'] / prof['launches'], 4),
                'gflop_per_launch': round(prof['flops'] / prof['launches'] / 1e9, 2)}
    if prof_up1['launches'] > 0:
        res['roofline'] = roof(prof_up1, 'conv3x3_sp_kernel<3> (3x3 64->256 @2x res, +bias +PixelShuffle(2) +PReLU, fused 64->1 tail taps)')
        res['roofline']['traffic'] = _pmc_traffic(res['roofline']['gflop_per_launch'])
    if prof_c2['launches'] > 0:
        res['roofline_trunk'] = roof(prof_c2, 'arsb_fused_kernel (one ARSB per launch: two 3x3 64->64 convs @1x res + PReLU + hi/lo residual stream)')

    # ---- sustained leg (power-capped part: a 0.5 s burst flatters the clock) -------------------------------
    if args.sustain > 0:
        n_s = max(args.steps, int(args.sustain / (ms_per_step / 1e3)) + 1)
        dts = timed(frames, n_s)
        res['sustained'] = {'seconds': round(dts, 2), 'steps': n_s, 'ms_per_step': round(dts / n_s * 1e3, 3), 'value': round(in_mp / (dts / n_s), 3), 'unit': 'MP/s'}

    # ---- second input: uniform uint8 noise ------------------------------------------------------
    inputs = {'natural': {'value': res['value'], 'ms_per_step': res['ms_per_step']}}
    noise_frames = None
    if not args.no_noise_input:
        noise_frames = make_frames('noise_u8')
        step(noise_frames)
        dtn = timed(noise_frames, args.steps)
        inputs['noise_u8'] = {'value': round(in_mp / (dtn / args.steps), 3), 'ms_per_step': round(dtn / args.steps * 1e3, 3)}
    res['inputs'] = inputs

    # ---- CPU baseline + parity gate (rank 0) -----------------------------------------------------------------
    parity_ok = True
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import nets as onets        # test infrastructure: the checker / baseline, never the product path
        import ctypes
        ncores, nthreads = _one_socket_cores()
        torch.set_num_threads(nthreads)
        # tile row 1 of the 5 x 8 grid (8 tiles incl. the ragged right column), then the ragged bottom-right corner
        ks = ([8, 9, 10, 11, 12, 13, 14, 15, 39] + [35, 0, 20])[:max(1, args.cpu_tiles)]
        if world > 1:
            ks = ks[1:2] + ks[8:9]            # multi-GPU runs: a bounded check (the other ranks wait)
        off = plan.tile_offsets(3)
        tile_px_total = sum((t[1] - t[0]) * (t[3] - t[2]) for t in plan.tiles)

        def check(fr, tiles):
            pool = torch.empty(plan.pool_elems(3), dtype=torch.float32, device=dev)
            y = torch.empty((3, plan.outH, plan.outW), dtype=torch.float16, device=dev)
            sC, sH, sW = fr.stride()
            _lib.check(_lib.lib().moe_run_plan_ex(model._h, plan._h, fr.data_ptr(), _lib.F16, sC, sH, sW, y.data_ptr(), _lib.F16, args.tiles_per_batch,
                                                  ctypes.c_void_p(pool.data_ptr()), 0, 1, 1, torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            x16 = fr.float().cpu().numpy()          # the same fp16-quantised input the engine saw
            px, cpu_s, errs = 0, 0.0, []
            for k in tiles:
                top, bottom, left, right = plan.tiles[k][:4]
                xt = np.ascontiguousarray(x16[:, None, top:bottom, left:right])
                c0 = time.perf_counter()
                want = onets.forward('net4x', sd_np, xt).numpy()[:, 0]
                cpu_s += time.perf_counter() - c0
                px += (bottom - top) * (right - left)
                got = pool[off[k]:off[k] + want.size].reshape(want.shape).cpu().numpy()
                errs.append(float(np.abs(got - want).max()))
            return px, cpu_s, errs
        px, cpu_s, errs = check(frames[0], ks)
        frame_s = cpu_s / px * tile_px_total              # all 40 tiles at the sampled per-pixel rate
        parity = {'natural': {'tiles': ks, 'worst_max_abs': float('{:.3e}'.format(max(errs))), 'per_tile': [float('{:.2e}'.format(e)) for e in errs]}}
        parity_ok = max(errs) <= PARITY_TOL
        if noise_frames is not None:
            kn = ks if world == 1 else ks[:1]
            _, cpu_n, errs_n = check(noise_frames[0], kn)
            parity['noise_u8'] = {'tiles': kn, 'worst_max_abs': float('{:.3e}'.format(max(errs_n))), 'per_tile': [float('{:.2e}'.format(e)) for e in errs_n]}
            parity['noise_u8']['ok'] = max(errs_n) <= PARITY_TOL
        res['cpu_baseline'] = {'value': round(FRAME[1] * FRAME[2] / 1e6 / frame_s, 5), 'unit': 'MP/s', 'cores': ncores, 'threads': nthreads,
                               'kind': 'port', 'sample': '{} of 40 tiles (tile row 1 + the ragged corner: {} tile pixels x 3 planes) of the headline frame through the fp32 '
                               'oracle (torch/oneDNN conv backend), {:.1f} s; extrapolated to the frame by tile pixels'.format(len(ks), px, cpu_s),
                               'cpu_model': _cpu_model(), 'host_logical_cpus': os.cpu_count()}
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
        res['config']['parity_max_abs_vs_oracle'] = parity['natural']['worst_max_abs']
        res['config']['parity_tolerance'] = PARITY_TOL
        res['config']['parity_ok'] = bool(parity_ok)
    if rank == 0:
        print(json.dumps(res))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not parity_ok:
        raise SystemExit('parity gate failed: the engine differs from the oracle by more than {} on the headline input'.format(PARITY_TOL))


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


def _pmc_traffic(gflop_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/), if present.  The counters were
    collected on launches of 12 planes of 256^2 (tools/prof_workload.py); the benchmark's launches carry more planes (8 tiles per
    launch set, split by the 32-bit offset range), so the figure is scaled by the launches' algorithmic FLOPs (traffic is linear in planes)."""
    p = os.path.join(ROOT, 'profiles', 'pmc_dominant.json')
    try:
        d = json.load(open(p))
        return int(d['hbm_bytes_per_launch'] * gflop_per_launch / d['gflop_per_launch'])
    except Exception:
        return None


if __name__ == '__main__':
    main()
