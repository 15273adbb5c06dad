#!/usr/bin/env python
"""Headline benchmark: megapixels/s of tiled 4x super-resolution of 1080p frames (model a4 = Net4x,
256-px tiles with 5-px overlap: BASELINE.json configs[1]) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path (doCrop: tile gather -> Net4x on the MFMA kernels -> stitch) over one batch
of N frames that are already resident in HBM.  N = 1: one frame.  N > 1: the N frames' tiles are sharded
round-robin over the ranks, exchanged with one RCCL all-to-all, and rank f % N stitches frame f (weak scaling:
every rank computes 40 tiles and stitches one frame per step).  `value` = input megapixels of all frames / s.

Weights: `a4` is absent from the reference mount (.MISSING_LARGE_BLOBS); a synthetic Net4x in the zoo's exact
schema/format is used (tests/golden_defs.py: a2's real trunk + its upsampler duplicated), written and re-read
through the legacy-format file path.  Input: seeded natural-image-like synthetic frame (the regime SR nets are
built for; DESIGN.md discusses white-noise input and the fp16x3 mode).

Extra objects on the JSON line:
  roofline      the dominant kernel (3x3 64->256 implicit-GEMM conv at 2x resolution, 59.8 % of the FLOPs):
                algorithmic FLOPs / launch time from hipEvents recorded on the launch stream inside the timed steps
  cpu_baseline  the oracle (a port of the reference's PyTorch-CPU fp32 path, proven equal to it on the goldens)
                timed on this host on a bounded sample of the same workload (rank 0, N = 1 only); the same
                sample doubles as the parity gate (max-abs error of the engine's tiles vs the oracle).
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--precision', default='fp16', choices=['fp16', 'fp16x3'])
    ap.add_argument('--tiles-per-batch', type=int, default=0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-tiles', type=int, default=1, help='tiles of the frame timed on the CPU oracle')
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
    sd_np = gd.synth_state_dict('a4', load_state_dict_file)
    if world > 1:
        from moephoto_amd.dist import broadcast_state_dict
        sd_t = broadcast_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()} if rank == 0 else None, src=0, device=dev)
        sd_np = type(sd_np)((k, v.numpy()) for k, v in sd_t.items())
    save_state_dict_file(sd_np, wpath)
    runSR.mode_switch['a4'] = (wpath, runSR.mode_switch['a4'][1])
    opt = runSR.getOpt({'op': 'SR', 'model': 'a', 'scale': SCALE, 'ensemble': 0})
    model = opt.modelCached.set_precision(args.precision)

    # ---- input frames, resident in HBM ------------------------------------------------------------------
    nframes = world
    frames_np = [gd.natural_image(1000 + f, FRAME) for f in range(nframes)]
    frames = list(torch.stack([torch.from_numpy(a) for a in frames_np]).to(dev).half().unbind(0))   # slices of one tensor
    plan = ip._plan_for(opt, frames[0].shape)
    assert plan.n_tiles == 40, plan.n_tiles

    def step():
        if world == 1:
            return ip.doCrop(opt, frames[0])
        from moephoto_amd.dist import run_frames
        return run_frames(opt, frames, out_dtype=torch.float16, max_tiles_per_batch=args.tiles_per_batch)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        out = step()
    model.set_profile('up1')          # hipEvent pairs around the 64->256 @2x convs (both branches), on the launch stream
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    dt = time.perf_counter() - t0
    prof = model.get_profile()
    model.set_profile(None)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == 'nccl' else None)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    in_mp = nframes * FRAME[1] * FRAME[2] / 1e6
    value = in_mp / (ms_per_step / 1e3)

    res = {
        'metric': 'megapixels/sec (input), 1080p 4x SR (Net4x a4), 256-px tiles with overlap',
        'value': round(value, 3), 'unit': 'MP/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'fp16' if args.precision == 'fp16' else 'fp16x3', 'data': 'synthetic',
        'config': {'workload': 'BASELINE configs[1]: {} frame(s) 1920x1080 RGB -> 7680x4320, model a4 (Net4x, synthetic weights in zoo format), '
                               'crop 256 pad 5 align 8 -> 40 tiles/frame, fp16 I/O, fp16 MFMA operands + fp32 accumulate'.format(nframes),
                   'frames_per_step': nframes, 'tiles_per_frame': plan.n_tiles, 'output_mp_per_s': round(value * SCALE * SCALE, 2),
                   'tflops_algorithmic': round(nframes * 3 * FRAME[1] * FRAME[2] * MFLOP_PER_PX_PLANE * 1e6 / (ms_per_step / 1e3) / 1e12, 2),
                   'parallelism': 'tile-parallel x{} (round-robin tiles, all-to-all of tile results, stitch on rank f%N)'.format(world) if world > 1 else 'single GPU',
                   'precision': args.precision},
    }
    if prof['launches'] > 0:
        ach = prof['flops'] / (prof['total_ms'] / 1e3) / 1e12
        res['roofline'] = {'bound': 'mfma', 'kernel': 'conv3x3_sp_kernel<3> (3x3 64->256 @2x res, +bias +PixelShuffle(2) +PReLU, fused 64->1 tail taps)',
                           'achieved': round(ach, 1), 'peak': PEAK_FP16_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(ach / PEAK_FP16_TFLOPS, 4),
                           'launches': prof['launches'], 'avg_launch_ms': round(prof['total_ms'] / prof['launches'], 4),
                           'gflop_per_launch': round(prof['flops'] / prof['launches'] / 1e9, 2), 'traffic': _pmc_traffic()}

    # ---- CPU baseline + parity gate (rank 0, N = 1) -----------------------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import nets as onets        # test infrastructure: the checker / baseline, never the product path
        import ctypes
        ks = [9, 10, 18, 39][:max(1, args.cpu_tiles)]
        pool = torch.empty(plan.pool_elems(3), dtype=torch.float32, device=dev)
        y = torch.empty((3, plan.outH, plan.outW), dtype=torch.float16, device=dev)
        xs = frames[0]
        sC, sH, sW = xs.stride()
        _lib.check(_lib.lib().moe_run_plan_ex(model._h, plan._h, xs.data_ptr(), _lib.F16, sC, sH, sW, y.data_ptr(), _lib.F16, args.tiles_per_batch,
                                              ctypes.c_void_p(pool.data_ptr()), 0, 1, 1, torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        off = plan.tile_offsets(3)
        x16 = frames[0].float().cpu().numpy()          # the same fp16-quantised input the engine saw
        torch.set_num_threads(max(1, os.cpu_count() or 1))
        px, cpu_s, err = 0, 0.0, 0.0
        for k in ks:
            top, bottom, left, right = plan.tiles[k][:4]
            xt = np.ascontiguousarray(x16[:, None, top:bottom, left:right])
            c0 = time.perf_counter()
            want = onets.forward('net4x', sd_np, xt).numpy()[:, 0]
            cpu_s += time.perf_counter() - c0
            px += (bottom - top) * (right - left)
            got = pool[off[k]:off[k] + want.size].reshape(want.shape).cpu().numpy()
            err = max(err, float(np.abs(got - want).max()))
        tile_px_total = sum((t[1] - t[0]) * (t[3] - t[2]) for t in plan.tiles)
        frame_s = cpu_s / px * tile_px_total              # all 40 tiles at the sampled per-pixel rate
        res['cpu_baseline'] = {'value': round(FRAME[1] * FRAME[2] / 1e6 / frame_s, 5), 'unit': 'MP/s', 'cores': torch.get_num_threads(),
                               'kind': 'port', 'sample': '{} of 40 tiles ({} tile pixels x 3 planes) of the same frame through the fp32 oracle (torch/oneDNN conv '
                               'backend), {:.1f} s; extrapolated to the frame by tile pixels'.format(len(ks), px, cpu_s),
                               'cpu_model': _cpu_model()}
        res['config']['parity_max_abs_vs_oracle'] = float('{:.3e}'.format(err))
        res['config']['parity_tolerance'] = 1e-3
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _cpu_model():
    try:
        for l in open('/proc/cpuinfo'):
            if l.startswith('model name'):
                return l.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def _pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/), if present."""
    p = os.path.join(ROOT, 'profiles', 'pmc_dominant.json')
    try:
        return json.load(open(p)).get('hbm_bytes_per_launch')
    except Exception:
        return None


if __name__ == '__main__':
    main()
