"""bench.py's other legs (imported by it; nothing here is product code):

* `python bench.py --config 3|4|5` -- BASELINE.json's configs[2..4] at full size, one JSON line each in bench.py's contract:
    3  4K frame, denoise l25 (SEDN, synthetic weights) -> x2 SR a2 chain (python/runDN.py:10-16, runSR.py:10-16), 144 + 144 tiles
    4  batch of 64 1080p frames, x4 SR a4: one GPU runs them one after the other (as the reference does: python/server.py:310-357, video.py:349-360);
       under --gpus N the (frame, tile) pairs are dealt round-robin to the ranks (dist.run_frames) -- the total work is fixed: "scaling": "strong"
    5  one 8K frame -> 32K, a4, 512-px tiles, 144 tiles, fp16 canvas of 3.19 GB: the HBM-bound stitch path, with the stitch kernel's own roofline object
* for the default config 2 line: `roofline_hbm_kernels` (the HBM-bound members of the path: tailadd, the stitch kernel, lite's conv1x1 layers, measured with
  hipEvents on the launch stream, algorithmic bytes over 8 TB/s) and `io_edges` (the uint8 -> fp16 and fp16 -> uint8 passes of python/imageProcess.py:245-263
  that the headline excludes: SURVEY section 8(d) wants them reported beside it).
"""
import ctypes
import json
import os
import sys
import time

PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def _events_ms(torch, fn, reps):
    torch.cuda.synchronize()
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _hbm_obj(kernel, what, alg_bytes, ms, extra=None):
    k = {'bound': 'hbm', 'kernel': kernel, 'what': what, 'achieved': round(alg_bytes / (ms / 1e3) / 1e9, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
         'frac': round(alg_bytes / (ms / 1e3) / 1e9 / PEAK_HBM_GBS, 4), 'ms': round(ms, 4), 'algorithmic_bytes': int(alg_bytes), 'traffic': None}
    if extra:
        k.update(extra)
    return k


def io_edges(torch, _lib, dev):
    """The two I/O edges of a 1080p -> 8K job, on the device: interleaved uint8 frame -> planar fp16 (toTorch), planar fp16 canvas -> interleaved uint8
    (toOutput: x 256, clamp, truncate).  Inputs resident in HBM; the PCIe copies around them are not part of it."""
    L = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = {}
    H, W = 1080, 1920
    src = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device=dev)
    dst = torch.empty((3, H, W), dtype=torch.float16, device=dev)
    ms = _events_ms(torch, lambda: _lib.check(L.moe_to_float(src.data_ptr(), _lib.U8, 8, H, W, 3, dst.data_ptr(), _lib.F16, dev.index or 0, stream)), 20)
    out['u8_to_fp16_1080p'] = _hbm_obj('to_float3_kernel<uint8, half> (eight pixels per thread)', '1920x1080x3 interleaved uint8 -> 3 planes fp16 (v / 255)', H * W * 3 * (1 + 2), ms)
    H, W = 4320, 7680
    can = torch.rand((3, H, W), device=dev).half()
    o8 = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    ms = _events_ms(torch, lambda: _lib.check(L.moe_to_output(can.data_ptr(), _lib.F16, H, W, 3, 8, o8.data_ptr(), _lib.U8, dev.index or 0, stream)), 10)
    out['fp16_to_u8_8k'] = _hbm_obj('to_output3_kernel<half, uint8> (eight pixels per thread)', '3 planes fp16 7680x4320 -> interleaved uint8 (x 256, clamp, truncate)', H * W * 3 * (2 + 1), ms)
    out['note'] = 'excluded from `value` (SURVEY 8(d)); beside a 24-ms frame the two passes add %.3f ms' % (out['u8_to_fp16_1080p']['ms'] + out['fp16_to_u8_8k']['ms'])
    return out


def hbm_members(torch, _lib, ip, gd, dev, opt, model, plan, frame, tailadd_prof, frames_timed, tiles_per_batch, load_state_dict_file):
    """roofline objects of the HBM-bound kernels of the headline frame (+ lite's conv1x1 on a lite2 frame of the same size)."""
    L = _lib.lib()
    out = []
    px_hr = 3.0 * plan.outH * plan.outW
    if tailadd_prof and tailadd_prof['launches'] > 0:
        ms = tailadd_prof['total_ms'] / frames_timed
        out.append(_hbm_obj('tailadd_kernel', 'sum of the two branches\' fp32 tail planes + column aprons -> fp32 tile pool: 8 B in + 4 B out per output pixel and plane '
                            '(round 3: tapsum4, 32 B in)', px_hr * 12, ms, {'launches_per_frame': tailadd_prof['launches'] / frames_timed, 'per': 'frame'}))
    # the stitch kernel alone, from a caller-owned tile pool
    pool = torch.empty(plan.pool_elems(3), dtype=torch.float32, device=dev)
    y = torch.empty((3, plan.outH, plan.outW), dtype=torch.float16, device=dev)
    sC, sH, sW = frame.stride()
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(L.moe_run_plan_ex(model._h, plan._h, frame.data_ptr(), _lib.F16, sC, sH, sW, y.data_ptr(), _lib.F16, tiles_per_batch, ctypes.c_void_p(pool.data_ptr()), 0, 1, 1, stream))
    ms = _events_ms(torch, lambda: _lib.check(L.moe_stitch(plan._h, 0, pool.data_ptr(), None, 3, y.data_ptr(), _lib.F16, stream)), 10)
    out.append(_hbm_obj('stitch8r_kernel<4>', "doCrop's blend fold (python/imageProcess.py:120-131,157-172) of 40 fp32 tiles into the fp16 8K canvas: 4 B in + 2 B out per output pixel and plane",
                        px_hr * 6, ms, {'per': 'frame'}))
    del pool, y
    # lite2 on the same frame: its 1x1 layers (conv1x1.hip)
    try:
        from moephoto_amd import runSR
        from moephoto_amd.config import config
        config.modelRoot = gd.ZOO
        ip.modelCache.pop('SRlite2', None)
        ol = runSR.getOpt({'op': 'SR', 'model': 'lite', 'scale': 2, 'ensemble': 0})
        ml = ol.modelCached
        for _ in range(2):
            ip.doCrop(ol, frame)
        keys = ['input2', 'ures.up0', 'uim.up0']
        ml.set_profile(','.join(keys))
        n = 5
        for _ in range(n):
            ip.doCrop(ol, frame)
        torch.cuda.synchronize()
        profs = ml.get_profile(all_keys=True)
        ml.set_profile(None)
        ms = sum(p['total_ms'] for p in profs) / n
        px = 3.0 * frame.shape[-2] * frame.shape[-1]
        # fp16x3 (lite's default): hi + lo in and out.  conv_input2: 256 B in + 256 B out per pixel and plane; each branch's upsampler stage with the folded
        # 48 -> 1 tail: 256 B in, 4 HR pixels x 4 B out
        alg = px * (512 + 2 * (256 + 16))
        out.append(_hbm_obj('conv1x1_kernel (lite2: conv_input2 + the two x2 upsampler stages with the folded 48->1 tail, split operands)',
                            'lite2 on the headline frame (1080p, 256-px tiles): three 1x1 layers per forward', alg, ms,
                            {'per': 'frame', 'launches_per_frame': sum(p['launches'] for p in profs) / n, 'precision': ml.resolved_precision()}))
        ip.modelCache.pop('SRlite2', None)
    except Exception as e:      # (the leg is an extra: the headline line must not die with it)
        out.append({'kernel': 'conv1x1_kernel', 'error': repr(e)})
    return out


# ---------------------------------------------------------------------------------------------------------------------------------------------
def _setup(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    import golden_defs as gd
    from moephoto_amd import _lib, imageProcess as ip, runDN, runSR
    from moephoto_amd.config import config
    from moephoto_amd.weights import load_state_dict_file, save_state_dict_file
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('MOE_FORCE_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    backend = os.environ.get('MOE_DIST_BACKEND', 'nccl')
    _lib.require_device()
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(backend)
    config.deviceId, config.fp16, config.modelRoot, config.tilesPerBatch = local, True, gd.ZOO, args.tiles_per_batch
    ns = dict(np=np, torch=torch, dist=dist, gd=gd, _lib=_lib, ip=ip, runDN=runDN, runSR=runSR, config=config, load_sd=load_state_dict_file, save_sd=save_state_dict_file,
              world=world, rank=rank, local=local, backend=backend, dev=dev)
    return ns


def _synth(ns, key, table, slot):
    path = '/tmp/moe_bench_{}_rank{}.pth'.format(key, ns['rank'])
    ns['save_sd'](ns['gd'].synth_state_dict(key, ns['load_sd']), path)
    table.mode_switch[slot] = (path,) + tuple(table.mode_switch[slot][1:])


def _fence(ns):
    ns['torch'].cuda.synchronize(ns['dev'])
    if ns['world'] > 1:
        ns['dist'].barrier()
        ns['torch'].cuda.synchronize(ns['dev'])


def _all_max(ns, v):
    if ns['world'] > 1:
        t = ns['torch'].tensor([v], dtype=ns['torch'].float64, device=ns['dev'] if ns['backend'] == 'nccl' else None)
        ns['dist'].all_reduce(t, op=ns['dist'].ReduceOp.MAX)
        v = float(t.item())
    return v


def _timed(ns, fn, steps):
    _fence(ns)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    _fence(ns)
    return _all_max(ns, time.perf_counter() - t0)


def _tile_parity(ns, arch, sd, xd, plan, pool, ks, C=3):
    """(seconds of CPU oracle, tile pixels, worst max-abs) of the tiles `ks` of a plan whose raw fp32 tile results are in `pool`."""
    from oracle import nets as onets        # test infrastructure: the checker / baseline, never the product path
    np = ns['np']
    off = plan.tile_offsets(C)
    x16 = xd.float().cpu().numpy()
    cpu_s, px, worst = 0.0, 0, 0.0
    for k in ks:
        top, bottom, left, right = plan.tiles[k][:4]
        xt = np.ascontiguousarray(x16[:, None, top:bottom, left:right])
        c0 = time.perf_counter()
        want = onets.forward(arch, sd, xt).numpy()[:, 0]
        cpu_s += time.perf_counter() - c0
        px += (bottom - top) * (right - left)
        got = pool[off[k]:off[k] + want.size].reshape(want.shape).cpu().numpy()
        worst = max(worst, float(np.abs(got - want).max()))
    return cpu_s, px, worst


def _pool_of(ns, opt, plan, xd, stitch_to=None):
    torch, _lib = ns['torch'], ns['_lib']
    C = xd.shape[0]
    pool = torch.empty(plan.pool_elems(C), dtype=torch.float32, device=xd.device)
    sC, sH, sW = xd.stride()
    dt = _lib.F16 if xd.dtype == torch.float16 else _lib.F32
    out_p, out_dt = (stitch_to.data_ptr(), _lib.F16 if stitch_to.dtype == torch.float16 else _lib.F32) if stitch_to is not None else (None, _lib.F32)
    _lib.check(_lib.lib().moe_run_plan_ex(opt.modelCached._h, plan._h, xd.data_ptr(), dt, sC, sH, sW, out_p, out_dt, 0, ctypes.c_void_p(pool.data_ptr()), 0, 1,
                                          1 if stitch_to is not None else 0, torch.cuda.current_stream().cuda_stream))
    return pool


def _base(ns, args, metric, value, ms_per_step, steps, scaling, dtype, workload, extra_cfg):
    res = {'metric': metric, 'value': round(value, 3), 'unit': 'MP/s', 'n_gpus': ns['world'], 'steps': steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
           'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None, 'dtype': dtype, 'data': 'synthetic', 'config': dict({'workload': workload}, **extra_cfg)}
    if ns['world'] > 1:
        res['config']['wire'] = args.wire
    d = ns['_lib'].device_info(ns['local'])
    peak = d['compute_units'] * 4 * 1024 * d['clock_khz'] * 1e3 / 1e12
    res['device'] = {'compute_units': d['compute_units'], 'max_clock_ghz': round(d['clock_khz'] / 1e6, 3), 'peak_fp16_mfma_tflops': round(peak, 1)}
    return res, peak


def _mfma_obj(label, key, prof, alg_flops, peak, steps):
    secs = prof['total_ms'] / 1e3
    return {'bound': 'mfma', 'kernel': label, 'layer_key': key, 'achieved': round(alg_flops / secs / 1e12, 1), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
            'frac': round(alg_flops / secs / 1e12 / peak, 4), 'achieved_executed': round(prof['flops'] / secs / 1e12, 1), 'launches': prof['launches'],
            'avg_launch_ms': round(prof['total_ms'] / max(1, prof['launches']), 4), 'ms_per_step': round(prof['total_ms'] / steps, 3), 'traffic': None}


def _finish(ns, res, ok):
    if ns['rank'] == 0:
        print(json.dumps(res))
        sys.stdout.flush()
    if ns['world'] > 1:
        torch, dist = ns['torch'], ns['dist']
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=ns['dev'] if ns['backend'] == 'nccl' else None)
        dist.broadcast(flag, src=0)
        ok = bool(int(flag.item()))
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit('parity gate failed')


def run_config(cfg, args):
    ns = _setup(args)
    {3: _config3, 4: _config4, 5: _config5}[cfg](ns, args)


def _run_frames(ns, opt, frames, args, bands=False):
    from moephoto_amd.dist import run_frames
    return run_frames(opt, frames, out_dtype=ns['torch'].float16, max_tiles_per_batch=args.tiles_per_batch, bands=bands, wire=args.wire)


def _config3(ns, args):
    torch, gd, ip, runDN, runSR, config, np = ns['torch'], ns['gd'], ns['ip'], ns['runDN'], ns['runSR'], ns['config'], ns['np']
    world, dev = ns['world'], ns['dev']
    config.crop_sr = config.crop_dn = config.crop_dns = 256
    _synth(ns, 'l25', runDN, '25')
    odn = runDN.getOpt({'op': 'DN', 'model': '25'})
    osr = runSR.getOpt({'op': 'SR', 'model': 'a', 'scale': 2, 'ensemble': 0})
    shape = (3, 2160, 3840)
    frames = list(torch.stack([torch.from_numpy(gd.natural_image(31 + f, shape)) for f in range(world)]).to(dev).half().unbind(0))

    def step():
        if world == 1:
            return ip.doCrop(osr, ip.doCrop(odn, frames[0]))
        mid = _run_frames(ns, odn, frames, args)                 # N frames in flight: the DN tiles are dealt over the ranks, rank r stitches frame r ...
        return {f: ip.doCrop(osr, v) for f, v in mid.items()}    # ... and super-resolves it (the denoised frame lives on that rank only)
    for _ in range(max(1, args.warmup)):
        step()
    steps = args.steps if args.steps_given else 5
    mdn, msr = odn.modelCached, osr.modelCached
    mdn.set_profile('rb0,rb2')
    msr.set_profile('up0,arsb')
    dt = _timed(ns, step, steps)
    pdn, psr = mdn.get_profile(all_keys=True), msr.get_profile(all_keys=True)
    mdn.set_profile(None)
    msr.set_profile(None)
    ms = dt / steps * 1e3
    in_mp = world * shape[1] * shape[2] / 1e6
    res, peak = _base(ns, args, 'megapixels/sec (input), 4K denoise (l25, SEDN) -> x2 SR (a2) chain, 256-px tiles', in_mp / (ms / 1e3), ms, steps, 'weak',
                      'fp16 (SEDN: fp16 operands; a2: fp16 operands + hi/lo stream, split operands on 4 ARSBs)',
                      'BASELINE configs[2]: {} frame(s) 3840x2160 RGB, [DN l25 (synthetic weights), SR a x2 (real weights)], 144 + 144 tiles per frame, fp16 I/O -> 7680x4320'.format(world),
                      {'frames_per_step': world, 'tiles_per_frame': [144, 144], 'tflops_algorithmic': round(in_mp * 1e6 * 3 * (7.6045 + 1.5587) * 1e6 / (ms / 1e3) / 1e12, 2),
                       'parallelism': 'single GPU' if world == 1 else 'DN tile-parallel x{} (one frame per rank in flight), SR on the stitching rank'.format(world)})
    px = 3.0 * shape[1] * shape[2] * world
    ks = []
    p = {'total_ms': pdn[0]['total_ms'] + pdn[1]['total_ms'], 'launches': pdn[0]['launches'] + pdn[1]['launches'], 'flops': pdn[0]['flops'] + pdn[1]['flops']}
    if p['launches']:
        ks.append(_mfma_obj("SEDN's 3x3 64->64 convs (rblock.0, rblock.2 of the 16 blocks: conv3x3_rw<1> / <4>)", 'rb0+rb2', p, px * steps * 16 * 2 * 2 * 64 * 64 * 9, peak, steps))
    if psr[0]['launches']:
        ks.append(_mfma_obj('a2: both branches\' 3x3 64->256 @1x res with the fused 64->1 tail (conv3x3_ps4)', 'up0', psr[0], px * steps * 2 * 2 * 256 * 64 * 9, peak, steps))
    ks.sort(key=lambda k: -k['ms_per_step'])
    if ks:
        res['roofline'] = ks[0]
        res['roofline_kernels'] = ks
    ok = True
    if ns['rank'] == 0 and not args.no_cpu_baseline:
        plan = ip._plan_for(odn, frames[0].shape)
        d16 = torch.empty(shape, dtype=torch.float16, device=dev)
        pool = _pool_of(ns, odn, plan, frames[0], stitch_to=d16)
        sd = gd.synth_state_dict('l25', ns['load_sd'])
        c1, px1, w1 = _tile_parity(ns, 'sedn', sd, frames[0], plan, pool, [143])          # the ragged bottom-right corner (112 x 240)
        del pool
        plan2 = ip._plan_for(osr, d16.shape)
        pool = _pool_of(ns, osr, plan2, d16)
        c2, px2, w2 = _tile_parity(ns, 'net2x', gd.state_dict_for('a2', ns['load_sd']), d16, plan2, pool, [40])
        del pool
        tpx1 = sum((t[1] - t[0]) * (t[3] - t[2]) for t in plan.tiles)
        tpx2 = sum((t[1] - t[0]) * (t[3] - t[2]) for t in plan2.tiles)
        frame_s = c1 / px1 * tpx1 + c2 / px2 * tpx2
        import bench
        ncores, nthreads = bench._one_socket_cores()
        res['cpu_baseline'] = {'value': round(shape[1] * shape[2] / 1e6 / frame_s, 5), 'unit': 'MP/s', 'cores': ncores, 'kind': 'port',
                               'sample': 'one tile of each step through the fp32 oracle (l25: tile 143, {} px, {:.1f} s; a2: tile 40, {} px, {:.1f} s), extrapolated by tile pixels'.format(px1, c1, px2, c2)}
        res['config']['parity_max_abs_vs_oracle'] = float('{:.3e}'.format(max(w1, w2)))
        res['config']['parity_tolerance'] = 1e-3
        ok = max(w1, w2) <= 1e-3
        res['config']['parity_ok'] = bool(ok)
    _finish(ns, res, ok)


def _config4(ns, args):
    torch, gd, ip, runSR, config, np = ns['torch'], ns['gd'], ns['ip'], ns['runSR'], ns['config'], ns['np']
    world, dev = ns['world'], ns['dev']
    config.crop_sr = 256
    _synth(ns, 'a4', runSR, 'a4')
    opt = runSR.getOpt({'op': 'SR', 'model': 'a', 'scale': 4, 'ensemble': 0})
    NF, shape = 64, (3, 1080, 1920)
    base = [torch.from_numpy(gd.natural_image(1000 + f, shape)) for f in range(4)]
    stack = torch.stack([base[f % 4] for f in range(NF)]).to(dev).half()          # 64 frames resident in HBM (four distinct images)
    frames = list(stack.unbind(0))

    def step():
        if world == 1:
            for f in frames:                  # the reference runs a batch's frames one after the other (python/video.py:349-360)
                ip.doCrop(opt, f)
            return
        # round-robin (frame, tile) ownership over the ranks, in groups of N frames: every rank computes one frame's worth of tiles per group and stitches one
        # frame of it; the all-to-all of a group runs behind the next group's convolutions (dist.run_frames_overlapped)
        from moephoto_amd.dist import run_frames_overlapped
        run_frames_overlapped(opt, frames, out_dtype=torch.float16, max_tiles_per_batch=args.tiles_per_batch, wire=args.wire)
    step()
    steps = args.steps if args.steps_given else 3
    model = opt.modelCached
    model.set_profile('convt_R1.up1,u.up1,arsb')
    dt = _timed(ns, step, steps)
    profs = model.get_profile(all_keys=True)
    model.set_profile(None)
    ms = dt / steps * 1e3
    in_mp = NF * shape[1] * shape[2] / 1e6
    res, peak = _base(ns, args, 'megapixels/sec (input), batch of 64 1080p frames, 4x SR (a4), 256-px tiles', in_mp / (ms / 1e3), ms, steps, 'strong', 'fp16',
                      'BASELINE configs[3]: 64 frames 1920x1080 RGB -> 7680x4320 per step, model a4 (synthetic weights), 2560 tiles per step'
                      + (' dealt round-robin over {} GPUs ((frame, tile) -> rank; groups of {} frames, the exchange of a group behind the next group\'s convolutions)'.format(world, world) if world > 1 else ', one GPU, frame after frame'),
                      {'frames_per_step': NF, 'tiles_per_step': 40 * NF, 'ms_per_frame': round(ms / NF, 3),
                       'tflops_algorithmic': round(in_mp * 1e6 * 3 * 3.9456e6 / (ms / 1e3) / 1e12, 2)})
    ks = []
    px = 3.0 * shape[1] * shape[2] * NF / world
    for (key, label, fl), p in zip((('convt_R1.up1', 'R-branch 3x3 64->256 @2x res + fused tail, split activations (conv3x3_ps4<2>)', 4 * 2 * 256 * 64 * 9),
                                    ('u.up1', 'U-branch 3x3 64->256 @2x res + fused tail (conv3x3_ps4<1>)', 4 * 2 * 256 * 64 * 9),
                                    ('arsb', 'arsb32c_kernel (5 of 6 ARSBs)', 5 * 2 * 2 * 64 * 64 * 9)), profs):
        if p['launches']:
            ks.append(_mfma_obj(label, key, p, px * steps * fl, peak, steps))
    ks.sort(key=lambda k: -k['ms_per_step'])
    if ks:
        res['roofline'] = ks[0]
        res['roofline_kernels'] = ks
    ok = True
    if ns['rank'] == 0 and not args.no_cpu_baseline:
        plan = ip._plan_for(opt, frames[0].shape)
        sd = gd.synth_state_dict('a4', ns['load_sd'])
        worst, cpu_s, px_s = 0.0, 0.0, 0
        for f, k in ((5, 9), (62, 39)):                # two tiles of two different frames of the batch
            pool = _pool_of(ns, opt, plan, frames[f])
            c, p_, w = _tile_parity(ns, 'net4x', sd, frames[f], plan, pool, [k])
            del pool
            worst, cpu_s, px_s = max(worst, w), cpu_s + c, px_s + p_
        tpx = sum((t[1] - t[0]) * (t[3] - t[2]) for t in plan.tiles)
        import bench
        ncores, _ = bench._one_socket_cores()
        res['cpu_baseline'] = {'value': round(shape[1] * shape[2] / 1e6 / (cpu_s / px_s * tpx), 5), 'unit': 'MP/s', 'cores': ncores, 'kind': 'port',
                               'sample': 'tile 9 of frame 5 and tile 39 of frame 62 through the fp32 oracle ({} px, {:.1f} s), extrapolated to a frame by tile pixels'.format(px_s, cpu_s)}
        res['config']['parity_max_abs_vs_oracle'] = float('{:.3e}'.format(worst))
        res['config']['parity_tolerance'] = 1e-3
        ok = worst <= 1e-3
        res['config']['parity_ok'] = bool(ok)
    _finish(ns, res, ok)


def _config5(ns, args):
    torch, gd, ip, runSR, config, np, _lib = ns['torch'], ns['gd'], ns['ip'], ns['runSR'], ns['config'], ns['np'], ns['_lib']
    world, dev = ns['world'], ns['dev']
    config.crop_sr = 512
    _synth(ns, 'a4', runSR, 'a4')
    opt = runSR.getOpt({'op': 'SR', 'model': 'a', 'scale': 4, 'ensemble': 0})
    shape = (3, 4320, 7680)
    frame = torch.from_numpy(gd.natural_image(51, shape)).to(dev).half()

    def step():
        if world == 1:
            return ip.doCrop(opt, frame)
        return _run_frames(ns, opt, [frame], args, bands=True)       # ONE frame: its 144 tiles over the ranks (strong scaling); the canvas stays sharded in bands (dist.py)
    step()
    steps = args.steps if args.steps_given else 3
    model = opt.modelCached
    model.set_profile('convt_R1.up1,u.up1,arsb')
    dt = _timed(ns, step, steps)
    profs = model.get_profile(all_keys=True)
    model.set_profile(None)
    ms = dt / steps * 1e3
    in_mp = shape[1] * shape[2] / 1e6
    res, peak = _base(ns, args, 'megapixels/sec (input), one 8K frame -> 32K, 4x SR (a4), 512-px tiles', in_mp / (ms / 1e3), ms, steps, 'strong', 'fp16',
                      'BASELINE configs[4]: 7680x4320 RGB -> 30720x17280 (fp16 canvas of 3.19 GB), model a4 (synthetic weights), crop 512 -> 144 tiles'
                      + (', tiles dealt round-robin over {} GPUs, canvas stitched in row bands'.format(world) if world > 1 else ', one GPU'),
                      {'tiles_per_frame': 144, 'tflops_algorithmic': round(in_mp * 1e6 * 3 * 3.9456e6 / (ms / 1e3) / 1e12, 2)})
    ks = []
    px = 3.0 * shape[1] * shape[2] / world
    for (key, label, fl), p in zip((('convt_R1.up1', 'R-branch 3x3 64->256 @2x res + fused tail, split activations (conv3x3_ps4<2>)', 4 * 2 * 256 * 64 * 9),
                                    ('u.up1', 'U-branch 3x3 64->256 @2x res + fused tail (conv3x3_ps4<1>)', 4 * 2 * 256 * 64 * 9),
                                    ('arsb', 'arsb32c_kernel (5 of 6 ARSBs)', 5 * 2 * 2 * 64 * 64 * 9)), profs):
        if p['launches']:
            ks.append(_mfma_obj(label, key, p, px * steps * fl, peak, steps))
    ks.sort(key=lambda k: -k['ms_per_step'])
    if ks:
        res['roofline'] = ks[0]
        res['roofline_kernels'] = ks
    ok = True
    if ns['rank'] == 0:
        plan = ip._plan_for(opt, frame.shape)
        canvas = torch.empty((3, plan.outH, plan.outW), dtype=torch.float16, device=dev)
        pool = _pool_of(ns, opt, plan, frame, stitch_to=canvas)
        stream = torch.cuda.current_stream().cuda_stream
        L = _lib.lib()
        sms = _events_ms(torch, lambda: _lib.check(L.moe_stitch(plan._h, 0, pool.data_ptr(), None, 3, canvas.data_ptr(), _lib.F16, stream)), 5)
        res['roofline_hbm'] = _hbm_obj('stitch8r_kernel<4>', "doCrop's blend fold of 144 fp32 tiles of 2048^2 into the 3.19-GB fp16 canvas: 4 B in + 2 B out per output pixel and plane",
                                       3.0 * plan.outH * plan.outW * 6, sms, {'per': 'frame (single GPU: the whole canvas)'})
        if not args.no_cpu_baseline:
            c, px_s, w = _tile_parity(ns, 'net4x', gd.synth_state_dict('a4', ns['load_sd']), frame, plan, pool, [143])      # the ragged corner: 232 x 192 px
            tpx = sum((t[1] - t[0]) * (t[3] - t[2]) for t in plan.tiles)
            import bench
            ncores, _ = bench._one_socket_cores()
            res['cpu_baseline'] = {'value': round(in_mp / (c / px_s * tpx), 5), 'unit': 'MP/s', 'cores': ncores, 'kind': 'port',
                                   'sample': 'tile 143 (the ragged corner, {} px) through the fp32 oracle, {:.1f} s, extrapolated by tile pixels'.format(px_s, c)}
            res['config']['parity_max_abs_vs_oracle'] = float('{:.3e}'.format(w))
            res['config']['parity_tolerance'] = 1e-3
            ok = w <= 1e-3
            res['config']['parity_ok'] = bool(ok)
    _finish(ns, res, ok)
