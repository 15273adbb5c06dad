#!/usr/bin/env python
"""GPU-box diagnostic sweep (one gpurun call = as much information as possible):
  1. device facts (name, CUs, clocks),
  2. every net family x precision mode x input kind on a small tile, layer-by-layer max-abs error vs the oracle,
  3. per-layer-class kernel timing of Net4x on full-size tiles (hipEvents on the launch stream),
  4. end-to-end 1080p frame timing at several batch sizes.
Writes gpurun_out/diag.txt (human) and gpurun_out/diag.json."""
import json
import os
import subprocess
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
OUT = os.path.join(ROOT, 'gpurun_out')
os.makedirs(OUT, exist_ok=True)
LOG = open(os.path.join(OUT, 'diag.txt'), 'w')
RES = {}


def say(*a):
    s = ' '.join(str(x) for x in a)
    print(s, flush=True)
    LOG.write(s + '\n')
    LOG.flush()


def section(name, fn):
    say('\n=== ' + name + ' ===')
    try:
        fn()
    except Exception:
        say('FAILED:', traceback.format_exc())
        RES.setdefault('failures', []).append(name)


import numpy as np  # noqa: E402
import torch  # noqa: E402
import golden_defs as gd  # noqa: E402
from moephoto_amd import _lib, models  # noqa: E402
from moephoto_amd.weights import load_state_dict_file  # noqa: E402
from oracle import nets as onets  # noqa: E402

dev = torch.device('cuda:0')
CTOR = {'net2x': models.Net2x, 'net3x': models.Net3x, 'net4x': models.Net4x, 'netdn': models.NetDN, 'sedn': models.SEDN,
        'lite2': lambda: models.Net(2), 'lite4': lambda: models.Net(4), 'lite8': lambda: models.Net(8)}


def make(key, precision):
    m = CTOR[gd.MODELS[key][0]]()
    m.load_state_dict({n: torch.from_numpy(v) for n, v in gd.state_dict_for(key, load_state_dict_file).items()})
    m.precision = precision
    return m.to(dtype=torch.float32, device=dev)


def facts():
    p = torch.cuda.get_device_properties(0)
    say('device', p.name, 'CUs', p.multi_processor_count, 'mem GB', round(p.total_memory / 2 ** 30, 1), 'clock MHz', getattr(p, 'clock_rate', 0) / 1e3)
    RES['device'] = dict(name=p.name, cus=p.multi_processor_count)
    try:
        o = subprocess.run('rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock Freq" | sort | uniq -c | head -12', shell=True, capture_output=True, text=True, timeout=60)
        say(o.stdout)
    except Exception as e:
        say('rocminfo failed', e)
    say('host cores', os.cpu_count(), 'torch threads', torch.get_num_threads())


def nets():
    RES['nets'] = {}
    for key in os.environ.get('DIAG_KEYS', 'a2,a4,a3,dn_lite5,l25,lite2,lite4').split(','):
        arch = gd.MODELS[key][0]
        sd = gd.state_dict_for(key, load_state_dict_file)
        for prec in os.environ.get('DIAG_PREC', 'auto,fp16,fp16x3').split(','):
            try:
                m = make(key, prec).set_debug(True)
                for kind in ('natural', 'noise'):
                    x = gd.natural_image(5, (3, 40, 48))[:, None] if kind == 'natural' else gd.noise_image(5, (3, 1, 40, 48))
                    taps = {}
                    want = onets.forward(arch, sd, x, 'torch', taps).numpy()
                    got = m(torch.from_numpy(x).to(dev))[-1].cpu().numpy()
                    torch.cuda.synchronize()
                    err = float(np.abs(got - want).max())
                    terr = {}
                    for name, w in taps.items():
                        try:
                            terr[name] = float(np.abs(m.debug_tap(name) - w.numpy()).max())
                        except Exception as e:
                            terr[name] = str(e)[:60]
                    RES['nets']['{}/{}/{}'.format(key, prec, kind)] = dict(out=err, taps=terr)
                    say('{:9s} {:12s} {:8s} out {:.3e}  taps {}'.format(key, prec, kind, err,
                        ' '.join('{}={:.1e}'.format(k, v) if isinstance(v, float) else '{}=?'.format(k) for k, v in terr.items())))
                del m
            except Exception:
                say(key, prec, 'FAILED', traceback.format_exc()[-1500:])
                RES.setdefault('failures', []).append('{}/{}'.format(key, prec))


def layer_timing():
    m = make('a4', os.environ.get('DIAG_LAYER_PREC', 'auto'))
    say('precision', m.resolved_precision())
    RES['layers'] = {}
    for B in (3, 12):
        x = torch.from_numpy(gd.natural_image(1, (B, 256, 256))[:, None]).to(dev).half()
        for _ in range(2):
            m(x)
        torch.cuda.synchronize()
        for sub in ('input2', 'c1_', 'c2_', 'up0', 'up1'):
            m.set_profile(sub)
            for _ in range(3):
                m(x)
            pr = m.get_profile()
            tf = pr['flops'] / (pr['total_ms'] / 1e3) / 1e12 if pr['total_ms'] > 0 else 0
            RES['layers']['B{}/{}'.format(B, sub)] = dict(avg_ms=pr['total_ms'] / max(1, pr['launches']), tflops=tf)
            say('B={:2d} layers *{:7s}*: {:3d} launches avg {:.4f} ms  {:.1f} TFLOP/s ({:.1f} % of 2500)'.format(
                B, sub, pr['launches'], pr['total_ms'] / max(1, pr['launches']), tf, tf / 25))
        m.set_profile(None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            m(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        tf = B * 65536 * 3.9456e6 / (ms / 1e3) / 1e12
        RES['layers']['B{}/forward'.format(B)] = dict(ms=ms, tflops=tf)
        say('B={:2d} whole Net4x forward on 256x256 tiles: {:.3f} ms  {:.1f} TFLOP/s algorithmic'.format(B, ms, tf))


def frame_timing():
    from moephoto_amd import imageProcess as ip, runSR
    from moephoto_amd.config import config
    from moephoto_amd.weights import save_state_dict_file
    config.deviceId, config.fp16, config.crop_sr = 0, True, 256
    path = '/tmp/moe_diag_a4.pth'
    save_state_dict_file(gd.synth_state_dict('a4', load_state_dict_file), path)
    runSR.mode_switch['a4'] = (path, runSR.mode_switch['a4'][1])
    opt = runSR.getOpt({'op': 'SR', 'model': 'a', 'scale': 4, 'ensemble': 0})
    x = torch.from_numpy(gd.natural_image(1000, (3, 1080, 1920))).to(dev).half()
    RES['frame'] = {}
    for per in (1, 2, 4, 8):
        config.tilesPerBatch = per
        for _ in range(2):
            ip.doCrop(opt, x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            ip.doCrop(opt, x)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 4 * 1e3
        RES['frame'][per] = ms
        say('1080p frame, {} tiles/batch: {:.2f} ms/frame  {:.2f} input MP/s  {:.1f} TFLOP/s'.format(per, ms, 2.0736 / ms * 1e3, 24.545 / ms * 1e3))


if __name__ == '__main__':
    which = sys.argv[1:] or ['facts', 'nets', 'layers', 'frame']
    if 'facts' in which:
        section('device facts', facts)
    if 'nets' in which:
        section('nets x precision x input, layer by layer', nets)
    if 'layers' in which:
        section('Net4x per-layer-class timing', layer_timing)
    if 'frame' in which:
        section('1080p frame timing', frame_timing)
    json.dump(RES, open(os.path.join(OUT, 'diag.json'), 'w'), indent=1)
    say('\nfailures:', RES.get('failures'))
