#!/usr/bin/env python
"""BASELINE config 5 at full size: 7680x4320 RGB -> 30720x17280, a4-synth, crop 512 (144 tiles).  Timing plus two tile checks
against the oracle (full CPU reference of this case is out of reach); config 3 (4K: DN l25 then SR a2) with --config3."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import golden_defs as gd  # noqa: E402
from moephoto_amd import imageProcess as ip, runDN, runSR  # noqa: E402
from moephoto_amd.config import config  # noqa: E402
from moephoto_amd.weights import load_state_dict_file, save_state_dict_file  # noqa: E402

config.deviceId, config.fp16, config.modelRoot = 0, True, gd.ZOO


def synth(key, table, tkey):
    path = '/tmp/moe_c5_{}.pth'.format(key)
    save_state_dict_file(gd.synth_state_dict(key, load_state_dict_file), path)
    table.mode_switch[tkey] = (path,) + tuple(table.mode_switch[tkey][1:])


def timed(f, n=2):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        y = f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, y


if '--config3' in sys.argv:
    config.crop_sr, config.crop_dn, config.crop_dns = 256, 256, 256
    synth('l25', runDN, '25')
    x = torch.from_numpy(gd.natural_image(3, (3, 2160, 3840))).cuda().half()
    dn, sr = runDN.getOpt({'model': '25'}), runSR.getOpt({'model': 'a', 'scale': 2})
    ms, y = timed(lambda: ip.doCrop(sr, ip.doCrop(dn, x)), 1)
    print('config 3: 3840x2160 DN l25 -> SR a2: {:.1f} ms  {:.2f} input MP/s  out {}'.format(ms, 8.2944 / ms * 1e3, tuple(y.shape)))
else:
    config.crop_sr = 512
    synth('a4', runSR, 'a4')
    x = torch.from_numpy(gd.natural_image(5, (3, 4320, 7680))).cuda().half()
    opt = runSR.getOpt({'model': 'a', 'scale': 4})
    plan = ip._plan_for(opt, x.shape)
    ms, y = timed(lambda: ip.doCrop(opt, x), 2)
    print('config 5: {} tiles, {:.1f} ms  {:.2f} input MP/s  {:.1f} TFLOP/s  out {}'.format(plan.n_tiles, ms, 33.1776 / ms * 1e3, 392.7 / ms * 1e3, tuple(y.shape)))
    print('output finite:', bool(torch.isfinite(y[:, ::64, ::64].float()).all()), 'mean', float(y[:, ::16, ::16].float().mean()), 'input mean', float(x.float().mean()))
