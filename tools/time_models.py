#!/usr/bin/env python
"""Frame timing of the other model families / precisions (1080p input, 256-px tiles): MP/s per model key and precision."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import golden_defs as gd  # noqa: E402
from moephoto_amd import imageProcess as ip, runDN, runSR  # noqa: E402
from moephoto_amd.config import config  # noqa: E402
from moephoto_amd.weights import load_state_dict_file, save_state_dict_file  # noqa: E402

config.deviceId, config.fp16, config.crop_sr, config.crop_dn, config.crop_dns, config.modelRoot = 0, True, 256, 256, 256, gd.ZOO
for key, table in (('a4', runSR), ('a3', runSR)):
    path = '/tmp/moe_tm_{}.pth'.format(key)
    save_state_dict_file(gd.synth_state_dict(key, load_state_dict_file), path)
    table.mode_switch[key] = (path, table.mode_switch[key][1])
path = '/tmp/moe_tm_l25.pth'
save_state_dict_file(gd.synth_state_dict('l25', load_state_dict_file), path)
runDN.mode_switch['25'] = (path,) + tuple(runDN.mode_switch['25'][1:])
x = torch.from_numpy(gd.natural_image(1000, (3, 1080, 1920))).cuda().half()
cases = [('SR a2', lambda: runSR.getOpt({'model': 'a', 'scale': 2})), ('SR a3', lambda: runSR.getOpt({'model': 'a', 'scale': 3})),
         ('SR a4', lambda: runSR.getOpt({'model': 'a', 'scale': 4})), ('SR lite2', lambda: runSR.getOpt({'model': 'lite', 'scale': 2})),
         ('SR lite4', lambda: runSR.getOpt({'model': 'lite', 'scale': 4})), ('SR lite8', lambda: runSR.getOpt({'model': 'lite', 'scale': 8})), ('DN lite5', lambda: runDN.getOpt({'model': 'lite5'})),
         ('DN l25', lambda: runDN.getOpt({'model': '25'}))]
only = os.environ.get('TM_ONLY')          # e.g. TM_ONLY='SR a3' TM_PREC=fp16 under rocprofv3
for name, mk in cases:
    if only and name != only:
        continue
    for prec in os.environ.get('TM_PREC', 'auto,fp16,fp16x3').split(','):
        ip.modelCache.clear()
        opt = mk()
        opt.modelCached.set_precision(prec)
        f = (lambda: ip.doCrop(opt, x))
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print('{:9s} {:7s} {:8.2f} ms/frame  {:7.2f} input MP/s'.format(name, prec, ms, 2.0736 / ms * 1e3), flush=True)
