#!/usr/bin/env python
"""Randomised shape sweep on the GPU: every net family, random plane counts and tile sizes (multiples of 8, the planner's
alignment), default ('auto') precision and single-pass fp16, against the oracle.  Prints the worst error per family; exits
non-zero on a tolerance violation.  Not part of the pytest suite (a few minutes of CPU oracle time); run through gpurun."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import golden_defs as gd  # noqa: E402
from moephoto_amd import models  # noqa: E402
from moephoto_amd.weights import load_state_dict_file  # noqa: E402
from oracle import nets as onets  # noqa: E402

CTOR = {'net2x': models.Net2x, 'net3x': models.Net3x, 'net4x': models.Net4x, 'netdn': models.NetDN, 'sedn': models.SEDN,
        'lite2': lambda: models.Net(2), 'lite4': lambda: models.Net(4), 'lite8': lambda: models.Net(8)}
N = int(os.environ.get('FUZZ_N', '6'))
rng = np.random.default_rng(int(os.environ.get('FUZZ_SEED', '1')))
bad = 0
for key in os.environ.get('FUZZ_KEYS', 'a2,a3,a4,dn_lite5,l25,lite2,lite4').split(','):
    arch = gd.MODELS[key][0]
    sd = gd.state_dict_for(key, load_state_dict_file)
    for prec in ('auto', 'fp16'):
        m = CTOR[arch]()
        m.load_state_dict({n: torch.from_numpy(v) for n, v in sd.items()})
        m.precision = prec
        m = m.to(dtype=torch.float32, device='cuda:0')
        x3 = m.resolved_precision() == 'fp16x3'
        # single-pass fp16 is not the parity mode of NetDN / lite: only gross errors are flagged there
        tol = 2e-5 if x3 else (1e-3 if prec == 'auto' else (8e-3 if arch.startswith('lite') else 3e-3))     # default arithmetic: the product's 1e-3
        worst, wcase = 0.0, None
        for i in range(N):
            B = int(rng.integers(1, 8))
            h, w = 8 * int(rng.integers(1, 9)), 8 * int(rng.integers(1, 13))
            if arch in ('net4x', 'lite8', 'lite4', 'sedn'):
                h, w = min(h, 40), min(w, 56)          # keep the CPU oracle quick
            x = gd.natural_image(100 + i, (B, h, w))[:, None]
            want = onets.forward(arch, sd, x).numpy()
            got = m(torch.from_numpy(x).cuda())[-1].cpu().numpy()
            err = float(np.abs(got - want).max())
            if err > worst:
                worst, wcase = err, (B, h, w)
        flag = '' if worst <= tol else '   <-- EXCEEDS {:g}'.format(tol)
        bad += worst > tol
        print('{:9s} {:5s} ({:6s}) worst {:.3e} at B,h,w={}{}'.format(key, prec, 'x3' if x3 else 'fp16', worst, wcase, flag), flush=True)

# ---- whole doCrop (planner + tile loop + stitch) on random image sizes / tile sizes, a2 (x2) and dn_lite5 ------------------------
from moephoto_amd import imageProcess as ip, runDN, runSR  # noqa: E402
from moephoto_amd.config import config  # noqa: E402
from oracle import planner as oplanner, stitch as ostitch  # noqa: E402
config.modelRoot, config.fp16, config.deviceId = gd.ZOO, False, 0
for i in range(int(os.environ.get('FUZZ_CROPS', '5'))):
    H, W = int(rng.integers(60, 200)), int(rng.integers(60, 260))
    crop = 8 * int(rng.integers(6, 14))
    C = int(rng.integers(1, 5))
    x = gd.natural_image(300 + i, (C, H, W))
    for kind in ('SR a2', 'DN lite5'):
        ip.modelCache.clear()
        if kind == 'SR a2':
            config.crop_sr = crop
            opt = runSR.getOpt({'model': 'a', 'scale': 2, 'ensemble': 0})
            arch, sd, pad, sc, tol = 'net2x', gd.state_dict_for('a2', load_state_dict_file), 5, 2, 1e-3
        else:
            config.crop_dn = crop
            opt = runDN.getOpt({'model': 'lite5'})
            arch, sd, pad, sc, tol = 'netdn', gd.state_dict_for('dn_lite5', load_state_dict_file), 7, 1, 1e-3
        got = ip.doCrop(opt, torch.from_numpy(x).cuda()).cpu().numpy()
        pl = oplanner.prepare((C, H, W), 1 << 40, 1e-3, pad, sc, 8, crop)
        want = ostitch.do_crop(x, pl, sc, onets.model_fn(arch, sd))
        err = float(np.abs(got - want).max())
        bad += err > tol
        print('doCrop {:8s} C={} {}x{} crop {:3d} tiles {:3d}: {:.3e}{}'.format(kind, C, H, W, crop, len(pl.tiles), err, '' if err <= tol else '   <-- EXCEEDS'), flush=True)
sys.exit(1 if bad else 0)
