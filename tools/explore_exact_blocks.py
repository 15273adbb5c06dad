#!/usr/bin/env python
"""GPU exploration: worst tile error (default arithmetic vs the engine's own exact mode 'fp16x3') and frame time of Net2x / Net4x / NetDN
for every setting of `exact_blocks` (leading ARSBs with split operands) -- the data behind exact_blocks_of() in engine.cpp."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import golden_defs as gd  # noqa: E402
import test_gpu_fullsize as T  # noqa: E402
from moephoto_amd import imageProcess as ip  # noqa: E402

dev = torch.device('cuda:0')
for (kind_model, scale, blocks) in ((('a', 2), 2, (1, 2, 3, 4, 6)), (('a', 4), 4, (0, 1, 2)), (('dn', 'lite5'), 1, (0, 1, 2))):
    if kind_model[0] == 'a':
        opt = T._opt_sr('a', scale, 256)
    else:
        from moephoto_amd.config import config
        config.fp16 = False
        opt = T._opt_dn('lite5', 256)
    m = opt.modelCached
    frames = [(k, s, torch.from_numpy(T._frames(k, (100 + s) if k == 'natural' else s, (3, 1080, 1920))).to(dev).half()) for k in ('natural', 'noise_u8') for s in (0, 1, 2)]
    plan = ip._plan_for(opt, frames[0][2].shape)
    m.set_precision('fp16x3')
    wants = [T._pool_of(opt, plan, x) for _, _, x in frames]
    m.set_precision('auto')
    for nb in blocks:
        m.set_exact_blocks(nb)
        worst = {'natural': 0.0, 'noise_u8': 0.0}
        for (k, s, x), w in zip(frames, wants):
            worst[k] = max(worst[k], float((T._pool_of(opt, plan, x) - w).abs().max()))
        ms = T._time_ms(lambda: ip.doCrop(opt, frames[0][2]), reps=5)
        print('%s%s exact_blocks=%d  natural %.3e  noise_u8 %.3e   %.2f ms/frame' % (kind_model[0], kind_model[1], nb, worst['natural'], worst['noise_u8'], ms), flush=True)
    m.set_exact_blocks(-1)
