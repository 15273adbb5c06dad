#!/usr/bin/env python
"""arsb_s (option arsb_impl = s) against arsb32c (v3): bit-equality of the net's output on ragged shapes, with few workgroups, 48-channel nets, fp16 and mixed;
then the frame timing of both."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import golden_defs as gd  # noqa: E402
from moephoto_amd import models  # noqa: E402
from moephoto_amd.weights import load_state_dict_file  # noqa: E402


def module_for(key, prec):
    ctor = {'net2x': models.Net2x, 'net4x': models.Net4x, 'netdn': models.NetDN}[gd.MODELS[key][0]]
    m = ctor()
    m.load_state_dict({n: torch.from_numpy(v) for n, v in gd.state_dict_for(key, load_state_dict_file).items()})
    m.eval()
    m.precision = prec
    return m.to(dtype=torch.float32, device='cuda:0')


bad = 0
for key, prec in (('a2', 'auto'), ('a2', 'fp16'), ('dn_lite5', 'auto'), ('dn_lite5', 'fp16'), ('a4', 'auto')):
    m = module_for(key, prec)
    for shape in ((3, 8, 16), (3, 24, 40), (2, 40, 264), (3, 16, 35), (5, 88, 64), (2, 128, 61), (1, 256, 256)):
        x = gd.noise_image(17, shape)[:, None]
        xd = torch.from_numpy(x).cuda()
        y3 = m.set_option('arsb_impl', 'v3')(xd)[-1]
        ys = m.set_option('arsb_impl', 's')(xd)[-1]
        ys2 = m(xd)[-1]
        yg = m.set_option('max_groups', 5)(xd)[-1]
        m.set_option('max_groups', 0)
        ok = torch.equal(y3, ys) and torch.equal(ys, ys2) and torch.equal(ys, yg)
        bad += not ok
        d = (y3 - ys).abs()
        i = np.unravel_index(int(d.argmax()), d.shape)
        print('%-9s %-5s %-14s s vs v3 %.3e %s | repeat %s | 5 groups %s' % (key, prec, shape, float(d.max()), ('at ' + str(tuple(int(v) for v in i))) if float(d.max()) else '',
              torch.equal(ys, ys2), torch.equal(ys, yg)), flush=True)
m = module_for('a4', 'auto')
x = torch.from_numpy(gd.noise_image(5, (48, 1, 256, 256))).cuda()
for impl in ('v3', 's', 'v3', 's'):
    m.set_option('arsb_impl', impl)
    for _ in range(2):
        m(x)
    m.set_profile('arsb')
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    p = m.get_profile()
    m.set_profile(None)
    print('a4 48 planes of 256x256, arsb_impl = %-2s: %.3f ms per launch set, the five one-launch ARSBs %.3f ms' % (impl, dt, p['total_ms'] / 5), flush=True)
print('diag_arsb_s: %d problem(s)' % bad)
sys.exit(1 if bad else 0)
