#!/usr/bin/env python
"""Diagnosis of conv64_q8 (option x3_impl = q8): the 'input2' tap (first split-operand conv) against the oracle's, beside conv64_x3's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
import golden_defs as gd
from moephoto_amd.weights import load_state_dict_file
from moephoto_amd import models
from oracle import nets as onets
key = sys.argv[1] if len(sys.argv) > 1 else 'a2'
arch = gd.MODELS[key][0]
sd = gd.state_dict_for(key, load_state_dict_file)
x = gd.natural_image(3, (2, 24, 40))[:, None]
taps = {}
onets.forward(arch, sd, x, 'torch', taps)
ctor = {'net2x': models.Net2x, 'net4x': models.Net4x, 'netdn': models.NetDN}[arch]
m = ctor(); m.load_state_dict({n: torch.from_numpy(v) for n, v in sd.items()}); m.eval(); m = m.to(device='cuda:0')
for impl in ('x3', 'q8'):
    m.set_option('x3_impl', impl).set_debug(True)
    m(torch.from_numpy(x).cuda()); torch.cuda.synchronize()
    for name in ('stem', 'input2', 'arsb1', 'arsb2'):
        if name in taps:
            got, want = m.debug_tap(name), taps[name].numpy()
            d = np.abs(got - want)
            print(impl, name, 'max err %.3e' % d.max(), 'mean err %.3e' % d.mean(), 'swing %.2f' % np.abs(want).max(), 'worst channel', int(np.argmax(d.reshape(d.shape[0], d.shape[1], -1).max(axis=(0, 2)))))
    m.set_debug(False)

# ---- fingerprint: which combination of products does the device's 'input2' tap match?
import torch.nn.functional as F
def q8(t, sh):
    return (torch.clamp(t * float(2.0 ** sh), -448, 448).to(torch.float8_e4m3fn).float()) * float(2.0 ** -sh)
a = taps['stem'].float()
w = torch.from_numpy(np.asarray(sd['conv_input2.weight'], dtype=np.float32))
ah, wh = a.half().float(), w.half().float()
al, wl = ((a - ah) * 2048).half().float(), ((w - wh) * 2048).half().float()
main = F.conv2d(ah, wh, padding=1)
c1 = F.conv2d(q8(ah, -2), q8(wl, 8), padding=1) / 2048
c2 = F.conv2d(q8(al, -2), q8(wh, 8), padding=1) / 2048
e1 = F.conv2d(ah, wl, padding=1) / 2048
e2 = F.conv2d(al, wh, padding=1) / 2048
m.set_option('x3_impl', 'q8').set_debug(True)
m(torch.from_numpy(x).cuda()); torch.cuda.synchronize()
got = torch.from_numpy(m.debug_tap('input2'))
for name, cand in (('main', main), ('main+c1', main + c1), ('main+c2', main + c2), ('main+c1+c2', main + c1 + c2), ('main+e1+e2 (exact)', main + e1 + e2), ('main-c1-c2', main - c1 - c2),
                   ('main+2c1+2c2', main + 2 * c1 + 2 * c2), ('main+c1/2+c2/2', main + c1 / 2 + c2 / 2), ('main+4c1+4c2', main + 4 * (c1 + c2)), ('main+(c1+c2)/4', main + (c1 + c2) / 4)):
    print('%-22s max |device - candidate| = %.3e' % (name, float((got - cand).abs().max())))
print('sizes: |c1| %.3e |c2| %.3e' % (float(c1.abs().max()), float(c2.abs().max())))
m8 = F.conv2d(q8(ah, -2), q8(wh, 8), padding=1) / 2048        # w_hi8 x a_hi8 (pass 2 on the wrong image)
l8 = F.conv2d(q8(al, -2), q8(wl, 8), padding=1) / 2048        # w_lo8 x a_lo8
for name, cand in (('main+c1+m8', main + c1 + m8), ('main+m8', main + m8), ('main+2*m8', main + 2 * m8), ('main+l8+c2', main + l8 + c2), ('main+c1+l8', main + c1 + l8), ('main+m8+l8', main + m8 + l8)):
    print('%-22s max |device - candidate| = %.3e' % (name, float((got - cand).abs().max())))
d = (got - main)
print('device - main: max %.3e; corr with c1 %.3f, c2 %.3f, m8 %.3f, l8 %.3f' % (float(d.abs().max()), *[float((d * t).sum() / (d.norm() * t.norm())) for t in (c1, c2, m8, l8)]))
# per output channel correlation with c1 + c2
cc = c1 + c2
print('per-channel corr(device - main, c1 + c2):', [round(float((d[:, k] * cc[:, k]).sum() / (d[:, k].norm() * cc[:, k].norm() + 1e-30)), 2) for k in range(0, 64, 4)])
cc = c1 + c2
def corr(u, v): return float((u * v).sum() / (u.norm() * v.norm() + 1e-30))
for dyy in (-2, -1, 0, 1, 2):
    for dxx in (-1, 0, 1):
        sh = torch.roll(cc, shifts=(dyy, dxx), dims=(2, 3))
        print('shift rows %+d cols %+d: corr %.3f' % (dyy, dxx, corr(d[:, :, 3:-3, 3:-3], sh[:, :, 3:-3, 3:-3])), end=' | ')
    print()
# by output row inside the 8-row patch
print('corr by image row:', [round(corr(d[:, :, r], cc[:, :, r]), 2) for r in range(d.shape[2])])
print('corr by 8-channel group:', [round(corr(d[:, k:k + 8], cc[:, k:k + 8]), 2) for k in range(0, 64, 8)])
