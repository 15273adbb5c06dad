#!/usr/bin/env python
"""GPU diagnostic of the fused ARSB kernel (arsb_fused.hip): trunk taps against the oracle with the error broken down by channel,
row-in-patch and column-in-patch (patches are 8 x 30 outputs), fused vs two-launch outputs, and timings of both forms."""
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

dev = torch.device('cuda:0')


def make(key, prec, nb=None):
    ctor = {'net2x': models.Net2x, 'net4x': models.Net4x, 'netdn': models.NetDN}[gd.MODELS[key][0]]
    m = ctor()
    m.load_state_dict({n: torch.from_numpy(v) for n, v in gd.state_dict_for(key, load_state_dict_file).items()})
    m.precision = prec
    m = m.to(dtype=torch.float32, device=dev)
    if nb is not None:
        m.set_exact_blocks(nb)
    return m


def breakdown(err, name):
    # err: (B, C, H, W)
    B, C, H, W = err.shape
    print('   {}: max {:.3e} at {}'.format(name, err.max(), np.unravel_index(err.argmax(), err.shape)))
    print('     per 16-channel group:', ['%.1e' % err[:, c:c + 16].max() for c in range(0, C, 16)])
    print('     per channel%16 quad :', ['%.1e' % max(err[:, c + 4 * qq:c + 4 * qq + 4].max() for c in range(0, C, 16)) for qq in range(4)])
    print('     per row % 8        :', ['%.1e' % err[:, :, r::8].max() for r in range(min(8, H))])
    print('     per col % 30       :', ['%.0e' % err[:, :, :, c::30].max() for c in range(min(30, W))])


for key, shape in (('a2', (3, 24, 40)), ('a2', (2, 40, 72)), ('dn_lite5', (3, 16, 64)), ('a2', (3, 9, 35))):
    arch = gd.MODELS[key][0]
    sd = gd.state_dict_for(key, load_state_dict_file)
    x = gd.natural_image(3, shape)[:, None]
    taps = {}
    want = onets.forward(arch, sd, x, 'torch', taps).numpy()
    for prec, nb in (('fp16', None), ('mixed', 0)):
        res = {}
        for fuse in ('0', '1'):
            os.environ['MOE_ARSB_FUSE'] = fuse
            m = make(key, prec, nb).set_debug(True)
            y = m(torch.from_numpy(x).to(dev))[-1].cpu().numpy()
            torch.cuda.synchronize()
            res[fuse] = (y, {k: m.debug_tap(k) for k in taps if k.startswith('arsb') or k == 'input2'})
            del m
        d = np.abs(res['1'][0] - res['0'][0]).max()
        e0, e1 = np.abs(res['0'][0] - want).max(), np.abs(res['1'][0] - want).max()
        print('{} {} {} nb={}: out err two-launch {:.3e} fused {:.3e}  |fused - two-launch| {:.3e}'.format(key, shape, prec, nb, e0, e1, d))
        for k in ('arsb1', 'arsb2', 'arsb6'):
            t0, t1, w = res['0'][1][k], res['1'][1][k], taps[k].numpy()
            print('   tap {}: two-launch {:.3e} fused {:.3e} (swing {:.2f})'.format(k, np.abs(t0 - w).max(), np.abs(t1 - w).max(), np.abs(w).max()))
            if np.abs(t1 - w).max() > 4 * max(np.abs(t0 - w).max(), 1e-4):
                breakdown(np.abs(t1 - w), k + ' fused')
                break

# ---- timings: B = 12 planes of 256 x 256 (one 4-tile batch of the benchmark) ---------------------------------------------
x = torch.from_numpy(gd.natural_image(1, (12, 256, 256))[:, None]).to(dev).half()
for prec, nb in (('fp16', None), ('mixed', 0), ('mixed', 1)):
    for fuse in ('0', '1'):
        os.environ['MOE_ARSB_FUSE'] = fuse
        m = make('a4', prec, nb)
        for _ in range(2):
            m(x)
        m.set_profile('arsb,c1_,c2_')
        for _ in range(3):
            m(x)
        pr = m.get_profile(all_keys=True)
        m.set_profile(None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            m(x)
        e1.record()
        torch.cuda.synchronize()
        parts = ['{} {} launches avg {:.4f} ms {:.0f} TF'.format(k, p['launches'], p['total_ms'] / max(1, p['launches']), p['flops'] / max(1e-9, p['total_ms']) / 1e9)
                 for k, p in zip(('arsb', 'c1_', 'c2_'), pr)]
        print('a4 B=12 256x256 {} nb={} fuse={}: forward {:.3f} ms | {}'.format(prec, nb, fuse, e0.elapsed_time(e1) / 5, ' | '.join(parts)))
        del m

# ---- split-operand layers: one launch (conv64_x3.hip) vs three ----------------------------------------------------------------
for fuse in ('0', '1'):
    os.environ['MOE_X3_FUSE'] = fuse
    m = make('a4', 'mixed', 1)
    for _ in range(2):
        m(x)
    m.set_profile('input2,c1_1,c2_1')
    for _ in range(3):
        m(x)
    pr = m.get_profile(all_keys=True)
    m.set_profile(None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        m(x)
    e1.record()
    torch.cuda.synchronize()
    parts = ['{} {} launches avg {:.4f} ms'.format(k, p['launches'], p['total_ms'] / max(1, p['launches'])) for k, p in zip(('input2', 'c1_1', 'c2_1'), pr)]
    print('a4 B=12 256x256 mixed nb=1 x3-fuse={}: forward {:.3f} ms | {}'.format(fuse, e0.elapsed_time(e1) / 5, ' | '.join(parts)))
    del m

# ---- fused ARSB, two forms ------------------------------------------------------------------------------------------------------
os.environ['MOE_ARSB_FUSE'] = '1'; os.environ['MOE_X3_FUSE'] = '1'
for impl in ('v1', 'pc'):
    os.environ['MOE_ARSB_IMPL'] = impl
    for prec, nb in (('fp16', None), ('mixed', 1)):
        m = make('a4', prec, nb)
        for _ in range(2):
            m(x)
        m.set_profile('arsb')
        for _ in range(3):
            m(x)
        pr = m.get_profile()
        m.set_profile(None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            m(x)
        e1.record()
        torch.cuda.synchronize()
        print('a4 B=12 256x256 {} nb={} arsb impl {}: forward {:.3f} ms | arsb {} launches avg {:.4f} ms {:.0f} TF'.format(
            prec, nb, impl, e0.elapsed_time(e1) / 5, pr['launches'], pr['total_ms'] / max(1, pr['launches']), pr['flops'] / max(1e-9, pr['total_ms']) / 1e9))
        del m
os.environ.pop('MOE_ARSB_IMPL')
