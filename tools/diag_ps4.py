#!/usr/bin/env python
"""conv3x3_ps4 (option up_impl = ps4) against round 3's form (up_impl = rw: conv3x3_rw per phase + tapsum4) and the oracle: per shape the largest
difference and where it sits (row / column inside the 4-row blocks and 32-pixel columns of the fused layer), then frame timings of both forms."""
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
from oracle import nets as onets  # noqa: E402


def module_for(key):
    ctor = {'net2x': models.Net2x, 'net4x': models.Net4x}[gd.MODELS[key][0]]
    m = ctor()
    m.load_state_dict({n: torch.from_numpy(v) for n, v in gd.state_dict_for(key, load_state_dict_file).items()})
    m.eval()
    return m.to(dtype=torch.float32, device='cuda:0')


def where(d, sc):
    i = np.unravel_index(np.argmax(d), d.shape)
    return 'at plane %d Y %d X %d (conv row %d %% 4 = %d, conv col %d %% 32 = %d)' % (i[0], i[-2], i[-1], i[-2] // 2, (i[-2] // 2) % 4, i[-1] // 2, (i[-1] // 2) % 32)


def main():
    bad = 0
    for key in ('a2', 'a4'):
        arch = gd.MODELS[key][0]
        sd = gd.state_dict_for(key, load_state_dict_file)
        m = module_for(key)
        shapes = ((3, 8, 8), (3, 24, 40), (2, 40, 264), (3, 16, 72), (1, 88, 64), (3, 64, 64)) if key == 'a4' else ((3, 24, 40), (2, 40, 264), (3, 8, 36), (1, 88, 64), (3, 128, 96), (4, 256, 256))
        for shape in shapes:
            for kind in ('natural', 'noise'):
                x = (gd.natural_image(31, shape) if kind == 'natural' else gd.noise_image(31, shape))[:, None]
                xd = torch.from_numpy(x).cuda()
                y_rw = m.set_option('up_impl', 'rw')(xd)[-1].cpu().numpy()
                y_ps = m.set_option('up_impl', 'ps4')(xd)[-1].cpu().numpy()
                y_ps2 = m(xd)[-1].cpu().numpy()
                y_g = m.set_option('max_groups', 7)(xd)[-1].cpu().numpy()
                m.set_option('max_groups', 0)
                d = np.abs(y_ps - y_rw)
                line = '%s %-7s %-14s ps4 vs rw %.3e' % (key, kind, shape, d.max())
                if d.max() > 2e-5:
                    line += ' ' + where(d[:, 0], 0)
                    bad += 1
                if shape[1] * shape[2] <= 128 * 128:
                    want = onets.forward(arch, sd, x).numpy()
                    line += ' | vs oracle: ps4 %.3e rw %.3e' % (np.abs(y_ps - want).max(), np.abs(y_rw - want).max())
                line += ' | repeat %s, 7 workgroups %s' % ('same bits' if np.array_equal(y_ps, y_ps2) else 'DIFFERS %.3e' % np.abs(y_ps - y_ps2).max(),
                                                             'same bits' if np.array_equal(y_ps, y_g) else 'DIFFERS %.3e' % np.abs(y_ps - y_g).max())
                bad += (not np.array_equal(y_ps, y_ps2)) + (not np.array_equal(y_ps, y_g)) + (not np.isfinite(y_ps).all())
                print(line, flush=True)
    # timing: 16 tiles of 256 x 256 x 3 planes, a4
    m = module_for('a4')
    x = torch.from_numpy(gd.noise_image(5, (48, 1, 256, 256))).cuda()
    for impl in ('rw', 'ps4', 'rw', 'ps4'):
        m.set_option('up_impl', impl)
        for _ in range(2):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        print('a4 48 planes of 256x256, up_impl = %-3s: %.3f ms per launch set' % (impl, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
    print('diag_ps4: %d problem(s)' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
