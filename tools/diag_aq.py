#!/usr/bin/env python
"""arsb_sq (option exact_fuse = 1: an exact ARSB of the chain in one launch, conv_1's rows in LDS) against the two-launch form on conv64_sq (exact_fuse = 0)
and the oracle: per shape the largest difference between the two forms and where it sits, each form's error against the oracle, repeatability and independence
of the workgroup count; then launch-set timings of both forms."""
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
    ctor = {'net2x': models.Net2x, 'net4x': models.Net4x, 'net3x': models.Net3x}[gd.MODELS[key][0]]
    m = ctor()
    m.load_state_dict({n: torch.from_numpy(v) for n, v in gd.state_dict_for(key, load_state_dict_file).items()})
    m.eval()
    return m.to(dtype=torch.float32, device='cuda:0')


def main():
    bad = 0
    for key in ('a2', 'a4'):
        arch = gd.MODELS[key][0]
        sd = gd.state_dict_for(key, load_state_dict_file)
        m = module_for(key)
        for shape in ((3, 8, 8), (2, 24, 40), (2, 40, 264), (3, 16, 72), (1, 88, 64), (3, 64, 64), (1, 6, 33)):
            for kind in ('natural', 'noise'):
                x = (gd.natural_image(31, shape) if kind == 'natural' else gd.noise_image(31, shape))[:, None]
                xd = torch.from_numpy(x).cuda()
                y_p = m.set_option('exact_fuse', 0)(xd)[-1].cpu().numpy()
                y_s = m.set_option('exact_fuse', 1)(xd)[-1].cpu().numpy()
                y_s2 = m(xd)[-1].cpu().numpy()
                y_g = m.set_option('max_groups', 7)(xd)[-1].cpu().numpy()
                m.set_option('max_groups', 0)
                d = np.abs(y_s - y_p)
                line = '%s %-7s %-12s s vs p %.3e' % (key, kind, shape, d.max())
                if not (d.max() <= 6e-4):      # (the fp8 low words between the layers turn fp32-rounding differences of a sum into differences of a few 1e-5 .. 1e-4 at the output)
                    i = np.unravel_index(np.argmax(d), d.shape)
                    line += ' at plane %d Y %d X %d' % (i[0], i[-2], i[-1])
                    bad += 1
                want = onets.forward(arch, sd, x).numpy()
                line += ' | vs oracle: s %.3e p %.3e' % (np.abs(y_s - want).max(), np.abs(y_p - want).max())
                bad += not (np.abs(y_s - want).max() <= 1e-3)
                line += ' | repeat %s, 7 workgroups %s' % ('same bits' if np.array_equal(y_s, y_s2) else 'DIFFERS %.3e' % np.abs(y_s - y_s2).max(),
                                                             'same bits' if np.array_equal(y_s, y_g) else 'DIFFERS %.3e' % np.abs(y_s - y_g).max())
                bad += (not np.array_equal(y_s, y_s2)) + (not np.array_equal(y_s, y_g)) + (not np.isfinite(y_s).all())
                print(line, flush=True)
    for key in ('a2', 'a4'):
        m = module_for(key)
        x = torch.from_numpy(gd.noise_image(5, (48, 1, 256, 256))).cuda()
        for impl in (0, 1, 0, 1):
            m.set_option('exact_fuse', impl)
            for _ in range(2):
                m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                m(x)
            torch.cuda.synchronize()
            print('%s 48 planes of 256x256, exact_fuse = %s: %.3f ms per launch set' % (key, impl, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
    print('diag_aq: %d problem(s)' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
