#!/usr/bin/env python
"""conv64_s (option s64 = 1: SEDN's rblock.0 and fused block tail streamed) against the round-3 kernels (s64 = 0) and the oracle; then frame-level timing"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
import golden_defs as gd
from moephoto_amd import models
from moephoto_amd.weights import load_state_dict_file
from oracle import nets as onets


def main():
    bad = 0
    key = 'l25'
    sd = gd.state_dict_for(key, load_state_dict_file)
    m = models.SEDN()
    m.load_state_dict({n: torch.from_numpy(v) for n, v in sd.items()})
    m.eval()
    m = m.to(dtype=torch.float32, device='cuda:0')
    for shape in ((3, 8, 8), (2, 24, 40), (2, 40, 264), (3, 16, 72), (1, 88, 64), (1, 6, 33), (2, 2, 40), (1, 64, 30)):
        for kind in ('natural', 'noise'):
            x = (gd.natural_image(31, shape) if kind == 'natural' else gd.noise_image(31, shape))[:, None]
            xd = torch.from_numpy(x).cuda()
            y0 = m.set_option('s64', 0)(xd)[-1].cpu().numpy()
            y1 = m.set_option('s64', 1)(xd)[-1].cpu().numpy()
            y1b = m(xd)[-1].cpu().numpy()
            y_g = m.set_option('max_groups', 5)(xd)[-1].cpu().numpy()
            m.set_option('max_groups', 0)
            want = onets.forward('sedn', sd, x).numpy()
            d = np.abs(y1 - y0)
            line = '%s %-7s %-12s s64 vs old %.3e | vs oracle: new %.3e old %.3e | repeat %s, 5 workgroups %s' % (
                key, kind, shape, d.max(), np.abs(y1 - want).max(), np.abs(y0 - want).max(),
                'same bits' if np.array_equal(y1, y1b) else 'DIFFERS %.3e' % np.abs(y1 - y1b).max(), 'same bits' if np.array_equal(y1, y_g) else 'DIFFERS %.3e' % np.abs(y1 - y_g).max())
            bad += (not np.abs(y1 - want).max() <= 1e-3) + (not np.array_equal(y1, y1b)) + (not np.array_equal(y1, y_g))
            print(line, flush=True)
    x = torch.from_numpy(gd.noise_image(5, (30, 1, 256, 256))).cuda()
    mh = m.to(dtype=torch.float16, device='cuda:0')
    xh = x.half()
    for impl in (0, 1, 0, 1):
        mh.set_option('s64', impl)
        for _ in range(2):
            mh(xh)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            mh(xh)
        torch.cuda.synchronize()
        print('l25 30 planes of 256x256, s64 = %d: %.3f ms per launch set' % (impl, (time.perf_counter() - t0) / 4 * 1e3), flush=True)
    print('diag_s64: %d problem(s)' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
