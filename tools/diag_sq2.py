#!/usr/bin/env python
"""error maps of conv64_sq against conv64_q8 (a2): per output row / column maxima for small shapes"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
import golden_defs as gd
from moephoto_amd import models
from moephoto_amd.weights import load_state_dict_file
np.set_printoptions(linewidth=250, precision=1, suppress=False)
m = models.Net2x()
m.load_state_dict({n: torch.from_numpy(v) for n, v in gd.state_dict_for('a2', load_state_dict_file).items()})
m.eval(); m = m.to(dtype=torch.float32, device='cuda:0')
eb = int(os.environ.get('SQ_EXACT', '-1'))
if eb >= 0:
    m.set_exact_blocks(eb)
for shape in ((1, 8, 8), (1, 16, 40), (1, 32, 72)):
    x = gd.natural_image(31, shape)[:, None]
    xd = torch.from_numpy(x).cuda()
    y_p = m.set_option('q8_impl', 'p')(xd)[-1].cpu().numpy()
    y_s = m.set_option('q8_impl', 's')(xd)[-1].cpu().numpy()
    d = np.abs(y_s - y_p)[0, 0]
    print(shape, 'max', d.max(), 'swing', np.abs(y_p).max())
    print(' rows', np.array2string(d.max(axis=1), formatter={'float_kind': lambda v: '%.0e' % v}))
    print(' cols', np.array2string(d.max(axis=0), formatter={'float_kind': lambda v: '%.0e' % v}))
