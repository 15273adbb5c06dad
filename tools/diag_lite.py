#!/usr/bin/env python
"""Layer-by-layer max-abs error of a lite net against the oracle (debug taps), for the 1x1 kernel A/B (MOE_CONV1X1=0|1)."""
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

key = sys.argv[1] if len(sys.argv) > 1 else 'lite4'
scale = int(key[4:])
sd = gd.state_dict_for(key, load_state_dict_file)
for prec in ('fp16', 'fp16x3'):
    for shape in ((2, 24, 40), (1, 16, 64)):
        x = gd.natural_image(3, shape)[:, None]
        taps = {}
        want = onets.forward(key, sd, x, taps=taps).numpy()
        m = models.Net(scale)
        m.load_state_dict({n: torch.from_numpy(v) for n, v in sd.items()})
        m.precision = prec
        m = m.to(dtype=torch.float32, device='cuda:0')
        m.set_debug(True)
        y = m(torch.from_numpy(x).cuda())[-1].cpu().numpy()
        line = '{} {} {}: out {:.2e}'.format(key, prec, shape, float(np.abs(y - want).max()))
        for name in ('stem', 'input2', 'lb3', 'r.up0', 'u.up0', 'r.up1', 'u.up1'):
            if name in taps:
                try:
                    got = m.debug_tap(name)
                    ref = taps[name].numpy()
                    line += ' | {} {:.2e}'.format(name, float(np.abs(got[:, :ref.shape[1]] - ref).max()))
                except Exception as e:      # tap not produced (fused layers)
                    line += ' | {} n/a'.format(name)
        print(line, flush=True)
