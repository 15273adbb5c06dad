#!/usr/bin/env python
"""What moe_net_finalize(MOE_PREC_AUTO) settles on per zoo key: the count of split-operand ARSBs, the worst noise-tile error of the calibration, the time of
`.to(device)` (first: with the measurement; second: cached), and the same for the trunk x 1.15 variants of a2 / a4."""
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

dev = torch.device('cuda', 0)
torch.zeros(1, device=dev)
CLS = {'a2': models.Net2x, 'p2': models.Net2x, 'a3': models.Net3x, 'a4': models.Net4x, 'dn_lite5': models.NetDN, 'dn_lite10': models.NetDN, 'dn_lite15': models.NetDN}
for key in ('a4', 'a2', 'p2', 'a3', 'dn_lite5', 'dn_lite10', 'dn_lite15'):
    for scale in ((1.0, 1.15) if key in ('a2', 'a4') else (1.0,)):
        sd = gd.state_dict_for(key, load_state_dict_file)
        sd = {k: (np.ascontiguousarray(a * np.float32(scale)) if (k.startswith('conv_input2') or (k.startswith('convt_F') and a.ndim == 4)) else a) for k, a in sd.items()}
        m = CLS[key]()
        m.load_state_dict({n: torch.from_numpy(np.ascontiguousarray(v, np.float32)) for n, v in sd.items()})
        t0 = time.perf_counter()
        m = m.eval().to(dtype=torch.float16, device=dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        m._finalized_key = None
        m.to(device=dev)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n0 = m.exact_blocks()
        errs = []
        for tgt in (1e-9,):                      # an unreachable target walks every count: the error per count
            pass
        line = '{:10s} trunk x{:.2f}: auto -> {} with {} split blocks | .to() {:.0f} ms with the measurement, {:.0f} ms cached'.format(key, scale, m.resolved_precision(), n0, (t1 - t0) * 1e3, (t2 - t1) * 1e3)
        r = m.calibrate()
        line += ' | calibrate() = {}'.format(None if r is None else (r[0], float('{:.3e}'.format(r[1]))))
        r = m.calibrate(7e-4)
        line += ' | target 7e-4: {}'.format(None if r is None else (r[0], float('{:.3e}'.format(r[1]))))
        print(line, flush=True)
