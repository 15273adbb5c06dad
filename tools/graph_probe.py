#!/usr/bin/env python
"""Is the per-tile forward launch-bound?  One a4 forward of 3 x 256 x 256 (the reference loop's call) eagerly, back to back, against the same forward captured into a
hipGraph (torch.cuda.CUDAGraph) and replayed: kernel launches per forward, ms per forward each way, bits compared.  (The engine takes part in a capture like any stream
work: the U-branch fork / join are event edges inside the captured stream; the overlap path of moe_net_forward_ex stands aside under capture.)"""
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
for key, ctor in (('a4', models.Net4x), ('a2', models.Net2x)):
    m = ctor()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in gd.state_dict_for(key, load_state_dict_file).items()})
    m = m.eval().to(dtype=torch.float16, device=dev)
    x = torch.from_numpy(gd.natural_image(3, (3, 256, 256))).to(dev).half()[:, None].contiguous()
    for fork in (1, 0):
        m.set_option('branch_streams', fork)
        m.set_option('overlap_calls', 0)
        for _ in range(3):
            y_ref = m(x)[-1]
        torch.cuda.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            m(x)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n * 1e3
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m(x)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        try:
            with torch.cuda.graph(g):
                y_g = m(x)[-1]
            g.replay()
            torch.cuda.synchronize()
            same = bool(torch.equal(y_g, y_ref))
            t0 = time.perf_counter()
            for _ in range(n):
                g.replay()
            torch.cuda.synchronize()
            rep = (time.perf_counter() - t0) / n * 1e3
            print('{} 3x256x256, branch_streams={}: eager {:.3f} ms per forward, hipGraph replay {:.3f} ms ({:+.1f} %), bit-identical {}'.format(key, fork, eager, rep, (rep / eager - 1) * 100, same), flush=True)
        except Exception as e:
            print('{} branch_streams={}: capture failed: {}'.format(key, fork, str(e).splitlines()[0][:300]), flush=True)
