#!/usr/bin/env python
"""launch sets of a2 (48 planes of 256 x 256) with q8_impl = p, then s: run under rocprofv3 --kernel-trace --stats to compare conv64_q8_kernel and conv64_sq_kernel per launch"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch
import golden_defs as gd
from moephoto_amd import models
from moephoto_amd.weights import load_state_dict_file
key = os.environ.get('SQ_KEY', 'a2')
m = {'a2': models.Net2x, 'a4': models.Net4x}[key]()
m.load_state_dict({n: torch.from_numpy(v) for n, v in gd.state_dict_for(key, load_state_dict_file).items()})
m.eval(); m = m.to(dtype=torch.float16, device='cuda:0')
x = torch.from_numpy(gd.noise_image(5, (48, 1, 256, 256))).cuda().half()
for impl in os.environ.get('SQ_IMPLS', 'p,s').split(','):
    m.set_option('q8_impl', impl)
    for _ in range(2):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(6):
        m(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 6 * 1e3
    m.set_profile('input2,c1_,c2_,xpair')
    for _ in range(6):
        m(x)
    torch.cuda.synchronize()
    pr = m.get_profile(all_keys=True)
    m.set_profile(None)
    print('%s %-8s q8_impl = %s: %.3f ms per launch set | split-operand layers: %s' % (key, os.environ.get('SQ_TAG', ''), impl, ms,
          '  '.join('%s %.1f us x %d' % (k, p['total_ms'] / max(1, p['launches']) * 1e3, p['launches'] // 6) for k, p in zip(('input2', 'conv_1', 'conv_2', 'fused exact ARSB'), pr) if p['launches'])), flush=True)
