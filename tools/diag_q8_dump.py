#!/usr/bin/env python
"""Dump the 'input2' tap of the loaded library (option x3_impl = q8) for offline fingerprinting: gpurun_out/q8diag_<tag>.npy"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
import golden_defs as gd
from moephoto_amd.weights import load_state_dict_file
from moephoto_amd import models
key, tag = 'a2', sys.argv[1]
sd = gd.state_dict_for(key, load_state_dict_file)
x = gd.natural_image(3, (2, 24, 40))[:, None]
m = models.Net2x(); m.load_state_dict({n: torch.from_numpy(v) for n, v in sd.items()}); m.eval(); m = m.to(device='cuda:0')
m.set_option('x3_impl', 'q8').set_debug(True)
m(torch.from_numpy(x).cuda()); torch.cuda.synchronize()
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
np.save(os.path.join(ROOT, 'gpurun_out', 'q8diag_%s.npy' % tag), m.debug_tap('input2'))
np.save(os.path.join(ROOT, 'gpurun_out', 'q8diag_stem.npy'), m.debug_tap('stem'))
