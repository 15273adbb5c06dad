#!/usr/bin/env python
"""Small fixed workload for rocprofv3 counter passes: Net4x (a4-synth) forward on B=12 planes of 256x256
(= 4 RGB tiles of BASELINE config 2), ITER times.  Used by tools/profile_gpu.sh."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import golden_defs as gd  # noqa: E402
from moephoto_amd import models  # noqa: E402
from moephoto_amd.weights import load_state_dict_file  # noqa: E402

B = int(os.environ.get('PROF_B', '12'))
ITER = int(os.environ.get('PROF_ITER', '2'))
KEY = os.environ.get('PROF_MODEL', 'a4')     # a4 | a3 | a2
m = {'a4': models.Net4x, 'a3': models.Net3x, 'a2': models.Net2x}[KEY]()
m.load_state_dict({n: torch.from_numpy(v) for n, v in gd.state_dict_for(KEY, load_state_dict_file).items()})
m.to(dtype=torch.float16, device='cuda:0')
x = torch.from_numpy(gd.natural_image(1, (B, 256, 256))[:, None]).cuda().half()
for _ in range(ITER):
    y = m(x)
torch.cuda.synchronize()
print('done', y[-1].shape)
