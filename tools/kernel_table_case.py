#!/usr/bin/env python
"""One case of the kernel-resolution table (tools/kernel_table.sh runs it under rocprofv3 --kernel-trace --stats):  KEY PRECISION CLASS
   CLASS frame: doCrop of a 3 x 300 x 420 fp16 image with 256-px tiles (full, ragged-right, ragged-bottom, corner tiles: batched launch sets)
         tile:  the per-tile call, 3 planes of 64 x 96 (a small launch set: the reference's own loop)
         odd:   forwards whose rows are not a multiple of four / tiny shapes (fp32 I/O): the fallbacks of the fast kernels"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import golden_defs as gd  # noqa: E402
from moephoto_amd import imageProcess as ip, models  # noqa: E402
from moephoto_amd.config import config  # noqa: E402
from moephoto_amd.weights import load_state_dict_file  # noqa: E402

key, prec, cls = sys.argv[1:4]
CTOR = {'net2x': models.Net2x, 'net3x': models.Net3x, 'net4x': models.Net4x, 'netdn': models.NetDN, 'sedn': models.SEDN,
        'lite2': lambda: models.Net(2), 'lite4': lambda: models.Net(4), 'lite8': lambda: models.Net(8)}
arch = gd.MODELS[key][0]
m = CTOR[arch]()
m.load_state_dict({n: torch.from_numpy(v) for n, v in gd.state_dict_for(key, load_state_dict_file).items()})
m.precision = prec
dev = torch.device('cuda', 0)
if cls == 'frame':
    config.deviceId, config.fp16 = 0, True
    opt = ip.Option()
    opt.fixChannel, opt.scale, opt.cropsize = 0, m.scale, 256
    opt.padding = 7 if m.scale == 1 else (9 if m.scale == 3 else 5)
    opt.modelDef = type(m)
    opt.modelCached = m.eval().to(dtype=torch.float16, device=dev)
    ip.doCrop(opt, torch.from_numpy(gd.natural_image(3, (3, 300, 420))).to(dev).half())
elif cls == 'tile':
    m = m.eval().to(dtype=torch.float16, device=dev)
    m(torch.from_numpy(gd.natural_image(4, (3, 64, 96))[:, None]).to(dev).half())
else:
    m = m.eval().to(dtype=torch.float32, device=dev)
    for shape in ((3, 22, 36), (2, 8, 16), (1, 10, 9)):
        m(torch.from_numpy(gd.natural_image(5, shape)[:, None]).to(dev))
torch.cuda.synchronize()
print('case', key, prec, cls, m.resolved_precision(), 'done')
