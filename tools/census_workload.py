#!/usr/bin/env python
"""The DEFAULTS workload of tools/kernel_census.py: every zoo key in its default arithmetic over a 1080p frame through doCrop (full + ragged tiles), one frame through the
reference-style per-tile loop (small launch sets), fp32 and fp16 I/O, an RGBA input, the denoise wrapper, the uint8 / uint16 I/O edges and the three resize modes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import golden_defs as gd  # noqa: E402
import bench  # noqa: E402
from moephoto_amd import _lib, imageProcess as ip, runDN, runSR  # noqa: E402
from moephoto_amd.config import config  # noqa: E402
from moephoto_amd.weights import load_state_dict_file, save_state_dict_file  # noqa: E402

config.deviceId, config.fp16, config.crop_sr, config.crop_dn, config.crop_dns, config.modelRoot = 0, True, 256, 256, 256, gd.ZOO
for key, table, slot in (('a4', runSR, 'a4'), ('a3', runSR, 'a3'), ('l25', runDN, '25')):
    path = '/tmp/moe_census_{}.pth'.format(key)
    save_state_dict_file(gd.synth_state_dict(key, load_state_dict_file), path)
    table.mode_switch[slot] = (path,) + tuple(table.mode_switch[slot][1:])
dev = torch.device('cuda', 0)
x16 = torch.from_numpy(gd.natural_image(1000, (3, 1080, 1920))).to(dev).half()
small = torch.from_numpy(gd.natural_image(5, (3, 300, 420))).to(dev)
rgba = torch.from_numpy(gd.natural_image(6, (4, 300, 420))).to(dev).half()
cases = [('SR a2', lambda: runSR.getOpt({'op': 'SR', 'model': 'a', 'scale': 2, 'ensemble': 0})), ('SR a3', lambda: runSR.getOpt({'op': 'SR', 'model': 'a', 'scale': 3, 'ensemble': 0})),
         ('SR a4', lambda: runSR.getOpt({'op': 'SR', 'model': 'a', 'scale': 4, 'ensemble': 0})), ('SR p2', lambda: runSR.getOpt({'op': 'SR', 'model': 'p', 'scale': 2, 'ensemble': 0})),
         ('SR lite2', lambda: runSR.getOpt({'op': 'SR', 'model': 'lite', 'scale': 2, 'ensemble': 0})), ('SR lite4', lambda: runSR.getOpt({'op': 'SR', 'model': 'lite', 'scale': 4, 'ensemble': 0})),
         ('SR lite8', lambda: runSR.getOpt({'op': 'SR', 'model': 'lite', 'scale': 8, 'ensemble': 0})),
         ('DN lite5', lambda: runDN.getOpt({'op': 'DN', 'model': 'lite5'})), ('DN lite10', lambda: runDN.getOpt({'op': 'DN', 'model': 'lite10'})), ('DN l25', lambda: runDN.getOpt({'op': 'DN', 'model': '25'}))]
for name, mk in cases:
    ip.modelCache.clear()
    opt = mk()
    ip.doCrop(opt, x16)                                  # the batched device-resident path, fp16 I/O
    config.fp16 = False
    ip.doCrop(opt, small)                                # fp32 I/O, small ragged tiles
    config.fp16 = True
    if name.startswith('SR'):
        ip.doCrop(opt, rgba)                             # alpha as a fourth plane
        plan = ip._plan_for(opt, small.half().shape)
        ramp = torch.from_numpy(plan.ramp.copy()).to(dev).half()
        bench._reference_style_loop(opt, small.half(), plan, ramp, torch)                              # per-tile calls: small launch sets
        bench._reference_style_loop(opt, small.half(), plan, ramp, torch, blend_tile=ip.blendTile)
    torch.cuda.synchronize()
    print('census workload:', name, opt.modelCached.resolved_precision(), flush=True)
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
u8 = torch.randint(0, 256, (300, 420, 3), dtype=torch.uint8, device=dev)
f16 = torch.empty((3, 300, 420), dtype=torch.float16, device=dev)
_lib.check(L.moe_to_float(u8.data_ptr(), _lib.U8, 8, 300, 420, 3, f16.data_ptr(), _lib.F16, 0, st))
_lib.check(L.moe_to_output(f16.data_ptr(), _lib.F16, 300, 420, 3, 8, u8.data_ptr(), _lib.U8, 0, st))
u16 = torch.empty((300, 420, 3), dtype=torch.int16, device=dev)
_lib.check(L.moe_to_output(f16.data_ptr(), _lib.F16, 300, 420, 3, 16, u16.data_ptr(), _lib.U16, 0, st))
for mode in (0, 1, 2):
    dst = torch.empty((3, 200, 333), dtype=torch.float16, device=dev)
    _lib.check(L.moe_resize(f16.data_ptr(), dst.data_ptr(), _lib.F16, 3, 300, 420, 200, 333, mode, 0, st))
torch.cuda.synchronize()
print('census workload: done')
