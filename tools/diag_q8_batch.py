#!/usr/bin/env python
"""conv64_q8: is a tile's result independent of its launch set?  Same frame, 1 / 4 / 16 tiles per launch set, and the same setting twice."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
import golden_defs as gd
sys.argv = sys.argv[:1]
import test_gpu_parity as T
from moephoto_amd import _lib, imageProcess as ip
dev = torch.device('cuda:0')
opt = T._opt_sr('a', 4, 256)
opt.modelCached.set_option('x3_impl', os.environ.get('IMPL', 'q8'))
x = gd.natural_image(0, (3, 1080, 1920))
xd = torch.from_numpy(x).to(dev)
plan = ip._plan_for(opt, xd.shape)
L, model = _lib.lib(), opt.modelCached
stream = torch.cuda.current_stream().cuda_stream
sC, sH, sW = xd.stride()
def run(per_batch):
    pool = torch.zeros(plan.pool_elems(3), dtype=torch.float32, device=dev)
    out = torch.empty((3, plan.outH, plan.outW), dtype=torch.float32, device=dev)
    _lib.check(L.moe_run_plan_ex(model._h, plan._h, xd.data_ptr(), _lib.F32, sC, sH, sW, out.data_ptr(), _lib.F32, per_batch, ctypes.c_void_p(pool.data_ptr()), 0, 1, 1, stream))
    torch.cuda.synchronize()
    return pool
a4, b4, a1, a16 = run(4), run(4), run(1), run(16)
off = plan.tile_offsets(3) + [plan.pool_elems(3)]
def rep(name, u, v):
    d = (u - v).abs()
    bad = [k for k in range(plan.n_tiles) if float(d[off[k]:off[k + 1]].max()) > 0]
    print(name, 'max diff %.3e' % float(d.max()), 'tiles that differ:', bad[:12], '(%d)' % len(bad), 'elements: %d' % int((d > 0).sum()))
rep('4 vs 4 again', a4, b4); rep('1 vs 4', a1, a4); rep('16 vs 4', a16, a4)
