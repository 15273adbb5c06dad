#!/usr/bin/env python
"""The reference's own per-tile loop around the drop-in class (bench.py: dropin_loop; python/imageProcess.py:157-172) taken apart:
   loop      the whole thing: 40 x (slice view -> Net4x.__call__ on 3 planes -> two torch blends -> slice-assign)
   engine    the 40 forwards alone (results dropped)
   blends    the torch blends + assigns alone (on precomputed tile results)
   device    moe_run_plan (the headline's device-resident doCrop) for comparison
each as wall time per frame (host-inclusive, synchronised at both ends) over N frames.  Under rocprofv3 --kernel-trace --stats (DROPIN_ONLY=loop) the
kernel-time sums split the loop's wall time into engine kernels, torch kernels and host gaps.

    python tools/prof_dropin.py [frames, default 8]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import golden_defs as gd  # noqa: E402
import bench  # noqa: E402
from moephoto_amd import imageProcess as ip, runSR  # noqa: E402
from moephoto_amd.config import config  # noqa: E402
from moephoto_amd.weights import load_state_dict_file, save_state_dict_file  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
only = os.environ.get('DROPIN_ONLY')
dev = torch.device('cuda', 0)
config.deviceId, config.fp16, config.crop_sr = 0, True, 256
wpath = '/tmp/moe_prof_dropin_a4.pth'
save_state_dict_file(gd.synth_state_dict('a4', load_state_dict_file), wpath)
runSR.mode_switch['a4'] = (wpath, runSR.mode_switch['a4'][1])
SCALE = int(os.environ.get('DROPIN_SCALE', '4'))      # 4: a4 (synthetic weights); 2 / 3: a2 / a3 from the zoo fixtures
if SCALE != 4:
    config.modelRoot = gd.ZOO
opt = runSR.getOpt({'op': 'SR', 'model': 'a', 'scale': SCALE, 'ensemble': 0})
x = torch.from_numpy(gd.natural_image(1000, bench.FRAME)).to(dev).half()
plan = ip._plan_for(opt, x.shape)
ramp = torch.from_numpy(plan.ramp.copy()).to(dev).half()
sc, psc = plan.sc, plan.padSc
xb = plan.padImage(x).unsqueeze(1)


def loop():
    return bench._reference_style_loop(opt, x, plan, ramp, torch)


def engine():
    for (top, bottom, left, right, tt, lt, bsc, rsc) in plan.tiles:
        opt(xb[..., top:bottom, left:right])


tiles = [opt(xb[..., t[0]:t[1], t[2]:t[3]]).squeeze(1).clone() for t in plan.tiles]


def blends():
    class Fixed(object):
        def __init__(self):
            self.k = 0

        def __call__(self, _):
            r = tiles[self.k].unsqueeze(1)
            self.k += 1
            return r
    return bench._reference_style_loop(Fixed(), x, plan, ramp, torch)


def device():
    return ip.doCrop(opt, x)


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


legs = (('loop', loop), ('engine', engine), ('blends', blends), ('device', device))
out = {}
for name, fn in legs:
    if only and name != only:
        continue
    out[name] = timed(fn, N)
    print('prof_dropin: {:7s} {:.3f} ms per frame'.format(name, out[name]), flush=True)
if not only:
    print('prof_dropin: loop - engine - blends = {:.3f} ms (what overlapping the two hides or serialising them adds)'.format(out['loop'] - out['engine'] - out['blends']))
