#!/bin/bash
# round 6, call ZJ: plan lanes (a plan's small launch sets on an internal stream beside the large ones): A/B of every family's 1080p frame, interleaved, then bit-identity and the GPU suite
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06zj
mkdir -p $OUT
{
for i in 1 2; do for v in 0 1; do echo "== MOE_PLAN_LANES=$v"; MOE_PLAN_LANES=$v TM_PREC=auto timeout 600 python tools/time_models.py 2>&1 | grep -E "^(SR|DN)"; done; done
for g in 32 128; do echo "== MOE_PLAN_LANES=1 MOE_PLAN_LANE_GROUPS=$g"; MOE_PLAN_LANE_GROUPS=$g TM_PREC=auto timeout 600 python tools/time_models.py 2>&1 | grep -E "^(SR|DN)"; done
} > $OUT/ab_plan_lanes.txt 2>&1; cat $OUT/ab_plan_lanes.txt
python - > $OUT/bits.txt 2>&1 <<'P'
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, golden_defs as gd
from moephoto_amd import imageProcess as ip, runDN, runSR
from moephoto_amd.config import config
config.deviceId, config.fp16, config.crop_sr, config.crop_dn, config.crop_dns, config.modelRoot = 0, True, 256, 256, 256, gd.ZOO
x = torch.from_numpy(gd.natural_image(1000, (3, 1080, 1920))).cuda().half()
for name, mk in (('SR a2', lambda: runSR.getOpt({'model': 'a', 'scale': 2})), ('DN lite5', lambda: runDN.getOpt({'model': 'lite5'})), ('SR lite2', lambda: runSR.getOpt({'model': 'lite', 'scale': 2}))):
    ip.modelCache.clear()
    opt = mk()
    m = opt.modelCached
    y1 = ip.doCrop(opt, x).clone()
    m.set_option('plan_lanes', 0)
    y0 = ip.doCrop(opt, x).clone()
    m.set_option('plan_lanes', 1)
    y2 = ip.doCrop(opt, x).clone()
    print(name, 'lanes on == off:', bool(torch.equal(y0, y1)), 'repeat:', bool(torch.equal(y1, y2)), flush=True)
P
cat $OUT/bits.txt | grep -v amdgpu
timeout 1500 python -m pytest tests -x -q -m gpu -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log | cut -c1-200
