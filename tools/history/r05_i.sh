#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05i
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lite or rccl_path or golden" > $OUT/pytest_subset.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.txt
python - <<P 2>&1 | grep -v amdgpu.ids | tee $OUT/lite_frames.txt
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, golden_defs as gd
from moephoto_amd import imageProcess as ip, runSR
from moephoto_amd.config import config
config.deviceId, config.fp16, config.crop_sr, config.modelRoot = 0, True, 256, gd.ZOO
x = torch.from_numpy(gd.natural_image(1000, (3, 1080, 1920))).cuda().half()
for sc in (2, 4, 8):
    opt = runSR.getOpt({'op': 'SR', 'model': 'lite', 'scale': sc, 'ensemble': 0})
    for _ in range(2): ip.doCrop(opt, x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): ip.doCrop(opt, x)
    torch.cuda.synchronize()
    print('lite%d 1080p frame: %.2f ms' % (sc, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
    del opt; ip.modelCache.clear(); torch.cuda.empty_cache()
P
