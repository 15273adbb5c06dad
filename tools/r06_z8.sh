#!/bin/bash
# round 6, call Z8: conv1x1_f2 (lite's last two upsampler stages + tail in one launch) -- bit-equality with the stage-by-stage form, oracle, A/B on lite4 / lite8 frames
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06z8
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -x -k "lite or golden or kernel_forms or stub" 2>&1 | tail -8 > $OUT/pytest_lite.txt; cat $OUT/pytest_lite.txt
{
for rep in 1 2; do for k in 1 0; do
  echo "== MOE_UP_FUSE2=$k"; MOE_UP_FUSE2=$k TM_PREC=auto timeout 400 python tools/time_models.py 2>&1 | grep -E "lite2|lite4|lite8"
done; done
} > $OUT/ab_fuse2_lite.txt 2>&1; cat $OUT/ab_fuse2_lite.txt
FUZZ_N=12 FUZZ_KEYS=lite2,lite4,lite8 FUZZ_SEED=73 FUZZ_CROPS=4 timeout 600 python tools/fuzz_gpu.py 2>&1 | grep -v amdgpu | grep lite > $OUT/fuzz_lite.txt; cat $OUT/fuzz_lite.txt
