#!/bin/bash
# round 6, call Z2: conv1x1 with register-resident weights and a four-tile ring for the split-operand stages of the 48-channel nets -- lite parity, frame times, kernel table
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06z2
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -x -k "lite or golden or kernel_forms or stub or fuzz" 2>&1 | tail -5 > $OUT/pytest_lite.txt; cat $OUT/pytest_lite.txt
{
for rep in 1 2; do for k in 1 0; do
  echo "== MOE_K48=$k"; MOE_K48=$k TM_PREC=auto timeout 400 python tools/time_models.py 2>&1 | grep -E "lite2|lite4|lite8"
done; done
} > $OUT/ab_k48_lite.txt 2>&1; cat $OUT/ab_k48_lite.txt
FUZZ_N=12 FUZZ_KEYS=lite2,lite4,lite8 FUZZ_SEED=71 FUZZ_CROPS=4 timeout 600 python tools/fuzz_gpu.py 2>&1 | grep -v amdgpu | tail -10 > $OUT/fuzz_lite.txt; cat $OUT/fuzz_lite.txt
bash tools/kernel_table.sh 2>&1 | tail -3; cp gpurun_out/ktable/kernel_resolution.json $OUT/
