#!/bin/bash
# round 6, call R: lite's conv1x1 with the zero k-slice of the 48-channel nets skipped (option k48) -- parity of the lite family, A/B on lite2 / lite4 / lite8 frames
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06r
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -x -k "lite or golden or kernel_forms or stub" 2>&1 | tail -5 > $OUT/pytest_lite.txt; cat $OUT/pytest_lite.txt
{
for rep in 1 2; do for k in 1 0; do
  echo "== MOE_K48=$k"; MOE_K48=$k TM_PREC=auto timeout 300 python tools/time_models.py 2>&1 | grep -E "lite2|lite4"
done; done
} > $OUT/ab_k48_lite.txt 2>&1; cat $OUT/ab_k48_lite.txt
