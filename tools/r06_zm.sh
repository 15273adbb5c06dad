#!/bin/bash
# round 6, call ZM: lite's conv_input2 in closed form inside the stem (option stem2): the whole GPU suite on this tree, A/B of the lite frames, kernel table
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06zm
mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log | cut -c1-250
{
for i in 1 2; do for v in 0 1; do echo "== MOE_STEM2=$v"; MOE_STEM2=$v TM_PREC=auto timeout 600 python tools/time_models.py 2>&1 | grep -E "^SR lite"; done; done
} > $OUT/ab_stem2.txt 2>&1; cat $OUT/ab_stem2.txt
bash tools/kernel_table.sh 2>&1 | tail -3
