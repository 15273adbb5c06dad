#!/bin/bash
# round 6, call ZI: the whole GPU suite on the final tree (the closing pass r06f4 stopped at a new test's own too-tight bound), then l25's frame time alone and in the all-models run
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06zi
mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu -rs > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
cp gpurun_out/fullsize_report.json $OUT/fullsize_report.json 2>/dev/null
{
TM_ONLY='DN l25' TM_PREC=auto timeout 300 python tools/time_models.py 2>&1 | grep -E "^(SR|DN)"
TM_PREC=auto timeout 600 python tools/time_models.py 2>&1 | grep -E "^(SR|DN)"
TM_ONLY='DN l25' TM_PREC=auto timeout 300 python tools/time_models.py 2>&1 | grep -E "^(SR|DN)"
} > $OUT/time_models.txt 2>&1; cat $OUT/time_models.txt
