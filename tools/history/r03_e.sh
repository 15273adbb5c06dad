#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${R03_TAG:-r03i}
mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu -k "${R03_K}" --durations=5 > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_sel.log
cp gpurun_out/fullsize_report.json $OUT/ 2>/dev/null
if [ -n "${R03_MODELS:-}" ]; then TM_PREC=auto timeout 600 python tools/time_models.py > $OUT/time_models.txt 2>&1; grep ms/frame $OUT/time_models.txt; fi
