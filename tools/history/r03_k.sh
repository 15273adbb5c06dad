#!/bin/bash
# 48-channel k-slice skip (option k48) on the NetDN family: the ARSB test, then 1080p frame times with k48 = 0 / 1 alternating
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${R03_TAG:-r04c}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused_arsb or golden or layer_by_layer" > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_sel.log
for rep in 1 2; do for k in 0 1; do echo -n "k48=$k "; MOE_K48=$k TM_ONLY="DN lite5" TM_PREC=auto timeout 300 python tools/time_models.py 2>/dev/null | tail -1; done; done
