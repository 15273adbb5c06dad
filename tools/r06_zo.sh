#!/bin/bash
# round 6, call ZO: the table kernel with four outputs per thread: lite parity + frame times
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06zo
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x -k "lite or stub or e2e or chain or ragged" 2>&1 | tail -4 > $OUT/pytest.txt; cat $OUT/pytest.txt | cut -c1-300
{ for i in 1 2; do TM_PREC=auto timeout 600 python tools/time_models.py 2>&1 | grep -E "^SR lite"; done; } > $OUT/times.txt 2>&1; cat $OUT/times.txt
