#!/bin/bash
# round 5, call E: the U branch of small launch sets on a second stream (option branch_streams): bit-equality test, the drop-in loop with and without, inside one call
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05e
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "second_stream or dropin_protocol or blend_tile or golden or ragged" > $OUT/pytest_subset.txt 2>&1
echo "pytest subset rc=$?"; tail -5 $OUT/pytest_subset.txt
for rep in 1 2; do for bs in 1 0; do echo "== MOE_BRANCH_STREAMS=$bs"; MOE_BRANCH_STREAMS=$bs timeout 200 python tools/prof_dropin.py 8 2>&1 | grep prof_dropin; done; done > $OUT/dropin_branch_streams.txt 2>&1; cat $OUT/dropin_branch_streams.txt
