#!/bin/bash
# round 6, call ZQ: the GPU suite twice more on a fresh box, final tree (flakiness check after the frame-time test's change)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06zq
mkdir -p $OUT
for i in 1 2; do timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -2 | cut -c1-200; done > $OUT/pytest_twice.txt 2>&1; cat $OUT/pytest_twice.txt
