#!/bin/bash
# round 6, call T: new tests (unaligned output planes, a3 full-size tile vs the oracle), counters of conv3x3_ps9 beside conv3x3_ps4
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06t
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -x -k "unaligned or x3_upconv or sedn" 2>&1 | tail -6 > $OUT/pytest_new.txt; cat $OUT/pytest_new.txt
PMC_TAG=r06t bash tools/pmc_ps9.sh > $OUT/pmc_ps9.txt 2>&1; cat $OUT/pmc_ps9.txt
