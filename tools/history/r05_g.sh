#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05g
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "second_stream or dropin_protocol or golden or ragged" > $OUT/pytest_subset.txt 2>&1
echo "pytest subset rc=$?"; tail -5 $OUT/pytest_subset.txt
for bg in 0 48 64 80 96 112; do echo "== MOE_BRANCH_GROUPS=$bg"; MOE_BRANCH_GROUPS=$bg DROPIN_ONLY=engine timeout 200 python tools/prof_dropin.py 6 2>&1 | grep prof_dropin; done > $OUT/dropin_branch_groups.txt 2>&1
echo "== MOE_BRANCH_STREAMS=0" >> $OUT/dropin_branch_groups.txt; MOE_BRANCH_STREAMS=0 DROPIN_ONLY=engine timeout 200 python tools/prof_dropin.py 6 2>&1 | grep prof_dropin >> $OUT/dropin_branch_groups.txt
cat $OUT/dropin_branch_groups.txt
