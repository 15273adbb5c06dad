#!/bin/bash
# how does a small launch set (3 planes of 256 x 256 per forward: the drop-in loop) scale with the number of persistent workgroups?  engine-only leg, one stream
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05f
mkdir -p $OUT
for g in 256 224 192 160 128 96 64; do echo "== MOE_MAX_GROUPS=$g"; MOE_BRANCH_STREAMS=0 MOE_MAX_GROUPS=$g DROPIN_ONLY=engine timeout 200 python tools/prof_dropin.py 6 2>&1 | grep prof_dropin; done > $OUT/dropin_max_groups.txt 2>&1; cat $OUT/dropin_max_groups.txt
