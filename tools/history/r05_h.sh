#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05g
mkdir -p $OUT
for sc in 2 3; do for bg in off 24 32 48 64 80; do
  if [ $bg = off ]; then e="MOE_BRANCH_STREAMS=0"; else e="MOE_BRANCH_GROUPS=$bg"; fi
  echo "== scale $sc $e"; env $e DROPIN_SCALE=$sc DROPIN_ONLY=engine timeout 200 python tools/prof_dropin.py 6 2>&1 | grep -E "prof_dropin|Error|error" | tail -2
done; done > $OUT/dropin_branch_groups_a2_a3.txt 2>&1
for bg in 72 80 88; do echo "== scale 4 MOE_BRANCH_GROUPS=$bg"; MOE_BRANCH_GROUPS=$bg DROPIN_ONLY=engine timeout 200 python tools/prof_dropin.py 6 2>&1 | grep prof_dropin; done >> $OUT/dropin_branch_groups_a2_a3.txt 2>&1
cat $OUT/dropin_branch_groups_a2_a3.txt
