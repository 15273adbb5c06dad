#!/bin/bash
# round 6, call U: conv3x3_ps9b (x3 nets' up-conv on 16-channel tiles, four SIMDs) -- parity, A/B against conv3x3_ps9 (x3_form = a) on the a3 frame
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06u
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -x -k "x3_upconv or kernel_forms_agree or unaligned" 2>&1 | tail -12 > $OUT/pytest_ps9b.txt; cat $OUT/pytest_ps9b.txt
{
for rep in 1 2; do for v in b a; do
  echo "== MOE_X3_FORM=$v: $(MOE_X3_FORM=$v TM_ONLY='SR a3' TM_PREC=auto timeout 300 python tools/time_models.py 2>&1 | grep 'SR a3')"
done; done
} > $OUT/ab_a3.txt 2>&1; cat $OUT/ab_a3.txt
TM_ONLY='SR a3' TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_a3 -o a3 -f csv -- python tools/time_models.py > $OUT/prof_a3.log 2>&1
f=$(find $OUT/prof_a3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_SR_a3.csv && head -8 $OUT/kernel_stats_SR_a3.csv | cut -c1-160
rm -rf $OUT/prof_a3
