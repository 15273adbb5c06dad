#!/bin/bash
# round 6, call ZC: NetDN (dn_lite5 / 10 / 15, real weights) with the fp8-correction form of the split-operand layers (x3_impl = q8) against the default (x3): time and calibration error
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06zc
mkdir -p $OUT
{
for v in auto q8 x3; do
  echo "== MOE_X3_IMPL=$v: $(MOE_X3_IMPL=$v TM_ONLY='DN lite5' TM_PREC=auto timeout 300 python tools/time_models.py 2>&1 | grep 'DN lite5')"
  MOE_X3_IMPL=$v timeout 300 python tools/calib_report.py 2>&1 | grep -E "dn_lite"
done
} > $OUT/dn_q8.txt 2>&1; cat $OUT/dn_q8.txt
MOE_X3_IMPL=q8 TM_ONLY='DN lite5' TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p -o t -f csv -- python tools/time_models.py > $OUT/p.log 2>&1
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -9 "$f" | cut -c1-150; rm -rf $OUT/p
