#!/bin/bash
# round 6, call ZD: NetDN on the fp8-correction chain (dual-form stem, conv64_sq + arsb_sq): parity of the dn family + everything that shares the trunk, calibration, frame times
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06zd
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x -k "not config and not dist and not bench_gpus" 2>&1 | tail -6 > $OUT/pytest.txt; cat $OUT/pytest.txt
{
for v in auto x3; do
  echo "== MOE_X3_IMPL=$v"; MOE_X3_IMPL=$v TM_PREC=auto timeout 300 python tools/time_models.py 2>&1 | grep -E "DN lite5|SR a2|SR a4"
  MOE_X3_IMPL=$v timeout 300 python tools/calib_report.py 2>&1 | grep -E "dn_lite|^a4 .*x1.00|^a2 .*x1.00"
done
} > $OUT/dn_chain.txt 2>&1; cat $OUT/dn_chain.txt
TM_ONLY='DN lite5' TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p -o t -f csv -- python tools/time_models.py > $OUT/p.log 2>&1
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_DN_lite5.csv && head -9 "$f" | cut -c1-150; rm -rf $OUT/p
timeout 600 python tools/margin_sweep.py dn_lite5 2>&1 | grep -v amdgpu | tail -8 > $OUT/margin_dn.txt; cat $OUT/margin_dn.txt
