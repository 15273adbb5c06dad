#!/bin/bash
# per-kernel stats of the other families on the final tree (1080p frame, 256-px tiles, default arithmetic)  -> gpurun_out/r05j/
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05j
mkdir -p $OUT
for m in "SR a2" "SR a3" "SR lite2" "SR lite4" "DN lite5" "DN l25"; do
  tag=$(echo $m | tr ' ' '_')
  TM_ONLY="$m" TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/$tag -o ks -f csv -- python tools/time_models.py > $OUT/$tag.log 2>&1
  grep "ms/frame" $OUT/$tag.log
  python tools/kstats.py $OUT/$tag/ks_kernel_stats.csv | head -12
  cp $OUT/$tag/ks_kernel_stats.csv $OUT/kernel_stats_$tag.csv; rm -rf $OUT/$tag
done 2>&1 | tee $OUT/summary.txt
