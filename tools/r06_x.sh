#!/bin/bash
# round 6, call X: per-kernel stats of the lite family (real weights) and dn_lite5 on the final tree
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06x
mkdir -p $OUT
for k in "SR lite2" "SR lite4" "SR lite8" "DN lite5" "SR a2"; do
  tag=$(echo $k | tr ' ' '_')
  TM_ONLY="$k" TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_$tag -o t -f csv -- python tools/time_models.py > $OUT/p_$tag.log 2>&1
  f=$(find $OUT/p_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$tag.csv
  rm -rf $OUT/p_$tag
  echo "== $k: $(grep "$k" $OUT/p_$tag.log)"; head -9 $OUT/kernel_stats_$tag.csv | cut -c1-150
done
