#!/bin/bash
# round 6, call Z9: kernel-resolution table with conv1x1_f2; per-kernel stats of lite4 / lite8 on the fused tree
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06z9
mkdir -p $OUT
bash tools/kernel_table.sh 2>&1 | tail -3; cp gpurun_out/ktable/kernel_resolution.json $OUT/
for k in "SR lite4" "SR lite8"; do
  tag=$(echo $k | tr ' ' '_')
  TM_ONLY="$k" TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p_$tag -o t -f csv -- python tools/time_models.py > $OUT/p_$tag.log 2>&1
  f=$(find $OUT/p_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$tag.csv
  rm -rf $OUT/p_$tag
  echo "== $k: $(grep "$k" $OUT/p_$tag.log)"; head -9 $OUT/kernel_stats_$tag.csv | cut -c1-150
done
