#!/bin/bash
# per-kernel breakdown of a model family on the 1080p frame: rocprofv3 --kernel-trace --stats      MODELS="DN lite5;SR lite2" tools/r03_g.sh
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${R03_TAG:-r03q}
mkdir -p $OUT
IFS=';' read -ra MS <<< "${MODELS:-DN lite5;SR lite2}"
for m in "${MS[@]}"; do
  t=$(echo $m | tr ' ' '_')
  TM_ONLY="$m" TM_PREC=auto timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$t -o p -f csv -- python tools/time_models.py > $OUT/tm_$t.log 2>&1
  grep "ms/frame" $OUT/tm_$t.log
  f=$(find $OUT/prof_$t -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats_$t.csv && head -14 $f | cut -c1-160
  rm -rf $OUT/prof_$t
done
