#!/bin/bash
# round 6, call S: conv3x3_ps1 (SEDN's rblock.0 in the row-streaming form) -- bit-equality with conv3x3_rw, l25 frame and config 3 A/B against up_impl = rw
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06s
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -x -k "sedn or l25 or golden or stub" 2>&1 | tail -6 > $OUT/pytest_sedn.txt; cat $OUT/pytest_sedn.txt
{
for rep in 1 2; do for v in ps4 rw; do
  echo "== MOE_UP_IMPL=$v: $(MOE_UP_IMPL=$v TM_ONLY='DN l25' TM_PREC=auto timeout 300 python tools/time_models.py 2>&1 | grep 'DN l25')"
done; done
} > $OUT/ab_l25.txt 2>&1; cat $OUT/ab_l25.txt
TM_ONLY='DN l25' TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o l25 -f csv -- python tools/time_models.py > $OUT/prof.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_DN_l25.csv && head -8 $OUT/kernel_stats_DN_l25.csv | cut -c1-160
rm -rf $OUT/prof
timeout 600 python bench.py --config 3 --steps 3 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config 3', d['ms_per_step'], d['value'], d.get('roofline'))" > $OUT/c3.txt 2>&1; cat $OUT/c3.txt
