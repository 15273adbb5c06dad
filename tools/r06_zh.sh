#!/bin/bash
# round 6, call ZH: sedn_weff on fp32 MFMAs: SEDN parity (goldens, config 3 chain, full-size config 3), l25 frame time, per-kernel stats
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06zh
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x -k "sedn or l25 or golden or layer_by_layer or stub" 2>&1 | tail -6 > $OUT/pytest.txt; cat $OUT/pytest.txt
{
for i in 1 2; do TM_ONLY='DN l25' TM_PREC=auto,fp16x3 timeout 300 python tools/time_models.py 2>&1 | grep -E "DN l25"; done
} > $OUT/l25.txt 2>&1; cat $OUT/l25.txt
TM_ONLY='DN l25' TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p -o t -f csv -- python tools/time_models.py > $OUT/p.log 2>&1
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_DN_l25.csv && head -12 "$f" | cut -c1-150; rm -rf $OUT/p
bash tools/kernel_table.sh 2>&1 | tail -3
