#!/bin/bash
# round 6, call ZG: tail3 hi + lo form on packed fp32 FMAs (ring-prefetched rows): parity, calibration, frame time
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06zg
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x -k "not config and not dist and not bench_gpus" 2>&1 | tail -6 > $OUT/pytest.txt; cat $OUT/pytest.txt
{
for i in 1 2; do TM_PREC=auto timeout 300 python tools/time_models.py 2>&1 | grep -E "DN lite5|SR a2|DN l25"; done
timeout 300 python tools/calib_report.py 2>&1 | grep -E "dn_lite"
} > $OUT/dn_tail_f32.txt 2>&1; cat $OUT/dn_tail_f32.txt
TM_ONLY='DN lite5' TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p -o t -f csv -- python tools/time_models.py > $OUT/p.log 2>&1
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_DN_lite5.csv && head -9 "$f" | cut -c1-150; rm -rf $OUT/p
