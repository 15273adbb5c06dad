#!/bin/bash
# round 6, call ZL: frm_pre / sedn_fmean with their load loops unrolled (sixteen / eight loads in flight): lite + SEDN parity, frame times, per-kernel stats
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06zl
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x -k "lite or sedn or l25 or stub or golden" 2>&1 | tail -4 > $OUT/pytest.txt; cat $OUT/pytest.txt | cut -c1-300
{
for i in 1 2; do TM_PREC=auto timeout 600 python tools/time_models.py 2>&1 | grep -E "^SR lite|^DN l25"; done
} > $OUT/times.txt 2>&1; cat $OUT/times.txt
for m in "SR lite2" "DN l25"; do
TM_ONLY="$m" TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p -o t -f csv -- python tools/time_models.py > $OUT/p.log 2>&1
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "frm_pre|sedn_" "$f" | cut -c1-150; rm -rf $OUT/p
done > $OUT/helpers.txt 2>&1; cat $OUT/helpers.txt
