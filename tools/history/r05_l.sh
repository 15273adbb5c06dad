#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05l
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "l25 or sedn or SEDN or config3 or golden" > $OUT/pytest_subset.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.txt
TM_ONLY="DN l25" TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/l25 -o ks -f csv -- python tools/time_models.py > $OUT/l25.log 2>&1
grep "ms/frame" $OUT/l25.log; python tools/kstats.py $OUT/l25/ks_kernel_stats.csv | head -8
cp $OUT/l25/ks_kernel_stats.csv $OUT/kernel_stats_DN_l25.csv; rm -rf $OUT/l25
