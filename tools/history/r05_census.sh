#!/bin/bash
# kernel census (tools/kernel_census.py): (a) the defaults workload, (b) the whole GPU test suite, each under rocprofv3 --kernel-trace --stats  -> gpurun_out/r05census/
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05census
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/defaults -o census -f csv -- python tools/census_workload.py > $OUT/defaults.log 2>&1; echo "defaults rc=$?"; grep "census workload" $OUT/defaults.log | tail -12
python tools/kernel_census.py $OUT/defaults/census_kernel_stats.csv "defaults workload (tools/census_workload.py)" > $OUT/census_defaults.txt 2>&1; head -70 $OUT/census_defaults.txt
timeout 1500 rocprofv3 --kernel-trace --stats -d $OUT/suite -o census -f csv -- python -m pytest tests -m gpu -x -q > $OUT/suite.log 2>&1; echo "suite rc=$?"; grep -E "passed|failed" $OUT/suite.log | tail -3
python tools/kernel_census.py $OUT/suite/census_kernel_stats.csv "the GPU test suite (pytest tests -m gpu)" > $OUT/census_suite.txt 2>&1; grep -A40 "never launched" $OUT/census_suite.txt
rm -rf $OUT/suite/*kernel_trace.csv $OUT/defaults/*kernel_trace.csv
