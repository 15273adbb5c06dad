#!/bin/bash
# Round-3 closing pass on the final tree: all GPU tests, the benchmark line, PMC over the bench's launches (tools/r03_a.sh), the kernel-trace summary of the
# same bench command, frame times of every model (x3_impl auto, and the SR nets with x3 for the A/B)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${R03_TAG:-r04h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
R03_TAG=$TAG bash tools/r03_a.sh
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -f csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sustain 0 --no-noise-input --no-dropin-loop > $OUT/stats_stdout.log 2>&1; echo "stats rc=$?"
cp $OUT/stats/*/bench_kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null || find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
head -12 $OUT/bench_kernel_stats.csv
TM_PREC=auto timeout 600 python tools/time_models.py > $OUT/time_models.txt 2>&1; cat $OUT/time_models.txt | grep -v "^$" | tail -12
MOE_X3_IMPL=x3 TM_PREC=auto timeout 600 python tools/time_models.py > $OUT/time_models_x3.txt 2>&1; grep -v "^$" $OUT/time_models_x3.txt | tail -12
