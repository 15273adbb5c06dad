#!/bin/bash
# kernel-trace stats of the benchmark command (mixed precision, fused ARSB)
export TMPDIR=/tmp
OUT=gpurun_out/r02e
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -f csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --sustain 0 --no-noise-input > $OUT/stats_stdout.log 2>&1
echo "stats rc=$?"
python tools/summarize_prof.py $OUT 2>&1 | head -40
