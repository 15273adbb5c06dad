#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05m
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or ragged or layer_by_layer or e2e or exact_mode" > $OUT/pytest_subset.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_subset.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -f csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sustain 0 --no-noise-input --no-dropin-loop --no-extras --no-configs --no-pmc > $OUT/stats_stdout.log 2>&1
python tools/kstats.py $OUT/stats/bench_kernel_stats.csv stem tailadd stitch; tail -1 $OUT/stats_stdout.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms', d['ms_per_step'])"
rm -rf $OUT/stats
