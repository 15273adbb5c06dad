#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06k
mkdir -p $OUT
python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -8 > $OUT/pytest_fullsize_1.txt; cat $OUT/pytest_fullsize_1.txt
for i in 1 2 3; do python -m pytest tests/test_gpu_fullsize.py -q -m gpu -x -k "bench_gpus" 2>&1 | tail -3; done
grep -n "Error\|rank1\|rank2" gpurun_out/bench_gpus3_p2p_failure.txt 2>/dev/null | head -40
python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -5 > $OUT/pytest_parity.txt; cat $OUT/pytest_parity.txt
