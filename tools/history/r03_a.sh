#!/bin/bash
# Round-3 pass A: GPU tests (incl. the new full-size / sweep / shared-GPU dist tests), the benchmark line, PMC over the bench's launches.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${R03_TAG:-r03a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu --durations=12 > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -25 $OUT/pytest_gpu.log
cp gpurun_out/fullsize_report.json $OUT/ 2>/dev/null
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 3000 $OUT/bench.json; echo; tail -5 $OUT/bench.err
if [ "${R03_PMC:-1}" = "1" ]; then PMC_TAG=$TAG tools/pmc_bench.sh; fi
