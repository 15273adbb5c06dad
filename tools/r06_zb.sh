#!/bin/bash
# round 6, call ZB: the GPU suite twice more on the final tree (the driver runs it with -x), smoke, the default bench once
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06zb
mkdir -p $OUT
for i in 1 2; do timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed" | tail -1; done > $OUT/pytest_gpu_repeats.txt 2>&1; cat $OUT/pytest_gpu_repeats.txt
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $OUT/smoke.txt; cat $OUT/smoke.txt
( time timeout 1200 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; grep real $OUT/bench_default.err
tail -1 $OUT/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
