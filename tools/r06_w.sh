#!/bin/bash
# round 6, call W: what the driver runs at round end, once more on a fresh box: GPU suite, build() + smoke() in one process, the default bench command
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06w
mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -3 > $OUT/smoke.txt; cat $OUT/smoke.txt
( time timeout 1200 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; grep real $OUT/bench_default.err
tail -1 $OUT/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['summary'])"
