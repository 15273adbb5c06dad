#!/bin/bash
# round 6, call ZP: what the driver runs at round end, on the final tree: build() + smoke() in one process, then the default bench
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06zp
mkdir -p $OUT
( time timeout 900 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" ) > $OUT/smoke.txt 2>&1; tail -4 $OUT/smoke.txt
( time timeout 1200 python bench.py ) > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"; grep real $OUT/bench_default.err; python - $OUT <<'P'
import json, sys
r = json.loads(open(sys.argv[1] + '/bench_default.json').read().strip().splitlines()[-1])
print(r['metric'], r['value'], r['unit'], r['ms_per_step'], 'roofline', r['roofline']['frac'], 'cpu', r['cpu_baseline']['value'], r['cpu_baseline']['cores'])
P
