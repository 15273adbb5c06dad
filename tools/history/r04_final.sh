#!/bin/bash
# Round-4 closing pass on ONE binary in ONE call: all GPU tests, the benchmark line (default = config 2, then configs 3 / 4 / 5), PMC over the bench's own launches,
# the kernel-trace summary of the same command, frame times of every model family.  Results under gpurun_out/$TAG; the caller copies them to profiles/r04.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${R04_TAG:-r04z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python - > $OUT/source_digest.txt <<'PY'
import sys; sys.path.insert(0, '.')
from moephoto_amd.build import source_digest
import hashlib
print('sources', source_digest()); print('library', hashlib.sha256(open('moephoto_amd/libmoephoto_amd.so', 'rb').read()).hexdigest())
PY
cat $OUT/source_digest.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
PMC_TAG=$TAG PMC_STEPS=2 bash tools/pmc_bench.sh > $OUT/pmc_stdout.log 2>&1; tail -14 $OUT/pmc_stdout.log
cp $OUT/pmc_bench.json profiles/pmc_bench.json 2>/dev/null      # (bench.py below reads it: same sources, same box)
timeout 900 python bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err; echo "bench c2 rc=$?"
for c in 3 4 5; do timeout 900 python bench.py --config $c > $OUT/bench_c$c.json 2> $OUT/bench_c$c.err; echo "bench c$c rc=$?"; done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -f csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sustain 0 --no-noise-input --no-dropin-loop --no-extras > $OUT/stats_stdout.log 2>&1; echo "stats rc=$?"
find $OUT/stats -name '*kernel_stats.csv' -exec cp {} $OUT/bench_kernel_stats.csv \;
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_grbm
head -14 $OUT/bench_kernel_stats.csv
TM_PREC=auto timeout 600 python tools/time_models.py > $OUT/time_models.txt 2>&1; grep -v "^$" $OUT/time_models.txt | grep -v amdgpu | tail -9
cp gpurun_out/fullsize_report.json $OUT/fullsize_report.json 2>/dev/null
python - $OUT <<'PY'
import json, sys
for c in (2, 3, 4, 5):
    try:
        r = json.loads(open('%s/bench_c%d.json' % (sys.argv[1], c)).read().strip().splitlines()[-1])
        print('config', c, r['ms_per_step'], 'ms', r['value'], r['unit'], '| roofline', r.get('roofline', {}).get('layer_key'), r.get('roofline', {}).get('frac'), '| parity', r['config'].get('parity_max_abs_vs_oracle'), r['config'].get('parity_ok'))
    except Exception as e:
        print('config', c, 'no line', e)
PY
