#!/bin/bash
# full GPU test suite + short bench + frame times of the other families
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${R03_TAG:-r03t}
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --sustain 0 --no-noise-input --no-dropin-loop 2>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench ms', d['ms_per_step'], [(k['layer_key'], k['ms_per_frame'], k['frac']) for k in d.get('roofline_kernels', [])])"
TM_PREC=auto timeout 600 python tools/time_models.py 2>/dev/null | tee $OUT/time_models.txt
