#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${R03_TAG:-r03j}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused_arsb" > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_sel.log
for impl in ${IMPLS:-v2 v3 v2 v3}; do
  MOE_ARSB_IMPL=$impl timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --sustain 0 --no-noise-input --no-dropin-loop 2>$OUT/bench_$impl.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$impl', d['ms_per_step'], [(k['layer_key'], k['ms_per_frame'], k['frac']) for k in d.get('roofline_kernels', [])])
except Exception as e: print('$impl failed', e)"
done
if [ -f moephoto_amd/_abl/lib_trace32.so ]; then TRACE_IMPL=${TRACE_IMPL:-v2} python tools/show_trace_a32.py 2>/dev/null | cut -c1-700 | head -${TRACE_LINES:-12}; fi
