#!/bin/bash
mkdir -p gpurun_out/r02r
MOE_SP_IMPL=rw timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "golden or fuzz or config5 or batches or ragged or tiny or whole" > gpurun_out/r02r/pytest_rw.log 2>&1; echo "pytest rw rc=$?"; tail -4 gpurun_out/r02r/pytest_rw.log
for impl in rw sp rw sp; do
MOE_SP_IMPL=$impl timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --no-noise-input 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$impl', 'ms_per_step', d['ms_per_step'], 'up1 avg ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'])"
done
