#!/bin/bash
mkdir -p gpurun_out/r02t
timeout 1200 python -m pytest tests -q -m gpu -x > gpurun_out/r02t/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02t/pytest.log
for impl in default sp default sp; do
e=""; [ $impl = sp ] && e="MOE_SP_IMPL=sp"
env $e timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --no-noise-input 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$impl', 'ms_per_step', d['ms_per_step'], 'up1 avg ms', d['roofline']['avg_launch_ms'])"
done
for impl in default sp; do e=""; [ $impl = sp ] && e="MOE_SP_IMPL=sp"; env $e TM_ONLY="DN l25" TM_PREC=auto timeout 300 python tools/time_models.py 2>&1 | grep ms/frame; done
