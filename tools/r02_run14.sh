#!/bin/bash
mkdir -p gpurun_out/r02m
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x > gpurun_out/r02m/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02m/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 > gpurun_out/r02m/bench.json 2> gpurun_out/r02m/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02m/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('roofline'), d.get('roofline_trunk'), d.get('inputs'))
PY
