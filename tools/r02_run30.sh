#!/bin/bash
mkdir -p gpurun_out/r02x
timeout 1200 python -m pytest tests -q -m gpu > gpurun_out/r02x/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r02x/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r02x/bench.json 2> gpurun_out/r02x/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02x/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_trunk']['frac'], d['sustained'], d['config']['parity']['noise_u8']['worst_max_abs'], d['config'].get('parity_ok'), d['cpu_baseline']['value'])
PY
