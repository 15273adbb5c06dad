#!/bin/bash
mkdir -p gpurun_out/r02h
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "net_forward_vs or docrop or ragged or full_size or fused_arsb or ensemble" > gpurun_out/r02h/pytest.log 2>&1; echo "pytest rc=$?"
tail -8 gpurun_out/r02h/pytest.log | cut -c1-300
for ts in 0 r ru; do
  MOE_TAIL_SPLIT=$ts timeout 600 python bench.py --steps 10 --sustain 0 > gpurun_out/r02h/bench_ts_$ts.json 2> gpurun_out/r02h/bench_ts_$ts.err; echo "bench $ts rc=$?"
  python - <<PY
import json
r=json.load(open('gpurun_out/r02h/bench_ts_$ts.json'))
print('$ts', r['ms_per_step'], r['inputs'], r['roofline']['avg_launch_ms'], 'natural', r['config']['parity']['natural']['worst_max_abs'], 'noise', r['config']['parity']['noise_u8']['worst_max_abs'], r['config']['parity']['noise_u8']['per_tile'])
PY
done
