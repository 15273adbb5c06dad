#!/bin/bash
# conv64_q8 (x3_impl = q8) as the exact layers' kernel: its own test, the all-tile parity sweep and the goldens with it, bench and frame times q8 vs x3
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${R03_TAG:-r04f}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fp8_corrections" 2>&1 | tail -2
MOE_X3_IMPL=q8 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -k "golden or sweep or ragged or step_chain or full_size_properties or ensemble" > $OUT/pytest_q8.log 2>&1; echo "pytest(q8 default) rc=$?"; tail -3 $OUT/pytest_q8.log
python - <<P
import json
d = json.load(open('gpurun_out/fullsize_report.json'))
print('sweep with q8:', {k: max(v.values()) for k, v in d.get('parity_sweep_max_abs_default_vs_fp16x3', {}).items()})
P
for rep in 1 2; do for impl in x3 q8; do
  MOE_X3_IMPL=$impl timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --sustain 0 --no-dropin-loop --no-noise-input 2>$OUT/bench_$impl.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$impl', d['ms_per_step'], 'parity', d['config'].get('parity_max_abs_vs_oracle'), d['config'].get('parity_ok'), [(k['layer_key'], k['ms_per_frame']) for k in d.get('roofline_kernels', [])])"
  echo -n "$impl "; MOE_X3_IMPL=$impl TM_ONLY="SR a2" TM_PREC=auto timeout 300 python tools/time_models.py 2>/dev/null | tail -1
done; done
