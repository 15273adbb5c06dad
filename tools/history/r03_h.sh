#!/bin/bash
# stitch kernel: the parity tests that exercise it + config 5's stitch rate + a short bench (r03r / r03s ran this with a build-time switch between
# the eight-pixel form of round 3's first half and 4 / 8 rows per block: profiles/r03/h_stitch_ab.txt)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${R03_TAG:-r03r}
mkdir -p $OUT
for form in 4; do
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -x -k "stitch or config5_full_size or full_size_properties_config2" > $OUT/pytest_form$form.log 2>&1; echo "form $form pytest rc=$?"; tail -3 $OUT/pytest_form$form.log
  python - <<P
import json
d = json.load(open('gpurun_out/fullsize_report.json'))
for k, v in d.items():
    if isinstance(v, dict) and 'stitch_ms' in v: print('form $form', k, v['stitch_ms'], v['stitch_algorithmic_gb_per_s'], v.get('stitch_pool_bytes_read_gb_per_s'))
P
  timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --sustain 0 --no-noise-input --no-dropin-loop 2>$OUT/bench_form$form.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('form $form bench ms', d['ms_per_step'])"
done
