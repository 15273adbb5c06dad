#!/bin/bash
# Round-3 pass B: the phase-class-sums fused tail -- targeted parity tests, then planes vs sums timing in one session.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${R03_TAG:-r03b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "kernel_forms or forward_vs_reference or ragged or docrop_vs or ensemble or full_size_properties or auto_cropsize or large_batches or dropin or e2e" > $OUT/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -30 $OUT/pytest_sel.log
for form in planes sums planes sums; do
  MOE_TAIL_FORM=$form timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sustain 4 --no-noise-input --no-dropin-loop > $OUT/bench_$form.json 2> $OUT/bench_$form.err
  echo "$form rc=$?"; python - <<P
import json
try:
    d = json.load(open('$OUT/bench_$form.json'))
    print('$form', d['ms_per_step'], 'sustained', d.get('sustained', {}).get('ms_per_step'), [(k['layer_key'], k['ms_per_frame'], k['frac']) for k in d.get('roofline_kernels', [])], d.get('clock', {}).get('sclk_ghz_mean'))
except Exception as e:
    print('parse failed', e); print(open('$OUT/bench_$form.err').read()[-1500:])
P
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -f csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sustain 0 --no-noise-input --no-dropin-loop > $OUT/stats_stdout.log 2>&1; echo "stats rc=$?"
python tools/summarize_prof.py $OUT 2>/dev/null | head -16
if [ "${R03_EXPLORE:-0}" = "1" ]; then timeout 400 python tools/explore_exact_blocks.py > $OUT/explore_exact_blocks.txt 2>&1; cat $OUT/explore_exact_blocks.txt | grep exact_blocks; fi
