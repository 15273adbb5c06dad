#!/bin/bash
# round 6, call C: overlapped consecutive forwards (moe_net_forward_ex): the new GPU test, the drop-in loop of bench.py with the option off / on / on half the chip;
# the 12-tile calibration against the full-frame sweeps
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06c
mkdir -p $OUT
python -m pytest tests -q -m gpu -x -k "consecutive_forwards or calibrate or integration_md or small_launch or blend_tile or dropin" 2>&1 | tail -15 > $OUT/pytest_subset.txt; cat $OUT/pytest_subset.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --no-noise-input --no-extras --no-configs --no-pmc"
for g in 0 96 128 160 192; do
  MOE_OVERLAP_GROUPS=$g $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); l=d['dropin_loop']
print('overlap_groups=$g: frame %.2f ms | drop-in loop %.2f ms (%.3f of the headline), with moe_blend_tile %.2f (%.3f) | overlap off: %.2f (%.3f), blend_tile %.2f | forwards only %.2f | bit-identical %s, vs doCrop %.1e' % (d['ms_per_step'], l['ms_per_step'], l['ratio_to_value'], l['with_moe_blend_tile']['ms_per_step'], l['with_moe_blend_tile']['ratio_to_value'], l['without_overlap_calls']['ms_per_step'], l['without_overlap_calls']['ratio_to_value'], l['without_overlap_calls']['with_moe_blend_tile_ms'], l['breakdown']['engine_forwards_only_ms'], l['without_overlap_calls']['bit_identical_to_overlapped'], l['max_abs_vs_device_docrop']))
print('   value_floor', d['config'].get('value_floor'), 'exact_blocks', d['config'].get('exact_blocks'))
"
done > $OUT/dropin_overlap_ab.txt 2>&1
cat $OUT/dropin_overlap_ab.txt
timeout 600 python tools/calib_report.py > $OUT/calib_report.txt 2>&1; grep -v amdgpu.ids $OUT/calib_report.txt
timeout 900 python tools/margin_sweep.py a4 a2 > $OUT/margin_sweep.txt 2>&1; grep -v amdgpu.ids $OUT/margin_sweep.txt
