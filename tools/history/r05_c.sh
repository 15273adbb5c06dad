#!/bin/bash
# round 5, call C (HISTORICAL: option stream8 / MOE_STREAM8 existed at commit 0486050 only -- measured, then dropped; profiles/r05/c_stream8_ab.txt): the fp8 low-part stream in the single-pass ARSBs -- parity subset, calibration errors with and without, A/B of the frame inside one call
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05c
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or noise or exact_mode or layer_by_layer or one_launch_arsb or fused_arsb or calibrate or ragged or integration_md" > $OUT/pytest_subset.txt 2>&1
echo "pytest subset rc=$?"; tail -3 $OUT/pytest_subset.txt
for s8 in 1 0; do echo "== MOE_STREAM8=$s8"; MOE_STREAM8=$s8 timeout 300 python tools/calib_report.py 2>&1 | grep -v amdgpu.ids; done > $OUT/calib_stream8.txt 2>&1; cat $OUT/calib_stream8.txt
for rep in 1 2; do for s8 in 1 0; do
  MOE_STREAM8=$s8 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --no-dropin-loop --no-extras --no-configs --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stream8=$s8', 'ms_per_step', d['ms_per_step'], 'noise', d['inputs'].get('noise_u8', {}).get('ms_per_step'), [(k['layer_key'], k['ms_per_frame'], k['frac']) for k in d.get('roofline_kernels', [])], 'split', d.get('roofline_split_operand', {}).get('ms_per_frame'))"
done; done > $OUT/ab_stream8.txt 2>&1; cat $OUT/ab_stream8.txt
MOE_AUTO_CALIBRATE=0 timeout 120 python tools/kernel_power.py 3 arsb3 stream8 2>&1 | grep arsb32c | tee $OUT/kernel_power_stream8.txt
MOE_STREAM8=0 MOE_AUTO_CALIBRATE=0 timeout 120 python tools/kernel_power.py 3 arsb3 fp16lo 2>&1 | grep arsb32c | tee -a $OUT/kernel_power_stream8.txt
