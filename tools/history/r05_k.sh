#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05k
mkdir -p $OUT
for rep in 1 2; do for bs in 1 0; do
  MOE_BRANCH_STREAMS=$bs timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --no-noise-input --no-extras --no-configs --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('branch_streams=$bs', 'ms_per_step', d['ms_per_step'], [(k['layer_key'], k['ms_per_frame'], k['frac']) for k in d.get('roofline_kernels', [])], 'dropin', d['dropin_loop']['ms_per_step'], d['dropin_loop']['with_moe_blend_tile']['ms_per_step'])"
done; done > $OUT/ab_branch_streams_bench.txt 2>&1; cat $OUT/ab_branch_streams_bench.txt
