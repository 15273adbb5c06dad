#!/bin/bash
# round 6, call O: conv3x3_ps9 (x3 nets' up-conv with the phase rows in three-wave workgroups) -- parity, a3 frame time against the per-phase form, kernel stats;
# the hand-slotted variants of the row-stationary 1-D Winograd probe
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06o
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -x -k "x3_upconv or kernel_forms_agree or golden or a3" 2>&1 | tail -15 > $OUT/pytest_ps9.txt; cat $OUT/pytest_ps9.txt
{
echo "== a3, up_impl = ps4 (conv3x3_ps9)"; TM_ONLY='SR a3' TM_PREC=auto timeout 300 python tools/time_models.py 2>&1 | grep -v amdgpu.ids
echo "== a3, up_impl = rw (conv3x3_sp + tap planes + tapsum<3>)"; MOE_UP_IMPL=rw TM_ONLY='SR a3' TM_PREC=auto timeout 300 python tools/time_models.py 2>&1 | grep -v amdgpu.ids
} > $OUT/time_a3.txt 2>&1; cat $OUT/time_a3.txt
TM_ONLY='SR a3' TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_a3 -o a3 -f csv -- python tools/time_models.py > $OUT/prof_a3.log 2>&1
f=$(find $OUT/prof_a3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_SR_a3.csv && head -12 $OUT/kernel_stats_SR_a3.csv | cut -c1-160
{
for v in wino1d_rs_probe_s0 wino1d_rs_probe_s wino1d_rs_probe_s4 wino1d_rs_probe_s6; do echo "== $v"; timeout 300 tools/micro/bin/$v 96 512 512 60 | grep -v "^reference"; done
} > $OUT/wino1d_rs_slot.txt 2>&1; cat $OUT/wino1d_rs_slot.txt
rm -rf $OUT/prof_a3
