#!/bin/bash
# round 6, call D: the 1-D Winograd probe; why two forwards in flight did not overlap (hardware queues?): GPU_MAX_HW_QUEUES and the fork inside overlapped forwards
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06d
mkdir -p $OUT
{
for v in "" _f8 _np; do echo "== wino1d_probe$v"; timeout 300 tools/micro/bin/wino1d_probe$v 96 512 512 40 | grep -v "^reference"; done
echo "== conv3x3_ps4<1> looped alone on this box"; timeout 300 python tools/kernel_power.py 3 u.up1 2>&1 | grep -v amdgpu.ids
} > $OUT/wino1d_probe.txt 2>&1
cat $OUT/wino1d_probe.txt
P=tools/micro/bin/wino1d_probe
for pass in "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" "sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "grbm GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex wino_kernel -d $OUT/pmc_$name -o pmc -f csv -- $P 96 512 512 3 > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
done
python tools/wino_pmc_report.py $OUT > $OUT/wino1d_pmc.txt 2>&1; cat $OUT/wino1d_pmc.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain 0 --no-noise-input --no-extras --no-configs --no-pmc"
run() {
  "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); l=d['dropin_loop']
print('frame %.2f ms | drop-in loop %.2f ms (%.3f of the headline), with moe_blend_tile %.2f (%.3f) | overlap off: %.2f (%.3f), blend_tile %.2f | forwards only %.2f | bit-identical %s' % (d['ms_per_step'], l['ms_per_step'], l['ratio_to_value'], l['with_moe_blend_tile']['ms_per_step'], l['with_moe_blend_tile']['ratio_to_value'], l['without_overlap_calls']['ms_per_step'], l['without_overlap_calls']['ratio_to_value'], l['without_overlap_calls']['with_moe_blend_tile_ms'], l['breakdown']['engine_forwards_only_ms'], l['without_overlap_calls']['bit_identical_to_overlapped']))
"
}
{
for q in 4 8 16; do for g in 0 128 160; do echo "GPU_MAX_HW_QUEUES=$q overlap_groups=$g fork=1:"; GPU_MAX_HW_QUEUES=$q MOE_OVERLAP_GROUPS=$g run $B; done; done
for g in 0 128 144 160 192; do echo "default queues, overlap_groups=$g fork=0:"; MOE_OVERLAP_FORK=0 MOE_OVERLAP_GROUPS=$g run $B; done
} > $OUT/dropin_overlap_queues.txt 2>&1
cat $OUT/dropin_overlap_queues.txt
python -m pytest tests -q -m gpu -x -k "consecutive_forwards" 2>&1 | tail -5 > $OUT/pytest_subset.txt; cat $OUT/pytest_subset.txt
