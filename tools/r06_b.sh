#!/bin/bash
# round 6, call B: Winograd probe variants (raw ring without bank conflicts; pin sizes; packed fp32 subtractions) beside conv3x3_ps4 on the same box; calibration report
# and margin sweep with the new calibrator; the GPU tests touched so far
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06b
mkdir -p $OUT
{
for v in a b c d e; do echo "== wino_probe_$v"; timeout 300 tools/micro/bin/wino_probe_$v 96 512 512 40 | grep -v "^reference"; done
echo "== conv3x3_ps4<1> / <2> looped alone on this box (tools/kernel_power.py)"; timeout 300 python tools/kernel_power.py 3 u.up1,convt_R1.up1 2>&1 | grep -v amdgpu.ids
} > $OUT/wino_probe_variants.txt 2>&1
cat $OUT/wino_probe_variants.txt
P=tools/micro/bin/wino_probe_a
for pass in "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" "sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "grbm GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex wino_kernel -d $OUT/pmc_$name -o pmc -f csv -- $P 96 512 512 3 > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
done
python tools/wino_pmc_report.py $OUT > $OUT/wino_pmc.txt 2>&1; cat $OUT/wino_pmc.txt
timeout 600 python tools/calib_report.py > $OUT/calib_report.txt 2>&1; grep -v amdgpu.ids $OUT/calib_report.txt
timeout 900 python tools/margin_sweep.py a4 a2 > $OUT/margin_sweep.txt 2>&1; grep -v amdgpu.ids $OUT/margin_sweep.txt
python -m pytest tests -q -m gpu -k "blend_tile or calibrate or small_launch or integration_md or dropin" 2>&1 | tail -15 > $OUT/pytest_subset.txt; cat $OUT/pytest_subset.txt
