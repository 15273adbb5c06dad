#!/bin/bash
# round 6, call M: the row-stationary 1-D Winograd probe (three MFMAs per fragment read) beside the others and conv3x3_ps4<1> on one box
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06m
mkdir -p $OUT
{
for v in wino1d_rs_probe wino1d_rs_probe_np wino1d_probe_v4 wino_probe_a; do echo "== $v"; timeout 300 tools/micro/bin/$v 96 512 512 60 | grep -v "^reference"; done
echo "== conv3x3_ps4<1> looped alone on this box"; timeout 300 python tools/kernel_power.py 3 u.up1 2>&1 | grep -v amdgpu.ids
} > $OUT/wino1d_rs_probe.txt 2>&1
cat $OUT/wino1d_rs_probe.txt
P=tools/micro/bin/wino1d_rs_probe
for pass in "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" "sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "grbm GRBM_GUI_ACTIVE" "fetch FETCH_SIZE"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex wino_kernel -d $OUT/pmc_$name -o pmc -f csv -- $P 96 512 512 3 > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
done
python tools/wino_pmc_report.py $OUT > $OUT/wino1d_rs_pmc.txt 2>&1; tail -7 $OUT/wino1d_rs_pmc.txt
