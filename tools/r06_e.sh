#!/bin/bash
# round 6, call E: 1-D Winograd probe v2 (double-steps: one barrier per 96 MFMAs, DMA a double-step ahead, fragments 3-5 slots ahead)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06e
mkdir -p $OUT
{
for v in v2 v2a5 v2f8 v2np; do echo "== wino1d_probe_$v"; timeout 300 tools/micro/bin/wino1d_probe_$v 96 512 512 40 | grep -v "^reference"; done
echo "== conv3x3_ps4<1> looped alone on this box"; timeout 300 python tools/kernel_power.py 3 u.up1 2>&1 | grep -v amdgpu.ids
} > $OUT/wino1d_probe_v2.txt 2>&1
cat $OUT/wino1d_probe_v2.txt
P=tools/micro/bin/wino1d_probe_v2
for pass in "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" "sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "grbm GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex wino_kernel -d $OUT/pmc_$name -o pmc -f csv -- $P 96 512 512 3 > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
done
python tools/wino_pmc_report.py $OUT > $OUT/wino1d_v2_pmc.txt 2>&1; cat $OUT/wino1d_v2_pmc.txt
