#!/bin/bash
# round 6, call G: 1-D Winograd probe with the two workgroups of a strip on one XCD (shared L2) vs not: time and FETCH_SIZE; a2's exact-block table
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06g
mkdir -p $OUT
{
for v in v4 v4x0 v4 v4x0; do echo "== wino1d_probe_$v"; timeout 300 tools/micro/bin/wino1d_probe_$v 96 512 512 60 | grep -v "^reference\|^validation"; done
} > $OUT/wino1d_probe_v4.txt 2>&1
cat $OUT/wino1d_probe_v4.txt
for v in v4 v4x0; do
for pass in "fetch FETCH_SIZE" "grbm GRBM_GUI_ACTIVE" "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex wino_kernel -d $OUT/$v/pmc_$name -o pmc -f csv -- tools/micro/bin/wino1d_probe_$v 96 512 512 3 > $OUT/pmc_${v}_$name.log 2>&1
done
echo "== $v"; python tools/wino_pmc_report.py $OUT/$v | tail -5
done > $OUT/wino1d_v4_pmc.txt 2>&1
cat $OUT/wino1d_v4_pmc.txt
timeout 900 python tools/explore_exact_blocks.py > $OUT/explore_exact_blocks.txt 2>&1; grep -v amdgpu.ids $OUT/explore_exact_blocks.txt
