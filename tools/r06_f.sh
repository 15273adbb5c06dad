#!/bin/bash
# round 6, call F: 1-D Winograd probe v3 (micro-ops dealt evenly) with package power / clock sampled while it loops; the same for the 2-D probe and conv3x3_ps4<1>
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06f
mkdir -p $OUT
smp() {   # command...: run it in the background, sample rocm-smi while it lives
  "$@" > $OUT/.run.txt 2>&1 &
  BP=$!
  n=0
  while kill -0 $BP 2>/dev/null && [ $n -lt 8 ]; do
    s=$(rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics Package Power" | tr -s ' \t' ' ' | sed 's/GPU\[0\] : //' | tr '\n' '|')
    case "$s" in *"(95Mhz)"*|*"(132Mhz)"*) ;; *) echo "   $s"; n=$((n+1));; esac
    sleep 0.5
  done
  wait $BP
  grep -v "^validation\|^reference" $OUT/.run.txt
}
{
for v in v3 v3f4 v3f6 v2; do echo "== wino1d_probe_$v (1500 launches, rocm-smi beside it)"; smp tools/micro/bin/wino1d_probe_$v 96 512 512 1500; done
echo "== wino_probe_a (2-D), 1500 launches"; smp tools/micro/bin/wino_probe_a 96 512 512 1500
echo "== conv3x3_ps4<1> looped alone on this box"; timeout 300 python tools/kernel_power.py 4 u.up1 2>&1 | grep -v amdgpu.ids
} > $OUT/wino1d_probe_v3_power.txt 2>&1
cat $OUT/wino1d_probe_v3_power.txt
P=tools/micro/bin/wino1d_probe_v3
for pass in "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT" "sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "grbm GRBM_GUI_ACTIVE"; do
  set -- $pass; name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex wino_kernel -d $OUT/pmc_$name -o pmc -f csv -- $P 96 512 512 3 > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
done
python tools/wino_pmc_report.py $OUT > $OUT/wino1d_v3_pmc.txt 2>&1; tail -6 $OUT/wino1d_v3_pmc.txt
