#!/bin/bash
# rocprofv3 recipe (run on the GPU box through gpurun).  Kernel-trace/stats and every counter set are SEPARATE runs
# (PMC + trace domains are never combined).  Summaries land in gpurun_out/prof/; copy what matters to profiles/.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
STEPS=${PROF_STEPS:-3}
# 1. per-kernel time of the benchmark command
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -f csv -- python bench.py --steps $STEPS --warmup 1 --no-cpu-baseline > $OUT/stats_stdout.log 2>&1
echo "stats rc=$?" >> $OUT/stats_stdout.log
# 2. counters on the fixed Net4x workload, conv kernel only
timeout 120 rocprofv3-avail list > $OUT/counters_avail.txt 2>&1 || timeout 120 rocprofv3 -L > $OUT/counters_avail.txt 2>&1
pass() {  # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "${PROF_KERNEL_RE:-conv3x3_sp}" -d $OUT/pmc_$name -o pmc -f csv -- python tools/prof_workload.py > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?" >> $OUT/pmc_$name.log
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM
pass sq3 SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_MFMA GRBM_GUI_ACTIVE GRBM_COUNT
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
tail -60 $OUT/summary.txt
