#!/bin/bash
# round-2 GPU call: PMC counters of the fused ARSB kernel on the fixed Net4x workload (B = 12 planes of 256 x 256)
export TMPDIR=/tmp
OUT=gpurun_out/r02c
mkdir -p $OUT
pass() {  # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "arsb_fused" -d $OUT/pmc_$name -o pmc -f csv -- python tools/prof_workload.py > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?" >> $OUT/pmc_$name.log
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM
pass sq3 SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_MFMA GRBM_GUI_ACTIVE GRBM_COUNT
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
