#!/bin/bash
# round 5, call A: (1) quick parity subset on the new tree, (2) calibration report, (3) "find the joules": SQ instruction-mix counters of the bench's own launches + per-kernel
# package power / clock, (4) the drop-in loop taken apart.   -> gpurun_out/r05a/
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05a
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blend_tile or calibrate or integration_md or one_launch_arsb or net_forward_vs_reference_golden or dropin_protocol or layer_by_layer" > $OUT/pytest_subset.txt 2>&1
echo "pytest subset rc=$?"; tail -3 $OUT/pytest_subset.txt
timeout 300 python tools/calib_report.py > $OUT/calib_report.txt 2>&1; echo "calib rc=$?"; cat $OUT/calib_report.txt | tail -20
timeout 200 python tools/kernel_power.py 4 > $OUT/kernel_power.txt 2>&1; echo "power rc=$?"; cat $OUT/kernel_power.txt
timeout 200 python tools/prof_dropin.py 8 > $OUT/prof_dropin.txt 2>&1; echo "dropin rc=$?"; cat $OUT/prof_dropin.txt
DROPIN_ONLY=loop timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/dropin_trace -o dropin -f csv -- python tools/prof_dropin.py 6 > $OUT/dropin_trace.log 2>&1
echo "dropin trace rc=$?"
python tools/kstats.py $OUT/dropin_trace 2>/dev/null | head -40 > $OUT/dropin_kernel_stats.txt; cat $OUT/dropin_kernel_stats.txt | head -30
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters.txt; wc -l $OUT/sq_counters.txt
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain 0 --no-noise-input --no-dropin-loop --no-extras"
RE='conv3x3_ps4|arsb32c|arsb_sq|conv64_sq'
pass() { name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "$RE" -d $OUT/pmc_$name -o pmc -f csv -- $CMD > $OUT/pmc_$name.log 2>&1; echo "pmc $name rc=$?"; }
pass mix SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT
pass lds SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass grbm GRBM_GUI_ACTIVE
python tools/pmc_mix.py $OUT > $OUT/pmc_mix.txt 2>&1; cat $OUT/pmc_mix.txt
