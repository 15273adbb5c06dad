#!/bin/bash
# PMC passes over bench.py's OWN launches (the launch shapes the headline is measured on), counters only, one pass per counter set
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots), then tools/pmc_collect.py condenses
# the CSVs into profiles-ready JSON (bytes per frame and launch, MFMA busy, effective clock per kernel).
#   PMC_TAG=r03a tools/pmc_bench.sh        -> gpurun_out/$PMC_TAG/{pmc_*,pmc_bench.json,pmc_bench.txt}
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${PMC_TAG:-pmc}
mkdir -p $OUT
STEPS=${PMC_STEPS:-2}
CMD="python bench.py --steps $STEPS --warmup 1 --no-cpu-baseline --sustain 0 --no-noise-input --no-dropin-loop --no-extras --no-configs --no-pmc --no-floor"
RE='conv3x3|arsb32c|arsb_sq|conv64_x3|conv64_q8|conv64_sq|conv64_s|tapsum|tailadd|stitch|stem_kernel'
pass() {  # name, counters...
  name=$1; shift
  timeout 420 rocprofv3 --pmc "$@" --kernel-include-regex "$RE" -d $OUT/pmc_$name -o pmc -f csv -- $CMD > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT
pass grbm GRBM_GUI_ACTIVE
python tools/pmc_collect.py $OUT $((STEPS + 1)) > $OUT/pmc_bench.txt 2>&1
cat $OUT/pmc_bench.txt | head -60
