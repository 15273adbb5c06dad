#!/bin/bash
# round-2 checkpoint: whole -m gpu suite, full bench, model timings, kernel stats, PMC of the dominant kernel
export TMPDIR=/tmp
OUT=gpurun_out/r02g
mkdir -p $OUT
python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -6 $OUT/pytest.log | cut -c1-300
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json | cut -c1-3000
TM_PREC=auto,fp16 python tools/time_models.py > $OUT/time_models.txt 2>&1; cat $OUT/time_models.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -f csv -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --sustain 0 --no-noise-input > $OUT/stats_stdout.log 2>&1
pass() {  # name, regex, counters...
  name=$1; re=$2; shift; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "$re" -d $OUT/pmc_$name -o pmc -f csv -- python tools/prof_workload.py > $OUT/pmc_$name.log 2>&1
}
pass fetch_sp "conv3x3_sp_kernelILi3" FETCH_SIZE
pass write_sp "conv3x3_sp_kernelILi3" WRITE_SIZE
pass sq_sp "conv3x3_sp_kernelILi3" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE
pass fetch_x3 "conv64_x3" FETCH_SIZE
pass write_x3 "conv64_x3" WRITE_SIZE
pass sq_x3 "conv64_x3" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -v "log:" $OUT/summary.txt | head -70
