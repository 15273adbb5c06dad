#!/bin/bash
# Round-2 measurement pass on the GPU box: full GPU tests, the benchmark line, per-kernel stats of the same command, PMC passes
# (separate runs, counters only) for the dominant conv kernel and the fused ARSB kernel, model-family timings.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${R02_TAG:-r02z}
mkdir -p $OUT
timeout 1200 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 600 $OUT/bench.json; echo
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o bench -f csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sustain 0 > $OUT/stats_stdout.log 2>&1; echo "stats rc=$?"
pass() {  # name, kernel regex, counters...
  name=$1; re=$2; shift; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-include-regex "$re" -d $OUT/pmc_$name -o pmc -f csv -- python tools/prof_workload.py > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
}
SP='conv3x3_(sp|rw)_kernel<[137]>'
AR='arsb_fused_kernel'
pass sp_fetch "$SP" FETCH_SIZE
pass sp_write "$SP" WRITE_SIZE
pass sp_sq "$SP" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT
pass sp_sq2 "$SP" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
pass arsb_fetch "$AR" FETCH_SIZE
pass arsb_write "$AR" WRITE_SIZE
pass arsb_sq "$AR" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -v "log:" $OUT/summary.txt | head -90
TM_PREC=auto,fp16 timeout 600 python tools/time_models.py > $OUT/time_models.txt 2>&1; cat $OUT/time_models.txt | grep ms/frame
