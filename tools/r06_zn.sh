#!/bin/bash
# round 6, call ZN: lite's U branch as a table over the fp16 bit patterns (option lite_lut): parity of the lite family + everything around it, A/B of the frames, table build time
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06zn
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x -k "lite or stub or golden or ragged or e2e or chain" 2>&1 | tail -6 > $OUT/pytest.txt; cat $OUT/pytest.txt | cut -c1-300
{
for i in 1 2; do for v in 0 1; do echo "== MOE_LITE_LUT=$v"; MOE_LITE_LUT=$v TM_PREC=auto timeout 600 python tools/time_models.py 2>&1 | grep -E "^SR lite"; done; done
} > $OUT/ab_lite_lut.txt 2>&1; cat $OUT/ab_lite_lut.txt
FUZZ_N=12 FUZZ_KEYS=lite2,lite4,lite8 FUZZ_SEED=37 FUZZ_CROPS=6 timeout 600 python tools/fuzz_gpu.py > $OUT/fuzz.txt 2>&1; echo "fuzz rc=$?"; grep -v amdgpu $OUT/fuzz.txt | tail -6
TM_ONLY='SR lite8' TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p -o t -f csv -- python tools/time_models.py > $OUT/p.log 2>&1
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_SR_lite8.csv && head -10 "$f" | cut -c1-150; rm -rf $OUT/p
