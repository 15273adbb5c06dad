#!/bin/bash
# round 6, call ZK: lite's FRM gate from conv_2's INPUT (frm_pre: conv64_x3 EPI 4 / 5 + frm_pre_kernel, no frm_apply pass): parity of the lite family, A/B of the frames, kernel table
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06zk
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x -k "lite or stub or golden or ragged or fuzz" 2>&1 | tail -8 > $OUT/pytest.txt; cat $OUT/pytest.txt | cut -c1-300
{
for i in 1 2; do for v in 0 1; do echo "== MOE_FRM_PRE=$v"; MOE_FRM_PRE=$v TM_PREC=auto timeout 600 python tools/time_models.py 2>&1 | grep -E "^SR lite"; done; done
} > $OUT/ab_frm_pre.txt 2>&1; cat $OUT/ab_frm_pre.txt
FUZZ_N=12 FUZZ_KEYS=lite2,lite4,lite8 FUZZ_SEED=31 FUZZ_CROPS=6 timeout 600 python tools/fuzz_gpu.py > $OUT/fuzz.txt 2>&1; echo "fuzz rc=$?"; grep -v amdgpu $OUT/fuzz.txt | tail -8
TM_ONLY='SR lite2' TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/p -o t -f csv -- python tools/time_models.py > $OUT/p.log 2>&1
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_SR_lite2.csv && head -10 "$f" | cut -c1-150; rm -rf $OUT/p
bash tools/kernel_table.sh 2>&1 | tail -3
