#!/bin/bash
# round 6, call P: conv3x3_ps9 with the zero-block skip and the whole-chip XCD-aware map -- parity, A/B of both against the variants without, kernel-resolution table
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06p
mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu -x -k "x3_upconv or kernel_forms_agree or golden or a3" 2>&1 | tail -15 > $OUT/pytest_ps9.txt; cat $OUT/pytest_ps9.txt
cp moephoto_amd/libmoephoto_amd.so /tmp/lib_product.so
{
for rep in 1 2; do for v in product ps9_nozskip ps9_noxcd; do
  [ $v = product ] && cp /tmp/lib_product.so moephoto_amd/libmoephoto_amd.so || cp moephoto_amd/_abl/lib_$v.so moephoto_amd/libmoephoto_amd.so
  echo "== $v: $(TM_ONLY='SR a3' TM_PREC=auto timeout 300 python tools/time_models.py 2>&1 | grep 'SR a3')"
done; done
cp /tmp/lib_product.so moephoto_amd/libmoephoto_amd.so
echo "== up_impl = rw: $(MOE_UP_IMPL=rw TM_ONLY='SR a3' TM_PREC=auto timeout 300 python tools/time_models.py 2>&1 | grep 'SR a3')"
} > $OUT/ab_a3.txt 2>&1; cat $OUT/ab_a3.txt
TM_ONLY='SR a3' TM_PREC=auto timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_a3 -o a3 -f csv -- python tools/time_models.py > $OUT/prof_a3.log 2>&1
f=$(find $OUT/prof_a3 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_SR_a3.csv && head -8 $OUT/kernel_stats_SR_a3.csv | cut -c1-160
rm -rf $OUT/prof_a3
bash tools/kernel_table.sh 2>&1 | tail -5; cp gpurun_out/ktable/kernel_resolution.json $OUT/
