#!/bin/bash
# round 6, call Z7: timing ablation of conv1x1's storing form: the same bytes with every store instruction 1 KiB contiguous (results wrong) -- is the 32-byte-per-pixel store pattern what holds it?
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r06z7
mkdir -p $OUT
cp moephoto_amd/libmoephoto_amd.so /tmp/lib_product.so
{
for rep in 1 2; do for v in product c1_coalesce; do
  [ $v = product ] && cp /tmp/lib_product.so moephoto_amd/libmoephoto_amd.so || cp moephoto_amd/_abl/lib_$v.so moephoto_amd/libmoephoto_amd.so
  echo "== $v"; TM_PREC=auto timeout 400 python tools/time_models.py 2>&1 | grep -E "lite4|lite8"
done; done
cp /tmp/lib_product.so moephoto_amd/libmoephoto_amd.so
} > $OUT/ab_coalesce.txt 2>&1; cat $OUT/ab_coalesce.txt
